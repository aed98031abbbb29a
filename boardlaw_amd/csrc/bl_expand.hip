// bl_expand.hip -- bl_sim_expand on the compacted policy rows: descend (mcts/cpp/cuda.cu:138-182) + tree expansion
// (mcts/__init__.py:117-129) + Hex step/observe (hex/cpp/cuda.cu:76-195) for one simulation, one wave per env.
//
// What bounds this kernel is not HBM and not VALU throughput but the LATENCY of one wave's dependent chain: the launch
// ends with its deepest env (25 levels at 9x9/64 sims against 5 on average), which runs alone on its SIMD for most of
// its life.  Everything here is arranged to shorten that chain, bit for bit the same arithmetic as the reference:
//
//  * per-node statistics live in registers.  Lane t loads {w[b,t,:], n[b,t], nk, seat, terminal, rand[b,t]} of node
//    slot t once, at the start (T <= 256: 1..4 registers each); a level looks its children up with ds_bpermute and the
//    next node's seat/terminal/uniform with v_readlane -- no second memory round trip per level.
//  * compacted rows (bl_device.h: compact_store): a level loads pi and (child, action) of the kept actions only -- no
//    exp-table gather, no in-kernel compaction, half the row bytes on a mid-game board.
//  * ONE dependent DPP chain per level.  The Newton sums S = sum s_a and g = sum g_a must be folded in ascending action
//    order (float addition is not associative and the drawn action depends on every rounding).  Element e of a block
//    of 32 kept actions is evaluated by lane e (s term: top/(alpha-q)) and by lane 32+e (g term: -top/(alpha-q)^2) --
//    one IEEE division per lane instead of two -- and `v_add_f32_dpp row_shr:1` advances both chains, in all four
//    16-lane rows, with one instruction per step; rows hand over with row_bcast:15, blocks with row_bcast:15 (lane 31
//    -> 32) and wave_ror:1 (lane 63 -> 0), the S/g halves swapping sides from block to block so that both carries are
//    "next row" moves.  Each lane's last update reads a neighbour that is already final, so after j steps lanes 0..j of
//    a row hold the reference's running totals exactly; the totals stay in registers for the draw (cuda.cu:157-176).
//  * wait states: the ISA asks for 2 between a VALU write and a DPP read of the same VGPR (no interlock).  The SAFE
//    variant pads every step with `s_nop 1`; the FAST one with `s_nop 0`, which tools/micro/fold_variants.hip measures
//    as sufficient on gfx950 (a padded step then takes as long as the hardware-interlocked dependent v_add: 8.8 vs 9.2
//    cycles).  FAST is used only after bl_selftest() has verified it on the device at hand (bl_abi.hip: bl_selftest).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <type_traits>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"

#pragma clang fp contract(off)

namespace bl {

#define BLX_ROWSTEP "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define BLX_STEP7(NOP) BLX_ROWSTEP NOP BLX_ROWSTEP NOP BLX_ROWSTEP NOP BLX_ROWSTEP NOP BLX_ROWSTEP NOP BLX_ROWSTEP NOP BLX_ROWSTEP NOP
#define BLX_STEP8(NOP) BLX_STEP7(NOP) BLX_ROWSTEP NOP

// 7 resp. 8 in-row steps of the chain, all four rows at once.  Steps beyond the ones a row needs recompute the same values,
// so a half row of up to 8 elements takes fold7 and a longer one fold7 + fold8: at most three uniform branches per block.
template <bool FAST>
__device__ __forceinline__ void fold7(float& x, const float t) {
    if constexpr (FAST) asm volatile("s_nop 0\n\t" BLX_STEP7("s_nop 0\n\t") : "+v"(x) : "v"(t));
    else asm volatile("s_nop 1\n\t" BLX_STEP7("s_nop 1\n\t") : "+v"(x) : "v"(t));
}
// Every asm block starts with its own wait state(s): the compiler may place a VALU write of x (a copy, a select) between two
// blocks, and the first DPP read of a block must not rely on the previous block's trailing nop.
template <bool FAST>
__device__ __forceinline__ void fold8(float& x, const float t) {
    if constexpr (FAST) asm volatile("s_nop 0\n\t" BLX_STEP8("s_nop 0\n\t") : "+v"(x) : "v"(t));
    else asm volatile("s_nop 1\n\t" BLX_STEP8("s_nop 1\n\t") : "+v"(x) : "v"(t));
}

// One block of m <= 32 elements whose lanes 0 and 32 already hold their final values.
template <bool FAST>
__device__ __forceinline__ void fold_block(float& x, const float t, const int m);

// rows 0 -> 1 and 2 -> 3 of one block: lane 16 <- x[15] + t[16], lane 48 <- x[47] + t[48]  (lanes 4, 8, 12 of those rows
// receive values that later steps overwrite)
template <bool FAST>
__device__ __forceinline__ void fold_mid(float& x, const float t) {
    if constexpr (FAST) asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0x1\n\ts_nop 0\n\t" : "+v"(x) : "v"(t));
    else asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %1 row_bcast:15 row_mask:0xa bank_mask:0x1\n\ts_nop 1\n\t" : "+v"(x) : "v"(t));
}

// block r-1 -> block r: lane 32 <- xprev[31] + t[32] (row_bcast:15 into row 2), lane 0 <- xprev[63] + t[0] (wave_ror:1)
template <bool FAST>
__device__ __forceinline__ void fold_carry(float& x, const float xprev, const float t) {
    if constexpr (FAST)
        asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %1, %2 row_bcast:15 row_mask:0x4 bank_mask:0x1\n\t"
                     "v_add_f32_dpp %0, %1, %2 wave_ror:1 row_mask:0x1 bank_mask:0x1\n\ts_nop 0\n\t" : "+v"(x) : "v"(xprev), "v"(t));
    else
        asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_bcast:15 row_mask:0x4 bank_mask:0x1\n\t"
                     "v_add_f32_dpp %0, %1, %2 wave_ror:1 row_mask:0x1 bank_mask:0x1\n\ts_nop 1\n\t" : "+v"(x) : "v"(xprev), "v"(t));
}

template <bool FAST>
__device__ __forceinline__ void fold_block(float& x, const float t, const int m) {
    fold7<FAST>(x, t);
    if (m > 8) fold8<FAST>(x, t);
    if (m > 16) {
        fold_mid<FAST>(x, t);
        fold7<FAST>(x, t);
        if (m > 24) fold8<FAST>(x, t);
    }
}

__device__ __forceinline__ int bperm_i(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// ------------------------------------------------------------------------------------------------------------------
// RMAX: blocks of 32 kept actions a node can have (ceil(A / 32));  KT: registers of node slots (ceil(T / 64));
// NW: waves per env.
//
// POWF: the reference's own JIT build (bl_tune_t.powf_libm): the derivative term divides by glibc's powf(bot, 2) -- its own instantiation
// (ISA-padded fold, two waves per env), so that the default kernels carry none of its f64 code.
//
// NW > 1 -- speculative batches.  What policy() computes at a node (the Newton solve and the drawn edge: the uniform is
// rands[b, node]) depends on that node alone, not on how the descent got there.  So the NW waves of an env (one per
// SIMD) evaluate, at the same time, the current node and its most likely continuation u1 = fav[u0], u2 = fav[u1], ...
// (fav[t] = t's most visited child, kept up to date below); the descent then follows the drawn edges through the results
// for as long as they match the guesses, and starts the next batch at the first node that was not guessed.  A guess is
// only ever a hint: every level's result is the exact evaluation of the node the descent is at.  Measured on the bench
// workload (MI355X, 9x9, 4096 envs x 64 sims): one wave per env 64 us per launch, two waves 57 us, four waves 82 us --
// 16384 waves no longer fit the chip's 8192 wave slots -- and guessing from the first level on beats waiting for the
// descent to get deep (`deep_thresh` 0 / 3 / 5 / 8: 57 / 62 / 65 / 69 us): a wasted guess costs nothing that matters,
// the kernel is bound by the latency of its longest descent, not by VALU throughput.
// ------------------------------------------------------------------------------------------------------------------
// The one-wave-per-env instantiation (16384 envs and up: bound by VALU issue) asks for SEVEN: with eight the compiler has 78 SGPRs
// and spills 28 of them -- v_writelane / v_readlane pairs, i.e. VALU instructions in the regime where those are what is short --,
// with seven 94 and 8 spills, with six 102 and none; 32768 envs: 185.8 / 182.9 / 187.2 us, 16384: 117.8 / 113.7 / 114.7
// (profiles/r06_expand_occupancy.txt).
#ifndef BLX_OCC1
#define BLX_OCC1 7
#endif
#ifndef BLX_PICK1
#define BLX_PICK1 1
#endif
#ifndef BLX_LATE_ARGS
#define BLX_LATE_ARGS 1
#endif
// sim_expand2_kernel's argument list as the kernarg segment lays it out (every argument at its natural alignment, in order).  KEEP IN STEP
// with the kernel's signature: the expansion tail reads its pointers through this view (BLX_LATE_ARGS below); a mismatch shows up as
// wrong leaves in every parity test, not as a compile error.
struct ExpandArgs {
    Search s; int sim; const uint16_t* rands; int16_t* leaves_out; void* obs_out; uint8_t* valid_out; int32_t* leaf_seats_out;
    unsigned long long* counters; int deep_thresh;
};
template <int RMAX, int KT, bool FAST, bool COUNT, int NW, bool POWF = false>
// 8 waves per SIMD for boards up to 9x9: 4096 envs x 2 waves are the chip's 8192 wave slots, and without the bound the
// kernel's 106 SGPRs admit 6 (a quarter of the envs would start only when others have finished)
__global__ void __launch_bounds__(BL_WAVE * NW, (RMAX <= 3 ? (NW == 1 ? BLX_OCC1 : 8) : 4)) sim_expand2_kernel(Search s, int sim, const uint16_t* rands, int16_t* leaves_out,
                                                                   void* obs_out, uint8_t* valid_out, int32_t* leaf_seats_out,
                                                                   unsigned long long* counters, int deep_thresh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int res[2][NW][4];
    uint8_t* cells = (uint8_t*)smem;
    const int S = s.S, A = S * S, T = s.T;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int slot = blockIdx.x;
    const int b = s.order ? s.order[slot] : slot;
    if (b >= active_envs(s)) return;                 // the whole workgroup: envs that sit this search out (bl_search_t.n_active)
    const long envbase = (long)b * T;
    const bool lowhalf = lane < 32;
    const int el = lane & 31;
    int16_t* path = s.path ? s.path + (long)b * (T + 2) : nullptr;
    // Which descent of which env goes deep cannot be told in advance (most envs own a long most-visited line, and a
    // descent's length is mostly decided by where it leaves that line), so every env keeps its helper waves; they sleep
    // at the batch barrier and guesses are evaluated only from level `deep_thresh` on, where a descent that is still
    // going is likely to go on -- the short majority never pays for a wasted guess.
    const int E = NW;
    if (s.prio_thresh > 0 && path) {
        // s_setprio ignores EXEC: the condition has to be an SGPR compare (readfirstlane), not a divergent branch
        if (__builtin_amdgcn_readfirstlane((int)path[0]) >= s.prio_thresh) __builtin_amdgcn_s_setprio(3);
    }
    long long tk0 = 0, tk1 = 0, tsetup = 0, tterms = 0, tfold = 0, tupd = 0;
    if (COUNT) tk0 = clock64();

    // ---- the env's node slots, lane t <-> slot kt*64 + t
    uint32_t wp[KT];      // w[b,t,0] | w[b,t,1] << 16
    int nn[KT];           // n[b,t]
    int info[KT];         // nk | seat << 16 | terminal << 17
    int rd[KT];           // rand[b,t] (f16 bits)
    int fav[KT];          // most visited child of slot t, or -1
    // every load of the prologue goes out before anything waits: the q-range slots and c_puct first (they depend on nothing), then the
    // node slots -- ONE memory round trip instead of three in a row at the start of every workgroup
    const uint2 qwords = qrange_words(s.qrange + (long)BL_QWORDS * sim);
    const uint16_t cpuct_bits = s.c_puct[b];
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const int tt = kt * 64 + lane;
        wp[kt] = 0; nn[kt] = 0; info[kt] = 0; rd[kt] = 0; fav[kt] = -1;
        if (tt < T) {
            wp[kt] = *(const uint32_t*)(s.w + (envbase + tt) * 2);
            nn[kt] = s.n[envbase + tt];
            info[kt] = (int)(uint16_t)s.nk[envbase + tt] | ((s.seats[envbase + tt] & 1) << 16) | ((s.terminal[envbase + tt] ? 1 : 0) << 17);
            rd[kt] = rands[envbase + tt];
            if (NW > 1) fav[kt] = s.fav[envbase + tt];
        }
    }
    float lo, hi;
    qrange_reduce(qwords, lo, hi);
    const float rden = hi - lo + 1.e-4f;
    const float cpuct = h2f(cpuct_bits);
    // transition_q (cuda.cu:101-105) of every node slot, both seats, once per launch: lane t normalises its own slot's
    // w/(n + 1e-4); a level then fetches a child's q with one bpermute instead of dividing twice per level
    uint32_t qp[KT];      // q[b,t,0] | q[b,t,1] << 16  (f16 bits)
#pragma unroll
    for (int kt = 0; kt < KT; kt++) {
        const float den = (float)nn[kt] + 1.e-4f;
        const float q0 = h2f((uint16_t)wp[kt]) / den, q1 = h2f((uint16_t)(wp[kt] >> 16)) / den;
        qp[kt] = (uint32_t)f2h((q0 - lo) / rden) | ((uint32_t)f2h((q1 - lo) / rden) << 16);
    }

    auto pick = [&](const int (&regs)[KT], int t) {       // regs[t], t wave-uniform
        if constexpr (KT == 1 && BLX_PICK1) return __builtin_amdgcn_readlane(regs[0], t & 63);      // T <= 64: no test, no branch around the readlane
        int v = 0;
#pragma unroll
        for (int kt = 0; kt < KT; kt++) if ((t >> 6) == kt) v = __builtin_amdgcn_readlane(regs[kt], t & 63);
        return __builtin_amdgcn_readfirstlane(v);
    };

    // policy() + the draw at node `t` (cuda.cu:70-99, 35-68, 157-176): returns the drawn action (-1: none has positive
    // probability), its child slot (-1: not expanded) and its index in t's compacted row.
    auto evaluate = [&](const int t, const int tinfo, int& action_o, int& child_o, int& sel_o, int& iters_o, int& nch_o) {
        long long tp0 = 0;
        if (COUNT) tp0 = clock64();
        const int nk = tinfo & 0xffff, seat = (tinfo >> 16) & 1;
        const int R = (nk + 31) >> 5;
        const float rnd = h2f((uint16_t)pick(rd, t));
        const long row = (envbase + t) * A;

        // element e = 32 r + (lane & 31) of the node's kept actions, in both halves of the wave
        float top[RMAX], q[RMAX], term[RMAX], x[RMAX];
        uint32_t cc[RMAX];
        bool in[RMAX];
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            const int e = 32 * r + el;
            in[r] = (r < R) && (e < nk);
            top[r] = 0.f; cc[r] = 0xffff0000u; q[r] = 0.f; term[r] = 0.f; x[r] = 0.f;
            if (in[r]) { top[r] = s.cpi[row + e]; cc[r] = s.cca[row + e]; }
        }
        int Nloc = 0, nch = 0;
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            if (r < R) {                                                  // wave-uniform: the bpermutes run with every lane enabled
                const int c = (int)(int16_t)(cc[r] >> 16);
                const bool ex = c >= 0;
                const int src = ex ? c : 0;
                uint32_t q2 = 0; int nv = 0;
#pragma unroll
                for (int kt = 0; kt < KT; kt++) {
                    const uint32_t a_ = (uint32_t)bperm_i(src & 63, (int)qp[kt]);
                    const int b_ = bperm_i(src & 63, nn[kt]);
                    if ((src >> 6) == kt) { q2 = a_; nv = b_; }
                }
                if (ex) q[r] = h2f((uint16_t)(seat ? (q2 >> 16) : q2));
                if (lowhalf && in[r]) { Nloc += ex ? nv : 1; nch += ex ? 1 : 0; }
            }
        }
        const int N = wave_sum_i32(Nloc) + (A - nk);                      // dropped actions are unexpanded: +1 each
        const float lam = (cpuct * (float)N) / (float)(unsigned)(N + A);
        float alpha = (nk < A) ? 1.e-4f : 0.f;                            // a dropped action's q + max(lambda pi, 1e-4)
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            top[r] = lam * top[r];
            if (in[r]) alpha = fmaxf(alpha, q[r] + fmaxf(top[r], 1.e-4f));
        }
        alpha = wave_max_f32(alpha);
        if (COUNT) tsetup += clock64() - tp0;

        // newton_search, cuda.cu:35-68.  The iteration body is instantiated per block count (`RR` = R, fixed for the call): with
        // the blocks behind run-time branches each block's IEEE division was a basic block of its own, one dependent chain
        // after the other; as straight-line code the chains of all blocks interleave, and block 0's fold starts while the
        // later blocks' quotients are still in flight.
        float err = INFINITY;
        int iters = 0;
        const int last_e = nk - 1, rl = last_e >> 5;
        const int laneS = (last_e & 31) + ((rl & 1) ? 32 : 0), laneG = (last_e & 31) + ((rl & 1) ? 0 : 32);
        auto newton = [&](auto rr_c) __attribute__((always_inline)) {
            constexpr int RR = decltype(rr_c)::value;
            for (int it = 0; it < 101 && nk > 0; it++) {
                long long ti0 = 0, ti1 = 0, ti2 = 0;
                if (COUNT) ti0 = clock64();
                // prob(a), cuda.cu:23-25, resp. its derivative term.  No `in[r] ?` on the quotient: lanes beyond the row's end hold
                // top = 0, q = 0, so they get +-0 / alpha^k -- and nothing ever reads them (prefix sums only flow upwards, the totals
                // are read at the last kept action, the draw tests in[r]); a select would put every quotient behind its own EXEC
                // branch, one dependent chain after the other.
                float num[RR], den[RR], quo[RR];
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const bool isS = lowhalf != ((r & 1) != 0);
                    const float bot = alpha - q[r];
                    num[r] = isS ? top[r] : -top[r];
                    den[r] = isS ? bot : (POWF ? g_denominator(bot, 1) : bot * bot);
                }
                ieee_div_n<RR>(num, den, quo);
#pragma unroll
                for (int r = 0; r < RR; r++) term[r] = quo[r];
                if (COUNT) { ti1 = clock64(); tterms += ti1 - ti0; }
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    x[r] = term[r];
                    if (r == 0) { if (el == 0) x[0] = 0.f + x[0]; }          // the sums start from 0.f (cuda.cu:44): (+0) + (-0) = +0
                    else fold_carry<FAST>(x[r], x[r - 1 < 0 ? 0 : r - 1], term[r]);
                    fold_block<FAST>(x[r], term[r], r + 1 < RR ? 32 : nk - 32 * r);
                }
                const float Ssum = readlane_f(x[RR - 1], laneS), gsum_ = readlane_f(x[RR - 1], laneG);      // rl == RR - 1
                if (COUNT) { ti2 = clock64(); tfold += ti2 - ti1; }
                if (it == 100) break;      // alpha moved after the 100th fold (cuda.cu:48-65): this pass only refreshed the terms
                iters++;
                const float ne = Ssum - 1.f;
                if ((ne < 1e-3f) || (err == ne)) break;
                alpha -= ne / gsum_; err = ne;
                if (COUNT) tupd += clock64() - ti2;
            }
        };
        if constexpr (RMAX <= 6) {
            if (R <= 1) newton(std::integral_constant<int, 1>{});
            else if (R == 2) newton(std::integral_constant<int, RMAX >= 2 ? 2 : 1>{});
            else if (R == 3) newton(std::integral_constant<int, RMAX >= 3 ? 3 : 1>{});
            else if (R == 4) newton(std::integral_constant<int, RMAX >= 4 ? 4 : 1>{});
            else if (R == 5) newton(std::integral_constant<int, RMAX >= 5 ? 5 : 1>{});
            else newton(std::integral_constant<int, RMAX >= 6 ? 6 : 1>{});
        } else {
        for (int it = 0; it < 101 && nk > 0; it++) {
            long long ti0 = 0, ti1 = 0, ti2 = 0;
            if (COUNT) ti0 = clock64();
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                if (r < R) {
                    const bool isS = lowhalf != ((r & 1) != 0);
                    const float bot = alpha - q[r];
                    const float num = isS ? top[r] : -top[r];
                    const float den = isS ? bot : (POWF ? g_denominator(bot, 1) : bot * bot);
                    term[r] = in[r] ? num / den : 0.f;                    // prob(a), cuda.cu:23-25, resp. its derivative term
                }
            }
            if (COUNT) { ti1 = clock64(); tterms += ti1 - ti0; }
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                if (r < R) {
                    x[r] = term[r];
                    if (r == 0) { if (el == 0) x[0] = 0.f + x[0]; }          // the sums start from 0.f (cuda.cu:44): (+0) + (-0) = +0
                    else fold_carry<FAST>(x[r], x[r - 1 < 0 ? 0 : r - 1], term[r]);
                    fold_block<FAST>(x[r], term[r], nk - 32 * r);
                }
            }
            float Ssum = 0.f, gsum_ = 0.f;
#pragma unroll
            for (int r = 0; r < RMAX; r++) if (r == rl) { Ssum = readlane_f(x[r], laneS); gsum_ = readlane_f(x[r], laneG); }
            if (COUNT) { ti2 = clock64(); tfold += ti2 - ti1; }
            if (it == 100) break;      // alpha moved after the 100th fold (cuda.cu:48-65): this pass only refreshed the terms
            iters++;
            const float ne = Ssum - 1.f;
            if ((ne < 1e-3f) || (err == ne)) break;
            alpha -= ne / gsum_; err = ne;
            if (COUNT) tupd += clock64() - ti2;
        }
        }

        // the draw, cuda.cu:157-176: first kept action (ascending) with prob > 0 and running total >= rand, else the last
        // with prob > 0.  The S half of block r holds prob in term[r] and the running totals in x[r].
        int sel_r = -1, sel_lane = 0, last_r = -1, last_lane = 0;
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            if (r < R) {
                const bool isS = lowhalf != ((r & 1) != 0);
                const bool pos = in[r] && isS && term[r] > 0.f;
                // (ballot_w64 on the compare itself: __ballot(bool) goes through a 0/1 VGPR and a second compare)
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(pos && x[r] >= rnd), anyp = __builtin_amdgcn_ballot_w64(pos);
                if (sel_r < 0 && hit) { sel_r = r; sel_lane = __builtin_ctzll(hit); }
                if (anyp) { last_r = r; last_lane = 63 - __builtin_clzll(anyp); }
            }
        }
        if (sel_r < 0) { sel_r = last_r; sel_lane = last_lane; }
        iters_o = iters;
        nch_o = COUNT ? wave_sum_i32(nch) : 0;
        action_o = -1; child_o = -1; sel_o = 0;
        if (sel_r >= 0) {
            uint32_t ccs = 0;
#pragma unroll
            for (int r = 0; r < RMAX; r++) if (r == sel_r) ccs = (uint32_t)__builtin_amdgcn_readlane((int)cc[r], sel_lane);
            action_o = (int)(ccs & 0xffffu);
            sel_o = 32 * sel_r + (sel_lane & 31);
            child_o = __builtin_amdgcn_readfirstlane((int)(int16_t)(ccs >> 16));
        }
    };

    // ---- descend_kernel's loop, cuda.cu:138-182, a batch of up to E guessed levels at a time
    int t = 0, parent = 0, action = -1, nlev = 0, sel_e = 0, evals = 0;
    int tinfo = pick(info, 0);
    bool live = true;
    for (int batch = 0; batch < T; batch++) {
        if (!live || t == -1 || ((tinfo >> 17) & 1) || nlev >= T) break;
        // the nodes of this batch: the current one and its guessed continuation
        int u[NW], uinfo[NW];
        u[0] = t; uinfo[0] = tinfo;
#pragma unroll
        for (int k = 1; k < NW; k++) {
            u[k] = -1; uinfo[k] = 0;
            if (nlev >= deep_thresh && u[k - 1] != -1 && !((uinfo[k - 1] >> 17) & 1)) {
                u[k] = pick(fav, u[k - 1]);
                if (u[k] != -1) uinfo[k] = pick(info, u[k]);
            }
        }
        int my = -1, myinfo = 0;
#pragma unroll
        for (int k = 0; k < NW; k++) if (wave == k) { my = u[k]; myinfo = uinfo[k]; }
        int ra = -2, rc = -1, rs = 0, ri = 0, rn = 0;
        if (my != -1 && !((myinfo >> 17) & 1)) { evaluate(my, myinfo, ra, rc, rs, ri, rn); evals++; }
        if (NW > 1 && E > 1) {
            if (lane == 0) { res[batch & 1][wave][0] = ra; res[batch & 1][wave][1] = rc; res[batch & 1][wave][2] = rs; res[batch & 1][wave][3] = ri | (rn << 8); }
            __syncthreads();
        }
        // follow the drawn edges through the batch
#pragma unroll
        for (int k = 0; k < NW; k++) {
            if (k < E) {
                int a_k = ra, c_k = rc, s_k = rs, in_k = ri | (rn << 8);
                if (NW > 1 && E > 1) {
                    a_k = __builtin_amdgcn_readfirstlane(res[batch & 1][k][0]); c_k = __builtin_amdgcn_readfirstlane(res[batch & 1][k][1]);
                    s_k = __builtin_amdgcn_readfirstlane(res[batch & 1][k][2]); in_k = __builtin_amdgcn_readfirstlane(res[batch & 1][k][3]);
                }
                const int node = u[k];
                if (path && wave == 0 && lane == 0) path[1 + nlev] = (int16_t)node;
                nlev++;
                parent = node; sel_e = s_k;
                if (COUNT && wave == 0 && lane == 0) {
                    unsigned long long* e = counters + 12 * (long)b;
                    const int iters = in_k & 0xff;
                    e[0] += 1; e[1] += iters; if ((unsigned long long)iters > e[2]) e[2] = iters; e[3] += in_k >> 8;
                }
                if (a_k < 0) { action = -1; live = false; break; }       // no action with positive probability: the reference would index [-1]
                action = a_k;
                if (NW > 1) {
                    // node's most visited child once this descent is backed up: the drawn child gains a visit (n += 2)
                    const int cnew = c_k == -1 ? sim : c_k;
                    const int f_old = pick(fav, node);
                    if (f_old == -1 || pick(nn, cnew) + 2 >= pick(nn, f_old)) {
#pragma unroll
                        for (int kt = 0; kt < KT; kt++) if ((node >> 6) == kt && lane == (node & 63)) fav[kt] = cnew;
                    }
                }
                t = c_k;
                if (t == -1) break;
                tinfo = pick(info, t);
                if ((tinfo >> 17) & 1) break;
                if (!(k + 1 < E && k + 1 < NW && u[k + 1 < NW ? k + 1 : 0] == t)) break;     // the guess ends here: next batch starts at t
            }
        }
    }
    if (NW > 1 && wave > 0) return;
    if (action < 0) action = 0;
    if (COUNT) tk1 = clock64();
    if (NW > 1) {
#pragma unroll
        for (int kt = 0; kt < KT; kt++) if (kt * 64 + lane < T) s.fav[envbase + kt * 64 + lane] = (int16_t)fav[kt];
    }
#if BLX_LATE_ARGS
    // The arguments only the expansion below needs (9 pointers of the search and the 4 outputs: 26 SGPRs that would be live -- or spilled
    // to VGPR lanes and read back -- across the whole descent) are read from the kernarg segment HERE, through a pointer the compiler
    // cannot see through; the names shadow the kernel's parameters, whose own loads are then dead.
    const __attribute__((address_space(4))) ExpandArgs* late = (const __attribute__((address_space(4))) ExpandArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("; expansion arguments from %0" : "+s"(late));
    {                                      // (closed at the end of the kernel)
    Search s;
    s.logits = late->s.logits; s.children = late->s.children; s.parents = late->s.parents; s.relation = late->s.relation; s.rewards = late->s.rewards;
    s.terminal = late->s.terminal; s.boards = late->s.boards; s.seats = late->s.seats; s.cca = late->s.cca; s.obs_f16 = late->s.obs_f16; s.lazy = late->s.lazy;
    int16_t* const leaves_out = late->leaves_out; void* const obs_out = late->obs_out; uint8_t* const valid_out = late->valid_out;
    int32_t* const leaf_seats_out = late->leaf_seats_out;
#endif

    // ---- leaves = children[envs, parents, actions]; leaves[leaves == -1] = sim   (mcts/__init__.py:117-122)
    const int nxt = t;
    const int leaf = (nxt == -1) ? sim : nxt;
    if (s.lazy) lazy_slot_reset(s, envbase, sim, A, nxt == -1, lane);
    if (lane == 0) {
        s.children[(envbase + parent) * A + action] = (int16_t)leaf;
        s.parents[envbase + leaf] = (int16_t)parent;
        s.relation[envbase + leaf] = (int16_t)action;
        if (nxt == -1 && nlev > 0) ((uint16_t*)(s.cca + (envbase + parent) * A + sel_e))[1] = (uint16_t)leaf;   // the compacted row's child field
    }
    const int seat = s.seats[envbase + parent];
    const uint8_t* src = s.boards + (envbase + parent) * A;
    // (A <= 32 RMAX: a compile-time number of cells per lane, so that a lane's loads are all in flight before the first is awaited --
    // as `for (a = lane; a < A; a += 64)` the loop ran load, wait, write once per 64 cells: a second memory round trip for 9x9 on
    // every env's tail)
    constexpr int CPL = (32 * RMAX + 63) / 64;
    {
        uint8_t c[CPL];
#pragma unroll
        for (int i = 0; i < CPL; i++) c[i] = lane + 64 * i < A ? src[lane + 64 * i] : (uint8_t)0;
#pragma unroll
        for (int i = 0; i < CPL; i++) if (lane + 64 * i < A) cells[lane + 64 * i] = c[i];
    }
    __syncthreads();
    const int win = hex_step_wave<(RMAX + 1) / 2>(cells, S, __builtin_amdgcn_readfirstlane(seat), __builtin_amdgcn_readfirstlane(action), lane);     // one wave is left: the flood as a bit-board fill
    // Hex.step tail, hex/__init__.py:183-190
    const bool term = win != 0;
    const int new_seat = term ? 0 : 1 - seat;
    uint8_t* dst = s.boards + (envbase + leaf) * A;
    const float invS = 1.0f / (float)S;
    const bool flip = new_seat == 1;
    {
        uint8_t own[CPL], seen[CPL];                     // the cell itself (the stored board) and the cell the mover's frame shows there
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int a = lane + 64 * k;
            const int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
            own[k] = a < A ? cells[a] : (uint8_t)0;
            seen[k] = a < A ? cells[flip ? j * S + i : a] : (uint8_t)0;
        }
#pragma unroll
        for (int k = 0; k < CPL; k++) {
            const int a = lane + 64 * k;
            if (a < A) {
                dst[a] = term ? (uint8_t)0 : own[k];
                const int color = term ? 2 : color_of(seen[k]);
                const int ch = color < 2 ? (flip ? 1 - color : color) : 2;
                if (s.obs_f16) ((uint32_t*)obs_out)[(long)b * A + a] = ch == 0 ? 0x00003c00u : (ch == 1 ? 0x3c000000u : 0u);   // f16 1.0 = 0x3c00
                else ((float2*)obs_out)[(long)b * A + a] = make_float2(ch == 0 ? 1.f : 0.f, ch == 1 ? 1.f : 0.f);
                valid_out[(long)b * A + a] = color == 2;
            }
        }
    }
    if (lane == 0) {
        s.seats[envbase + leaf] = new_seat;
        s.terminal[envbase + leaf] = term;
        s.rewards[(envbase + leaf) * 2 + 0] = f2h((float)win);
        s.rewards[(envbase + leaf) * 2 + 1] = f2h((float)(-win));
        leaves_out[b] = (int16_t)leaf;
        leaf_seats_out[b] = new_seat;
        if (path) {
            path[1 + nlev] = (int16_t)leaf;
            path[0] = (int16_t)(nlev + 1);
        }
    }
    if (COUNT && lane == 0) {
        unsigned long long* e = counters + 12 * (long)b;
        const long long tk2 = clock64();
        e[4] += tsetup; e[5] += tterms; e[6] += tfold; e[7] += tupd;
        e[8] += tk1 - tk0; e[9] += tk2 - tk1; e[10] += evals;
    }
#if BLX_LATE_ARGS
    }
#endif
}

// (Round 4's shared-workgroup kernel -- several envs per workgroup, the waves of finished descents helping the ones still going --
// was bit-exact and slower at every batch size (profiles/r04_shared_wg_*.txt, HISTORY.md); it was removed in round 5: its
// protocol trapped on an exhausted poll budget, which aborts the HIP context, and nothing used it.  bl_tune_t.expand_envs > 1 is
// now BL_EINVAL.)

// ------------------------------------------------------------------------------------------------------------------
// Two nodes per wave.  While every env still descends, the kernel is bound by VALU issue (4.3 k VALU instructions per env,
// 8 waves per SIMD), and more than half of those are the folds' v_add_f32_dpp -- whose count depends on the chain length, not
// on how many chains a step advances.  Here ONE wave per env evaluates the current node in lanes 0..31 and its guessed
// continuation in lanes 32..63: rows 0/1 carry node A's S/g chains, rows 2/3 node B's, in blocks of 16 kept actions; a step
// still is one `row_shr:1` instruction, now advancing four chains, and a block hands over with one `row_ror:1` (lane 15 -> lane
// 0 of every row).  Same arithmetic per element, same order per chain: bit-identical results.  Per batch of two nodes the
// folds cost one wave's 64 steps instead of two waves' 64 each, nothing is exchanged through LDS and no barrier is needed.
// RB: blocks of 16 kept actions (ceil(A / 16)); T <= 64.
// ------------------------------------------------------------------------------------------------------------------
template <bool FAST>
__device__ __forceinline__ void fold_ror(float& x, const float xprev, const float t) {     // lane 0 of every row <- xprev[15 of the row] + t
    if constexpr (FAST) asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0x1\n\ts_nop 0\n\t" : "+v"(x) : "v"(xprev), "v"(t));
    else asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 row_ror:1 row_mask:0xf bank_mask:0x1\n\ts_nop 1\n\t" : "+v"(x) : "v"(xprev), "v"(t));
}
__device__ __forceinline__ int row_sum_i32(int v) {        // lane 15 of every row ends with the row's sum
    v += dpp_i<0x111, 0xf>(0, v); v += dpp_i<0x112, 0xf>(0, v); v += dpp_i<0x114, 0xf>(0, v); v += dpp_i<0x118, 0xf>(0, v);
    return v;
}
__device__ __forceinline__ float row_max_f32(float v) {    // lane 15 of every row ends with the row's max
    v = fmaxf(v, dpp_f<0x111, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x112, 0xf>(v, v)); v = fmaxf(v, dpp_f<0x114, 0xf>(v, v));
    v = fmaxf(v, dpp_f<0x118, 0xf>(v, v));
    return v;
}

template <int RB, bool FAST>
__global__ void __launch_bounds__(BL_WAVE) sim_expand3_kernel(Search s, int sim, const uint16_t* rands, int16_t* leaves_out,
                                                              void* obs_out, uint8_t* valid_out, int32_t* leaf_seats_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* cells = (uint8_t*)smem;
    const int S = s.S, A = S * S, T = s.T;
    const int lane = threadIdx.x & 63;
    const int b = s.order ? s.order[blockIdx.x] : blockIdx.x;
    if (b >= active_envs(s)) return;
    const long envbase = (long)b * T;
    const bool hiNode = lane >= 32;                 // which of the batch's two nodes this lane works for
    const bool isS = !((lane >> 4) & 1);            // rows 0, 2: S chain (and prob); rows 1, 3: g chain
    const int el = lane & 15;
    int16_t* path = s.path ? s.path + (long)b * (T + 2) : nullptr;

    // ---- the env's node slots, lane t <-> slot t (T <= 64)
    uint32_t wp = 0; int nn = 0, info = 0, rd = 0, fav = -1;
    if (lane < T) {
        wp = *(const uint32_t*)(s.w + (envbase + lane) * 2);
        nn = s.n[envbase + lane];
        info = (int)(uint16_t)s.nk[envbase + lane] | ((s.seats[envbase + lane] & 1) << 16) | ((s.terminal[envbase + lane] ? 1 : 0) << 17);
        rd = rands[envbase + lane];
        fav = s.fav[envbase + lane];
    }
    float lo, hi;
    load_qrange(s.qrange + (long)BL_QWORDS * sim, lo, hi);
    const float rden = hi - lo + 1.e-4f;
    const float cpuct = h2f(s.c_puct[b]);
    uint32_t qp;                                    // transition_q (cuda.cu:101-105) of slot `lane`, both seats, f16 bits
    {
        const float den = (float)nn + 1.e-4f;
        const float q0 = h2f((uint16_t)wp) / den, q1 = h2f((uint16_t)(wp >> 16)) / den;
        qp = (uint32_t)f2h((q0 - lo) / rden) | ((uint32_t)f2h((q1 - lo) / rden) << 16);
    }
    auto pick = [&](int reg, int t) { return __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(reg, t & 63)); };

    // policy() + the draw (cuda.cu:70-99, 35-68, 157-176) at nodes tA (lanes 0..31) and tB (lanes 32..63; -1: none).
    // res[k] = {action (-1: none with positive probability; -2: node not evaluated), child slot (-1: not expanded), index
    // in the compacted row}
    auto evaluate2 = [&](const int tA, const int infoA, const int tB, const int infoB, int (&res)[2][3]) {
        const bool evB = tB >= 0 && !((infoB >> 17) & 1);
        const int nkA = infoA & 0xffff, nkB = evB ? (infoB & 0xffff) : 0;
        const int nk = hiNode ? nkB : nkA;
        const int seat = ((hiNode ? infoB : infoA) >> 16) & 1;
        const int tmine = hiNode ? (evB ? tB : 0) : tA;
        const int nkmax = nkA > nkB ? nkA : nkB;
        const int R = (nkmax + 15) >> 4;
        const float rnd = h2f((uint16_t)(hiNode ? pick(rd, evB ? tB : 0) : pick(rd, tA)));
        const long row = (envbase + tmine) * A;

        float top[RB], q[RB], term[RB], x[RB];
        uint32_t cc[RB];
        bool in[RB];
#pragma unroll
        for (int r = 0; r < RB; r++) {
            const int e = 16 * r + el;
            in[r] = (r < R) && (e < nk);
            top[r] = 0.f; cc[r] = 0xffff0000u; q[r] = 0.f; term[r] = 0.f; x[r] = 0.f;
            if (in[r]) { top[r] = s.cpi[row + e]; cc[r] = s.cca[row + e]; }
        }
        int Nloc = 0;
#pragma unroll
        for (int r = 0; r < RB; r++) {
            if (r < R) {                                                  // wave-uniform: the bpermutes run with every lane enabled
                const int c = (int)(int16_t)(cc[r] >> 16);
                const bool ex = c >= 0;
                const int src = ex ? c : 0;
                const uint32_t q2 = (uint32_t)bperm_i(src & 63, (int)qp);
                const int nv = bperm_i(src & 63, nn);
                if (ex) q[r] = h2f((uint16_t)(seat ? (q2 >> 16) : q2));
                if (isS && in[r]) Nloc += ex ? nv : 1;
            }
        }
        Nloc = row_sum_i32(Nloc);                                         // rows 0 and 2 hold the two nodes' sums (rows 1, 3: zero)
        const int NA = __builtin_amdgcn_readlane(Nloc, 15) + (A - nkA), NB = __builtin_amdgcn_readlane(Nloc, 47) + (A - nkB);
        const int N = hiNode ? NB : NA;                                   // dropped actions are unexpanded: +1 each
        const float lam = (cpuct * (float)N) / (float)(unsigned)(N + A);
        float alpha = (nk < A) ? 1.e-4f : 0.f;                            // a dropped action's q + max(lambda pi, 1e-4)
#pragma unroll
        for (int r = 0; r < RB; r++) {
            top[r] = lam * top[r];
            if (in[r]) alpha = fmaxf(alpha, q[r] + fmaxf(top[r], 1.e-4f));
        }
        alpha = row_max_f32(alpha);                                       // both rows of a node hold the same elements
        {
            const float aA = readlane_f(alpha, 15), aB = readlane_f(alpha, 47);
            alpha = hiNode ? aB : aA;
        }

        // newton_search, cuda.cu:35-68, both nodes in step; a node that has converged keeps its alpha (its terms and totals are
        // then recomputed unchanged) until the other one has, too
        bool doneA = nkA == 0, doneB = nkB == 0;
        float errA = INFINITY, errB = INFINITY;
        const int lastA = nkA - 1, lastB = nkB - 1;
        const int rlA = lastA >> 4, rlB = lastB >> 4;
        auto newton = [&](auto rr_c) __attribute__((always_inline)) {      // body per block count, quotients side by side: see sim_expand2_kernel
            constexpr int RR = decltype(rr_c)::value;
            for (int it = 0; it < 101 && !(doneA && doneB); it++) {
                float num[RR], den[RR], quo[RR];
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    const float bot = alpha - q[r];
                    num[r] = isS ? top[r] : -top[r];
                    den[r] = isS ? bot : bot * bot;
                }
                ieee_div_n<RR>(num, den, quo);                            // prob(a), cuda.cu:23-25, resp. its derivative term
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    term[r] = quo[r];
                    x[r] = term[r];
                    if (r == 0) { if (el == 0) x[0] = 0.f + x[0]; }          // the sums start from 0.f (cuda.cu:44): (+0) + (-0) = +0
                    else fold_ror<FAST>(x[r], x[r - 1 < 0 ? 0 : r - 1], term[r]);
                    fold7<FAST>(x[r], term[r]);
                    if (r + 1 < RR || nkmax - 16 * r > 8) fold8<FAST>(x[r], term[r]);
                }
                float SA = 0.f, gA = 0.f, SB = 0.f, gB = 0.f;
#pragma unroll
                for (int r = 0; r < RR; r++) {
                    if (r == rlA) { SA = readlane_f(x[r], lastA & 15); gA = readlane_f(x[r], 16 + (lastA & 15)); }
                    if (r == rlB) { SB = readlane_f(x[r], 32 + (lastB & 15)); gB = readlane_f(x[r], 48 + (lastB & 15)); }
                }
                if (it == 100) break;      // alpha moved after the 100th fold (cuda.cu:48-65): this pass only refreshed the terms
                const float neA = SA - 1.f, neB = SB - 1.f;
                if (!doneA && ((neA < 1e-3f) || (errA == neA))) doneA = true;
                if (!doneB && ((neB < 1e-3f) || (errB == neB))) doneB = true;
                const float ne = hiNode ? neB : neA, gs = hiNode ? gB : gA;
                const bool frozen = hiNode ? doneB : doneA;
                const float step = ne / gs;
                if (!frozen) alpha -= step;
                if (!doneA) errA = neA;
                if (!doneB) errB = neB;
            }
        };
        if (R <= 1) newton(std::integral_constant<int, 1>{});
        else if (R == 2) newton(std::integral_constant<int, RB >= 2 ? 2 : 1>{});
        else if (R == 3) newton(std::integral_constant<int, RB >= 3 ? 3 : 1>{});
        else if (R == 4) newton(std::integral_constant<int, RB >= 4 ? 4 : 1>{});
        else if (R == 5) newton(std::integral_constant<int, RB >= 5 ? 5 : 1>{});
        else newton(std::integral_constant<int, RB >= 6 ? 6 : 1>{});

        // the draw, cuda.cu:157-176: first kept action (ascending) with prob > 0 and running total >= rand, else the last
        // with prob > 0.  Rows 0 / 2 hold prob in term[r] and the running totals in x[r].
        int selr[2] = {-1, -1}, sell[2] = {0, 0}, lastr[2] = {-1, -1}, lastl[2] = {0, 0};
#pragma unroll
        for (int r = 0; r < RB; r++) {
            if (r < R) {
                const bool pos = in[r] && isS && term[r] > 0.f;
                // (ballot_w64 on the compare itself: __ballot(bool) goes through a 0/1 VGPR and a second compare)
                const unsigned long long hit = __builtin_amdgcn_ballot_w64(pos && x[r] >= rnd), anyp = __builtin_amdgcn_ballot_w64(pos);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t h = (uint32_t)(hit >> (32 * k)) & 0xffffu, p = (uint32_t)(anyp >> (32 * k)) & 0xffffu;
                    if (selr[k] < 0 && h) { selr[k] = r; sell[k] = 32 * k + __builtin_ctz(h); }
                    if (p) { lastr[k] = r; lastl[k] = 32 * k + 31 - __builtin_clz(p); }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (selr[k] < 0) { selr[k] = lastr[k]; sell[k] = lastl[k]; }
            res[k][0] = -1; res[k][1] = -1; res[k][2] = 0;
            if (selr[k] >= 0) {
                uint32_t ccs = 0;
#pragma unroll
                for (int r = 0; r < RB; r++) if (r == selr[k]) ccs = (uint32_t)__builtin_amdgcn_readlane((int)cc[r], sell[k]);
                res[k][0] = (int)(ccs & 0xffffu);
                res[k][2] = 16 * selr[k] + (sell[k] & 15);
                res[k][1] = __builtin_amdgcn_readfirstlane((int)(int16_t)(ccs >> 16));
            }
        }
        if (!evB) res[1][0] = -2;
    };

    // ---- descend_kernel's loop, cuda.cu:138-182, two guessed levels at a time
    int t = 0, parent = 0, action = -1, nlev = 0, sel_e = 0;
    int tinfo = pick(info, 0);
    bool live = true;
    for (int batch = 0; batch < T; batch++) {
        if (!live || t == -1 || ((tinfo >> 17) & 1) || nlev >= T) break;
        const int u0 = t, u0info = tinfo;
        int u1 = pick(fav, u0), u1info = 0;
        if (u1 != -1) u1info = pick(info, u1);
        int res[2][3];
        evaluate2(u0, u0info, u1, u1info, res);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int node = k ? u1 : u0;
            if (path && lane == 0) path[1 + nlev] = (int16_t)node;
            nlev++;
            parent = node; sel_e = res[k][2];
            if (res[k][0] < 0) { action = -1; live = false; break; }     // no action with positive probability: the reference would index [-1]
            action = res[k][0];
            {
                // node's most visited child once this descent is backed up: the drawn child gains a visit (n += 2)
                const int cnew = res[k][1] == -1 ? sim : res[k][1];
                const int f_old = pick(fav, node);
                if (f_old == -1 || pick(nn, cnew) + 2 >= pick(nn, f_old)) { if (lane == node) fav = cnew; }
            }
            t = res[k][1];
            if (t == -1) break;
            tinfo = pick(info, t);
            if ((tinfo >> 17) & 1) break;
            if (!(k == 0 && u1 == t && res[1][0] != -2)) break;          // the guess ends here: next batch starts at t
        }
    }
    if (action < 0) action = 0;
    if (lane < T) s.fav[envbase + lane] = (int16_t)fav;

    // ---- leaves = children[envs, parents, actions]; leaves[leaves == -1] = sim   (mcts/__init__.py:117-122); Hex.step + observe
    const int nxt = t;
    const int leaf = (nxt == -1) ? sim : nxt;
    if (s.lazy) lazy_slot_reset(s, envbase, sim, A, nxt == -1, lane);
    if (lane == 0) {
        s.children[(envbase + parent) * A + action] = (int16_t)leaf;
        s.parents[envbase + leaf] = (int16_t)parent;
        s.relation[envbase + leaf] = (int16_t)action;
        if (nxt == -1 && nlev > 0) ((uint16_t*)(s.cca + (envbase + parent) * A + sel_e))[1] = (uint16_t)leaf;   // the compacted row's child field
    }
    const int seat = s.seats[envbase + parent];
    const uint8_t* src = s.boards + (envbase + parent) * A;
    for (int a = lane; a < A; a += 64) cells[a] = src[a];
    __syncthreads();
    const int win = hex_step_group<64>(cells, S, seat, action, true, lane);
    const bool term = win != 0;                                          // Hex.step tail, hex/__init__.py:183-190
    const int new_seat = term ? 0 : 1 - seat;
    uint8_t* dst = s.boards + (envbase + leaf) * A;
    const float invS = 1.0f / (float)S;
    const bool flip = new_seat == 1;
    for (int a = lane; a < A; a += 64) dst[a] = term ? (uint8_t)0 : cells[a];
    for (int a = lane; a < A; a += 64) {
        const int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
        const int color = term ? 2 : color_of(cells[flip ? j * S + i : a]);
        const int ch = color < 2 ? (flip ? 1 - color : color) : 2;
        if (s.obs_f16) ((uint32_t*)obs_out)[(long)b * A + a] = ch == 0 ? 0x00003c00u : (ch == 1 ? 0x3c000000u : 0u);   // f16 1.0 = 0x3c00
        else ((float2*)obs_out)[(long)b * A + a] = make_float2(ch == 0 ? 1.f : 0.f, ch == 1 ? 1.f : 0.f);
        valid_out[(long)b * A + a] = color == 2;
    }
    if (lane == 0) {
        s.seats[envbase + leaf] = new_seat;
        s.terminal[envbase + leaf] = term;
        s.rewards[(envbase + leaf) * 2 + 0] = f2h((float)win);
        s.rewards[(envbase + leaf) * 2 + 1] = f2h((float)(-win));
        leaves_out[b] = (int16_t)leaf;
        leaf_seats_out[b] = new_seat;
        if (path) {
            path[1 + nlev] = (int16_t)leaf;
            path[0] = (int16_t)(nlev + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Self-test of the FAST fold on the device at hand: random positive/negative terms through fold_carry / fold_rows /
// fold_mid exactly as the kernel chains them, against a serial sum by lane 0 through LDS.  out[0] += mismatching totals.
// ------------------------------------------------------------------------------------------------------------------
template <bool FAST>
__global__ void __launch_bounds__(BL_WAVE) fold_selftest_kernel(uint32_t seed, int nk, unsigned int* bad) {
    __shared__ float ts[3][64];
    const int lane = threadIdx.x, el = lane & 31;
    const bool lowhalf = lane < 32;
    uint32_t h = seed * 2654435761u + blockIdx.x * 40503u + 12345u;
    float term[3], x[3];
    const int R = (nk + 31) >> 5;
    for (int r = 0; r < 3; r++) {
        const int e = 32 * r + el;
        const bool isS = lowhalf != ((r & 1) != 0);
        uint32_t k = (h ^ (uint32_t)(e * 2 + (isS ? 0 : 1)) * 0x9E3779B1u); k ^= k >> 15; k *= 0x85EBCA6Bu; k ^= k >> 13;
        const float mag = __builtin_bit_cast(float, 0x3a000000u + (k & 0x03ffffffu));       // ~[5e-4, 8) spread over 7 binades
        term[r] = (e < nk && r < R) ? (isS ? mag : -mag) : 0.f;
        ts[r][lane] = term[r];
        x[r] = 0.f;
    }
    __syncthreads();
    for (int r = 0; r < 3; r++) {
        if (r < R) {
            x[r] = term[r];
            if (r == 0) { if (el == 0) x[0] = 0.f + x[0]; }
            else fold_carry<FAST>(x[r], x[r - 1 < 0 ? 0 : r - 1], term[r]);
            fold_block<FAST>(x[r], term[r], nk - 32 * r);
        }
    }
    // serial reference, lane 0: S elements then g elements, every prefix
    __shared__ float want[2][96];
    if (lane == 0) {
        float aS = 0.f, aG = 0.f;
        for (int e = 0; e < nk; e++) {
            const int r = e >> 5, i = e & 31;
            const int lS = i + ((r & 1) ? 32 : 0), lG = i + ((r & 1) ? 0 : 32);
            aS = aS + ts[r][lS]; aG = aG + ts[r][lG];
            want[0][e] = aS; want[1][e] = aG;
        }
    }
    __syncthreads();
    unsigned wrong = 0;
    for (int r = 0; r < 3; r++) {
        const int e = 32 * r + el;
        if (r < R && e < nk) {
            const bool isS = lowhalf != ((r & 1) != 0);
            if (__builtin_bit_cast(uint32_t, x[r]) != __builtin_bit_cast(uint32_t, want[isS ? 0 : 1][e])) wrong++;
        }
    }
    if (wrong) atomicAdd(bad, wrong);
}

}  // namespace bl

using namespace bl;

// Launches the compact-row kernel; returns BL_ETOOBIG when the shape is outside its template set (the caller then uses
// the general kernel of bl_sim.hip).  waves: 1, or 4 = speculative batches for envs whose last descent had at least
// `deep_thresh` nodes (needs s.fav).
int bl_expand2_launch(const Search& ss, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid, int32_t* leaf_seats,
                      unsigned long long* counters, int fast, int waves, int deep_thresh, int envs, int help_thresh, hipStream_t stream) {
    const int A = ss.S * ss.S, T = ss.T;
    if (!ss.cpi || !ss.cca || !ss.nk || A > 384 || T > 256) return BL_ETOOBIG;
    if (envs > 1) return BL_EINVAL;       // the shared-workgroup kernel of round 4 is gone (see above)
    (void)help_thresh;
    if (waves == 21 && ss.fav && A <= 96 && T <= 64 && !counters && !ss.powf_libm) {
        // two nodes per wave (sim_expand3_kernel)
        const dim3 grid3(ss.B), block3(64);
        const size_t lds3 = (size_t)al16(A);
#define BLX3(RB_) { if (fast) hipLaunchKernelGGL((sim_expand3_kernel<RB_, true>), grid3, block3, lds3, stream, ss, sim, (const uint16_t*)rands, leaves, obs, valid, leaf_seats); \
                    else hipLaunchKernelGGL((sim_expand3_kernel<RB_, false>), grid3, block3, lds3, stream, ss, sim, (const uint16_t*)rands, leaves, obs, valid, leaf_seats); }
        const int rb = (A + 15) / 16;
        if (rb <= 1) BLX3(1) else if (rb <= 2) BLX3(2) else if (rb <= 4) BLX3(4) else BLX3(6)
#undef BLX3
        return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
    }
    if ((waves != 4 && waves != 2) || !ss.fav) waves = 1;     // (eight waves per env: measured in round 4, slower at every batch size -- profiles/r04_waves8.txt)
    if (ss.powf_libm && counters) return BL_EINVAL;      // the counting build exists for the default target only
    if (ss.powf_libm) {
        // the second parity target (bl_tune_t.powf_libm): one instantiation per shape -- ISA-padded fold, two waves per env (one without fav)
        const int needp = (A + 31) / 32;
        const dim3 gridp(ss.B);
        const size_t ldsp = (size_t)al16(A);
#define BLXP(R_, K_, W_) hipLaunchKernelGGL((sim_expand2_kernel<R_, K_, false, false, W_, true>), gridp, dim3(64 * W_), ldsp, stream, ss, sim, \
                                            (const uint16_t*)rands, leaves, obs, valid, leaf_seats, (unsigned long long*)nullptr, deep_thresh)
#define BLXPK(R_) { if (T <= 64) { if (ss.fav) BLXP(R_, 1, 2); else BLXP(R_, 1, 1); } else { if (ss.fav) BLXP(R_, 4, 2); else BLXP(R_, 4, 1); } }
        if (needp <= 1) BLXPK(1) else if (needp <= 2) BLXPK(2) else if (needp <= 3) BLXPK(3) else if (needp <= 6) BLXPK(6) else BLXPK(12)
#undef BLXPK
#undef BLXP
        return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
    }
    const int need = (A + 31) / 32;
    const int rmax = need <= 1 ? 1 : need <= 2 ? 2 : need <= 3 ? 3 : need <= 6 ? 6 : 12;
    const int kt = T <= 64 ? 1 : 4;
    const size_t lds = (size_t)al16(A);
    const dim3 grid(ss.B), block(64 * waves);
#define BLX_LAUNCH(R_, K_, F_, C_, W_) hipLaunchKernelGGL((sim_expand2_kernel<R_, K_, F_, C_, W_>), grid, block, lds, stream, ss, sim, \
                                                          (const uint16_t*)rands, leaves, obs, valid, leaf_seats, counters, deep_thresh)
#define BLX_MODE(R_, K_, W_) { if (counters) { if (fast) BLX_LAUNCH(R_, K_, true, true, W_); else BLX_LAUNCH(R_, K_, false, true, W_); } \
                               else if (fast) BLX_LAUNCH(R_, K_, true, false, W_); else BLX_LAUNCH(R_, K_, false, false, W_); }
#define BLX_KT(R_, W_) { if (kt == 1) BLX_MODE(R_, 1, W_) else BLX_MODE(R_, 4, W_) }
#define BLX_W(R_) { if (waves == 4) BLX_KT(R_, 4) else if (waves == 2) BLX_KT(R_, 2) else BLX_KT(R_, 1) }
    switch (rmax) {
        case 1: BLX_W(1) break;
        case 2: BLX_W(2) break;
        case 3: BLX_W(3) break;
        case 6: BLX_W(6) break;
        default: BLX_W(12) break;
    }
#undef BLX_W
#undef BLX_KT
#undef BLX_MODE
#undef BLX_LAUNCH
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

// Runs the fold self-test (both variants) and returns the number of wrong prefix totals of the FAST one (0 = the FAST
// fold is exact on this device), or a negative BL_E* code.  Synchronises the stream.
int bl_fold_selftest(int use_fast, hipStream_t stream) {
    unsigned int* bad = nullptr;
    if (hipMalloc(&bad, sizeof(unsigned int)) != hipSuccess) return BL_ELAUNCH;
    unsigned int h = 0;
    if (hipMemcpyAsync(bad, &h, sizeof(h), hipMemcpyHostToDevice, stream) != hipSuccess) { hipFree(bad); return BL_ELAUNCH; }
    for (int rep = 0; rep < 6; rep++) {
        for (int nk : {1, 2, 15, 16, 17, 31, 32, 33, 47, 48, 49, 54, 63, 64, 65, 80, 81, 96}) {
            // 4096 waves: four per SIMD, the contention the search kernel runs under
            if (use_fast) hipLaunchKernelGGL((fold_selftest_kernel<true>), dim3(4096), dim3(64), 0, stream, (uint32_t)(rep * 131 + nk), nk, bad);
            else hipLaunchKernelGGL((fold_selftest_kernel<false>), dim3(4096), dim3(64), 0, stream, (uint32_t)(rep * 131 + nk), nk, bad);
        }
    }
    hipError_t e = hipMemcpyAsync(&h, bad, sizeof(h), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    hipFree(bad);
    if (e != hipSuccess || hipGetLastError() != hipSuccess) return BL_ELAUNCH;
    return (int)(h > 0x3fffffffu ? 0x3fffffffu : h);
}
