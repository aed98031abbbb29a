// bl_rows.hip -- bl_sim_expand with FOUR envs per wave, one per 16-lane DPP row: descend (mcts/cpp/cuda.cu:138-182), then the
// expansion (mcts/__init__.py:117-129, hex/cpp/cuda.cu:76-195) as a second launch.  A MEASURED ALTERNATIVE (bl_tune_t.expand_waves
// = 16), not a default: bit-exact in every oracle comparison, and slower than bl_expand.hip's kernel in the regime it was built for.
//
// The idea.  bl_expand.hip's kernel gives every env a wave (or two) and is arranged around the LATENCY of one env's dependent chain
// -- right while the launch fits the chip.  At 32 768 envs (the reference's own actor shape, boardlaw/main.py:147) the same kernel is
// bound by VALU ISSUE: 101 M VALU instructions per launch on 1024 SIMDs = 165 of its 186 us (profiles/r06_pmc32k_SQ1.csv), most of
// them the ordered fold of the Newton sums, one 64-lane instruction per step of ONE env's chain.  Here a fold step is one
// instruction for four envs: lane l of a row holds kept actions P l .. P l + P - 1 of its env's node (P = ceil(nk / 16)), a sweep
// step is one `v_add_f32_dpp row_shr:1` per chain (element 0 <- the neighbour's last element) plus P - 1 in-lane adds, the S and g
// chains interleave, and after j sweep steps lanes 0 .. j of every row hold the reference's running totals exactly (each update
// reads a neighbour that is already final).  Same additions in the same order as cuda.cu:35-68: bit-identical results.  (Gated in
// round 4 as a micro-benchmark: tools/micro/fold_rows.hip.)
//
// The four rows of a wave run in lock step per tree LEVEL, not per env: a row whose descent has ended writes its record and takes
// its next env (envs w, w + stride, ... -- stride = 4 x waves) while the other rows go on.  Within a level the Newton iterations run
// until every row has converged; a row that has keeps its alpha, so its terms and totals are recomputed unchanged.  Every DPP read
// has the ISA's two wait states (this kernel does not depend on bl_selftest()).
//
// What was measured (MI355X, 9x9, 64 sims, 32 768 envs; profiles/r06_rows_kernel.txt): descent 218 us + expansion 31 us against 186
// us for one wave per env.  The instruction count did NOT fall -- 91.5 M + 7.5 M VALU instructions per simulation against 101.5 M
// -- because the lock step costs what the packing saves: a level takes the iterations of its slowest row (about 6 against 4.06 on
// average), a wave the levels of its busiest row (x 1.3 with two envs per row), an iteration still 280 instructions for four
// evaluations against 117 for one (the in-lane adds are redone in every sweep step), and every change of env re-stages a row's slot
// tables under a quarter-full EXEC mask.  And a SIMD with four 111-register waves issues 68 % of the time where eight 50-register
// waves of the per-env kernel issue 89 %.  More waves help (8192: 226 us), fewer hurt (2048: 327 us).
//
// What a descent leaves behind for the expansion launch: path[1 .. nlev] (the nodes it evaluated) and one word per env in
// leaf_seats_out -- action | index in the compacted row << 8 | (next node + 1) << 15 | nlev << 22 -- which the expansion kernel
// replaces by the leaf's seat.  T <= 64 node slots, A <= 96 actions, default parity target; anything else stays with bl_expand.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"

#pragma clang fp contract(off)

namespace bl {

#define BLR_RS " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define BLR_A(d, a, b) "v_add_f32 %" #d ", %" #a ", %" #b "\n\t"
#define BLR_D(d, a, b) "v_add_f32_dpp %" #d ", %" #a ", %" #b BLR_RS

// Four sweep steps of the ordered fold, P elements per lane: xs / xg running totals of the S / g chain, s / g the terms.
// bound_ctrl:0: lane 0 of a row reads 0.0 for its missing neighbour, i.e. 0.f + t0 -- the reference's `float S = 0.f` start.
// Wait states: the DPP read of xs[P-1] follows its write by xg[P-1]'s add and one s_nop, the read of xg[P-1] by the s_nop and the
// S chain's DPP add: two each, what the ISA asks for.
template <int P> struct RowFold;
template <> struct RowFold<1> {
    static __device__ __forceinline__ void steps4(float (&xs)[1], float (&xg)[1], const float (&s)[1], const float (&g)[1]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 0, 2) BLR_D(1, 1, 3)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST : "+v"(xs[0]), "+v"(xg[0]) : "v"(s[0]), "v"(g[0]));
#undef BLR_ST
    }
};
template <> struct RowFold<2> {
    static __device__ __forceinline__ void steps4(float (&xs)[2], float (&xg)[2], const float (&s)[2], const float (&g)[2]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 1, 4) BLR_D(2, 3, 6) BLR_A(1, 0, 5) BLR_A(3, 2, 7)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST : "+v"(xs[0]), "+v"(xs[1]), "+v"(xg[0]), "+v"(xg[1]) : "v"(s[0]), "v"(s[1]), "v"(g[0]), "v"(g[1]));
#undef BLR_ST
    }
};
template <> struct RowFold<3> {
    static __device__ __forceinline__ void steps4(float (&xs)[3], float (&xg)[3], const float (&s)[3], const float (&g)[3]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 2, 6) BLR_D(3, 5, 9) BLR_A(1, 0, 7) BLR_A(4, 3, 10) BLR_A(2, 1, 8) BLR_A(5, 4, 11)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2])
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(g[0]), "v"(g[1]), "v"(g[2]));
#undef BLR_ST
    }
};
template <> struct RowFold<4> {
    static __device__ __forceinline__ void steps4(float (&xs)[4], float (&xg)[4], const float (&s)[4], const float (&g)[4]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 3, 8) BLR_D(4, 7, 12) BLR_A(1, 0, 9) BLR_A(5, 4, 13) BLR_A(2, 1, 10) BLR_A(6, 5, 14) BLR_A(3, 2, 11) BLR_A(7, 6, 15)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3])
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
#undef BLR_ST
    }
};
template <> struct RowFold<5> {
    static __device__ __forceinline__ void steps4(float (&xs)[5], float (&xg)[5], const float (&s)[5], const float (&g)[5]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 4, 10) BLR_D(5, 9, 15) BLR_A(1, 0, 11) BLR_A(6, 5, 16) BLR_A(2, 1, 12) BLR_A(7, 6, 17) BLR_A(3, 2, 13) BLR_A(8, 7, 18) \
                             BLR_A(4, 3, 14) BLR_A(9, 8, 19)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST
                     : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xs[4]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3]), "+v"(xg[4])
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]));
#undef BLR_ST
    }
};
template <> struct RowFold<6> {
    static __device__ __forceinline__ void steps4(float (&xs)[6], float (&xg)[6], const float (&s)[6], const float (&g)[6]) {
#define BLR_ST "s_nop 0\n\t" BLR_D(0, 5, 12) BLR_D(6, 11, 18) BLR_A(1, 0, 13) BLR_A(7, 6, 19) BLR_A(2, 1, 14) BLR_A(8, 7, 20) BLR_A(3, 2, 15) BLR_A(9, 8, 21) \
                             BLR_A(4, 3, 16) BLR_A(10, 9, 22) BLR_A(5, 4, 17) BLR_A(11, 10, 23)
        asm volatile("s_nop 1\n\t" BLR_ST BLR_ST BLR_ST BLR_ST
                     : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xs[4]), "+v"(xs[5]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3]), "+v"(xg[4]), "+v"(xg[5])
                     : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(s[4]), "v"(s[5]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]));
#undef BLR_ST
    }
};

// row-wise reductions: every lane of a 16-lane row ends with its row's value
__device__ __forceinline__ int bperm(int src_lane, int v) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ int row_sum_all_i32(int v, int lane) {        // integer: any order is exact
    v += dpp_i<0x111, 0xf>(0, v); v += dpp_i<0x112, 0xf>(0, v); v += dpp_i<0x114, 0xf>(0, v); v += dpp_i<0x118, 0xf>(0, v);
    return bperm(lane | 15, v);
}
__device__ __forceinline__ float row_max_all_f32(float v, int lane) {    // max: any order is exact; operands are never NaN
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_bit_cast(float, bperm(lane | 15, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int wave_max_all_i32(int v) {
    v = max(v, dpp_i<0x111, 0xf>(0, v)); v = max(v, dpp_i<0x112, 0xf>(0, v)); v = max(v, dpp_i<0x114, 0xf>(0, v)); v = max(v, dpp_i<0x118, 0xf>(0, v));
    v = max(v, dpp_i<0x142, 0xa>(0, v)); v = max(v, dpp_i<0x143, 0xc>(0, v));
    return __builtin_amdgcn_readlane(v, 63);
}

// One slot of an env's tables in LDS: A = {q[slot, seat 0] | q[slot, seat 1] << 16 (transition_q, f16 bits), n[slot]};
// B = {nk | seat << 16 | terminal << 17, rand[slot] (f16 bits)}
struct RowTables { uint2 a[4][64]; uint2 b[4][64]; };

// policy() + the draw (cuda.cu:70-99, 35-68, 157-176) at one node per row, P kept actions per lane.  `ev`: this lane's row has a
// node to evaluate (node, nk, seat, rnd, cpuct are the row's).  Returns per row the drawn action (-1: none has positive
// probability), its child slot (-1: not expanded) and its index in the node's compacted row.
template <int P>
__device__ __forceinline__ void rows_eval(const Search& s, const int A, const bool ev, const long node, const int nk, const int seat, const float rnd,
                                          const float cpuct, const uint2* tabA, const int lane, int& action_o, int& child_o, int& sel_o) {
    const int l = lane & 15;
    const long row = node * A;
    float top[P], qv[P];
    uint32_t cc[P];
    bool in[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int e = P * l + p;
        in[p] = ev && e < nk;
        top[p] = 0.f; cc[p] = 0xffff0000u;
        if (in[p]) { top[p] = s.cpi[row + e]; cc[p] = s.cca[row + e]; }
    }
    int Nloc = 0;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int c = (int)(int16_t)(cc[p] >> 16);
        const bool ex = c >= 0;                                           // (pads carry child 0xffff = -1)
        const uint2 st = tabA[ex ? c : 0];                                // unconditional: one LDS read per element, no EXEC branch
        qv[p] = ex ? h2f((uint16_t)(seat ? (st.x >> 16) : st.x)) : 0.f;
        Nloc += in[p] ? (ex ? (int)st.y : 1) : 0;
    }
    const int N = row_sum_all_i32(Nloc, lane) + (A - nk);                 // dropped actions are unexpanded: +1 each
    const float lam = (cpuct * (float)N) / (float)(unsigned)(N + A);
    float alpha = (nk < A) ? 1.e-4f : 0.f;                                // a dropped action's q + max(lambda pi, 1e-4)
#pragma unroll
    for (int p = 0; p < P; p++) {
        top[p] = lam * top[p];
        if (in[p]) alpha = fmaxf(alpha, qv[p] + fmaxf(top[p], 1.e-4f));
    }
    alpha = row_max_all_f32(alpha, lane);

    // the last lane of any evaluating row that holds a kept action: the sweeps run that many steps, and every row reads its totals
    // there (pads are top = 0, q = 0: s = +0 and g = -0, which leave every partial sum as it is)
    const int L = __builtin_amdgcn_readfirstlane(wave_max_all_i32((ev && nk > 0) ? (nk - 1) / P : 0));
    const int rdlane = (lane & 48) | L;
    // newton_search, cuda.cu:35-68, all rows in step.  `upd` = alpha updates so far = the reference's `it`; a row that is done keeps
    // its alpha, so every later pass recomputes its terms and totals unchanged.
    bool done = !ev || nk == 0;
    float err = INFINITY;
    int upd = 0;
    float sv[P], gv[P], xs[P], xg[P];
#pragma unroll
    for (int p = 0; p < P; p++) { sv[p] = 0.f; gv[p] = 0.f; xs[p] = 0.f; xg[p] = 0.f; }
    while (__builtin_amdgcn_ballot_w64(!done) != 0) {
        float num[2 * P], den[2 * P], quo[2 * P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float bot = alpha - qv[p];
            num[p] = top[p]; den[p] = bot; num[P + p] = -top[p]; den[P + p] = bot * bot;       // prob(a), cuda.cu:23-25, resp. its derivative term
        }
        ieee_div_n<2 * P>(num, den, quo);
#pragma unroll
        for (int p = 0; p < P; p++) { sv[p] = quo[p]; gv[p] = quo[P + p]; }
        xs[0] = 0.f + sv[0]; xg[0] = 0.f + gv[0];                          // the sums start from 0.f (cuda.cu:44): (+0) + (-0) = +0
#pragma unroll
        for (int p = 1; p < P; p++) { xs[p] = xs[p - 1] + sv[p]; xg[p] = xg[p - 1] + gv[p]; }
        RowFold<P>::steps4(xs, xg, sv, gv);
        if (L > 4) RowFold<P>::steps4(xs, xg, sv, gv);
        if (L > 8) RowFold<P>::steps4(xs, xg, sv, gv);
        if (L > 12) RowFold<P>::steps4(xs, xg, sv, gv);
        const float S = __builtin_bit_cast(float, bperm(rdlane, __builtin_bit_cast(int, xs[P - 1])));
        const float G = __builtin_bit_cast(float, bperm(rdlane, __builtin_bit_cast(int, xg[P - 1])));
        const float ne = S - 1.f;
        const float step = ieee_div(ne, G);
        if (!done) {
            if (upd == 100) done = true;                                  // alpha moved after the 100th fold: this pass only refreshed the terms
            else if ((ne < 1e-3f) || (err == ne)) done = true;
            else { alpha -= step; err = ne; upd++; }
        }
    }

    // the draw, cuda.cu:157-176: first kept action (ascending) with prob > 0 and running total >= rand, else the last with prob > 0
    int pf = -1, pl = -1;
#pragma unroll
    for (int p = P - 1; p >= 0; p--) if (in[p] && sv[p] > 0.f && xs[p] >= rnd) pf = p;
#pragma unroll
    for (int p = 0; p < P; p++) if (in[p] && sv[p] > 0.f) pl = p;
    const unsigned long long bh = __builtin_amdgcn_ballot_w64(pf >= 0), bp = __builtin_amdgcn_ballot_w64(pl >= 0);
    const uint32_t h16 = (uint32_t)(bh >> (lane & 48)) & 0xffffu, p16 = (uint32_t)(bp >> (lane & 48)) & 0xffffu;
    const bool first = h16 != 0;
    const int mine_p = first ? pf : pl;
    uint32_t mine_cc = 0;
#pragma unroll
    for (int p = 0; p < P; p++) if (p == mine_p) mine_cc = cc[p];
    const int mine_e = P * l + (mine_p < 0 ? 0 : mine_p);
    const int sl = first ? __builtin_ctz(h16) : (p16 ? 31 - __builtin_clz(p16) : 0);
    const uint32_t ccs = (uint32_t)bperm((lane & 48) | sl, (int)mine_cc);
    const int es = bperm((lane & 48) | sl, mine_e);
    const bool any = first || p16 != 0;
    action_o = any ? (int)(ccs & 0xffffu) : -1;
    child_o = any ? (int)(int16_t)(ccs >> 16) : -1;
    sel_o = any ? es : 0;
}

#define BLR_REC(action, sel, nxt, nlev) ((uint32_t)(action) | ((uint32_t)(sel) << 8) | ((uint32_t)((nxt) + 1) << 15) | ((uint32_t)(nlev) << 22))

// PMAX: ceil(A / 16), the kept actions a lane can hold (templates 1 .. PMAX are instantiated)
template <int PMAX>
__global__ void __launch_bounds__(BL_WAVE, 4) sim_descend_rows_kernel(Search s, int sim, const uint16_t* rands, uint32_t* records) {
    __shared__ RowTables tab;
    const int S = s.S, A = S * S, T = s.T;
    const int lane = threadIdx.x & 63, rowi = lane >> 4, l = lane & 15;
    const int nact = active_envs(s);
    const int stride = (int)gridDim.x * 4;
    const uint2 qwords = qrange_words(s.qrange + (long)BL_QWORDS * sim);
    float lo, hi;
    qrange_reduce(qwords, lo, hi);
    const float rden = hi - lo + 1.e-4f;
    uint2* tabA = tab.a[rowi];
    uint2* tabB = tab.b[rowi];

    // per row (the same value in its 16 lanes)
    int b = (int)blockIdx.x * 4 + rowi;
    int t = 0, nlev = 0, action = -1, sel_e = 0;
    float cpuct = 0.f;
    bool active = false;
    // rows that need an env: stage its slot tables (transition_q of every slot, cuda.cu:101-105; lane l takes slots l, 16 + l, ...),
    // and if its root is terminal (a descent that ends where it starts) write the record at once and go on to the next one
    bool want = true;
    while (__builtin_amdgcn_ballot_w64(want) != 0) {
        const bool go = want && b < nact;
        if (want && !go) { want = false; active = false; }
        if (go) {
            const long envbase = (long)b * T;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int tt = 16 * j + l;
                if (tt < T) {
                    const uint32_t wp = *(const uint32_t*)(s.w + (envbase + tt) * 2);
                    const int nn = s.n[envbase + tt];
                    const uint32_t info = (uint32_t)(uint16_t)s.nk[envbase + tt] | ((uint32_t)(s.seats[envbase + tt] & 1) << 16) | ((s.terminal[envbase + tt] ? 1u : 0u) << 17);
                    const uint32_t rd = rands[envbase + tt];
                    const float den = (float)nn + 1.e-4f;
                    const float q0 = h2f((uint16_t)wp) / den, q1 = h2f((uint16_t)(wp >> 16)) / den;
                    tabA[tt] = make_uint2((uint32_t)f2h((q0 - lo) / rden) | ((uint32_t)f2h((q1 - lo) / rden) << 16), (uint32_t)nn);
                    tabB[tt] = make_uint2(info, rd);
                }
            }
            cpuct = h2f(s.c_puct[b]);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (go) {
            t = 0; nlev = 0; action = -1; sel_e = 0;
            if ((tabB[0].x >> 17) & 1u) {                                 // terminal root: parent 0, no action, the root is re-visited
                if (l == 0) records[b] = BLR_REC(0, 0, 0, 0);
                b += stride;
            } else { want = false; active = true; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    }

    // descend_kernel's loop, cuda.cu:138-182: one level of every active row per trip
    while (__builtin_amdgcn_ballot_w64(active) != 0) {
        const uint2 tb = tabB[active ? t : 0];
        const int nk = (int)(tb.x & 0xffffu), seat = (int)((tb.x >> 16) & 1u);
        const float rnd = h2f((uint16_t)tb.y);
        const long node = (long)(active ? b : 0) * T + (active ? t : 0);
        const int Pw = __builtin_amdgcn_readfirstlane(wave_max_all_i32(active ? (nk + 15) >> 4 : 1));
        int ra = -1, rc = -1, rs = 0;
        if (Pw <= 1) rows_eval<1>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);
        else if (Pw == 2) rows_eval<(PMAX >= 2 ? 2 : 1)>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);
        else if (Pw == 3) rows_eval<(PMAX >= 3 ? 3 : 1)>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);
        else if (Pw == 4) rows_eval<(PMAX >= 4 ? 4 : 1)>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);
        else if (Pw == 5) rows_eval<(PMAX >= 5 ? 5 : 1)>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);
        else rows_eval<(PMAX >= 6 ? 6 : 1)>(s, A, active, node, nk, seat, rnd, cpuct, tabA, lane, ra, rc, rs);

        bool ended = false;
        if (active) {
            if (l == 0) s.path[(long)b * (T + 2) + 1 + nlev] = (int16_t)t;
            nlev++;
            sel_e = rs;
            if (ra < 0) { action = -1; ended = true; }                    // no action with positive probability: the reference would index [-1]
            else {
                action = ra;
                t = rc;
                if (t == -1 || nlev >= T) ended = true;
                else if ((tabB[t].x >> 17) & 1u) ended = true;
            }
            if (ended) {
                // (after `ra < 0` t is still the node just evaluated: it is re-visited, like a terminal one)
                if (l == 0) records[b] = BLR_REC(action < 0 ? 0 : action, sel_e, t, nlev);
                b += stride;
            }
        }
        want = active && ended;
        // the rows that ended take their next env (see above)
        while (__builtin_amdgcn_ballot_w64(want) != 0) {
            const bool go = want && b < nact;
            if (want && !go) { want = false; active = false; }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();     // nobody still reads the row's old tables
            if (go) {
                const long envbase = (long)b * T;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int tt = 16 * j + l;
                    if (tt < T) {
                        const uint32_t wp = *(const uint32_t*)(s.w + (envbase + tt) * 2);
                        const int nn = s.n[envbase + tt];
                        const uint32_t info = (uint32_t)(uint16_t)s.nk[envbase + tt] | ((uint32_t)(s.seats[envbase + tt] & 1) << 16) | ((s.terminal[envbase + tt] ? 1u : 0u) << 17);
                        const uint32_t rd = rands[envbase + tt];
                        const float den = (float)nn + 1.e-4f;
                        const float q0 = h2f((uint16_t)wp) / den, q1 = h2f((uint16_t)(wp >> 16)) / den;
                        tabA[tt] = make_uint2((uint32_t)f2h((q0 - lo) / rden) | ((uint32_t)f2h((q1 - lo) / rden) << 16), (uint32_t)nn);
                        tabB[tt] = make_uint2(info, rd);
                    }
                }
                cpuct = h2f(s.c_puct[b]);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
            if (go) {
                t = 0; nlev = 0; action = -1; sel_e = 0;
                if ((tabB[0].x >> 17) & 1u) {
                    if (l == 0) records[b] = BLR_REC(0, 0, 0, 0);
                    b += stride;
                } else want = false;
            }
        }
    }
}

// The expansion (the tail of bl_expand.hip's kernel): leaves = children[envs, parents, actions]; leaves[leaves == -1] = sim
// (mcts/__init__.py:117-122), Hex.step on the parent's board (hex/__init__.py:181-195; the flood as a bit-board fill), observe +
// valid for the network (hex/cpp/cuda.cu:154-195), one wave per env.
template <int NW64>
__global__ void __launch_bounds__(BL_WAVE) sim_expand_tail_kernel(Search s, int sim, const uint32_t* records, int16_t* leaves_out, void* obs_out,
                                                                  uint8_t* valid_out, int32_t* leaf_seats_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* cells = (uint8_t*)smem;
    const int S = s.S, A = S * S, T = s.T;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x;
    if (b >= active_envs(s)) return;
    const long envbase = (long)b * T;
    int16_t* path = s.path + (long)b * (T + 2);
    const uint32_t rec = __builtin_amdgcn_readfirstlane((int)records[b]);
    const int action = (int)(rec & 0xffu), sel_e = (int)((rec >> 8) & 0x7fu), nxt = (int)((rec >> 15) & 0x7fu) - 1, nlev = (int)((rec >> 22) & 0x7fu);
    const int parent = nlev > 0 ? __builtin_amdgcn_readfirstlane((int)path[nlev]) : 0;
    const int leaf = (nxt == -1) ? sim : nxt;
    if (s.lazy) lazy_slot_reset(s, envbase, sim, A, nxt == -1, lane);
    if (lane == 0) {
        s.children[(envbase + parent) * A + action] = (int16_t)leaf;
        s.parents[envbase + leaf] = (int16_t)parent;
        s.relation[envbase + leaf] = (int16_t)action;
        if (nxt == -1 && nlev > 0) ((uint16_t*)(s.cca + (envbase + parent) * A + sel_e))[1] = (uint16_t)leaf;   // the compacted row's child field
    }
    const int seat = __builtin_amdgcn_readfirstlane(s.seats[envbase + parent]);
    const uint8_t* src = s.boards + (envbase + parent) * A;
    for (int a = lane; a < A; a += 64) cells[a] = src[a];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    const int win = hex_step_wave<NW64>(cells, S, seat, action, lane);
    const bool term = win != 0;                                           // Hex.step tail, hex/__init__.py:183-190
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
        path[1 + nlev] = (int16_t)leaf;
        path[0] = (int16_t)(nlev + 1);
    }
}

}  // namespace bl

using namespace bl;

// Descend + expand as two launches.  BL_ETOOBIG when the shape is outside this path (the caller then uses bl_expand.hip's kernel).
// waves: workgroups (= waves) of the descent launch, 0 = by the batch.
int bl_expand_rows_launch(const Search& ss, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid, int32_t* leaf_seats,
                          int waves, hipStream_t stream) {
    const int A = ss.S * ss.S, T = ss.T;
    if (!ss.cpi || !ss.cca || !ss.nk || !ss.path || ss.order || ss.powf_libm || A > 96 || T > 64 || ss.B <= 0) return BL_ETOOBIG;
    const int rows = (ss.B + 3) / 4;
    // four waves per SIMD: the fold's dependent steps leave a SIMD idle with fewer (tools/micro/fold_rows.hip: 1575 / 973 / 689
    // cycles per wave-iteration and SIMD at one / two / four), and with more a row has too few envs to even out their depths
    if (waves <= 0) waves = 4096;
    if (waves > rows) waves = rows;
    uint32_t* records = (uint32_t*)leaf_seats;
    const int pmax = (A + 15) / 16;
#define BLR_LAUNCH(P_) hipLaunchKernelGGL((sim_descend_rows_kernel<P_>), dim3(waves), dim3(64), 0, stream, ss, sim, (const uint16_t*)rands, records)
    if (pmax <= 1) BLR_LAUNCH(1); else if (pmax <= 2) BLR_LAUNCH(2); else if (pmax <= 4) BLR_LAUNCH(4); else BLR_LAUNCH(6);
#undef BLR_LAUNCH
    const size_t lds = (size_t)al16(A);
    if (A <= 64) hipLaunchKernelGGL((sim_expand_tail_kernel<1>), dim3(ss.B), dim3(64), lds, stream, ss, sim, records, leaves, obs, valid, leaf_seats);
    else hipLaunchKernelGGL((sim_expand_tail_kernel<2>), dim3(ss.B), dim3(64), lds, stream, ss, sim, records, leaves, obs, valid, leaf_seats);
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}
