// bl_sim.hip -- the fused simulation step on the reference's arrays for Hex (boardlaw/mcts/__init__.py:108-140, hex/__init__.py:148-195)
// in its general form and everything around it: the general bl_sim_expand kernel (lanes per env; bl_expand.hip's compact-row kernel is
// the one that normally runs), bl_sim_backup, the finish step (heads + store + backup + next q range), the compacted rows' writer,
// root planting, tree reset, n_leaves -- and their C entry points (bl_sim_*).  Split out of bl_search.hip in round 6.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"
#include "bl_dispatch.h"
#include "bl_policy.h"

#pragma clang fp contract(off)

namespace bl {

// ------------------------------------------------------------------------------------------------------------------
// Fused simulation step for Hex: mcts/__init__.py:113-129 + hex/__init__.py:148-195.
// ------------------------------------------------------------------------------------------------------------------

template <int G, int K, bool COUNT>
__global__ void __launch_bounds__(BL_WAVE) sim_expand_kernel(Search s, int sim, const uint16_t* rands, int16_t* leaves_out,
                                                             void* obs_out, uint8_t* valid_out, int32_t* leaf_seats_out,
                                                             unsigned long long* counters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = s.S, A = S * S, T = s.T;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int slot = blockIdx.x * (BL_WAVE / G) + grp;
    const bool inb = slot < s.B;
    // launch slot -> env: with an `order` the host decides which envs are dispatched first (oldest wave on a SIMD wins
    // the issue arbitration); results do not depend on it
    const int b = (s.order && inb) ? s.order[slot] : slot;
    const bool act = inb && b < active_envs(s);
    const GroupLds L(smem + (size_t)grp * lds_bytes(A, true), A);
    uint8_t* cells = L.cells;
    if (s.prio_thresh > 0 && s.path && act && s.path[(long)b * (T + 2)] >= s.prio_thresh) __builtin_amdgcn_s_setprio(3);

    Tree m;
    m.logits = s.logits; m.w = s.w; m.n = s.n; m.c_puct = s.c_puct; m.seats = s.seats; m.terminal = s.terminal;
    m.children = s.children; m.qrange = s.qrange + (long)BL_QWORDS * sim; m.exp_table = s.exp_table;
    m.B = s.B; m.T = T; m.A = A; m.S = 2; m.seats_i32 = 1; m.powf_libm = s.powf_libm;

    long long tk0 = 0, tk1 = 0;
    if (COUNT) tk0 = clock64();
    int parent, action, nxt, nlev;
    int16_t* path = s.path ? s.path + (long)b * (T + 2) : nullptr;
    descend_group<G, K, COUNT>(m, b, act, gl, rands, L, counters, act ? path : nullptr, parent, action, nxt, nlev);
    if (action < 0) action = 0;
    if (COUNT) tk1 = clock64();

    // leaves = children[envs, parents, actions]; leaves[leaves == -1] = sim   (mcts/__init__.py:117-122)
    const int leaf = (nxt == -1) ? sim : nxt;
    const long envbase = (long)b * T;
    int seat = 0;
    if (act && s.lazy) lazy_slot_reset(s, envbase, sim, A, nxt == -1, gl, G);
    if (act) {
        if (gl == 0) {
            s.children[(envbase + parent) * A + action] = (int16_t)leaf;
            s.parents[envbase + leaf] = (int16_t)parent;
            s.relation[envbase + leaf] = (int16_t)action;
        }
        seat = s.seats[envbase + parent];
        const uint8_t* src = s.boards + (envbase + parent) * A;
        for (int a = gl; a < A; a += G) cells[a] = src[a];
    }
    __syncthreads();
    const int win = hex_step_group<G>(cells, S, seat, action, act, gl);
    if (!act) return;
    // Hex.step tail, hex/__init__.py:183-190
    const bool term = win != 0;
    const int new_seat = term ? 0 : 1 - seat;
    uint8_t* dst = s.boards + (envbase + leaf) * A;
    const float invS = 1.0f / (float)S;
    const bool flip = new_seat == 1;
    for (int a = gl; a < A; a += G) {
        const uint8_t c = term ? (uint8_t)0 : cells[a];
        dst[a] = c;
    }
    for (int a = gl; a < A; a += G) {
        const int i = (int)(((float)a + 0.5f) * invS), j = a - i * S;
        const int color = term ? 2 : color_of(cells[flip ? j * S + i : a]);
        const int ch = color < 2 ? (flip ? 1 - color : color) : 2;
        if (s.obs_f16) ((uint32_t*)obs_out)[(long)b * A + a] = ch == 0 ? 0x00003c00u : (ch == 1 ? 0x3c000000u : 0u);   // f16 1.0 = 0x3c00
        else ((float2*)obs_out)[(long)b * A + a] = make_float2(ch == 0 ? 1.f : 0.f, ch == 1 ? 1.f : 0.f);
        valid_out[(long)b * A + a] = color == 2;
    }
    if (gl == 0) {
        s.seats[envbase + leaf] = new_seat;
        s.terminal[envbase + leaf] = term;
        // transition.rewards.half(): +-1 and 0 are exact in f16
        s.rewards[(envbase + leaf) * 2 + 0] = f2h((float)win);
        s.rewards[(envbase + leaf) * 2 + 1] = f2h((float)(-win));
        leaves_out[b] = (int16_t)leaf;
        leaf_seats_out[b] = new_seat;
        if (path) {
            // evaluated nodes root-first, then the leaf (new, or the terminal node the descent stopped at)
            path[1 + nlev] = (int16_t)leaf;
            path[0] = (int16_t)(nlev + 1);
        }
    }
    if (COUNT && gl == 0) {
        unsigned long long* e = counters + 12 * (long)b;
        const long long tk2 = clock64();
        e[8] += tk1 - tk0;    // descent, all levels
        e[9] += tk2 - tk1;    // expansion: step + flood + observe + stores
    }
}

// mcts/__init__.py:135-140: store logits/v for the leaf, back up, then reduce next sim's q-range.  16 lanes per env.
__global__ void __launch_bounds__(BL_WAVE) sim_backup_kernel(Search s, int sim, const int16_t* leaves, const void* leaf_logits,
                                                            int logits_f16, const void* leaf_v, int v_f16) {
    constexpr int G = 16;
    const int S = s.S, A = S * S, T = s.T;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int b = blockIdx.x * (BL_WAVE / G) + grp;
    const bool act = b < active_envs(s);
    const long envbase = (long)b * T;
    uint32_t nmin = 0, vmax = 0;
    if (act) {
        const int leaf = leaves[b];
        uint16_t* dst = s.logits + (envbase + leaf) * A;
        if (logits_f16) { const uint16_t* src = (const uint16_t*)leaf_logits + (long)b * A; for (int a = gl; a < A; a += G) dst[a] = src[a]; }
        else { const float* src = (const float*)leaf_logits + (long)b * A; for (int a = gl; a < A; a += G) dst[a] = f2h(src[a]); }
        if (gl < 2) {
            const uint16_t vb = v_f16 ? ((const uint16_t*)leaf_v)[2 * b + gl] : f2h(((const float*)leaf_v)[2 * b + gl]);
            s.v[(envbase + leaf) * 2 + gl] = vb;
            backup_walk(s.rewards, s.parents, s.terminal, s.w, s.n, envbase, 2, gl, leaf, h2f(vb));
        }
    }
    __syncthreads();   // workgroup-scope release/acquire: the walk's stores are visible to the scan below
    if (act) {
        for (int e = gl; e < T; e += G) {
            const float den = (float)s.n[envbase + e] + 1.e-4f;
            const float qa_[2] = {h2f(s.w[(envbase + e) * 2]), h2f(s.w[(envbase + e) * 2 + 1])}, qb_[2] = {den, den};
            float qq_[2];
            ieee_div_n<2>(qa_, qb_, qq_);                      // both seats' quotients side by side (bl_device.h)
            const uint32_t e0 = enc(qq_[0]), e1 = enc(qq_[1]);
            nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
        }
    }
    qrange_publish(s.qrange + (long)BL_QWORDS * (sim + 1), nmin, vmax, blockIdx.x % BL_QSLOTS);
}

// ------------------------------------------------------------------------------------------------------------------
// bl_sim_finish: the network's heads + mcts/__init__.py:135-140 in one launch, one wave per env.
//   logits = log_softmax(masked_fill(policy_raw, ~valid, -inf)) in f32, stored as f16   (heads.py:101-104 under autocast,
//            then decisions.logits.half()): the arithmetic follows torch's persistent-softmax kernel operation for
//            operation (lane l holds elements l, l+W, ...; per-lane sequential exp-sum; xor-butterfly over W lanes;
//            out = (x - max) - log(sum)), so the stored bits equal what F.log_softmax(...).half() gives on this device;
//   v      = scatter_values(tanh(value_raw), seats) (heads.py:122-142): tanhf in f32, rounded to f16, negated for the
//            other seat.
// tests/test_gpu_parity.py::test_finish_heads_match_torch checks both against torch bit for bit.
// ------------------------------------------------------------------------------------------------------------------
// RAW = uint16_t: the pre-head outputs are f16 (fp16 autocast, the reference's GPU configuration); RAW = float: they are f32
// (bl_sim_finish_f32: the reference's CPU configuration, where autocast is a no-op and the heads run in f32 before `.half()`).
__device__ __forceinline__ float raw2f(uint16_t x) { return h2f(x); }
__device__ __forceinline__ float raw2f(float x) { return x; }
template <typename RAW>
__global__ void __launch_bounds__(BL_WAVE) sim_finish_kernel(Search s, int sim, const int16_t* leaves, const RAW* policy_raw,
                                                            const RAW* value_raw, const uint8_t* valid,
                                                            const int32_t* leaf_seats, int W, int iters) {
    const int S = s.S, A = S * S, T = s.T;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= active_envs(s)) return;
    const long envbase = (long)b * T;
    const int leaf = leaves[b];
    uint16_t lb[16];
#pragma unroll
    for (int it = 0; it < 16; it++) lb[it] = 0;
    // ---- policy head
    if (lane < W) {
        float e[16];
        float mx = -INFINITY;
#pragma unroll
        for (int it = 0; it < 16; it++) {
            e[it] = -INFINITY;
            if (it < iters) {
                const int a = lane + it * W;
                if (a < A) e[it] = valid[(long)b * A + a] ? raw2f(policy_raw[(long)b * A + a]) : -INFINITY;
                mx = (it == 0) ? e[0] : ((mx > e[it]) ? mx : e[it]);
            }
        }
        for (int off = W / 2; off > 0; off /= 2) { const float o = __shfl_xor(mx, off, W); mx = (mx < o) ? o : mx; }
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < 16; it++) if (it < iters) sum += expf(e[it] - mx);
        for (int off = W / 2; off > 0; off /= 2) sum = sum + __shfl_xor(sum, off, W);
        const float lsum = logf(sum);
        uint16_t* dst = s.logits + (envbase + leaf) * A;
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int a = lane + it * W;
            if (it < iters && a < A) { lb[it] = f2h(e[it] - mx - lsum); dst[a] = lb[it]; }
        }
    }
    if (s.cpi) {
        int count = 0;
#pragma unroll
        for (int it = 0; it < 16; it++)
            if (it < iters) count = compact_store(s, envbase + leaf, A, lane + it * W, lane < W && lane + it * W < A, lb[it], count);
        if (lane == 0) s.nk[envbase + leaf] = (int16_t)count;
    }
    // ---- value head
    const uint16_t tv = f2h(tanhf(raw2f(value_raw[b])));
    const int mover = leaf_seats[b];
    const uint16_t vb0 = (mover == 0) ? tv : (uint16_t)(tv ^ 0x8000u), vb1 = (uint16_t)(vb0 ^ 0x8000u);
    if (lane == 0) { s.v[(envbase + leaf) * 2] = vb0; s.v[(envbase + leaf) * 2 + 1] = vb1; }
    // ---- backup (cuda.cu:205-236) along the path bl_sim_expand recorded (root first, leaf last): lane j loads node j's
    // fields in one round trip instead of chasing parents[] leaf-to-root; v then flows leaf -> root through registers.
    const int16_t* path = s.path + (long)b * (T + 2);
    const int len = path[0];
    float v0 = h2f(vb0), v1 = h2f(vb1);
    for (int base = ((len - 1) / BL_WAVE) * BL_WAVE; base >= 0; base -= BL_WAVE) {
        const int j = base + lane;
        const bool in = j < len;
        long i = envbase;
        int term = 0, nn = 0;
        float r0 = 0.f, r1 = 0.f, w0 = 0.f, w1 = 0.f;
        if (in) {
            i = envbase + path[1 + j];
            term = s.terminal[i]; nn = s.n[i];
            r0 = h2f(s.rewards[i * 2]); r1 = h2f(s.rewards[i * 2 + 1]);
            w0 = h2f(s.w[i * 2]); w1 = h2f(s.w[i * 2 + 1]);
        }
        const int top_j = min(len - 1 - base, BL_WAVE - 1);
        for (int l = top_j; l >= 0; l--) {
            if (__builtin_amdgcn_readlane(term, l)) { v0 = 0.f; v1 = 0.f; }
            v0 += readlane_f(r0, l); v1 += readlane_f(r1, l);
            if (lane == l) { w0 = h2f(f2h(w0 + h2f(f2h(v0)))); w1 = h2f(f2h(w1 + h2f(f2h(v1)))); }
        }
        if (in) {
            s.w[i * 2] = f2h(w0); s.w[i * 2 + 1] = f2h(w1);
            s.n[i] = (int16_t)(nn + 2);      // n += 1 once per seat (cuda.cu:230)
        }
    }
    __syncthreads();   // workgroup-scope release/acquire: the stores above are visible to the scan below
    uint32_t nmin = 0, vmax = 0;
    for (int e = lane; e < T; e += BL_WAVE) {
        const float den = (float)s.n[envbase + e] + 1.e-4f;
        const float qa_[2] = {h2f(s.w[(envbase + e) * 2]), h2f(s.w[(envbase + e) * 2 + 1])}, qb_[2] = {den, den};
        float qq_[2];
        ieee_div_n<2>(qa_, qb_, qq_);                          // both seats' quotients side by side (bl_device.h)
        const uint32_t e0 = enc(qq_[0]), e1 = enc(qq_[1]);
        nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
    }
    nmin = wave_max_u32(nmin); vmax = wave_max_u32(vmax);
    if (lane == 0) {
        uint32_t* p = s.qrange + (long)BL_QWORDS * (sim + 1) + BL_QSTRIDE * (blockIdx.x % BL_QSLOTS);
        q_atomic_max_checked(p, nmin);
        q_atomic_max_checked(p + 1, vmax);
    }
}

// The same step with its memory round trips counted (round 5).  sim_finish_kernel above chases its data: leaves[b]; valid, then the
// policy entry behind a branch on it, per iteration; the exp-table gather per iteration; path[0], then path[1 + j], then that node's
// fields; after the stores a barrier and every slot's (w, n) again for the q range; a checking load before each atomic -- about ten
// dependent trips of ~1 us for a wave that has nothing else to do (13.5 us per launch at config 4's shape, 17.4 in the torch-GEMM
// and fp32 plans of config 2).  Here everything that depends on nothing but b goes out at once (leaf, value, mover, path length, the
// path, every slot's (w, n), valid and policy rows), the second trip fetches what depends on the path (terminal, rewards) and the
// exp-table entries, the path nodes' (w, n) come out of the slot registers through LDS, and the q range reuses the slot registers
// with the path's nodes replaced -- no barrier + reload.  Same arithmetic in the same order, same stores.  T <= 64 KT, A <= W ITERS.
template <typename RAW, int KT, int ITERS>
__global__ void __launch_bounds__(BL_WAVE) sim_finish_fast_kernel(Search s, int sim, const int16_t* leaves, const RAW* policy_raw,
                                                                 const RAW* value_raw, const uint8_t* valid,
                                                                 const int32_t* leaf_seats, int W) {
    __shared__ uint32_t lw[64 * KT];
    __shared__ int16_t ln[64 * KT];
    const int S = s.S, A = S * S, T = s.T;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= active_envs(s)) return;
    const long envbase = (long)b * T;
    // ---- trip 1: everything that depends on b alone
    const int leaf = leaves[b];
    const RAW vraw = value_raw[b];
    const int mover = leaf_seats[b];
    const int16_t* path = s.path + (long)b * (T + 2);
    const int len = path[0];
    int pj[KT];
    uint32_t sw[KT]; int sn[KT];
#pragma unroll
    for (int c = 0; c < KT; c++) {
        const int t = 64 * c + lane;
        pj[c] = 0; sw[c] = 0; sn[c] = 0;
        if (t < T) { pj[c] = path[1 + t]; sw[c] = *(const uint32_t*)(s.w + (envbase + t) * 2); sn[c] = s.n[envbase + t]; }
    }
    float pe[ITERS]; uint8_t vd[ITERS]; bool in[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int a = lane + it * W;
        in[it] = lane < W && a < A;
        pe[it] = 0.f; vd[it] = 0;
        if (in[it]) { vd[it] = valid[(long)b * A + a]; pe[it] = raw2f(policy_raw[(long)b * A + a]); }
    }
    // ---- policy head (sim_finish_kernel's arithmetic, operation for operation)
    uint16_t lb[ITERS];
    {
        float e[ITERS];
        float mx = -INFINITY;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            e[it] = (in[it] && vd[it]) ? pe[it] : -INFINITY;
            mx = (it == 0) ? e[0] : ((mx > e[it]) ? mx : e[it]);
        }
        for (int off = W / 2; off > 0; off /= 2) { const float o = __shfl_xor(mx, off, W); mx = (mx < o) ? o : mx; }
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; it++) sum += expf(e[it] - mx);
        for (int off = W / 2; off > 0; off /= 2) sum = sum + __shfl_xor(sum, off, W);
        const float lsum = logf(sum);
#pragma unroll
        for (int it = 0; it < ITERS; it++) lb[it] = in[it] ? f2h(e[it] - mx - lsum) : (uint16_t)0;
    }
    // ---- trip 2: the exp-table entries of the row, and what depends on the path (terminal, rewards); the slots' (w, n) go into LDS
    float pi[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; it++) pi[it] = (s.cpi && in[it]) ? s.exp_table[lb[it]] : 0.f;
    int term[KT]; uint32_t rr[KT];
#pragma unroll
    for (int c = 0; c < KT; c++) {
        const int t = 64 * c + lane;
        term[c] = 0; rr[c] = 0;
        if (t < len) { const long i = envbase + pj[c]; term[c] = s.terminal[i]; rr[c] = *(const uint32_t*)(s.rewards + i * 2); }
        if (t < T) { lw[t] = sw[c]; ln[t] = (int16_t)sn[c]; }
    }
    {   // stores of the policy row
        uint16_t* dst = s.logits + (envbase + leaf) * A;
#pragma unroll
        for (int it = 0; it < ITERS; it++) if (in[it]) dst[lane + it * W] = lb[it];
    }
    // ---- value head
    const uint16_t tv = f2h(tanhf(raw2f(vraw)));
    const uint16_t vb0 = (mover == 0) ? tv : (uint16_t)(tv ^ 0x8000u), vb1 = (uint16_t)(vb0 ^ 0x8000u);
    if (lane == 0) { s.v[(envbase + leaf) * 2] = vb0; s.v[(envbase + leaf) * 2 + 1] = vb1; }
    __syncthreads();
    // ---- backup (cuda.cu:205-236), leaf -> root; node j's old (w, n) from the slot values
    float v0 = h2f(vb0), v1 = h2f(vb1);
#pragma unroll
    for (int c = KT - 1; c >= 0; c--) {
        const int base = 64 * c;
        if (base < len) {                                       // wave-uniform
            const int j = base + lane;
            const bool onp = j < len;
            const uint32_t wold = onp ? lw[pj[c]] : 0u;
            const int nold = onp ? (int)ln[pj[c]] : 0;
            float w0 = h2f((uint16_t)wold), w1 = h2f((uint16_t)(wold >> 16));
            const float r0 = h2f((uint16_t)rr[c]), r1 = h2f((uint16_t)(rr[c] >> 16));
            const int top_j = min(len - 1 - base, BL_WAVE - 1);
            for (int l = top_j; l >= 0; l--) {
                if (__builtin_amdgcn_readlane(term[c], l)) { v0 = 0.f; v1 = 0.f; }
                v0 += readlane_f(r0, l); v1 += readlane_f(r1, l);
                if (lane == l) { w0 = h2f(f2h(w0 + h2f(f2h(v0)))); w1 = h2f(f2h(w1 + h2f(f2h(v1)))); }
            }
            if (onp) {
                const long i = envbase + pj[c];
                const uint32_t wnew = (uint32_t)f2h(w0) | ((uint32_t)f2h(w1) << 16);
                const int16_t nnew = (int16_t)(nold + 2);        // n += 1 once per seat (cuda.cu:230)
                *(uint32_t*)(s.w + i * 2) = wnew; s.n[i] = nnew;
                lw[pj[c]] = wnew; ln[pj[c]] = nnew;
            }
        }
    }
    // ---- the compacted row (compact_store's order: iterations ascending, lanes ascending)
    if (s.cpi) {
        int count = 0;
        const long node = envbase + leaf;
#pragma unroll
        for (int it = 0; it < ITERS; it++) {
            const bool keep = in[it] && pi[it] != 0.f;
            const unsigned long long mk = __ballot(keep);
            if (keep) {
                const int jj = count + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
                s.cpi[node * A + jj] = pi[it];
                s.cca[node * A + jj] = 0xffff0000u | (uint32_t)(lane + it * W);
            }
            count += __builtin_popcountll(mk);
        }
        if (lane == 0) s.nk[node] = (int16_t)count;
    }
    __syncthreads();
    // ---- transition_q's range over the env's slots, the path's nodes with their new statistics
    uint32_t nmin = 0, vmax = 0;
#pragma unroll
    for (int c = 0; c < KT; c++) {
        const int t = 64 * c + lane;
        if (t < T) {
            const uint32_t wv = lw[t];
            const float den = (float)ln[t] + 1.e-4f;
            const float qa_[2] = {h2f((uint16_t)wv), h2f((uint16_t)(wv >> 16))}, qb_[2] = {den, den};
            float qq_[2];
            ieee_div_n<2>(qa_, qb_, qq_);
            const uint32_t e0 = enc(qq_[0]), e1 = enc(qq_[1]);
            nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
        }
    }
    nmin = wave_max_u32(nmin); vmax = wave_max_u32(vmax);
    if (lane == 0) {
        uint32_t* p = s.qrange + (long)BL_QWORDS * (sim + 1) + BL_QSTRIDE * (blockIdx.x % BL_QSLOTS);
        q_atomic_max_checked(p, nmin);
        q_atomic_max_checked(p + 1, vmax);
    }
}

// Compacted rows (bl_device.h: compact_store) for logits somebody else stored: node leaves[b] of every env, or node 0
// when leaves is null (a planted root).  One wave per env.
__global__ void __launch_bounds__(BL_WAVE) compact_rows_kernel(Search s, const int16_t* leaves) {
    const int A = s.S * s.S, b = blockIdx.x, lane = threadIdx.x;
    if (b >= active_envs(s)) return;
    const long node = (long)b * s.T + (leaves ? (int)leaves[b] : 0);
    int count = 0;
    for (int a0 = 0; a0 < A; a0 += BL_WAVE) {
        const int a = a0 + lane;
        const bool in = a < A;
        count = compact_store(s, node, A, a, in, in ? s.logits[node * A + a] : (uint16_t)0, count);
    }
    if (lane == 0) s.nk[node] = (int16_t)count;
}

// ------------------------------------------------------------------------------------------------------------------
// bl_sim_plant_root: what MCTS.initialize does with the root network's fp32 pre-head outputs (mcts/__init__.py:72-80,
// 13-24; heads.py:101-104,122-142), one wave per env:
//   logits = log_softmax(masked_fill(policy_raw, ~valid, -inf))            torch's persistent-softmax operation order
//   draw[~valid] = 0; draw /= draw.sum();  logits = log(exp(logits)*(1-eps) + draw*eps)      dirichlet_noise
//   v = scatter_values(tanh(value_raw), seats)
//   decisions.logits[:, 0] = logits.half(); decisions.v[:, 0] = v.half()
// ------------------------------------------------------------------------------------------------------------------
// gamma_Wr > 0: `draw` holds torch's standard-gamma variates, not a finished Dirichlet sample -- the kernel first does what
// at::_sample_dirichlet does after its gamma kernel (Distributions.cu: ret = gamma / gamma.sum(-1, keepdim); clamped to
// [FLT_MIN, 1 - FLT_EPSILON]) with torch's summation order (torch_row_sum, A < 128), i.e. two launches fewer per move and
// the same bits.
__global__ void __launch_bounds__(BL_WAVE) sim_plant_root_kernel(Search s, const float* policy_raw, const float* value_raw,
                                                                const uint8_t* valid, const int32_t* seats, const float* draw,
                                                                float eps, int W, int iters, int gamma_Wr) {
    const int S = s.S, A = S * S, T = s.T;
    const int b = blockIdx.x, lane = threadIdx.x;
    const long envbase = (long)b * T;
    uint16_t lb[16];
#pragma unroll
    for (int it = 0; it < 16; it++) lb[it] = 0;
    if (lane < W) {
        float e[16], d[16];
        float mx = -INFINITY, dsum = 0.f;
#pragma unroll
        for (int it = 0; it < 16; it++) {
            e[it] = -INFINITY; d[it] = 0.f;
            if (it < iters) {
                const int a = lane + it * W;
                if (a < A) {
                    const bool ok = valid[(long)b * A + a];
                    e[it] = ok ? policy_raw[(long)b * A + a] : -INFINITY;
                    if (draw) d[it] = (ok || gamma_Wr) ? draw[(long)b * A + a] : 0.f;
                }
                mx = (it == 0) ? e[0] : ((mx > e[it]) ? mx : e[it]);
            }
        }
        if (gamma_Wr) {
            // the Dirichlet sample from its gamma variates, over ALL actions (iters <= 2 here): lane x < Wr sums elements x and x + Wr
            float own, second;
            if (gamma_Wr == 64) { own = d[0]; second = d[1]; }                           // A >= 64: Wr == W == 64, the lane's own two elements
            else if (gamma_Wr == W) { own = d[0]; second = 0.f; }                        // A a power of two: one element per lane
            else { const float up = __shfl(d[0], (lane + gamma_Wr) & 63, BL_WAVE); own = lane < gamma_Wr ? d[0] : 0.f; second = (lane < gamma_Wr && lane + gamma_Wr < A) ? up : 0.f; }
            const float gsum = torch_row_sum(own, second, gamma_Wr);
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const int a = lane + it * W;
                if (it < iters && a < A) {
                    float r = d[it] / gsum;
                    r = (1.17549435e-38f > r) ? 1.17549435e-38f : r;
                    r = ((1.f - 1.1920929e-07f) < r) ? (1.f - 1.1920929e-07f) : r;
                    d[it] = valid[(long)b * A + a] ? r : 0.f;                             // dirichlet_noise: draw[~valid] = 0
                } else d[it] = 0.f;
            }
        }
#pragma unroll
        for (int it = 0; it < 16; it++) if (it < iters) dsum += d[it];
        for (int off = W / 2; off > 0; off /= 2) { const float o = __shfl_xor(mx, off, W); mx = (mx < o) ? o : mx; }
        float sum = 0.f;
#pragma unroll
        for (int it = 0; it < 16; it++) if (it < iters) sum += expf(e[it] - mx);
        for (int off = W / 2; off > 0; off /= 2) { sum = sum + __shfl_xor(sum, off, W); dsum = dsum + __shfl_xor(dsum, off, W); }
        const float lsum = logf(sum);
        const float keep = 1.f - eps;
        uint16_t* dst = s.logits + envbase * A;          // node 0
#pragma unroll
        for (int it = 0; it < 16; it++) {
            const int a = lane + it * W;
            if (it < iters && a < A) {
                float l = e[it] - mx - lsum;
                if (draw) l = logf(expf(l) * keep + (d[it] / dsum) * eps);
                lb[it] = f2h(l);
                dst[a] = lb[it];
            }
        }
    }
    if (s.cpi) {
        int count = 0;
#pragma unroll
        for (int it = 0; it < 16; it++)
            if (it < iters) count = compact_store(s, envbase, A, lane + it * W, lane < W && lane + it * W < A, lb[it], count);
        if (lane == 0) s.nk[envbase] = (int16_t)count;
    }
    if (lane == 0) {
        const float tv = tanhf(value_raw[b]);
        const int mover = seats[b];
        s.v[envbase * 2 + mover] = f2h(tv);
        s.v[envbase * 2 + 1 - mover] = f2h(-tv);
    }
}

// Fills nbytes at p (16-B aligned, as torch allocations are) with a repeating 16-bit pattern; whole grid cooperates.
__device__ __forceinline__ void grid_fill(void* p, size_t nbytes, uint16_t pat) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const uint32_t w = (uint32_t)pat | ((uint32_t)pat << 16);
    const uint4 v = make_uint4(w, w, w, w);
    const size_t n16 = nbytes / 16;
    for (size_t i = tid; i < n16; i += nth) ((uint4*)p)[i] = v;
    for (size_t i = n16 * 16 + tid; i < nbytes; i += nth) ((uint8_t*)p)[i] = (uint8_t)((i & 1) ? (pat >> 8) : pat);
}

// MCTS.__init__ (mcts/__init__.py:43-67) as ONE kernel.  (hipMemsetAsync nodes captured into a HIP graph were observed
// to execute on the first replay only with the ROCm runtime PyTorch bundles, so the reset is a kernel, not memsets.)
__global__ void __launch_bounds__(256) sim_init_kernel(Search s, const uint8_t* root_board, const int32_t* root_seats) {
    const size_t B = s.B, T = s.T, A = (size_t)s.S * s.S;
    if (!s.lazy) {                                // lazy: bl_sim_expand #sim resets slot sim's rows, node 0's are written below / by the root evaluation
        grid_fill(s.children, B * T * A * 2, 0xffff);
        grid_fill(s.logits, B * T * A * 2, 0x7e00);   // f16 NaN, mcts/__init__.py:56
    }
    grid_fill(s.parents, B * T * 2, 0xffff);
    grid_fill(s.relation, B * T * 2, 0xffff);
    grid_fill(s.v, B * T * 2 * 2, 0x7e00);
    grid_fill(s.w, B * T * 2 * 2, 0);
    grid_fill(s.n, B * T * 2, 0);
    grid_fill(s.rewards, B * T * 2 * 2, 0);
    grid_fill(s.terminal, B * T, 0);
    {   // every q-range word = the identity of the signed MAX (bl_device.h: BL_QBIAS)
        const size_t words = (T + 1) * (size_t)BL_QWORDS, step = (size_t)gridDim.x * blockDim.x;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += step) s.qrange[i] = BL_QBIAS;
    }
    if (s.nk) grid_fill(s.nk, B * T * 2, 0);
    if (s.fav) grid_fill(s.fav, B * T * 2, 0xffff);
    if (s.path) { const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x; if (tid < B) s.path[tid * (T + 2)] = 0; }   // no previous descent
}

// worlds = stack([world] * T) (mcts/__init__.py:62): every node slot of env b starts as a copy of the root board and
// seat.  One workgroup per env: the T*A bytes of its slots are the root board repeated, written as 32-bit words.
__global__ void __launch_bounds__(256) sim_init_worlds_kernel(Search s, const uint8_t* root_board, const int32_t* root_seats) {
    __shared__ uint8_t root[1024];
    const int b = blockIdx.x, T = s.T, A = s.S * s.S;
    for (int a = threadIdx.x; a < A; a += blockDim.x) root[a] = root_board[(long)b * A + a];
    __syncthreads();
    if (s.lazy) {
        // node 0 only: its board, its (empty) children row and -- until the root evaluation stores the real ones -- NaN logits
        for (int a = threadIdx.x; a < A; a += blockDim.x) {
            s.boards[(long)b * T * A + a] = root[a];
            s.children[(long)b * T * A + a] = (int16_t)-1;
            s.logits[(long)b * T * A + a] = 0x7e00u;
        }
        const int seat0 = root_seats[b];
        for (int t = threadIdx.x; t < T; t += blockDim.x) s.seats[(long)b * T + t] = seat0;
        return;
    }
    const long bytes = (long)T * A;
    uint8_t* dst = s.boards + (long)b * bytes;            // torch allocations are >= 16-B aligned and T*A*b keeps 1-B steps:
    const long head = (4 - ((uintptr_t)dst & 3)) & 3;     // bytes before the first aligned word of this env's block
    for (long i = threadIdx.x; i < head && i < bytes; i += blockDim.x) dst[i] = root[i % A];
    const long words = bytes > head ? (bytes - head) / 4 : 0;
    for (long k = threadIdx.x; k < words; k += blockDim.x) {
        int o = (int)((head + 4 * k) % A);
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { v |= (uint32_t)root[o] << (8 * j); o = (o + 1 == A) ? 0 : o + 1; }
        *(uint32_t*)(dst + head + 4 * k) = v;
    }
    for (long i = head + 4 * words + threadIdx.x; i < bytes; i += blockDim.x) dst[i] = root[i % A];
    const int seat = root_seats[b];
    for (int t = threadIdx.x; t < T; t += blockDim.x) s.seats[(long)b * T + t] = seat;
}

// descend #1 sees the untouched stats: every q is 0/1e-4 = 0, so its range is {0, 0}.  Separate launch: it must land
// after the grid-wide zeroing of qrange above.
__global__ void sim_init_qrange_kernel(Search s) {
    s.qrange[BL_QWORDS * 1 + 0] = ~enc(0.f) ^ BL_QBIAS;
    s.qrange[BL_QWORDS * 1 + 1] = enc(0.f) ^ BL_QBIAS;
}

// The three launches above as ONE, a workgroup per env (the lazy-reset move's form: 35 -> ~10 us of kernel time per move).  Env b's
// workgroup writes its own T slots of every (B,T) array, node 0's board / children / logits rows, its share of the q-range rows'
// zeroes -- and the thread whose share holds row 1's first slot writes descend #1's range {0, 0} there instead of zero, so no
// ordering between the zeroing and that write is needed.
__global__ void __launch_bounds__(256) sim_init_env_kernel(Search s, const uint8_t* root_board, const int32_t* root_seats) {
    const int b = blockIdx.x, T = s.T, A = s.S * s.S, tid = threadIdx.x;
    const long eb = (long)b * T;
    const int seat0 = root_seats[b];
    for (int t = tid; t < T; t += 256) {
        s.parents[eb + t] = (int16_t)-1; s.relation[eb + t] = (int16_t)-1;
        *(uint32_t*)(s.v + (eb + t) * 2) = 0x7e007e00u; *(uint32_t*)(s.w + (eb + t) * 2) = 0u;
        s.n[eb + t] = 0; *(uint32_t*)(s.rewards + (eb + t) * 2) = 0u; s.terminal[eb + t] = 0;
        if (s.nk) s.nk[eb + t] = 0;
        if (s.fav) s.fav[eb + t] = (int16_t)-1;
        s.seats[eb + t] = seat0;
    }
    if (s.path && tid == 0) s.path[(long)b * (T + 2)] = 0;                  // no previous descent
    for (int a = tid; a < A; a += 256) {
        s.boards[eb * A + a] = root_board[(long)b * A + a];
        s.children[eb * A + a] = (int16_t)-1;
        s.logits[eb * A + a] = 0x7e00u;                                    // until the root evaluation stores the real ones
    }
    const size_t words = (size_t)(T + 1) * BL_QWORDS, step = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)b * 256 + tid; i < words; i += step)
        s.qrange[i] = (i == (size_t)BL_QWORDS ? ~enc(0.f) : (i == (size_t)BL_QWORDS + 1 ? enc(0.f) : 0u)) ^ BL_QBIAS;
}

// MCTS.n_leaves (mcts/__init__.py:151-152): nodes that exist (parents != -1) and have no child.  A node has a child
// exactly when some node names it as its parent, so the (B,T) parents array suffices.  One wave per env; LDS flags.
__global__ void __launch_bounds__(BL_WAVE) sim_n_leaves_kernel(const int16_t* parents, long long* out, int T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* has_child = (uint8_t*)smem;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int16_t* p = parents + (long)b * T;
    for (int t = lane; t < T; t += BL_WAVE) has_child[t] = 0;
    __syncthreads();
    for (int t = lane; t < T; t += BL_WAVE) { const int q = p[t]; if (q >= 0) has_child[q] = 1; }
    __syncthreads();
    int count = 0;
    for (int t = lane; t < T; t += BL_WAVE) count += (p[t] != -1) && !has_child[t];
    count = wave_sum_i32(count);
    if (lane == 0) out[b] = count;
}

}  // namespace bl

using namespace bl;

int bl_expand2_launch(const Search& ss, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid, int32_t* leaf_seats,
                      unsigned long long* counters, int fast, int waves, int deep_thresh, int envs, int help_thresh, hipStream_t stream);     // bl_expand.hip
int bl_fold_selftest(int use_fast, hipStream_t stream);
int bl_expand_rows_launch(const Search& ss, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid, int32_t* leaf_seats,
                          int waves, hipStream_t stream);      // bl_rows.hip

extern "C" {

static int search_check(const bl_search_t* s) {
    if (!s || !s->logits || !s->v || !s->w || !s->n || !s->children || !s->parents || !s->relation || !s->rewards ||
        !s->terminal || !s->boards || !s->seats || !s->c_puct || !s->qrange || !s->exp_table) return BL_EINVAL;
    if (s->B <= 0 || s->T <= 0 || s->boardsize <= 0) return BL_EINVAL;
    if (s->boardsize > 32 || s->T > 32767) return BL_ETOOBIG;
    return BL_OK;
}

static Search to_search(const bl_search_t* s) {
    return Search{(uint16_t*)s->logits, (uint16_t*)s->v, (uint16_t*)s->w, s->n, s->children, s->parents, s->relation,
                  (uint16_t*)s->rewards, s->terminal, s->boards, s->seats, (const uint16_t*)s->c_puct, s->qrange,
                  s->exp_table, s->B, s->T, s->boardsize, s->obs_f16, s->path, s->order, s->prio_thresh,
                  s->cpi, s->cca, s->nk, s->fav, s->n_active, s->tune.lazy_init, s->tune.powf_libm};
}

static int sim_expand_impl(const bl_search_t* s, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid,
                           int32_t* leaf_seats, unsigned long long* counters, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!rands || !leaves || !obs || !valid || !leaf_seats || sim < 1 || sim >= s->T) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize;
    const bl_tune_t& tune = s->tune;
    if (!tune.expand_legacy && !tune.group && s->cpi && s->cca && s->nk) {
        // compacted rows + node statistics in registers + one DPP chain per level (bl_expand.hip); shapes outside its
        // template set (A > 384 or T > 256) fall through to the general kernel
        // waves per env: two fill the chip's 8192 wave slots at 4096 envs; up to 1024 envs four fit twice over and a batch then
        // covers four guessed levels (13x13, 1024 envs x 256 sims: 44.2 -> 40.3 ms per move; 9x9 at 2048 envs: no gain, at 4096 a loss)
        // From 16384 envs on the launch is several times the chip's wave slots and what binds is VALU issue, not one env's chain: the
        // helper wave's guessed evaluations (5.6 evaluated nodes for 4.8 needed) then cost more than they hide -- one wave per env
        // (9x9 x 64 sims, us per launch with 2 / 1 waves: 8192 envs 77.7 / 78.2, 16384 125.0 / 113.3, 32768 221.4 / 185.6;
        // profiles/r06_envs_sweep_waves.txt).  Results do not depend on the choice.
        const int waves = tune.expand_waves ? tune.expand_waves : (s->B <= 1024 ? 4 : s->B < 16384 ? 2 : 1);
        // expand_waves = 16: four ENVS per wave, one per DPP row (bl_rows.hip; expand_deep then = the descent launch's waves, 0 = its
        // default).  Built in round 6 for the VALU-bound regime above, bit-exact, and SLOWER there (32768 envs: 244 us against 186;
        // profiles/r06_rows_kernel.txt): opt-in only, parity-tested in a child process.
        if (!counters && tune.expand_waves == 16) {
            rc = bl_expand_rows_launch(to_search(s), sim, rands, leaves, obs, valid, leaf_seats, tune.expand_deep, (hipStream_t)stream);
            if (rc != BL_ETOOBIG) return rc;
        }
        // (expand_envs > 1 was round 4's shared-workgroup kernel: removed, BL_EINVAL)
        const int envs = tune.expand_envs ? tune.expand_envs : 1;
        rc = bl_expand2_launch(to_search(s), sim, rands, leaves, obs, valid, leaf_seats, counters, tune.fold_fast != 0,
                               waves, tune.expand_deep, tune.expand_waves ? 1 : envs, 0, (hipStream_t)stream);
        if (rc != BL_ETOOBIG) return rc;
    }
    const int G = pick_group(s->B, A, tune.group), K = pick_k(A, G);
    const int per = lds_bytes(A, true);
    const int blocks = (s->B + 64 / G - 1) / (64 / G);
    Search ss = to_search(s);
    if (counters) {
#define CALL(g, k) hipLaunchKernelGGL((sim_expand_kernel<g, k, true>), dim3(blocks), dim3(64), (size_t)per * (64 / g), \
                                      (hipStream_t)stream, ss, sim, (const uint16_t*)rands, leaves, (void*)obs, valid, leaf_seats, counters)
        BL_DISPATCH_GK(G, K, CALL)
#undef CALL
    } else {
#define CALL(g, k) hipLaunchKernelGGL((sim_expand_kernel<g, k, false>), dim3(blocks), dim3(64), (size_t)per * (64 / g), \
                                      (hipStream_t)stream, ss, sim, (const uint16_t*)rands, leaves, (void*)obs, valid, leaf_seats, nullptr)
        BL_DISPATCH_GK(G, K, CALL)
#undef CALL
    }
    return check_launch();
}

int bl_sim_expand(const bl_search_t* s, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid,
                  int32_t* leaf_seats, bl_stream_t stream) {
    return sim_expand_impl(s, sim, rands, leaves, obs, valid, leaf_seats, nullptr, stream);
}

int bl_sim_expand_counted(const bl_search_t* s, int sim, const void* rands, int16_t* leaves, void* obs, uint8_t* valid,
                          int32_t* leaf_seats, unsigned long long* counters, bl_stream_t stream) {
    if (!counters) return BL_EINVAL;
    return sim_expand_impl(s, sim, rands, leaves, obs, valid, leaf_seats, counters, stream);
}

int bl_sim_backup(const bl_search_t* s, int sim, const int16_t* leaves, const void* leaf_logits, int logits_dtype,
                  const void* leaf_v, int v_dtype, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!leaves || !leaf_logits || !leaf_v || sim < 1 || sim >= s->T) return BL_EINVAL;
    const int blocks = (s->B + 3) / 4;
    hipLaunchKernelGGL(sim_backup_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, to_search(s), sim, leaves,
                       leaf_logits, logits_dtype, leaf_v, v_dtype);
    if (s->cpi && s->cca && s->nk) hipLaunchKernelGGL(compact_rows_kernel, dim3(s->B), dim3(64), 0, (hipStream_t)stream, to_search(s), leaves);
    return check_launch();
}

int bl_sim_compact(const bl_search_t* s, const int16_t* leaves, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!s->cpi || !s->cca || !s->nk) return BL_EINVAL;
    hipLaunchKernelGGL(compact_rows_kernel, dim3(s->B), dim3(64), 0, (hipStream_t)stream, to_search(s), leaves);
    return check_launch();
}

static int sim_finish_impl(int f32, const bl_search_t* s, int sim, const int16_t* leaves, const void* policy_raw, const void* value_raw,
                           const uint8_t* valid, const int32_t* leaf_seats, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!leaves || !policy_raw || !value_raw || !valid || !leaf_seats || !s->path || sim < 1 || sim >= s->T) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize;
    int np2 = 1; while (np2 < A) np2 *= 2;
    const int W = np2 < 64 ? np2 : 64, iters = np2 / W;
    if (iters > 16) return BL_ETOOBIG;
    if (s->T <= 256 && iters <= 4) {
        // the common shapes: the version with its loads batched (sim_finish_fast_kernel)
        const int KT = (s->T + 63) / 64;
        hipStream_t hs = (hipStream_t)stream;
        const Search ss = to_search(s);
#define FIN(R_, K_, I_) hipLaunchKernelGGL((sim_finish_fast_kernel<R_, K_, I_>), dim3(s->B), dim3(64), 0, hs, ss, sim, leaves, (const R_*)policy_raw, (const R_*)value_raw, valid, leaf_seats, W)
#define FIN_I(R_, K_) { if (iters == 1) FIN(R_, K_, 1); else if (iters == 2) FIN(R_, K_, 2); else FIN(R_, K_, 4); }
#define FIN_K(R_) { if (KT == 1) FIN_I(R_, 1) else if (KT == 2) FIN_I(R_, 2) else FIN_I(R_, 4) }
        if (f32) FIN_K(float) else FIN_K(uint16_t)
#undef FIN_K
#undef FIN_I
#undef FIN
        return check_launch();
    }
    if (f32)
        hipLaunchKernelGGL(sim_finish_kernel<float>, dim3(s->B), dim3(64), 0, (hipStream_t)stream, to_search(s), sim, leaves,
                           (const float*)policy_raw, (const float*)value_raw, valid, leaf_seats, W, iters);
    else
        hipLaunchKernelGGL(sim_finish_kernel<uint16_t>, dim3(s->B), dim3(64), 0, (hipStream_t)stream, to_search(s), sim, leaves,
                           (const uint16_t*)policy_raw, (const uint16_t*)value_raw, valid, leaf_seats, W, iters);
    return check_launch();
}

int bl_sim_finish(const bl_search_t* s, int sim, const int16_t* leaves, const void* policy_raw, const void* value_raw,
                  const uint8_t* valid, const int32_t* leaf_seats, bl_stream_t stream) {
    return sim_finish_impl(0, s, sim, leaves, policy_raw, value_raw, valid, leaf_seats, stream);
}

int bl_sim_finish_f32(const bl_search_t* s, int sim, const int16_t* leaves, const float* policy_raw, const float* value_raw,
                      const uint8_t* valid, const int32_t* leaf_seats, bl_stream_t stream) {
    return sim_finish_impl(1, s, sim, leaves, policy_raw, value_raw, valid, leaf_seats, stream);
}

static int plant_root_impl(const bl_search_t* s, const float* policy_raw, const float* value_raw, const uint8_t* valid,
                           const int32_t* seats, const float* draw, float eps, int gamma, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!policy_raw || !value_raw || !valid || !seats || (gamma && !draw)) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize;
    int np2 = 1; while (np2 < A) np2 *= 2;
    const int W = np2 < 64 ? np2 : 64, iters = np2 / W;
    if (iters > 16 || (gamma && A >= 128)) return BL_ETOOBIG;
    const int Wr = gamma ? (last_pow2_le(A) < 64 ? last_pow2_le(A) : 64) : 0;
    hipLaunchKernelGGL(sim_plant_root_kernel, dim3(s->B), dim3(64), 0, (hipStream_t)stream, to_search(s), policy_raw, value_raw,
                       valid, seats, draw, eps, W, iters, Wr);
    return check_launch();
}

int bl_sim_plant_root(const bl_search_t* s, const float* policy_raw, const float* value_raw, const uint8_t* valid,
                      const int32_t* seats, const float* draw, float eps, bl_stream_t stream) {
    return plant_root_impl(s, policy_raw, value_raw, valid, seats, draw, eps, 0, stream);
}

int bl_sim_plant_root_gamma(const bl_search_t* s, const float* policy_raw, const float* value_raw, const uint8_t* valid,
                            const int32_t* seats, const float* gamma, float eps, bl_stream_t stream) {
    return plant_root_impl(s, policy_raw, value_raw, valid, seats, gamma, eps, 1, stream);
}

int bl_sim_root(const bl_search_t* s, int sim, void* probs, const void* log_table, void* logits, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!probs || sim < 1 || sim > s->T || (logits && !log_table)) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize;
    Tree m{(const uint16_t*)s->logits, (const uint16_t*)s->w, s->n, (const uint16_t*)s->c_puct, s->seats, s->terminal,
           s->children, s->qrange + (long)BL_QWORDS * sim, s->exp_table, s->B, s->T, A, 2, 1, s->tune.powf_libm};
    const int G = pick_group(s->B, A, s->tune.group), K = pick_k(A, G);
    const int per = lds_bytes(A, false);
    const int blocks = (s->B + 64 / G - 1) / (64 / G);
#define CALL(g, k) hipLaunchKernelGGL((root_kernel<g, k>), dim3(blocks), dim3(64), (size_t)per * (64 / g), \
                                      (hipStream_t)stream, m, (uint16_t*)probs, (const uint16_t*)log_table, (uint16_t*)logits)
    BL_DISPATCH_GK(G, K, CALL)
#undef CALL
    return check_launch();
}

int bl_sim_n_leaves(const bl_search_t* s, long long* out, bl_stream_t stream) {
    if (int rc = search_check(s)) return rc;
    if (!out) return BL_EINVAL;
    hipLaunchKernelGGL(sim_n_leaves_kernel, dim3(s->B), dim3(64), (size_t)((s->T + 15) & ~15), (hipStream_t)stream, s->parents, out, s->T);
    return check_launch();
}

int bl_sim_init(const bl_search_t* s, const uint8_t* root_board, const int32_t* root_seats, bl_stream_t stream) {
    int rc = search_check(s);
    if (rc) return rc;
    if (!root_board || !root_seats) return BL_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    if (s->tune.lazy_init) {             // the (B,T,A) arrays are reset slot by slot by the simulations: everything else in one launch
        hipLaunchKernelGGL(sim_init_env_kernel, dim3(s->B), dim3(256), 0, hs, to_search(s), root_board, root_seats);
        return check_launch();
    }
    hipLaunchKernelGGL(sim_init_kernel, dim3(2048), dim3(256), 0, hs, to_search(s), root_board, root_seats);
    hipLaunchKernelGGL(sim_init_worlds_kernel, dim3(s->B), dim3(256), 0, hs, to_search(s), root_board, root_seats);
    hipLaunchKernelGGL(sim_init_qrange_kernel, dim3(1), dim3(1), 0, hs, to_search(s));
    return check_launch();
}

}  // extern "C"
