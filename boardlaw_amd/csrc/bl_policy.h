// bl_policy.h -- the general (lanes-per-env) policy evaluation shared by bl_search.hip and bl_sim.hip: the view of the reference's
// `struct MCTS`, the per-group LDS carve-up, policy() / newton_search / prob() (boardlaw/mcts/cpp/cuda.cu:8-99) for a group of G lanes
// and for a whole wave, descend_kernel and root_kernel (cuda.cu:107-182), the q-range publication and the backup walk
// (cuda.cu:205-236).  Split out of bl_search.hip in round 6; the header comment of that file describes the lane mapping.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"

#pragma clang fp contract(off)

namespace bl {


// View of the reference's `struct MCTS` (boardlaw/mcts/cpp/common.h:25-33) plus what transition_q needs.
struct Tree {
    const uint16_t* logits;   // (B,T,A) f16
    const uint16_t* w;        // (B,T,S) f16
    const int16_t* n;         // (B,T)
    const uint16_t* c_puct;   // (B) f16
    const void* seats;        // (B,T) i16 (reference struct) or i32 (worlds.seats, fused path)
    const uint8_t* terminal;  // (B,T)
    const int16_t* children;  // (B,T,A)
    const uint32_t* qrange;   // BL_QWORDS
    const float* exp_table;   // 65536
    int B, T, A, S;
    int seats_i32;
    int powf_libm;            // bl_tune_t.powf_libm (bl_device.h: g_denominator)
};

__device__ __forceinline__ int load_seat(const Tree& m, long i) {
    return m.seats_i32 ? ((const int32_t*)m.seats)[i] : (int)((const int16_t*)m.seats)[i];
}


// ------------------------------------------------------------------------------------------------------------------
// Per-group LDS carve-up (bytes, every array 16-B aligned):  s[A] f32 | g[A] f32 | child[A] i16 | info[A] u8 | cells[A] u8
// ------------------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int lds_bytes(int A, bool with_cells) { return 2 * al16(4 * A) + al16(2 * A) + al16(A) + (with_cells ? al16(A) : 0); }

struct GroupLds {
    float* s;        // Newton terms lambda*pi/(alpha-q)          == prob(a) once converged (cuda.cu:23-25)
    float* g;        // derivative terms -lambda*pi/(alpha-q)^2
    int16_t* child;  // children[b,t,:] of the node being evaluated
    uint8_t* info;   // per child: bit0 = terminal[b,child], bits1.. = seats[b,child] (prefetched for the next level)
    uint8_t* cells;  // board scratch for the fused step
    __device__ __forceinline__ GroupLds(char* base, int A) {
        s = (float*)base; g = (float*)(base + al16(4 * A)); child = (int16_t*)(base + 2 * al16(4 * A));
        info = (uint8_t*)(base + 2 * al16(4 * A) + al16(2 * A)); cells = info + al16(A);
    }
};

// One serial fold over a = 0..A-1 in the reference's order, leaving the running totals in place of the terms.
// Lane 0 of the group folds the s terms, lane 1 the g terms (same instruction stream, different array).  Float addition
// is not associative, so this 1-add-per-action dependent chain IS the algorithm's critical path; everything else in
// the kernel is arranged to keep other instructions out of it.
__device__ __forceinline__ float serial_prefix(float* arr, int A) {
    float acc = 0.f;
    float4* v4 = (float4*)arr;
    int a = 0;
#pragma unroll 4
    for (; a + 3 < A; a += 4) {
        float4 x = v4[a >> 2];
        acc += x.x; x.x = acc;
        acc += x.y; x.y = acc;
        acc += x.z; x.z = acc;
        acc += x.w; x.w = acc;
        v4[a >> 2] = x;
    }
    for (; a < A; a++) { acc += arr[a]; arr[a] = acc; }
    return acc;
}

// ------------------------------------------------------------------------------------------------------------------
// policy(): cuda.cu:70-99 + newton_search cuda.cu:35-68 + the action draw of descend_kernel cuda.cu:157-176, for the
// group's env b at node t (whose mover is `seat`).  On return prob[k] == prob(a = k*G+gl) for the final alpha
// (cuda.cu:23-25), L.child[a] == children[b,t,a], L.info[a] describes that child, and the return value is the sampled
// edge for uniform r (group-uniform).  `go` is group-uniform; idle groups only keep the wave's barriers company.
// ------------------------------------------------------------------------------------------------------------------
template <int G, int K, bool COUNT>
__device__ __forceinline__ int policy_eval(const Tree& m, int b, int t, int seat, bool go, int gl, float lo, float rden,
                                           float r, const GroupLds& L, float (&prob)[K], unsigned long long* counters) {
    const int A = m.A, T = m.T, S = m.S;
    float top[K], q[K];
    int child[K];
    uint16_t lb[K];
    int Nloc = 0, nch = 0;
    const long envbase = (long)b * T;
    const long row = (envbase + t) * A;
    long long tp0 = 0, tdiv = 0, tfold = 0, tupd = 0;     // phase clocks, COUNT builds only
    if (COUNT) tp0 = clock64();
    // round trip 1: the node's two rows, coalesced across the group
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int a = k * G + gl;
        child[k] = -1; lb[k] = 0;
        if (go && a < A) { child[k] = m.children[row + a]; lb[k] = m.logits[row + a]; }
    }
    // round trip 2: per-child statistics + what the next level needs to know about each child
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int a = k * G + gl;
        float pi = 0.f, qa = 0.f;
        if (go && a < A) {
            pi = m.exp_table[lb[k]];
            L.child[a] = (int16_t)child[k];
            if (child[k] > -1) {
                const long i = envbase + child[k];
                const float wv = h2f(m.w[i * S + seat]);
                const int nv = m.n[i];
                L.info[a] = (uint8_t)((m.terminal[i] ? 1 : 0) | (load_seat(m, i) << 1));
                const float q32 = wv / ((float)nv + 1.e-4f);
                qa = h2f(f2h((q32 - lo) / rden));
                Nloc += nv;
                nch++;
            } else {
                Nloc += 1;
            }
        }
        top[k] = pi; q[k] = qa; prob[k] = 0.f;
    }
    const int N = gsum<G>(Nloc);
    const float lam = go ? (h2f(m.c_puct[b]) * (float)N) / (float)(unsigned)(N + A) : 0.f;
    float alpha = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
        top[k] = lam * top[k];
        if (k * G + gl < A) alpha = fmaxf(alpha, q[k] + fmaxf(top[k], 1.e-4f));
    }
    alpha = gmaxf<G>(alpha);
    long long tload = 0;
    if (COUNT) { tload = clock64() - tp0; }

    float err = INFINITY;
    bool conv = !go;      // group-uniform
    int iters = 0;
    for (int it = 0; it < 101; it++) {
        long long ti0 = 0;
        if (COUNT) ti0 = clock64();
        // iteration 100 only happens for groups that ran out of Newton steps: their alpha moved after the last fold
        // (cuda.cu:48-65), so the probabilities are evaluated once more at the final alpha for the draw.
        if (!__any(!conv)) break;
        if (!conv) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                const int a = k * G + gl;
                if (a < A) {
                    const float bot = alpha - q[k];
                    prob[k] = top[k] / bot;
                    L.s[a] = prob[k];
                    L.g[a] = (-top[k]) / g_denominator(bot, m.powf_libm);
                }
            }
        }
        __syncthreads();
        long long ti1 = 0;
        if (COUNT) { ti1 = clock64(); tdiv += ti1 - ti0; }
        float acc = 0.f;
        if (!conv && gl < 2) acc = serial_prefix(gl == 0 ? L.s : L.g, A);
        const float Ssum = __shfl(acc, 0, G), gsum_ = __shfl(acc, 1, G);
        long long ti2 = 0;
        if (COUNT) { ti2 = clock64(); tfold += ti2 - ti1; }
        if (!conv) {
            if (it == 100) { conv = true; }
            else {
                iters++;
                const float ne = Ssum - 1.f;
                if ((ne < 1e-3f) || (err == ne)) { conv = true; }
                else { alpha -= ne / gsum_; err = ne; }
            }
        }
        __syncthreads();
        if (COUNT) tupd += clock64() - ti2;
    }
    // The draw, cuda.cu:157-176: first a (ascending) with prob > 0 and running total >= r, else the last a with prob > 0.
    // L.s now holds the running totals in the reference's summation order; every lane tests its own actions.
    int first = 0x7fffffff, last = -1;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int a = k * G + gl;
        if (go && a < A) {
            const bool pos = prob[k] > 0.f;
            if (pos && L.s[a] >= r && a < first) first = a;
            if (pos) last = a;
        }
    }
#pragma unroll
    for (int msk = G / 2; msk > 0; msk >>= 1) {
        first = min(first, __shfl_xor(first, msk, G));
        last = max(last, __shfl_xor(last, msk, G));
    }
    if (COUNT && go) {
        // diagnostics, per env, plain stores (atomics would perturb the memory timings being measured):
        // {levels, Newton iterations, most iterations in a level, child look-ups, clocks: loads, terms, folds, update}
        const int nc = gsum<G>(nch);
        if (gl == 0) {
            unsigned long long* e = counters + 12 * (long)b;
            e[0] += 1; e[1] += iters; if ((unsigned long long)iters > e[2]) e[2] = iters; e[3] += nc;
            e[4] += tload; e[5] += tdiv; e[6] += tfold; e[7] += tupd;
        }
    }
    return first != 0x7fffffff ? first : last;
}

// ------------------------------------------------------------------------------------------------------------------
// One-wave-per-env (G == 64) specialisation of policy_eval: no LDS, no barriers.
//
// The serial fold runs across LANES with DPP: after  x <- t  and  x[lane 0] <- carry + t[0],  the step
//     x[i] <- x[i-1] + t[i]   for every lane i >= 1 at once     (v_add_f32_dpp ... wave_shr:1, lane 0 keeps its value)
// applied j times makes lanes 0..j hold the reference's running total  ((carry + t0) + t1) + ...  exactly -- each
// lane's last update reads a neighbour that is already final, and later updates recompute the same sum.  63 steps
// finish a 64-action register; the totals stay in registers, which is what the draw needs.  S and g chains interleave,
// filling each other's DPP wait states.
// ------------------------------------------------------------------------------------------------------------------

// One step of both chains.  The ISA asks for 2 wait states between a VALU write and a DPP read of the same VGPR (there is
// no interlock): each chain's next step is separated from its previous one by the other chain's instruction plus one
// s_nop.  (tools/micro/dpp_hazard.hip measures that the other chain's instruction alone is enough on gfx950; this
// general kernel -- bl_mcts_descend/root, bl_sim_root, and bl_sim_expand outside bl_expand.hip's shapes -- does not rely
// on it.  The hot path's fold lives in bl_expand.hip and picks its padding after a device self-test.)
#define BL_FOLD_STEP "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t"
// 8 fold steps of both chains.  The leading s_nop covers the VALU-write -> DPP-read hazard against whatever wrote x/y.
#define BL_FOLD8(x, y, ts, tg)                                                                              \
    asm volatile("s_nop 1\n\t" BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP \
                 BL_FOLD_STEP BL_FOLD_STEP : "+v"(x), "+v"(y) : "v"(ts), "v"(tg))
#define BL_FOLD16(x, y, ts, tg)                                                                             \
    asm volatile("s_nop 1\n\t" BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP \
                 BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP     \
                 BL_FOLD_STEP BL_FOLD_STEP BL_FOLD_STEP : "+v"(x), "+v"(y) : "v"(ts), "v"(tg))

template <int K, bool COUNT, bool WANT_PROB>
__device__ __forceinline__ int policy_eval_wave(const Tree& m, int b, int t, int seat, float lo, float rden, float r,
                                                const GroupLds& L, float (&prob)[K], int& next_child, int& next_info,
                                                unsigned long long* counters) {
    const int A = m.A, T = m.T, S = m.S;
    const int lane = threadIdx.x & 63;
    float top[K], q[K], tg[K];
    int child[K], info[K];
    uint16_t lb[K];
    int Nloc = 0, nch = 0;
    const long envbase = (long)b * T;
    const long row = (envbase + t) * A;
    long long tp0 = 0, tdiv = 0, tfold = 0, tupd = 0;
    if (COUNT) tp0 = clock64();
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int a = k * 64 + lane;
        child[k] = -1; lb[k] = 0; info[k] = 0;
        if (a < A) { child[k] = m.children[row + a]; lb[k] = m.logits[row + a]; }
    }
    long long trt1 = 0, tq = 0;
    if (COUNT) { __builtin_amdgcn_s_waitcnt(0); trt1 = clock64() - tp0; }     // rows of the node have landed
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int a = k * 64 + lane;
        float pi = 0.f, qa = 0.f;
        if (a < A) {
            pi = m.exp_table[lb[k]];
            if (child[k] > -1) {
                const long i = envbase + child[k];
                const float wv = h2f(m.w[i * S + seat]);
                const int nv = m.n[i];
                info[k] = (m.terminal[i] ? 1 : 0) | (load_seat(m, i) << 1);
                const float q32 = wv / ((float)nv + 1.e-4f);
                qa = h2f(f2h((q32 - lo) / rden));
                Nloc += nv;
                nch++;
            } else {
                Nloc += 1;
            }
        }
        top[k] = pi; q[k] = qa; prob[k] = 0.f; tg[k] = 0.f;
    }
    if (COUNT) { __builtin_amdgcn_s_waitcnt(0); tq = clock64() - tp0; }       // + children's statistics and q
    const int N = wave_sum_i32(Nloc);
    const float lam = (h2f(m.c_puct[b]) * (float)N) / (float)(unsigned)(N + A);
    float alpha = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) {
        top[k] = lam * top[k];
        if (k * 64 + lane < A) alpha = fmaxf(alpha, q[k] + fmaxf(top[k], 1.e-4f));
    }
    alpha = wave_max_f32(alpha);

    // Compact the actions whose terms are not identically zero (top != 0) to the front, keeping their order.  A term
    // with top == 0 (an illegal move: logit -inf) contributes s = +0 and g = -0 to the folds, and x + (+-0) == x for
    // every partial sum the folds can hold, so dropping those steps leaves every rounding unchanged -- and a mid-game
    // board's legal moves usually fit one 64-lane register, halving the divisions and shortening the serial chain.
    float ctop[K], cq[K];
    int ca[K];
    int n_c = 0;
    {
        int rank[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool present = (k * 64 + lane < A) && (top[k] != 0.f);
            const unsigned long long mk = __ballot(present);
            rank[k] = present ? n_c + __builtin_popcountll(mk & ((1ull << lane) - 1ull)) : -1;
            n_c += __builtin_popcountll(mk);
        }
#pragma unroll
        for (int k = 0; k < K; k++) if (rank[k] >= 0) { L.s[rank[k]] = top[k]; L.g[rank[k]] = q[k]; L.child[rank[k]] = (int16_t)(k * 64 + lane); }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = k * 64 + lane;
            const bool in = j < n_c;
            ctop[k] = in ? L.s[j] : 0.f; cq[k] = in ? L.g[j] : 0.f; ca[k] = in ? (int)L.child[j] : -1;
        }
        __syncthreads();
    }
    long long tload = 0;
    if (COUNT) tload = clock64() - tp0;

    float err = INFINITY;
    int iters = 0;
    float tot[K], cprob[K];
#pragma unroll
    for (int k = 0; k < K; k++) { tot[k] = 0.f; cprob[k] = 0.f; }
    const int last_k = (n_c - 1) >> 6, last_lane = (n_c - 1) & 63;
    for (int it = 0; it < 101 && n_c > 0; it++) {
        long long ti0 = 0, ti1 = 0, ti2 = 0;
        if (COUNT) ti0 = clock64();
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k <= last_k) {
                const float bot = alpha - cq[k];
                const bool in = k * 64 + lane < n_c;
                cprob[k] = in ? ctop[k] / bot : 0.f;        // lanes past the end fold +0: harmless to every earlier lane
                tg[k] = in ? (-ctop[k]) / g_denominator(bot, m.powf_libm) : 0.f;
            }
        }
        if (COUNT) { ti1 = clock64(); tdiv += ti1 - ti0; }
        float cs = 0.f, cg = 0.f, Ssum = 0.f, gsum_ = 0.f;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k <= last_k) {
                float x = cprob[k], y = tg[k];
                if (lane == 0) { x = cs + x; y = cg + y; }
                const int steps = (k == last_k) ? last_lane : 63;
                int j = 0;
                for (; j + 8 < steps; j += 16) BL_FOLD16(x, y, cprob[k], tg[k]);     // extra steps past `steps` are harmless
                for (; j < steps; j += 8) BL_FOLD8(x, y, cprob[k], tg[k]);
                tot[k] = x;
                if (k == last_k) { Ssum = readlane_f(x, last_lane); gsum_ = readlane_f(y, last_lane); }
                else { cs = readlane_f(x, 63); cg = readlane_f(y, 63); }
            }
        }
        if (COUNT) { ti2 = clock64(); tfold += ti2 - ti1; }
        if (it == 100) break;     // alpha had moved after the 100th fold (cuda.cu:48-65): this pass only refreshed prob/tot
        iters++;
        const float ne = Ssum - 1.f;
        if ((ne < 1e-3f) || (err == ne)) break;
        alpha -= ne / gsum_; err = ne;
        if (COUNT) tupd += clock64() - ti2;
    }
    // The draw, cuda.cu:157-176, on the running totals each lane holds for its own (compacted) actions: the first with
    // prob > 0 and total >= r, else the last with prob > 0.  Dropped actions have prob == 0 and can never be drawn.
    int action = -1, lastpos = -1;
#pragma unroll
    for (int k = 0; k < K; k++) {
        if (k <= last_k) {
            const bool pos = (k * 64 + lane < n_c) && cprob[k] > 0.f;
            const unsigned long long hit = __ballot(pos && tot[k] >= r), anyp = __ballot(pos);
            if (action < 0 && hit) action = __builtin_amdgcn_readlane(ca[k], __builtin_ctzll(hit));
            if (anyp) lastpos = __builtin_amdgcn_readlane(ca[k], 63 - __builtin_clzll(anyp));
        }
    }
    if (action < 0) action = lastpos;
    next_child = -1; next_info = 0;
    if (action >= 0) {
#pragma unroll
        for (int k = 0; k < K; k++) if ((action >> 6) == k) {
            next_child = __builtin_amdgcn_readlane(child[k], action & 63);
            next_info = __builtin_amdgcn_readlane(info[k], action & 63);
        }
    }
    if (WANT_PROB) {
        // un-compact prob(a) for the caller (root read-out): every action reads its compacted slot back
        int rank2 = 0;
#pragma unroll
        for (int k = 0; k < K; k++) { const int j = k * 64 + lane; if (j < n_c) L.s[j] = cprob[k]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < K; k++) {
            const bool present = (k * 64 + lane < A) && (top[k] != 0.f);
            const unsigned long long mk = __ballot(present);
            prob[k] = present ? L.s[rank2 + __builtin_popcountll(mk & ((1ull << lane) - 1ull))] : 0.f;
            rank2 += __builtin_popcountll(mk);
        }
        // an action whose top is 0 has prob 0/(alpha-q) = +0 unless alpha == q (0/0 = NaN in the reference);
        // alpha >= q + 1e-4 always (cuda.cu:37-41), so +0 it is.
    }
    if (COUNT) {
        const int nc = wave_sum_i32(nch);
        if (lane == 0) {
            unsigned long long* e = counters + 12 * (long)b;
            e[0] += 1; e[1] += iters; if ((unsigned long long)iters > e[2]) e[2] = iters; e[3] += nc;
            e[4] += tload; e[5] += tdiv; e[6] += tfold; e[7] += tupd; e[10] += trt1; e[11] += tq;
        }
    }
    return action;
}

// descend_kernel's per-env loop, cuda.cu:138-182.  Returns group-uniform (parent, action, next) where next ==
// children[b,parent,action] (-1 for an unexpanded edge, a terminal node's id otherwise).
template <int G, int K, bool COUNT>
__device__ __forceinline__ void descend_group(const Tree& m, int b, bool act, int gl, const uint16_t* rands,
                                              const GroupLds& L, unsigned long long* counters, int16_t* path,
                                              int& parent_out, int& action_out, int& next_out, int& depth_out) {
    float lo, hi;
    load_qrange(m.qrange, lo, hi);
    const float rden = hi - lo + 1.e-4f;
    const long envbase = (long)b * m.T;
    int t = 0, parent = 0, action = -1;
    bool term = false;
    int seat = 0;
    if (act) { term = m.terminal[envbase]; seat = load_seat(m, envbase); }
    // A root-to-leaf path in a T-slot tree has at most T nodes; the bound only matters for a corrupted tree, where the
    // reference's while(true) (cuda.cu:149) would spin forever.
    int nlevels = 0;
    for (int depth = 0; depth < m.T; depth++) {
        const bool go = act && (t != -1) && !term;
        if (!__any(go)) break;
        const float r = go ? h2f(rands[envbase + t]) : 0.f;
        if (go) {
            // the visited nodes, root first: bl_sim_finish walks them without chasing parents[]
            if (path && gl == 0) path[1 + depth] = (int16_t)t;
            nlevels = depth + 1;
        }
        float prob[K];
        if constexpr (G == 64) {
            // one env per wave: `go` is wave-uniform, the whole wave is here
            int nchild, ninfo;
            action = policy_eval_wave<K, COUNT, false>(m, b, t, seat, lo, rden, r, L, prob, nchild, ninfo, counters);
            parent = t;
            if (action < 0) { act = false; }
            else { t = nchild; term = (t != -1) && (ninfo & 1); seat = ninfo >> 1; }
        } else {
            const int a = policy_eval<G, K, COUNT>(m, b, t, seat, go, gl, lo, rden, r, L, prob, counters);
            if (go) {
                action = a;
                parent = t;
                if (action < 0) { act = false; }   // reference would index children[b][t][-1]; unreachable with a finite logit
                else {
                    t = L.child[action];
                    const int info = L.info[action];
                    term = (t != -1) && (info & 1);
                    seat = info >> 1;
                }
            }
            __syncthreads();
        }
    }
    parent_out = parent; action_out = action; next_out = t; depth_out = nlevels;
}

template <int G, int K, bool COUNT>
__global__ void __launch_bounds__(BL_WAVE) descend_kernel(Tree m, const uint16_t* rands, int16_t* parents,
                                                          int16_t* actions, unsigned long long* counters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int b = blockIdx.x * (BL_WAVE / G) + grp;
    const GroupLds L(smem + (size_t)grp * lds_bytes(m.A, false), m.A);
    int parent, action, nxt, nlev;
    descend_group<G, K, COUNT>(m, b, b < m.B, gl, rands, L, counters, nullptr, parent, action, nxt, nlev);
    if (b < m.B && gl == 0) { parents[b] = (int16_t)parent; actions[b] = (int16_t)action; }
}

// root_kernel, cuda.cu:107-118
template <int G, int K>
__global__ void __launch_bounds__(BL_WAVE) root_kernel(Tree m, uint16_t* probs, const uint16_t* log_table, uint16_t* logits) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int b = blockIdx.x * (BL_WAVE / G) + grp;
    const GroupLds L(smem + (size_t)grp * lds_bytes(m.A, false), m.A);
    float lo, hi;
    load_qrange(m.qrange, lo, hi);
    const bool go = b < m.B;
    const int seat = go ? load_seat(m, (long)b * m.T) : 0;
    float prob[K];
    if constexpr (G == 64) {
        int c, i;
        if (go) policy_eval_wave<K, false, true>(m, b, 0, seat, lo, hi - lo + 1.e-4f, 2.f, L, prob, c, i, nullptr);
    } else {
        policy_eval<G, K, false>(m, b, 0, seat, go, gl, lo, hi - lo + 1.e-4f, 2.f, L, prob, nullptr);
    }
    if (go) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int a = k * G + gl;
            if (a < m.A) {
                const uint16_t pb = f2h(prob[k]);
                probs[(long)b * m.A + a] = pb;
                if (logits) logits[(long)b * m.A + a] = log_table[pb];       // MCTS.root's r.log() (mcts/__init__.py:147), per f16 bit pattern
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// transition_q's range.  One thread per (b,t) node; per-wave reduce; one conditional atomic pair per wave.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void qrange_publish(uint32_t* qr, uint32_t nmin, uint32_t vmax, int slot) {
    nmin = gmaxu<64>(nmin); vmax = gmaxu<64>(vmax);
    if ((threadIdx.x & 63) == 0) {
        uint32_t* p = qr + BL_QSTRIDE * slot;
        q_atomic_max_checked(p, nmin);
        q_atomic_max_checked(p + 1, vmax);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backup_kernel, cuda.cu:205-236: one lane per (env, seat); lane s == 0 also owns n.
// n += 1 sits inside the seat loop in the reference, so a visit adds S to n (int16 wrap-around kept).
// w = rn16(f32(w) + f32(rn16(v))): c10::Half += float rounds v to f16 first.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void backup_walk(const uint16_t* rewards, const int16_t* parents, const uint8_t* terminal,
                                            uint16_t* w, int16_t* n, long envbase, int S, int s, int leaf, float v) {
    int cur = leaf;
    while (cur != -1) {
        const long i = envbase + cur;
        if (terminal[i]) v = 0.f;
        v += h2f(rewards[i * S + s]);
        if (s == 0) n[i] = (int16_t)(n[i] + S);
        w[i * S + s] = f2h(h2f(w[i * S + s]) + h2f(f2h(v)));
        cur = parents[i];
    }
}

}  // namespace bl
