// bl_search.hip -- the general search kernels of boardlaw's vectorised-MCTS hot path for gfx950 (MI355X, CDNA4): descend / root /
// backup / transition_q on the reference's arrays and their C entry points (bl_mcts_*).  The device code is in bl_policy.h, shared
// with bl_sim.hip (the fused simulation step).  (Until round 6 all of it was bl_kernels.hip, with what is now bl_hex.hip and bl_abi.hip.)
//
// Written for 64-wide wavefronts: every kernel assigns a GROUP of G lanes (G in {8,16,32,64}, chosen by the host
// from B and A) to one env, so a wave carries 64/G envs.  The lanes of a group stride the action axis, which makes
// every children[b,t,:] / logits[b,t,:] row a coalesced burst; the per-action Newton terms are staged in LDS and the
// group's lane 0 folds them in the reference's serial order (float addition is not associative and parity is
// bit-exact: see DESIGN.md "Exact arithmetic").  The reference (boardlaw/mcts/cpp/cuda.cu) uses one THREAD per env
// in 8-thread blocks, i.e. 8 of 64 lanes and a row stride of T*A*2 bytes between neighbouring lanes.
//
// Arithmetic contract: IEEE binary32, RNE, no contraction (-ffp-contract=off and the pragma below), correctly
// rounded division (hipcc default), denormals kept; expf through the caller's 65536-entry table (host libm).
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

__global__ void __launch_bounds__(256) qrange_kernel(const uint16_t* w, const int16_t* n, long nodes, int S, uint32_t* qr) {
    uint32_t nmin = 0, vmax = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nodes; i += (long)gridDim.x * blockDim.x) {
        const float den = (float)n[i] + 1.e-4f;
        for (int s = 0; s < S; s++) {
            const uint32_t e = enc(h2f(w[i * S + s]) / den);
            nmin = max(nmin, ~e); vmax = max(vmax, e);
        }
    }
    qrange_publish(qr, nmin, vmax, (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) % BL_QSLOTS);
}

__global__ void __launch_bounds__(256) backup_kernel(const uint16_t* v, uint16_t* w, int16_t* n, const uint16_t* rewards,
                                                     const int16_t* parents, const uint8_t* terminal,
                                                     const int16_t* leaves, int B, int T, int S) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * S) return;
    const int b = (int)(idx / S), s = (int)(idx % S);
    const int leaf = leaves[b];
    const long envbase = (long)b * T;
    backup_walk(rewards, parents, terminal, w, n, envbase, S, s, leaf, h2f(v[(envbase + leaf) * S + s]));
}

__global__ void __launch_bounds__(256) zero_words_kernel(uint32_t* p, int n, uint32_t word) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = word;
}

}  // namespace bl

using namespace bl;

extern "C" {

int bl_mcts_qrange(const void* w, const int16_t* n, int B, int T, int S, uint32_t* st, bl_stream_t stream) {
    if (!w || !n || !st || B <= 0 || T <= 0 || S <= 0) return BL_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, hs, st, BL_QWORDS, BL_QBIAS);
    const long nodes = (long)B * T;
    int blocks = (int)((nodes + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(qrange_kernel, dim3(blocks), dim3(256), 0, hs, (const uint16_t*)w, n, nodes, S, st);
    return check_launch();
}

static int tree_check(const void* logits, const void* w, const void* n, const void* c, const void* seats, const void* term,
                      const void* ch, const void* qr, const void* et, int B, int T, int A, int S) {
    if (!logits || !w || !n || !c || !seats || !term || !ch || !qr || !et || B <= 0 || T <= 0 || A <= 0 || S <= 0) return BL_EINVAL;
    if (A > 1024 || T > 32767 || S > 8) return BL_ETOOBIG;
    return BL_OK;
}

int bl_mcts_descend(const void* logits, const void* w, const int16_t* n, const void* c_puct, const int16_t* seats,
                    const uint8_t* terminal, const int16_t* children, const void* rands, const uint32_t* qr,
                    const float* exp_table, int B, int T, int A, int S, int16_t* parents, int16_t* actions,
                    bl_stream_t stream) {
    return bl_mcts_descend_tuned(nullptr, logits, w, n, c_puct, seats, terminal, children, rands, qr, exp_table, B, T, A, S, parents,
                                 actions, stream);
}

int bl_mcts_descend_tuned(const bl_tune_t* tune, const void* logits, const void* w, const int16_t* n, const void* c_puct,
                          const int16_t* seats, const uint8_t* terminal, const int16_t* children, const void* rands,
                          const uint32_t* qr, const float* exp_table, int B, int T, int A, int S, int16_t* parents,
                          int16_t* actions, bl_stream_t stream) {
    int rc = tree_check(logits, w, n, c_puct, seats, terminal, children, qr, exp_table, B, T, A, S);
    if (rc) return rc;
    if (!rands || !parents || !actions) return BL_EINVAL;
    Tree m{(const uint16_t*)logits, (const uint16_t*)w, n, (const uint16_t*)c_puct, seats, terminal, children, qr,
           exp_table, B, T, A, S, 0, tune ? tune->powf_libm : 0};
    const int G = pick_group(B, A, tune ? tune->group : 0), K = pick_k(A, G);
    const int per = lds_bytes(A, false);
    const int blocks = (B + 64 / G - 1) / (64 / G);
#define CALL(g, k) hipLaunchKernelGGL((descend_kernel<g, k, false>), dim3(blocks), dim3(64), (size_t)per * (64 / g), \
                                      (hipStream_t)stream, m, (const uint16_t*)rands, parents, actions, nullptr)
    BL_DISPATCH_GK(G, K, CALL)
#undef CALL
    return check_launch();
}

int bl_mcts_root(const void* logits, const void* w, const int16_t* n, const void* c_puct, const int16_t* seats,
                 const uint8_t* terminal, const int16_t* children, const uint32_t* qr, const float* exp_table,
                 int B, int T, int A, int S, void* probs, bl_stream_t stream) {
    return bl_mcts_root_tuned(nullptr, logits, w, n, c_puct, seats, terminal, children, qr, exp_table, B, T, A, S, probs, stream);
}

int bl_mcts_root_tuned(const bl_tune_t* tune, const void* logits, const void* w, const int16_t* n, const void* c_puct,
                       const int16_t* seats, const uint8_t* terminal, const int16_t* children, const uint32_t* qr,
                       const float* exp_table, int B, int T, int A, int S, void* probs, bl_stream_t stream) {
    int rc = tree_check(logits, w, n, c_puct, seats, terminal, children, qr, exp_table, B, T, A, S);
    if (rc) return rc;
    if (!probs) return BL_EINVAL;
    Tree m{(const uint16_t*)logits, (const uint16_t*)w, n, (const uint16_t*)c_puct, seats, terminal, children, qr,
           exp_table, B, T, A, S, 0, tune ? tune->powf_libm : 0};
    const int G = pick_group(B, A, tune ? tune->group : 0), K = pick_k(A, G);
    const int per = lds_bytes(A, false);
    const int blocks = (B + 64 / G - 1) / (64 / G);
#define CALL(g, k) hipLaunchKernelGGL((root_kernel<g, k>), dim3(blocks), dim3(64), (size_t)per * (64 / g), \
                                      (hipStream_t)stream, m, (uint16_t*)probs, (const uint16_t*)nullptr, (uint16_t*)nullptr)
    BL_DISPATCH_GK(G, K, CALL)
#undef CALL
    return check_launch();
}

int bl_mcts_backup(const void* v, void* w, int16_t* n, const void* rewards, const int16_t* parents, const uint8_t* terminal,
                   const int16_t* leaves, int B, int T, int S, bl_stream_t stream) {
    if (!v || !w || !n || !rewards || !parents || !terminal || !leaves || B <= 0 || T <= 0 || S <= 0) return BL_EINVAL;
    if (T > 32767 || S > 8) return BL_ETOOBIG;
    const long threads = (long)B * S;
    hipLaunchKernelGGL(backup_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)v, (uint16_t*)w, n, (const uint16_t*)rewards, parents, terminal, leaves, B, T, S);
    return check_launch();
}

}  // extern "C"
