// bl_root.hip -- the ROOT evaluation's Linears (boardlaw/networks.py:10-40) in fp32 as ONE gfx950 kernel.
//
// MCTS.initialize runs the network outside autocast (boardlaw/mcts/__init__.py:72-76), so once per move the whole batch
// goes through the body and both heads in fp32.  As torch launches that is 6 hipBLASLt GEMMs plus the elementwise ReZero
// tails (142 us at 4096 x 512x4).  Here a workgroup of 8 waves takes 16 rows through every layer: activations in
// registers (the residual stream x, fp32) and LDS (relu(x)), weights streamed from L2 straight into the A operands of
// v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulation), bias / alpha*y / x + . / relu as the epilogue with
// torch's rounding points (each a separately rounded fp32 operation).  Only the K-summation order inside a Linear
// differs from the library GEMM's, so outputs agree with the module to fp32 rounding (~1e-6 relative;
// tests/test_gpu_parity.py::test_root_plan_matches_module).  The heads' nonlinearities stay in bl_sim_plant_root.
//
// Bounds at 4096 rows of 512x4 (256 workgroups, one per CU): 9.6 GFLOP at the measured 146 TFLOP/s of
// v_mfma_f32_16x16x4_f32 (tools/micro/mfma_f32_rate.hip: 32 cycles per instruction and SIMD) = 66 us -- 77 us with the
// weight loads compiled out; 256 workgroups x 4.5 MiB of weights through the L2 = 1.15 GB -- 79 us with the MFMAs compiled
// out (14.6 TB/s of L2 reads, what this access pattern gets); both together 102 us (the two streams do not overlap
// perfectly), against 135 us for the library plan.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/boardlaw_amd.h"
#include "bl_host.h"

namespace blroot {

typedef float float4v __attribute__((ext_vector_type(4)));

struct Params {
    const float* obs;        // (M, K0) f32
    const float* w0;         // intake, fragment-major packed (W/16 tiles, K0pad/16 blocks, 64 lanes, 4)
    const float* b0;         // (W)
    const float* wb;         // (D, W/16, W/16, 64, 4)
    const float* bb;         // (D, W)
    const float* alphas;     // (D)
    const float* wh;         // (NHpad/16, W/16, 64, 4): rows 0..NH-2 policy, row NH-1 value, rest zero
    const float* bh;         // (NHpad)
    float* policy;           // (M, NH-1)
    float* value;            // (M)
    int M, K0, K0pad, W, D, NH, NHpad;
};


// acc[t] += Wp[tile0 + t] . in^T over K: D[feature][batch row], features 16*tile + 4*(lane >> 4) + 0..3 of row lane & 15 in
// a lane's four accumulator registers.  Packing (host, networks.pack_fragment_major_f32):
//     Wp[tile][kb][lane][s] = W[n = 16*tile + (lane & 15)][k = 16*kb + 4*(lane >> 4) + s]
// so a (tile, k block) is one coalesced 1 KiB load per wave and MFMA s of the block multiplies pieces s; the activations
// use the same k assignment (one ds_read_b128 per k block, shared by the wave's tiles).
template <int NT, int DEPTH>
__device__ __forceinline__ void gemm(const float* in, int ld, const float* Wp, int K, int tile0, int ntiles, float4v (&acc)[NT]) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int KB = K >> 4;
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = float4v{0.f, 0.f, 0.f, 0.f};
    float4v wq[DEPTH][NT];
    const float* wl = Wp + (long)tile0 * KB * 256 + lane * 4;
    auto load = [&](float4v (&slot)[NT], int kb) {
#pragma unroll
        for (int t = 0; t < NT; t++) if (t < ntiles) slot[t] = *(const float4v*)(wl + ((long)t * KB + kb) * 256);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) { load(wq[d], d); __builtin_amdgcn_sched_barrier(0); }   // K % 64 == 0: whole rounds only
    const float* xrow = in + n * ld + 4 * g;
    float4v xb = *(const float4v*)xrow;
    auto compute = [&](float4v (&slot)[NT], int kb) {
        // the next block's activations are requested before this block's MFMAs: their LDS round trip runs under them
        const float4v xn = *(const float4v*)(xrow + 16 * (kb + 1 < KB ? kb + 1 : kb));
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < NT; t++) if (t < ntiles) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(slot[t][s], xb[s], acc[t], 0, 0, 0);
        }
        xb = xn;
    };
    // A slot is refilled only after its MFMAs are issued (no register copies): DEPTH - 1 blocks stay in flight.  No
    // conditionals inside a round, and the refills pinned in place (the scheduler otherwise sinks a round's loads to its end
    // and the next round starts by waiting for all of them).
    int kb0 = 0;
    for (; kb0 + DEPTH < KB; kb0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            compute(wq[d], kb0 + d);
            __builtin_amdgcn_sched_barrier(0);
            load(wq[d], kb0 + d + DEPTH);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) compute(wq[d], kb0 + d);
}

// 8 waves x NT tiles x 16 == W: every wave owns the same NT 16-feature tiles of every body layer's output.
template <int NT>
__global__ void __launch_bounds__(512) root_mlp_kernel(Params p) {
#ifndef BL_ROOT_DEPTH
#define BL_ROOT_DEPTH 8
#endif
    constexpr int BODY_DEPTH = NT >= 7 ? 2 : (NT >= 5 ? 4 : BL_ROOT_DEPTH);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = p.W, ld = W + 4;                 // +4 floats: rows 16 B apart in bank space, ds_read_b128 conflict-free
    float* R0 = (float*)smem;                      // two activation buffers [16][ld]: layer l reads R(l & 1), writes the other
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 16;
    for (int i = tid; i < 16 * p.K0pad; i += 512) {
        const int r = i / p.K0pad, c = i - r * p.K0pad;
        R0[r * ld + c] = (c < p.K0 && row0 + r < p.M) ? p.obs[(long)(row0 + r) * p.K0 + c] : 0.f;
    }
    __syncthreads();
    const int tile0 = wave * NT;
    float4v x[NT], acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) x[t] = float4v{0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l <= p.D; l++) {
        const float* Wl = l == 0 ? p.w0 : p.wb + (long)(l - 1) * W * W;
        const float* bl = l == 0 ? p.b0 : p.bb + (long)(l - 1) * W;
        const float al = l == 0 ? 0.f : p.alphas[l - 1];
        const float* Rin = R0 + 16 * ld * (l & 1);
        float* Rn = R0 + 16 * ld * ((l + 1) & 1);
        float4v bias[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) bias[t] = *(const float4v*)(bl + 16 * (tile0 + t) + 4 * g);
        // k blocks of weight fragments in flight per tile (4 * NT * DEPTH registers); W % 128 == 0, K0pad % 64 == 0
        if (l == 0) gemm<NT, (NT >= 7 ? 2 : 4)>(Rin, ld, Wl, p.K0pad, tile0, NT, acc);
        else gemm<NT, BODY_DEPTH>(Rin, ld, Wl, W, tile0, NT, acc);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const float4v y = acc[t] + bias[t];                       // the Linear's output (torch: GEMM, then + bias)
            x[t] = l == 0 ? y : x[t] + al * y;                        // networks.py:17-18; -ffp-contract=off: two roundings
            float4v o = x[t];
            if (l < p.D) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
            *(float4v*)(Rn + n * ld + 16 * (tile0 + t) + 4 * g) = o;  // the heads read the neck itself, the blocks relu(x)
        }
        __syncthreads();
    }
    // heads' Linears on the neck: the NHpad/16 output tiles (6 for 9x9) go round the waves
    const float* X = R0 + 16 * ld * ((p.D + 1) & 1);
    for (int tile = wave; tile < p.NHpad / 16; tile += 8) {
        float4v hacc[1];
        gemm<1, 8>(X, ld, p.wh, W, tile, 1, hacc);
        const float4v b = *(const float4v*)(p.bh + 16 * tile + 4 * g);
        const float4v o = hacc[0] + b;
        if (row0 + n < p.M) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int f = 16 * tile + 4 * g + r;
                if (f < p.NH - 1) p.policy[(long)(row0 + n) * (p.NH - 1) + f] = o[r];
                else if (f == p.NH - 1) p.value[row0 + n] = o[r];
            }
        }
    }
}

}  // namespace blroot

extern "C" int bl_root_mlp_f32(const float* obs, int M, int K0, const float* w0, const float* b0, const float* wb, const float* bb,
                               const float* alphas, const float* wh, const float* bh, int W, int D, int K0pad, int NH, int NHpad,
                               float* policy_out, float* value_out, bl_stream_t stream) {
    using namespace blroot;
    if (!obs || !w0 || !b0 || !wh || !bh || !policy_out || !value_out || M <= 0 || K0 <= 0 || D < 0 || NH < 2) return BL_EINVAL;
    if (D > 0 && (!wb || !bb || !alphas)) return BL_EINVAL;
    if (W % 128 != 0 || W < 128 || W > 1024 || K0pad % 64 != 0 || K0pad < K0 || K0pad > W || NHpad % 16 != 0 || NHpad < NH) return BL_ETOOBIG;
    const size_t lds = (size_t)2 * 16 * (W + 4) * sizeof(float);
    if (lds > 160 * 1024) return BL_ETOOBIG;
    Params p{obs, w0, b0, wb, bb, alphas, wh, bh, policy_out, value_out, M, K0, K0pad, W, D, NH, NHpad};
    const dim3 grid((M + 15) / 16);
    hipStream_t hs = (hipStream_t)stream;
    // above the 64 KiB default the dynamic LDS limit has to be raised per kernel (gfx950 has 160 KiB per CU)
#define BL_ROOT_LAUNCH(NT)                                                                                              \
    {                                                                                                                   \
        static size_t raised[64];                                                                                       \
        if (!bl_raise_lds_limit((const void*)root_mlp_kernel<NT>, lds, raised)) return BL_ELAUNCH;                     \
        hipLaunchKernelGGL((root_mlp_kernel<NT>), grid, dim3(512), lds, hs, p);                                         \
    }
    switch (W / 128) {
        case 1: BL_ROOT_LAUNCH(1) break;
        case 2: BL_ROOT_LAUNCH(2) break;
        case 3: BL_ROOT_LAUNCH(3) break;
        case 4: BL_ROOT_LAUNCH(4) break;
        case 5: BL_ROOT_LAUNCH(5) break;
        case 6: BL_ROOT_LAUNCH(6) break;
        case 7: BL_ROOT_LAUNCH(7) break;
        case 8: BL_ROOT_LAUNCH(8) break;
        default: return BL_ETOOBIG;
    }
#undef BL_ROOT_LAUNCH
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}
