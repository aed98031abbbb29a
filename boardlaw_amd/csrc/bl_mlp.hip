// bl_mlp.hip -- the leaf-evaluation network's body + head Linears (boardlaw/networks.py:10-40) as ONE gfx950 kernel.
//
// PyTorch runs this fp16-autocast MLP as 6 GEMM launches + elementwise launches (about 95 us per 4096-row batch on an
// MI355X, launch- and epilogue-bound at this size).  Here a workgroup of 4-8 waves takes 32 rows through every layer:
// activations live in registers (residual stream x) and LDS (relu(x)), weights stream from L2 straight into MFMA fragments
// (v_mfma_f32_32x32x16_f16), and the ReZero tail x + alpha*y / relu are the epilogue.  Rounding points are torch's
// (Linear output, alpha*y, x + ., each rounded to f16); only the K-summation order inside a GEMM differs, so results
// agree with the autocast module to f16 rounding (tests/test_gpu_parity.py::test_fused_mlp_matches_autocast).
// The heads' nonlinearities (masked log-softmax, tanh) stay in bl_sim_finish.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/boardlaw_amd.h"

namespace blmlp {

#ifdef BL_MLP_CLK
__device__ long long g_debug_clk[64];
#define CLK(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_debug_clk[i] = clock64();
#else
#define CLK(i)
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 f16;

__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (f16)f); }
__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(f16, b); }

struct Params {
    const uint16_t* obs;      // (M, K0) f16
    const uint16_t* w0;       // (W, K0pad) f16, zero padded in K; this and the other matrices are fragment-major packed
    const uint16_t* b0;       // (W)
    const uint16_t* wb;       // (D, W, W)
    const uint16_t* bb;       // (D, W)
    const float* alphas;      // (D) f32
    const uint16_t* wh;       // (NHpad, W): rows 0..NH-2 policy, row NH-1 value, rest zero
    const uint16_t* bh;       // (NHpad)
    uint16_t* policy;         // (M, NH-1)
    uint16_t* value;          // (M)
    int M, K0, K0pad, W, D, NH, NHpad;
};

// One layer, transposed: acc[t] = W[32 features of tile t][K] . in[32 rows][K]^T, i.e. D[feature][batch row].  With the
// weights as the A operand, a lane's accumulator registers are 4 groups of 4 CONSECUTIVE features of ONE batch row
// (feature = 32*tile + (i & 3) + 8*(i >> 2) + 4*(lane >> 5), row = lane & 31), so the epilogue moves 8 bytes at a time.
// `in` is LDS, row stride `ldin` halves.  Weights are PRE-PACKED fragment-major by the host (networks.Inference.refresh):
//     Wp[ntile][kblock][s][lane][8]  =  W[n = 32*ntile + (lane & 31)][k = 64*kblock + 32*(lane >> 5) + 8*s + 0..7]
// so each of a wave's B-fragment loads is one perfectly coalesced 1 KiB read, and the four MFMAs of a 64-wide k block
// consume pieces s = 0..3.  (Row-major weights made every load instruction touch 32 cache lines: 97 us per forward.)
// A fragments use the same k assignment from LDS.  K % 64 == 0.
template <int NT> struct Ring { half8 b0[NT][4], b1[NT][4], b2[NT][4]; };

template <int NT>
__device__ __forceinline__ void ring_load(half8 (&b)[NT][4], const uint16_t* Wp, int KB, int tile0, int ntiles_valid, int kb) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < NT; t++) if (t < ntiles_valid) {
        const uint16_t* bt = Wp + (long)(tile0 + t) * KB * 2048 + lane * 8 + kb * 2048;
#pragma unroll
        for (int s = 0; s < 4; s++) b[t][s] = *(const half8*)(bt + s * 512);
    }
}

// Starts a layer's weight stream (k blocks 0 and 1).  Called BEFORE the previous layer's epilogue and barriers: weights
// do not depend on activations, so their L2 latency hides behind that work.
template <int NT>
__device__ __forceinline__ void gemm_prefetch(Ring<NT>& rg, const uint16_t* Wp, int K, int tile0, int ntiles_valid) {
    const int KB = K >> 6;
    ring_load<NT>(rg.b0, Wp, KB, tile0, ntiles_valid, 0);
    if (KB > 1) ring_load<NT>(rg.b1, Wp, KB, tile0, ntiles_valid, 1);
}

// Runs the layer: three k blocks of weight fragments in flight (their L2 latency, 1-2k cycles under load, is several
// blocks of MFMA work and a wave has only one partner on its SIMD to hide behind).
template <int NT>
__device__ __forceinline__ void gemm_run(Ring<NT>& rg, const uint16_t* in, int ldin, const uint16_t* Wp, int K, int tile0,
                                         int ntiles_valid, float16v (&acc)[NT]) {
    const int lane = threadIdx.x & 63, r = lane & 31, hf = lane >> 5;
    const int KB = K >> 6;
#pragma unroll
    for (int t = 0; t < NT; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const uint16_t* arow = in + r * ldin + 32 * hf;
    auto compute = [&](half8 (&b)[NT][4], int kb) {
        half8 a[4];
#pragma unroll
        for (int s = 0; s < 4; s++) a[s] = *(const half8*)(arow + kb * 64 + 8 * s);
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int t = 0; t < NT; t++) if (t < ntiles_valid) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[t][s], a[s], acc[t], 0, 0, 0);
        }
    };
    for (int kb = 0; kb < KB; kb += 3) {
        if (kb + 2 < KB) ring_load<NT>(rg.b2, Wp, KB, tile0, ntiles_valid, kb + 2);
        compute(rg.b0, kb);
        if (kb + 1 < KB) {
            if (kb + 3 < KB) ring_load<NT>(rg.b0, Wp, KB, tile0, ntiles_valid, kb + 3);
            compute(rg.b1, kb + 1);
        }
        if (kb + 2 < KB) {
            if (kb + 4 < KB) ring_load<NT>(rg.b1, Wp, KB, tile0, ntiles_valid, kb + 4);
            compute(rg.b2, kb + 2);
        }
    }
}

typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// 4 consecutive features of one batch row: y = rn16(acc + bias); x' = x + alpha*y; r = relu(x'), all with torch's f16
// rounding points.  The f16 products/sums are done with packed f16 instructions: for binary16 operands, computing in f32
// and rounding to f16 (what torch does) equals the correctly rounded f16 operation (24 >= 2*11 + 2 bits), so the bits
// are the same at a quarter of the instructions.
__device__ __forceinline__ void rezero4(const float* acc4, uint2 bias, uint2 xold, half2v al2, bool first, uint2& xout, uint2& rout) {
    const half2v b01 = __builtin_bit_cast(half2v, bias.x), b23 = __builtin_bit_cast(half2v, bias.y);
    half2v y01, y23;
    y01[0] = (f16)(acc4[0] + (float)b01[0]); y01[1] = (f16)(acc4[1] + (float)b01[1]);
    y23[0] = (f16)(acc4[2] + (float)b23[0]); y23[1] = (f16)(acc4[3] + (float)b23[1]);
    half2v x01 = y01, x23 = y23;
    if (!first) {
        x01 = __builtin_bit_cast(half2v, xold.x) + al2 * y01;      // -ffp-contract=off: mul and add round separately
        x23 = __builtin_bit_cast(half2v, xold.y) + al2 * y23;
    }
    const half2v z = {(f16)0.f, (f16)0.f};
    const half2v r01 = __builtin_elementwise_max(x01, z), r23 = __builtin_elementwise_max(x23, z);
    xout = make_uint2(__builtin_bit_cast(uint32_t, x01), __builtin_bit_cast(uint32_t, x23));
    rout = make_uint2(__builtin_bit_cast(uint32_t, r01), __builtin_bit_cast(uint32_t, r23));
}

// WAVES x PASSES x NT x 32 == W: every wave owns PASSES groups of NT 32-column tiles of a body layer's output and works
// through them one group at a time (accumulators and weight ring sized for NT tiles; W = 1024 would not fit otherwise).
template <int NT, int PASSES, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) mlp_kernel(Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = p.W, ld = W + 8;                  // +8 halves: rows 16 B apart in bank space, ds_read_b128 conflict-free
    // Two activation buffers R(0), R(1) of [32][ld] f16: layer l reads R((l + par0) & 1) and writes the other one, so
    // one barrier per layer suffices; the last layer writes the un-rectified neck (for the heads) instead of relu, and
    // par0 is chosen so that the neck lands in R(0), leaving everything from R(1) on to the heads' staging.  (Offsets
    // from the one LDS base, not an array of pointers: the latter decays to generic pointers and turns every LDS
    // access into a flat_load/flat_store.)
    uint16_t* R0 = (uint16_t*)smem;
    const int par0 = (p.D + 1) & 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * 32;
    constexpr int NTHREADS = WAVES * 64;
    const int brow = lane & 31, hf = lane >> 5;     // this lane's batch row within the tile, and its feature half

    CLK(0)
    float16v acc[NT];
    Ring<NT> rg;
    gemm_prefetch<NT>(rg, p.w0, p.K0pad, wave * PASSES * NT, NT);   // weights first: their latency hides behind the staging
    // stage the observation tile as 32-bit words (K0 is even: 2 planes per cell), zero-padding K0 -> K0pad and rows >= M
    {
        const int wpr = p.K0pad >> 1, wvalid = p.K0 >> 1;           // words per staged row / per real row
        const uint32_t* src = (const uint32_t*)p.obs;               // row r starts at word r * K0 / 2 (K0 even)
        uint32_t* dst = (uint32_t*)(R0 + 32 * ld * par0);
        for (int r = tid >> 5; r < 32; r += NTHREADS / 32)
            for (int w = tid & 31; w < wpr; w += 32)
                dst[r * (ld >> 1) + w] = (w < wvalid && row0 + r < p.M) ? src[(long)(row0 + r) * wvalid + w] : 0u;
    }
    __syncthreads();
    CLK(1)

    // intake Linear, then the ReZero blocks (networks.py:17-18).  A wave owns the same columns of the same rows in every
    // layer, so its slice of the residual stream x stays in registers (packed f16) from layer to layer; only relu(x)
    // -- the next GEMM's input -- goes through LDS.
    uint2 xreg[PASSES][NT][4];
#pragma unroll
    for (int ps = 0; ps < PASSES; ps++) for (int t = 0; t < NT; t++) for (int g = 0; g < 4; g++) xreg[ps][t][g] = make_uint2(0, 0);
    for (int l = 0; l <= p.D; l++) {
        const uint16_t* Wl = l == 0 ? p.w0 : p.wb + (long)(l - 1) * W * W;
        const uint16_t* bl = l == 0 ? p.b0 : p.bb + (long)(l - 1) * W;
        const int Kl = l == 0 ? p.K0pad : W;
        half2v al2 = {(f16)0.f, (f16)0.f};
        if (l > 0) { const f16 a = (f16)p.alphas[l - 1]; al2[0] = a; al2[1] = a; }   // torch casts the f32 0-dim parameter to f16
        const uint16_t* Rin = R0 + 32 * ld * ((l + par0) & 1);
        uint16_t* Rn = R0 + 32 * ld * ((l + 1 + par0) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int tile0 = (wave * PASSES + ps) * NT, n0 = tile0 * 32;
            uint2 biasr[NT][4];                                       // issued now, needed after the GEMM
#pragma unroll
            for (int t = 0; t < NT; t++) for (int g = 0; g < 4; g++) biasr[t][g] = *(const uint2*)(bl + n0 + 32 * t + 8 * g + 4 * hf);
            gemm_run<NT>(rg, Rin, ld, Wl, Kl, tile0, NT, acc);
            CLK(2 + 3 * l)
            // next weights in flight before the epilogue: this layer's next pass, or the next layer's first pass
            if (ps + 1 < PASSES) gemm_prefetch<NT>(rg, Wl, Kl, tile0 + NT, NT);
            else if (l < p.D) gemm_prefetch<NT>(rg, p.wb + (long)l * W * W, W, wave * PASSES * NT, NT);
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int f0 = n0 + 32 * t + 8 * g + 4 * hf;             // 4 consecutive features of batch row `brow`
                    const float a4[4] = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    uint2 xo, ro;
                    rezero4(a4, biasr[t][g], xreg[ps][t][g], al2, l == 0, xo, ro);
                    xreg[ps][t][g] = xo;
                    *(uint2*)(Rn + brow * ld + f0) = (l == p.D) ? xo : ro;    // the heads read the neck itself
                }
            }
        }
        CLK(3 + 3 * l)
        __syncthreads();
        CLK(4 + 3 * l)
    }
    const uint16_t* X = R0;                                        // the neck (par0 makes the last layer write R(0))
    // heads' Linears on the un-rectified neck.  The NHpad/32 output tiles are few (3 for 9x9), so each tile's K range is
    // split over two waves (waves 2t and 2t+1); the upper half's partial sums go through LDS (R(1) is free now) to the
    // lower half's wave, which adds them in a fixed order and stores.
    const int htiles = p.NHpad / 32, KBh = W >> 6;
    float* Part = (float*)(R0 + 32 * ld);                         // [unit][16][64] f32, 4 KiB per unit
    for (int round = 0; round * WAVES < 2 * htiles; round++) {    // same trip count for every wave: uniform barriers
        const int u = round * WAVES + wave;
        const bool active = u < 2 * htiles;
        const int t0 = active ? (u >> 1) : 0, khalf = u & 1;
        const int kb0 = khalf ? KBh / 2 : 0, kb1 = khalf ? KBh : KBh / 2;
        float16v hacc;
        for (int i = 0; i < 16; i++) hacc[i] = 0.f;
        if (active) {
            const uint16_t* arow = X + brow * ld + 32 * hf;
            const uint16_t* bt = p.wh + (long)t0 * KBh * 2048 + lane * 8;
            auto loadb = [&](half8 (&b)[4], int kb) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) b[s2] = *(const half8*)(bt + kb * 2048 + s2 * 512);
            };
            auto step = [&](half8 (&b)[4], int kb) {
                half8 a[4];
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) a[s2] = *(const half8*)(arow + kb * 64 + 8 * s2);
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) hacc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[s2], a[s2], hacc, 0, 0, 0);
            };
            half8 be[4], bo[4];                                   // two k blocks in flight, alternating
            loadb(be, kb0);
            for (int kb = kb0; kb < kb1; kb += 2) {
                if (kb + 1 < kb1) loadb(bo, kb + 1);
                step(be, kb);
                if (kb + 1 < kb1) {
                    if (kb + 2 < kb1) loadb(be, kb + 2);
                    step(bo, kb + 1);
                }
            }
        }
        __syncthreads();
        if (active && khalf == 1) {
#pragma unroll
            for (int i = 0; i < 16; i++) Part[(t0 * 16 + i) * 64 + lane] = hacc[i];
        }
        __syncthreads();
        uint16_t* Out = (uint16_t*)(Part + htiles * 16 * 64);      // [32 rows][NHpad] f16 staging for coalesced stores
        if (active && khalf == 0) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int f0 = 32 * t0 + 8 * g + 4 * hf;              // 4 consecutive output features of row `brow`
                uint16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = f2h(hacc[4 * g + j] + Part[(t0 * 16 + 4 * g + j) * 64 + lane] + h2f(p.bh[f0 + j]));
                *(uint2*)(Out + brow * p.NHpad + f0) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            }
        }
        __syncthreads();
    }
    // coalesced stores: a wave writes one row's NH-1 policy outputs as consecutive halves
    {
        const uint16_t* Out = (const uint16_t*)(Part + htiles * 16 * 64);
        for (int r = wave; r < 32; r += WAVES) {
            if (row0 + r < p.M) {
                for (int f = lane; f < p.NH - 1; f += 64) p.policy[(long)(row0 + r) * (p.NH - 1) + f] = Out[r * p.NHpad + f];
            }
        }
        if (tid < 32 && row0 + tid < p.M) p.value[row0 + tid] = Out[tid * p.NHpad + p.NH - 1];
    }
    CLK(40)
}

}  // namespace blmlp

#ifdef BL_MLP_CLK
extern "C" int bl_mlp_debug_clk(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(blmlp::g_debug_clk), 64 * 8) == hipSuccess ? 0 : -3; }
#endif
extern "C" int bl_mlp_forward_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                  const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                  int K0pad, int NH, int NHpad, void* policy_out, void* value_out, bl_stream_t stream) {
    using namespace blmlp;
    if (!obs || !w0 || !b0 || !wh || !bh || !policy_out || !value_out || M <= 0 || K0 <= 0 || D < 0 || NH < 2) return BL_EINVAL;
    if (D > 0 && (!wb || !bb || !alphas)) return BL_EINVAL;
    if (W % 128 != 0 || W < 128 || W > 1024 || K0pad % 64 != 0 || K0pad < K0 || K0pad > W || NHpad % 32 != 0 || NHpad < NH) return BL_ETOOBIG;
    Params p{(const uint16_t*)obs, (const uint16_t*)w0, (const uint16_t*)b0, (const uint16_t*)wb, (const uint16_t*)bb, alphas,
             (const uint16_t*)wh, (const uint16_t*)bh, (uint16_t*)policy_out, (uint16_t*)value_out, M, K0, K0pad, W, D, NH, NHpad};
    // two activation buffers; the heads keep the neck in the first and stage split-K partials + outputs after it
    const size_t buf = (size_t)32 * (W + 8) * 2;
    const size_t staging = (size_t)(NHpad / 32) * 16 * 64 * 4 + (size_t)32 * NHpad * 2;
    const size_t lds = buf + (staging > buf ? staging : buf);
    if (lds > 160 * 1024) return BL_ETOOBIG;
    const dim3 grid((M + 31) / 32);
    hipStream_t hs = (hipStream_t)stream;
    // above the 64 KiB default the limit has to be raised per kernel (gfx950 has 160 KiB per CU)
#define BL_MLP_LAUNCH(NT, PASSES, WAVES)                                                                               \
    {                                                                                                                  \
        static size_t raised = 65536;                                                                                  \
        if (lds > raised) {                                                                                            \
            if (hipFuncSetAttribute((const void*)mlp_kernel<NT, PASSES, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return BL_ELAUNCH; \
            raised = lds;                                                                                              \
        }                                                                                                              \
        hipLaunchKernelGGL((mlp_kernel<NT, PASSES, WAVES>), grid, dim3(WAVES * 64), lds, hs, p);                       \
    }
    // 8 waves (two per SIMD) from W = 256 up: while one wave waits for its weight fragments the other issues MFMAs
    switch (W / 128) {
        case 1: BL_MLP_LAUNCH(1, 1, 4) break;
        case 2: BL_MLP_LAUNCH(1, 1, 8) break;
        case 4: BL_MLP_LAUNCH(2, 1, 8) break;
        case 6: BL_MLP_LAUNCH(1, 3, 8) break;
        case 8: BL_MLP_LAUNCH(2, 2, 8) break;
        default: return BL_ETOOBIG;
    }
#undef BL_MLP_LAUNCH
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}
