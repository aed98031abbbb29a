// bl_mlp.hip -- the leaf-evaluation network's body + head Linears (boardlaw/networks.py:10-40) as ONE gfx950 kernel.
//
// PyTorch runs this fp16-autocast MLP as 6 GEMM launches + elementwise launches (about 95 us per 4096-row batch on an
// MI355X, launch- and epilogue-bound at this size).  Here a workgroup of 4-8 waves takes 32 rows through every layer:
// activations live in LDS (the residual stream x, rectified as the next GEMM reads it), weights stream from L2 straight into MFMA fragments
// (v_mfma_f32_32x32x16_f16), and the ReZero tail x + alpha*y / relu are the epilogue.  Rounding points are torch's
// (Linear output, alpha*y, x + ., each rounded to f16); only the K-summation order inside a GEMM differs, so results
// agree with the autocast module to f16 rounding (tests/test_gpu_parity.py::test_fused_mlp_matches_autocast).
// The heads' nonlinearities (masked log-softmax, tanh) stay in bl_sim_finish.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/boardlaw_amd.h"
#include "bl_host.h"

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
    int xcd_rows;             // 1: tile i takes rows 256*(i/8) + 8*r + i%8 (rows whose index is i mod 8), else rows 32*i + r
    const int32_t* n_active;  // device scalar or null: rows >= *n_active are treated like rows >= M (bl_search_t.n_active)
    int bias_off;             // set by mlp_launch: where, in halves from the LDS base, the (D + 1) x W biases are staged
};

// What bl_sim_finish does for a leaf (heads, store, backup, next q range), as this kernel's epilogue: bl_sim_infer_finish.
// Field meanings as in bl_search_t; qrange points at the row the NEXT descent reads.
struct FinArgs {
    uint16_t* logits; uint16_t* v; uint16_t* w; int16_t* n; const uint16_t* rewards; const uint8_t* terminal;
    const int16_t* path; uint32_t* qrange; const int16_t* leaves; const int32_t* leaf_seats; const uint8_t* valid;
    int T, A, Wsm, iters;
    float* cpi; uint32_t* cca; int16_t* nk; const float* exp_table;      // compacted policy rows (bl_device.h), or cpi == null
};
#ifndef BLM_RD64
#define BLM_RD64 3          // weight-ring depth of the 512-wide, 64-row instantiation (measurement switch)
#endif
#ifndef BLM_RD32
#define BLM_RD32 3          // ... of the 512-wide, 32-row instantiation
#endif
#define BLM_QSLOTS 64
#define BLM_QSTRIDE 64
__device__ __forceinline__ uint32_t enc(float f) {      // order-preserving float -> u32 (as in bl_device.h)
    const uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
template <int CTRL, int RM>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {      // row_shr 1,2,4,8, row_bcast 15/31: lane 63 ends with the max
    v = max(v, (uint32_t)dpp_i<0x111, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x112, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x114, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x118, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x142, 0xa>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x143, 0xc>(0, (int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// x[lane ^ off] for the softmax butterflies, off a power of two: DPP within a row (1, 2, 8), ds_swizzle within 32 lanes (4, 16),
// v_permlane32_swap across the halves -- the same values __shfl_xor's ds_bpermute returns, without its address arithmetic and
// LDS round trip (a butterfly of 6 steps over four envs: 1.5k -> 0.4k cycles)
template <int OFF>
__device__ __forceinline__ float xor_lane(float x) {
    const int v = __builtin_bit_cast(int, x);
    int r;
    if constexpr (OFF == 1) r = __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false);            // quad_perm [1,0,3,2]
    else if constexpr (OFF == 2) r = __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false);       // quad_perm [2,3,0,1]
    else if constexpr (OFF == 4) r = __builtin_amdgcn_ds_swizzle(v, (4 << 10) | 0x1f);
    else if constexpr (OFF == 8) r = __builtin_amdgcn_update_dpp(v, v, 0x128, 0xf, 0xf, false);      // row_ror:8
    else if constexpr (OFF == 16) r = __builtin_amdgcn_ds_swizzle(v, (16 << 10) | 0x1f);
    else {
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);   // {[lo,lo], [hi,hi]}
        r = (int)((threadIdx.x & 32) ? sw[0] : sw[1]);
    }
    return __builtin_bit_cast(float, r);
}

__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v) {       // as wave_max_u32, with OR
    v |= (uint32_t)dpp_i<0x111, 0xf>(0, (int)v); v |= (uint32_t)dpp_i<0x112, 0xf>(0, (int)v);
    v |= (uint32_t)dpp_i<0x114, 0xf>(0, (int)v); v |= (uint32_t)dpp_i<0x118, 0xf>(0, (int)v);
    v |= (uint32_t)dpp_i<0x142, 0xa>(0, (int)v); v |= (uint32_t)dpp_i<0x143, 0xc>(0, (int)v);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ float dpp_next_lane(float beyond, float x) {   // lane j <- x[j + 1]; lane 63 <- `beyond`
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, beyond), __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// One layer, transposed: acc[t] = W[32 features of tile t][K] . in[32 rows][K]^T, i.e. D[feature][batch row].  With the
// weights as the A operand, a lane's accumulator registers are 4 groups of 4 CONSECUTIVE features of ONE batch row
// (feature = 32*tile + (i & 3) + 8*(i >> 2) + 4*(lane >> 5), row = lane & 31), so the epilogue moves 8 bytes at a time.
// `in` is LDS, row stride `ldin` halves.  Weights are PRE-PACKED fragment-major by the host (networks.Inference.refresh):
//     Wp[ntile][kblock][s][lane][8]  =  W[n = 32*ntile + (lane & 31)][k = 64*kblock + 32*(lane >> 5) + 8*s + 0..7]
// so each of a wave's B-fragment loads is one perfectly coalesced 1 KiB read, and the four MFMAs of a 64-wide k block
// consume pieces s = 0..3.  (Row-major weights made every load instruction touch 32 cache lines: 97 us per forward.)
// A fragments use the same k assignment from LDS.  K % 64 == 0.
template <int NT, int RD> struct Ring { half8 b[RD][NT][4]; };      // RD k blocks of weight fragments: RD - 1 in flight, one in use
// BL_MLP_RING_FULL (round 4, measured neutral, off): fill all RD slots at a layer boundary, see gemm_prefetch
#ifdef BL_MLP_RING_FULL
#define BLM_RING_FULL 1
#else
#define BLM_RING_FULL 0
#endif

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

// Hand-placed weight stream (round 6).  The compiler's s_waitcnt insertion cannot follow a ring of fragment registers through the
// block loop: in the rolled loop it waits with vmcnt(0) once per RD blocks -- it drains the blocks it has just requested, i.e. a
// whole L2 round trip per RD blocks with nothing in flight (read off the ISA: vmcnt(7), (3), (2), (1), (0) in the first block of
// every trip) -- and fully unrolled it hoists until it spills.  So here the loads are inline asm the compiler does not recognise as
// pending memory operations (it inserts no waits for them), and every block is preceded by ONE s_waitcnt vmcnt(8 x blocks requested
// after it) that names the block's registers as in/out operands: nothing that reads them can be scheduled above the wait.  vmcnt
// retires in order and counts every vector-memory load, so a count computed from the ring's own requests can only over-wait when
// other loads are in flight, never under-wait.  The kernel must not spill (a spilled ring register would be stored while pending).
template <int NT>
__device__ __forceinline__ void ring_load_asm(half8 (&b)[NT][4], const uint16_t* Wp, int KB, int tile0, int kb) {
    // scalar base (the tile's and block's start: wave-uniform) + one 32-bit lane offset shared by every load of the kernel:
    // no 64-bit address registers per block
    const uint32_t voff = (threadIdx.x & 63) * 16;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const unsigned long long a = (unsigned long long)(Wp + (long)(tile0 + t) * KB * 2048 + (long)kb * 2048);
        const unsigned long long base = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a) |
                                        ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32);
        // (s_nop 4: should the base have come out of a v_readfirstlane, a VMEM instruction may read an SGPR a VALU instruction
        // wrote only after five wait states -- and the compiler's hazard recognizer does not look inside inline asm.  Found as a
        // memory fault: the first version read the base right behind its readfirstlane.)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(b[t][0]) : "v"(voff), "s"(base));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024" : "=v"(b[t][1]) : "v"(voff), "s"(base));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(b[t][2]) : "v"(voff), "s"(base));
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:3072" : "=v"(b[t][3]) : "v"(voff), "s"(base));
    }
}
// waits until at most 4 * NT * AFTER ring loads are outstanding, i.e. until the block requested AFTER + 1 requests ago has landed
template <int AFTER, int NT>
__device__ __forceinline__ void ring_wait(half8 (&b)[NT][4]) {
    static_assert(NT == 1 || NT == 2, "operand lists for one or two tiles");
    if constexpr (NT == 1)
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]) : "n"(4 * AFTER));
    else
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]),
                                             "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3]) : "n"(8 * AFTER));
}

// Starts a layer's weight stream (its first RD - 1 k blocks).  Called BEFORE the previous layer's epilogue and barriers:
// weights do not depend on activations, so their L2 latency hides behind that work.
// ASM: for a layer that gemm_run consumes with hand-placed waits (all NT tiles valid).
template <int NT, int RD, bool ASM = false>
__device__ __forceinline__ void gemm_prefetch(Ring<NT, RD>& rg, const uint16_t* Wp, int K, int tile0, int ntiles_valid, int rot = 0) {
    const int KB = K >> 6;
    if constexpr (ASM) {
#pragma unroll
        for (int d = 0; d < RD - 1; d++) ring_load_asm<NT>(rg.b[d], Wp, KB, tile0, rot + d < KB ? rot + d : rot + d - KB);       // KB >= RD - 1
        return;
    }
    // Every call site sits between two GEMMs (before the staging barrier, before an epilogue): no ring slot is in use then, so
    // all RD of them COULD take a block -- the slot the last GEMM step has just released would travel under the epilogue too,
    // instead of being requested by the next GEMM's first step.  Built in round 4 (-DBL_MLP_RING_FULL), bit-exact, 238 VGPRs, and
    // within the noise of RD - 1 (profiles/r04_mlp_ring.txt): the stream is at the L1's rate, not short of requests.  Off.
#pragma unroll
    for (int d = 0; d < RD - 1 + BLM_RING_FULL; d++)
        if (d == 0 || d < KB) ring_load<NT>(rg.b[d], Wp, KB, tile0, ntiles_valid, rot + d < KB ? rot + d : rot + d - KB);      // KB >= 1
}

// Runs the layer with RD - 1 k blocks of weight fragments in flight beside the one in use.  A wave's request rate is
// (blocks in flight) / (L2 latency, 2-2.5k cycles under load): with two in flight the eight waves pull 43 B/clk through the
// CU's L1, with three its 64 B/clk -- so the 512-wide kernel, which has the registers, runs RD = 4.
// `rot`: the k blocks are taken in the order rot, rot+1, ... (mod KB).  With `own_first` the wave's first block is the one it
// wrote itself in the previous layer's epilogue (its 64 output features ARE k block `wave` of this layer when W = 512), so
// it is consumed BEFORE the layer barrier, which then hides behind 1/8 of the GEMM instead of standing in front of it.
// KBC > 0: K / 64 known at compile time -- the block loop is then straight-line code.  That matters more than it looks: behind
// the branches of the run-time loop the compiler's s_waitcnt insertion loses count and waits with vmcnt(0) both before
// each block's last MFMA and before re-using a ring buffer, i.e. it drains the weight stream once per block; in straight-line
// code it waits for exactly the fragment an MFMA needs (vmcnt(16 + 7), ...) and the blocks in flight stay in flight.
// RG > 1 (round 6): the workgroup takes RG groups of 32 rows, and every weight fragment that arrives feeds RG MFMAs (one per row
// group, accumulators acc[g * NT + t]) -- the weight bytes a CU pulls through its L1 per row fall by RG, which is what bounds the
// kernel once a launch is several workgroups per CU.  The k order of every accumulator is unchanged: same bits.
template <int NT, int RD, int KBC = 0, bool OWN = false, int RG = 1, bool RELUR = false, bool ASMW = false>
__device__ __forceinline__ void gemm_run(Ring<NT, RD>& rg, const uint16_t* in, int ldin, const uint16_t* Wp, int K, int tile0,
                                         int ntiles_valid, float16v (&acc)[RG * NT], int rot = 0, bool own_first = false, bool relu_in = false) {
    const int lane = threadIdx.x & 63, r = lane & 31, hf = lane >> 5;
    const int KB = KBC > 0 ? KBC : K >> 6;
#pragma unroll
    for (int t = 0; t < RG * NT; t++) for (int i = 0; i < 16; i++) acc[t][i] = 0.f;
    const uint16_t* arow = in + r * ldin + 32 * hf;
    if constexpr (KBC > 0 && ASMW) {
        // the straight-line block loop sits inside the (rolled) loop over the layers, and everything it derives from `rot` --
        // eight LDS addresses per row group, eight scalar bases per tile -- is loop-invariant: hoisted, it is live across the
        // whole layer loop and spills.  An opaque redefinition per call keeps that arithmetic where it is used.
        rot = __builtin_amdgcn_readfirstlane(rot);
        asm volatile("; rot = %0" : "+s"(rot));
    }
    auto blk = [&](int i) { const int kb = rot + i; return kb < KB ? kb : kb - KB; };
    // RELUR / relu_in: `in` holds the residual stream x itself and the Linear's input is relu(x) (networks.py:17-18), applied to
    // the fragments as they are read.  On the bit patterns, as SIGNED 16-bit integers: a binary16 with its sign bit clear is a
    // non-negative integer and stays, one with the sign bit set is a negative integer and becomes +0 -- max(bits, floor) with
    // floor = 0, or -32768 for "as it is" (no branch in the block loop).  One v_pk_max_i16 per register; the f16 maximum costs
    // three issue slots (a canonicalising max(x, x) first, a wait state between the two).  Against relu on the values this maps
    // -0 to +0 (a zero product either way) and a NaN with its sign bit set to 0 (torch keeps it; no finite network produces one).
    typedef short short8 __attribute__((ext_vector_type(8)));
    short8 floor8;
#pragma unroll
    for (int i = 0; i < 8; i++) floor8[i] = relu_in ? (short)0 : (short)-32768;
    auto compute = [&](half8 (&b)[NT][4], int kb) {
        // row group by row group: one group's activation fragments (16 registers) live at a time, the weight fragments stay put
#pragma unroll
        for (int g = 0; g < RG; g++) {
            half8 a[4];
#pragma unroll
            for (int s = 0; s < 4; s++) {
                a[s] = *(const half8*)(arow + g * 32 * ldin + kb * 64 + 8 * s);
                if constexpr (RELUR) a[s] = __builtin_bit_cast(half8, __builtin_elementwise_max(__builtin_bit_cast(short8, a[s]), floor8));
            }
#pragma unroll
            for (int s = 0; s < 4; s++) {
#pragma unroll
                for (int t = 0; t < NT; t++) if (t < ntiles_valid) acc[g * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[t][s], a[s], acc[g * NT + t], 0, 0, 0);
            }
        }
    };
    if constexpr (KBC > 0 && ASMW) {
        static_assert(KBC >= RD - 1, "the prefetch requests RD - 1 blocks");
#pragma unroll
        for (int i = 0; i < KBC; i++) {
            constexpr int LOADS = KBC - (RD - 1);             // steps 0 .. LOADS - 1 request block i + RD - 1
            if (i < LOADS) ring_load_asm<NT>(rg.b[(i + RD - 1) % RD], Wp, KBC, tile0, blk(i + RD - 1));
            // blocks requested after block i at this point: i + 1 .. min(i + RD - 1, KBC - 1)
            const int after = (i + RD - 1 < KBC ? i + RD - 1 : KBC - 1) - i;
            if (after >= 3) ring_wait<3, NT>(rg.b[i % RD]);
            else if (after == 2) ring_wait<2, NT>(rg.b[i % RD]);
            else if (after == 1) ring_wait<1, NT>(rg.b[i % RD]);
            else ring_wait<0, NT>(rg.b[i % RD]);
            compute(rg.b[i % RD], blk(i));
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0 && OWN) __syncthreads();
        }
    } else if constexpr (KBC > 0) {
#pragma unroll
        for (int i = 0; i < KBC; i++) {
            // (sched_barrier: left to itself the scheduler sinks each load to just before its use to save registers, which
            // is the opposite of a prefetch)
            if (i + RD - 1 < KBC && !(BLM_RING_FULL && i == 0)) ring_load<NT>(rg.b[(i + RD - 1) % RD], Wp, KBC, tile0, NT, blk(i + RD - 1));   // block RD - 1 came with the prefetch
            __builtin_amdgcn_sched_barrier(0);
            compute(rg.b[i % RD], blk(i));
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0 && OWN) __syncthreads();
        }
    } else
    for (int i = 0; i < KB; i += RD) {
#pragma unroll
        for (int d = 0; d < RD; d++) {
            if (i + d < KB) {
                if (i + d + RD - 1 < KB && !(BLM_RING_FULL && i == 0 && d == 0))       // block RD - 1 came with the prefetch
                    ring_load<NT>(rg.b[(d + RD - 1) % RD], Wp, KB, tile0, ntiles_valid, blk(i + d + RD - 1));
                compute(rg.b[d], blk(i + d));
                if (i == 0 && d == 0 && own_first) __syncthreads();
            }
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
// RG: row groups of 32 rows per workgroup (1, or 2 for launches of more than one workgroup per CU; see gemm_run).
template <int NT, int PASSES, int WAVES, bool FINISH, int RD, int RG = 1>
__global__ void __launch_bounds__(WAVES * 64) mlp_kernel(Params p, FinArgs f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ROWS = 32 * RG;
    // widths 256 and 512: the body layers' weight stream on hand-placed waits (ring_load_asm)
#ifdef BL_MLP_NO_HANDW
    constexpr bool HANDW = false;
#else
    constexpr bool HANDW = PASSES == 1 && (WAVES * NT * 32 == 512 || WAVES * NT * 32 == 256);
#endif
    const int W = p.W, ld = W + 8;                  // +8 halves: rows 16 B apart in bank space, ds_read_b128 conflict-free
    // Two activation buffers R(0), R(1) of [ROWS][ld] f16 holding the residual stream x: layer l reads x_l from R((l + par0) & 1)
    // (its GEMM rectifies the fragments on the way, its epilogue re-reads its own slice) and writes x_{l+1} to the other one, so
    // one barrier per layer suffices; the last layer's output is the neck the heads read, and
    // par0 is chosen so that the neck lands in R(0), leaving everything from R(1) on to the heads' staging.  (Offsets
    // from the one LDS base, not an array of pointers: the latter decays to generic pointers and turns every LDS
    // access into a flat_load/flat_store.)
    uint16_t* R0 = (uint16_t*)smem;
    const int par0 = (p.D + 1) & 1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;      // wave: provably uniform (scalar address arithmetic)
    // Which 32 batch rows this workgroup takes.  Workgroup i runs on XCD i % 8 and so does bl_sim_expand's workgroup for env
    // b = i mod 8 (one workgroup per env, placed the same way): with xcd_rows a tile is made of envs of its own XCD, so what
    // the search kernel just wrote for them (observation, path, leaf) and what this kernel writes for the next descent
    // (logits, compacted row, w, n) stay within one XCD's L2 instead of crossing the fabric.  Placement is a speed matter
    // only: any mapping gives the same results.
    const int tile_j = blockIdx.x >> 3, tile_x = blockIdx.x & 7;
    auto grow = [&](int r) { return p.xcd_rows ? 8 * ROWS * tile_j + 8 * r + tile_x : (int)blockIdx.x * ROWS + r; };
    // rows that exist: M, or fewer when the search runs with only its first *n_active envs (bl_search_t.n_active)
    int Mrows = p.M;
    if (p.n_active) { const int na = __builtin_amdgcn_readfirstlane(*p.n_active); Mrows = na < Mrows ? na : Mrows; }
    if (grow(0) >= Mrows) return;                   // nothing in this tile (its smallest row index is row 0's)
    constexpr int NTHREADS = WAVES * 64;
    const int brow = lane & 31, hf = lane >> 5;     // this lane's batch row within a row group of the tile, and its feature half

    CLK(0)
    // FINISH: this wave finishes the tile's envs 4*wave .. 4*wave+3 after the heads (WAVES == 8).  Everything that step
    // reads from the tree is requested after the staging barrier (paths, valid masks) and after the second layer (the
    // nodes those paths name), so that the round trips run under the GEMMs without holding up the staging (vmcnt
    // retires in order): lane j holds the j-th node of the env's recorded descent (bl_sim_expand's path) and,
    // separately, node slot `lane` of the env (T <= 64) for the q range.
#ifndef BLM_FIN_PASSES
#define BLM_FIN_PASSES 2    // 64-row tiles: a wave's eight envs as two passes of four (2) or phase by phase all at once (1) -- measurement switch
#endif
    constexpr int FPASS = RG > 1 ? BLM_FIN_PASSES : 1;
    constexpr int EPA = 4 * RG;                   // envs a wave finishes in all: tile rows EPA * wave .. EPA * wave + EPA - 1 ...
    constexpr int EPW = EPA / FPASS;              // ... EPW of them at a time
    // Held across the GEMMs, where registers are scarce (ring 96 + accumulators 32 + x 16 + ...): the per-env scalars are wave
    // uniform (SGPRs), and the envs' valid bits share one register.
    int fb[EPA], fleaf[EPA], fmover[EPA], flen[EPA];
    int fnode[EPA];                               // lane j's node of env e's path (left untouched until after the last GEMM:
                                                  // any use would wait for the load, and with it for the weights in flight)
    uint32_t fvbits = 0;                          // bit 2e + k: valid[lane + 64 k] of env e
    int fterm[EPA], fn[EPA], fallN[EPA];
    uint32_t frew[EPA], fw[EPA], fallW[EPA];
    float16v acc[RG * NT];
    Ring<NT, RD> rg;
    // stage the observation tile as 32-bit words (K0 is even: 2 planes per cell), zero-padding K0 -> K0pad and rows >= M.
    // The observation loads are issued BEFORE the first weight fragments: vmcnt retires in order, so the other way round
    // the staging barrier would also wait for the (cold, just evicted by the search kernel) weights.
    {
        const int wpr = p.K0pad >> 1, wvalid = p.K0 >> 1;           // words per staged row / per real row
        const uint32_t* src = (const uint32_t*)p.obs;               // row r starts at word r * K0 / 2 (K0 even)
        uint32_t* dst = (uint32_t*)(R0 + ROWS * ld * par0);
        constexpr int RPT = ROWS / (NTHREADS / 32);                 // rows per thread: a thread owns column words w, w+32, ...
        constexpr int WMAX = 4;                                     // ... up to 4 of them in registers (K0pad <= 256)
        const int c = tid & 31, r_first = tid >> 5;
        if (wpr <= 32 * WMAX) {
            uint32_t st[RPT][WMAX];
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int r = r_first + i * (NTHREADS / 32);
#pragma unroll
                for (int k = 0; k < WMAX; k++) {
                    const int w = c + 32 * k;
                    st[i][k] = (w < wvalid && grow(r) < Mrows) ? src[(long)grow(r) * wvalid + w] : 0u;
                }
            }
            CLK(50)
            gemm_prefetch<NT, RD>(rg, p.w0, p.K0pad, wave * PASSES * NT, NT);
            CLK(51)
#pragma unroll
            for (int i = 0; i < RPT; i++) {
                const int r = r_first + i * (NTHREADS / 32);
#pragma unroll
                for (int k = 0; k < WMAX; k++) { const int w = c + 32 * k; if (w < wpr) dst[r * (ld >> 1) + w] = st[i][k]; }
            }
        } else {
            gemm_prefetch<NT, RD>(rg, p.w0, p.K0pad, wave * PASSES * NT, NT);
            for (int r = r_first; r < ROWS; r += NTHREADS / 32)
                for (int w = c; w < wpr; w += 32)
                    dst[r * (ld >> 1) + w] = (w < wvalid && grow(r) < Mrows) ? src[(long)grow(r) * wvalid + w] : 0u;
        }
    }
    // every layer's bias: (D + 1) x W halves behind the two activation buffers, read back by the epilogues
    uint16_t* BiasL = R0 + p.bias_off;
    {
        const uint32_t* b0w = (const uint32_t*)p.b0; const uint32_t* bbw = (const uint32_t*)p.bb;
        uint32_t* dstw = (uint32_t*)BiasL;
        const int hw = W >> 1;
        for (int i = tid; i < (p.D + 1) * hw; i += NTHREADS) dstw[i] = i < hw ? b0w[i] : bbw[i - hw];
    }
    CLK(52)
    __syncthreads();
    CLK(53)
    // a zero in a VGPR the compiler cannot see through: added to wave-uniform addresses below so that the loaded values stay vector
    // loads (with a provably uniform address it would move each loaded value to an SGPR at once, and the s_waitcnt for that also
    // waits for whatever was requested before)
    int zero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
    if constexpr (FINISH) {
#pragma unroll
        for (int e = 0; e < EPA; e++) {
            const int b = grow(EPA * wave + e);
            fb[e] = b < Mrows ? b : -1;
            const long bb = (b < Mrows ? b : 0) + zero;
            const int16_t* path = f.path + bb * (f.T + 2);
            flen[e] = path[0];
            fnode[e] = lane < f.T ? (int)path[1 + lane] : 0;
            // valid(a) = both observation planes of cell a empty (hex/__init__.py:154-159), read from the tile staged above
            // (word a of a row = the two f16 planes of cell a; layer 1's epilogue is the first to overwrite this buffer).
            // Loading f.valid here instead cost 4.5k cycles: the compiler tests the byte at once, and the s_waitcnt vmcnt(0)
            // it needs for that also waits for the cold weight fragments requested just before.
            const uint32_t* stg = (const uint32_t*)(R0 + ROWS * ld * par0) + (EPA * wave + e) * (ld >> 1);
            if (lane < f.A && stg[lane] == 0u) fvbits |= 1u << (2 * e);
            if (lane + 64 < f.A && stg[lane + 64] == 0u) fvbits |= 2u << (2 * e);
        }
    }
    CLK(1)

    // intake Linear, then the ReZero blocks (networks.py:17-18).  A wave owns the same columns of the same rows in every
    // layer.  (Through round 5 its slice of the residual stream stayed in registers and only relu(x) went through LDS; since
    // round 6 x itself lives in LDS -- rectified as the next GEMM reads it, re-read by the owner's epilogue -- which frees 16 RG
    // registers per wave across the GEMMs for the weight ring.)
    // a wave's 64 output features are exactly one k block of the next layer: it can start on it before the layer barrier
#ifdef BL_MLP_NO_OWN_FIRST
    constexpr bool OWNC = false;
#else
    constexpr bool OWNC = PASSES == 1 && NT == 2;                // then WAVES * 64 == W
#endif
    const bool own = OWNC;
    // heads' Linears on the un-rectified neck.  The NHpad/32 output tiles are few (3 for 9x9), so each tile's K range is split
    // over two waves (2t: the even k blocks, 2t+1: the odd ones); the odd wave's partial sums go through LDS (R(1) is free
    // by then) to the even one, which adds them in a fixed order and stores.  Wave w's set contains k block w -- the 64
    // features it wrote itself in the last layer -- so with `own` it starts there, like the layers do, ahead of the barrier.
    const int htiles = p.NHpad / 32, KBh = W >> 6, nbh = KBh >> 1;
    auto hblk = [&](int j) {                                      // j-th k block of this wave's unit
        const int j0 = own ? (wave >> 1) : 0;
        const int jj = j0 + j < nbh ? j0 + j : j0 + j - nbh;
        return (wave & 1) + 2 * jj;
    };
    auto heads_prefetch = [&](half8 (&be)[4], half8 (&bo)[4]) {
        if (wave < 2 * htiles) {
            const uint16_t* bt = p.wh + (long)(wave >> 1) * KBh * 2048 + lane * 8;
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) be[s2] = *(const half8*)(bt + hblk(0) * 2048 + s2 * 512);
            if (nbh > 1) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) bo[s2] = *(const half8*)(bt + hblk(1) * 2048 + s2 * 512);
            }
        }
    };
    // What the finish step reads from the tree for env e of this wave (its path's nodes fnode[e] were requested after the staging
    // and have long landed).  Leaf, mover and the path's length are direct loads by env: requested here with the rest (held across
    // the GEMMs they were 3 x EPA more registers).  Lanes beyond the path's length hold whatever the path row held before --
    // clamped to a slot of the env, read, and never used.
    auto finish_request = [&](const int e) __attribute__((always_inline)) {
            const long envbase = (long)(fb[e] < 0 ? 0 : fb[e]) * f.T;
            const long bb = (fb[e] < 0 ? 0 : fb[e]) + zero;
            fleaf[e] = f.leaves[bb]; fmover[e] = f.leaf_seats[bb]; flen[e] = f.path[bb * (f.T + 2)];
            const int nd = fnode[e] < 0 ? 0 : (fnode[e] < f.T ? fnode[e] : f.T - 1);
            const long i = envbase + nd;
            fterm[e] = f.terminal[i]; fn[e] = f.n[i];
            frew[e] = *(const uint32_t*)(f.rewards + i * 2); fw[e] = *(const uint32_t*)(f.w + i * 2);
            const long t = envbase + (lane < f.T ? lane : 0);
            fallW[e] = *(const uint32_t*)(f.w + t * 2); fallN[e] = f.n[t];
    };
    auto layer = [&](const int l, auto first_c, auto last_c) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        constexpr int WC = WAVES * PASSES * NT * 32;                  // == W (mlp_launch picks the instantiation by it)
        const uint16_t* Wl = l == 0 ? p.w0 : p.wb + (long)(l - 1) * W * W;
        const int Kl = l == 0 ? p.K0pad : W;
        half2v al2 = {(f16)0.f, (f16)0.f};
        // (read through the scalar cache: as a vector load its s_waitcnt vmcnt(0) at the top of every layer also waited for the
        // whole weight prefetch instead of its first block)
        if (l > 0) { const f16 a = (f16)((const __attribute__((address_space(4))) float*)p.alphas)[l - 1]; al2[0] = a; al2[1] = a; }   // torch casts the f32 0-dim parameter to f16
        const uint16_t* Rin = R0 + ROWS * ld * ((l + par0) & 1);
        uint16_t* Rn = R0 + ROWS * ld * ((l + 1 + par0) & 1);
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int tile0 = (wave * PASSES + ps) * NT, n0 = tile0 * 32;
            if constexpr (HANDW) {
                // the intake's K is the board's (a run-time loop on compiler-placed waits); the body layers' block loop is
                // straight-line code on the hand-placed weight stream
                if constexpr (FIRST) gemm_run<NT, RD, 0, false, RG, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, 0, false, false);
                else gemm_run<NT, RD, WC / 64, OWNC, RG, true, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, own ? wave : 0, false, true);
            } else
                gemm_run<NT, RD, 0, false, RG, true>(rg, Rin, ld, Wl, Kl, tile0, NT, acc, (own && l > 0) ? wave : 0, own && l > 0, l > 0);
            CLK(2 + 3 * l)
            if constexpr (FINISH && LAST) {
                // what the finish step reads from the tree (the paths requested after the staging have long landed) is requested
                // here, after the last GEMM: the last epilogue and the heads hide the trip, and the GEMMs above do not carry
                // these 24 registers
#pragma unroll
                for (int e = 0; e < EPW; e++) finish_request(e);      // the first pass's envs; the second pass's after this epilogue (registers)
            }
            // next weights in flight before the epilogue: this layer's next pass, or the next layer's first pass
            if (ps + 1 < PASSES) gemm_prefetch<NT, RD>(rg, Wl, Kl, tile0 + NT, NT);
            else if constexpr (!LAST) gemm_prefetch<NT, RD, HANDW>(rg, p.wb + (long)l * W * W, W, wave * PASSES * NT, NT, own ? wave : 0);
            else heads_prefetch(rg.b[0][0], rg.b[1][0]);        // the heads' first two k blocks travel under the last epilogue
#pragma unroll
            for (int rgi = 0; rgi < RG; rgi++) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int f0 = n0 + 32 * t + 8 * g + 4 * hf;             // 4 consecutive features of batch row `32 rgi + brow`
                    const float a4[4] = {acc[rgi * NT + t][4 * g], acc[rgi * NT + t][4 * g + 1], acc[rgi * NT + t][4 * g + 2], acc[rgi * NT + t][4 * g + 3]};
                    uint2 xo, ro;
                    // the layer's bias and this wave's slice of the residual stream come from LDS (round 6; they used to be held in
                    // 16 + 16 RG registers across the GEMM): x_l sits in the buffer the GEMM has just read, x_{l+1} goes to the other
                    const uint2 xold = l == 0 ? make_uint2(0, 0) : *(const uint2*)(Rin + (32 * rgi + brow) * ld + f0);
                    rezero4(a4, *(const uint2*)(BiasL + l * W + f0), xold, al2, l == 0, xo, ro);
                    *(uint2*)(Rn + (32 * rgi + brow) * ld + f0) = xo;          // x itself: the next GEMM rectifies as it reads, the heads read the neck
                }
            }
            }
        }
        CLK(3 + 3 * l)
        if (!own) __syncthreads();                       // else the barrier sits inside the next layer's GEMM (or the heads'), after its first block
        CLK(4 + 3 * l)
    };
    // the last layer is its own copy of the body: it prefetches the heads' weights instead of a next layer's (one loop body
    // doing either keeps both sets of registers alive around the back edge)
    if constexpr (HANDW) {
        if (p.D == 0) layer(0, std::true_type{}, std::true_type{});
        else {
            layer(0, std::true_type{}, std::false_type{});
            for (int l = 1; l < p.D; l++) layer(l, std::false_type{}, std::false_type{});
            layer(p.D, std::false_type{}, std::true_type{});
        }
    } else {
        for (int l = 0; l < p.D; l++) layer(l, std::false_type{}, std::false_type{});
        layer(p.D, std::false_type{}, std::true_type{});
    }
    if constexpr (FINISH && RG > 1) {
#pragma unroll
        for (int e = EPW; e < EPA; e++) finish_request(e);            // consumed after the first pass of the finish
    }
    const uint16_t* X = R0;                                        // the neck (par0 makes the last layer write R(0))
    float* Part = (float*)(R0 + ROWS * ld);                       // [row group][unit][16][64] f32, 4 KiB per unit and row group
    for (int round = 0; round * WAVES < 2 * htiles; round++) {    // same trip count for every wave: uniform barriers
        const int u = round * WAVES + wave;
        const bool active = u < 2 * htiles;
        const int t0 = active ? (u >> 1) : 0, khalf = u & 1;
        const bool first = round == 0;                            // round 0's first two blocks are in flight already
        float16v hacc[RG];
#pragma unroll
        for (int g = 0; g < RG; g++) for (int i = 0; i < 16; i++) hacc[g][i] = 0.f;
        const uint16_t* arow = X + brow * ld + 32 * hf;
        const uint16_t* bt = p.wh + (long)t0 * KBh * 2048 + lane * 8;
        auto loadb = [&](half8 (&b)[4], int kb) {
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) b[s2] = *(const half8*)(bt + kb * 2048 + s2 * 512);
        };
        auto step = [&](half8 (&b)[4], int kb) {
            half8 a[RG][4];
#pragma unroll
            for (int g = 0; g < RG; g++) {
#pragma unroll
                for (int s2 = 0; s2 < 4; s2++) a[g][s2] = *(const half8*)(arow + g * 32 * ld + kb * 64 + 8 * s2);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; s2++) {
#pragma unroll
                for (int g = 0; g < RG; g++) hacc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[s2], a[g][s2], hacc[g], 0, 0, 0);
            }
        };
        half8 (&be)[4] = rg.b[0][0], (&bo)[4] = rg.b[1][0];           // two k blocks in flight, alternating (the layers' ring is free)
        if (active) {
            if (!first) { loadb(be, hblk(0)); if (nbh > 1) loadb(bo, hblk(1)); }
            step(be, hblk(0));
            if (2 < nbh) loadb(be, hblk(2));
        }
        if (first && own) __syncthreads();                        // the last layer's barrier, behind the wave's own block
        if (active) {
            for (int j = 1; j < nbh; j += 2) {
                step(bo, hblk(j));
                if (j + 2 < nbh) loadb(bo, hblk(j + 2));
                if (j + 1 < nbh) {
                    step(be, hblk(j + 1));
                    if (j + 3 < nbh) loadb(be, hblk(j + 3));
                }
            }
        }
        // no barrier here: Part lies in R(1), which nobody has read since the last layer's barrier (the heads read the neck in R(0))
        if (active && khalf == 1) {
#pragma unroll
            for (int rgi = 0; rgi < RG; rgi++) {
#pragma unroll
                for (int i = 0; i < 16; i++) Part[((rgi * htiles + t0) * 16 + i) * 64 + lane] = hacc[rgi][i];
            }
        }
        __syncthreads();
        uint16_t* Out = (uint16_t*)(Part + RG * htiles * 16 * 64);  // [ROWS][NHpad] f16 staging for coalesced stores
        if (active && khalf == 0) {
#pragma unroll
            for (int rgi = 0; rgi < RG; rgi++) {
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int f0 = 32 * t0 + 8 * g + 4 * hf;              // 4 consecutive output features of row `32 rgi + brow`
                uint16_t o[4];
#pragma unroll
                for (int j = 0; j < 4; j++) o[j] = f2h(hacc[rgi][4 * g + j] + Part[((rgi * htiles + t0) * 16 + 4 * g + j) * 64 + lane] + h2f(p.bh[f0 + j]));
                *(uint2*)(Out + (32 * rgi + brow) * p.NHpad + f0) = make_uint2(o[0] | ((uint32_t)o[1] << 16), o[2] | ((uint32_t)o[3] << 16));
            }
            }
        }
        __syncthreads();
    }
    const uint16_t* Out = (const uint16_t*)(Part + RG * htiles * 16 * 64);
    CLK(34)
    if constexpr (!FINISH) {
        // coalesced stores: a wave writes one row's NH-1 policy outputs as consecutive halves
        for (int r = wave; r < ROWS; r += WAVES) {
            if (grow(r) < Mrows) {
                for (int fi = lane; fi < p.NH - 1; fi += 64) p.policy[(long)grow(r) * (p.NH - 1) + fi] = Out[r * p.NHpad + fi];
            }
        }
        if (tid < ROWS && grow(tid) < Mrows) p.value[grow(tid)] = Out[tid * p.NHpad + p.NH - 1];
    } else {
        // ---- bl_sim_finish's work for this wave's four envs, operation for operation as in bl_search.hip:
        // sim_finish_kernel (heads with torch's order; backup cuda.cu:205-236; transition_q's range), but PHASE by phase
        // across the four envs so that their cross-lane exchanges and LDS trips overlap instead of queueing.
        const int A = f.A, T = f.T, Wsm = f.Wsm, iters = f.iters;
        const bool two = iters > 1;
        uint32_t nmin = 0, vmax = 0;
        // the wave's envs EPW at a time (RG passes): what a pass holds in registers is what the RG = 1 kernel holds
        auto finish_envs = [&](auto h_c) __attribute__((always_inline)) {
        constexpr int E0 = EPW * decltype(h_c)::value;
#pragma unroll
        for (int e = 0; e < EPW; e++) flen[E0 + e] = __builtin_amdgcn_readfirstlane(flen[E0 + e]);
        long envbase[EPW]; int leaf[EPW];
        float ev[EPW][2], mx[EPW], sum[EPW];
        // policy head: lane l < Wsm holds actions l + it * Wsm; A <= 128 here, so that is register `it` of the valid bytes
        // the heads' outputs of the four envs: all twelve LDS reads in flight at once, selected afterwards (as `cond ? h2f(Out[..]) : -inf`
        // under `if (lane < A)` every read sat behind its own EXEC branch with a wait: eight LDS round trips in a row; a read past a
        // row's NH entries -- lane + Wsm up to 127 -- stays inside Out + the scratch behind it and is discarded)
        uint16_t o0[EPW], o1[EPW], ov[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const int r = EPA * wave + E0 + e;
            o0[e] = Out[r * p.NHpad + lane]; o1[e] = Out[r * p.NHpad + lane + Wsm]; ov[e] = Out[r * p.NHpad + p.NH - 1];
        }
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            envbase[e] = (long)(fb[E0 + e] < 0 ? 0 : fb[E0 + e]) * T;
            leaf[e] = __builtin_amdgcn_readfirstlane(fleaf[E0 + e]);
            const bool k0 = lane < Wsm && lane < A && ((fvbits >> (2 * (E0 + e))) & 1u), k1 = lane < Wsm && two && lane + Wsm < A && ((fvbits >> (2 * (E0 + e) + 1)) & 1u);
            ev[e][0] = k0 ? h2f(o0[e]) : -INFINITY; ev[e][1] = k1 ? h2f(o1[e]) : -INFINITY;
            mx[e] = two ? ((ev[e][0] > ev[e][1]) ? ev[e][0] : ev[e][1]) : ev[e][0];
        }
        CLK(E0 ? 41 : 54)
#define BLM_MAXSTEP(OFF) if (Wsm > OFF) { _Pragma("unroll") for (int e = 0; e < EPW; e++) { const float o = xor_lane<OFF>(mx[e]); mx[e] = (mx[e] < o) ? o : mx[e]; } }
        BLM_MAXSTEP(32) BLM_MAXSTEP(16) BLM_MAXSTEP(8) BLM_MAXSTEP(4) BLM_MAXSTEP(2) BLM_MAXSTEP(1)
#undef BLM_MAXSTEP
        CLK(E0 ? 42 : 55)
#pragma unroll
        for (int e = 0; e < EPW; e++) { sum[e] = 0.f; sum[e] += expf(ev[e][0] - mx[e]); if (two) sum[e] += expf(ev[e][1] - mx[e]); }
        CLK(E0 ? 43 : 56)
#define BLM_SUMSTEP(OFF) if (Wsm > OFF) { _Pragma("unroll") for (int e = 0; e < EPW; e++) sum[e] = sum[e] + xor_lane<OFF>(sum[e]); }
        BLM_SUMSTEP(32) BLM_SUMSTEP(16) BLM_SUMSTEP(8) BLM_SUMSTEP(4) BLM_SUMSTEP(2) BLM_SUMSTEP(1)
#undef BLM_SUMSTEP
        CLK(E0 ? 44 : 57)
        uint16_t vb0[EPW], vb1[EPW];
        uint16_t lb[EPW][2];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const float lsum = logf(sum[e]);
            lb[e][0] = f2h(ev[e][0] - mx[e] - lsum); lb[e][1] = f2h(ev[e][1] - mx[e] - lsum);
        }
        // the leaf's compacted policy row (bl_device.h: compact_store): pi = exp_table[logit bits] of the kept actions.  The
        // gathers go out FIRST -- ahead of the stores below, so that waiting for them later does not also wait for the
        // stores' acknowledgements (vmcnt retires in order) -- and are consumed after the backup scan, which hides their trip.
        CLK(E0 ? 45 : 58)
        float pi[EPW][2];
        bool in[EPW][2];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            in[e][0] = f.cpi && fb[E0 + e] >= 0 && lane < Wsm && lane < A; in[e][1] = f.cpi && fb[E0 + e] >= 0 && two && lane < Wsm && lane + Wsm < A;
            pi[e][0] = in[e][0] ? f.exp_table[lb[e][0]] : 0.f; pi[e][1] = in[e][1] ? f.exp_table[lb[e][1]] : 0.f;
        }
        CLK(E0 ? 46 : 59)
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            const int r = EPA * wave + E0 + e;
            if (fb[E0 + e] >= 0 && lane < Wsm) {
                uint16_t* dst = f.logits + (envbase[e] + leaf[e]) * A;
                if (lane < A) dst[lane] = lb[e][0];
                if (two && lane + Wsm < A) dst[lane + Wsm] = lb[e][1];
            }
            // value head
            const uint16_t tv = f2h(tanhf(h2f(ov[e])));
            const int mover = __builtin_amdgcn_readfirstlane(fmover[E0 + e]);
            vb0[e] = (mover == 0) ? tv : (uint16_t)(tv ^ 0x8000u); vb1[e] = (uint16_t)(vb0[e] ^ 0x8000u);
            if (fb[E0 + e] >= 0 && lane == 0) { f.v[(envbase[e] + leaf[e]) * 2] = vb0[e]; f.v[(envbase[e] + leaf[e]) * 2 + 1] = vb1[e]; }
        }
        CLK(E0 ? 20 : 35)
        CLK(E0 ? 21 : 36)
        // backup (cuda.cu:205-236), leaf -> root: node j's value is v_j = (terminal_j ? 0 : v_{j+1}) + r_j with v_len the
        // leaf evaluation.  Every lane applies that step to its right neighbour's current value at once; after k rounds
        // the last k nodes of the path are final (each re-evaluation reads a final neighbour and recomputes the same
        // sum), so maxlen rounds finish all four envs' paths -- two DPP instructions per round and env instead of a
        // scalar walk.  w_j = rn16(w_j + rn16(v_j)) then needs no order at all.
        float x0[EPW], x1[EPW], r0[EPW], r1[EPW];
        int maxlen = 0;
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            x0[e] = h2f(vb0[e]); x1[e] = h2f(vb1[e]);                          // lanes >= len keep the leaf evaluation
            r0[e] = h2f((uint16_t)frew[E0 + e]); r1[e] = h2f((uint16_t)(frew[E0 + e] >> 16));
            if (fb[E0 + e] < 0) flen[E0 + e] = 0;
            maxlen = flen[E0 + e] > maxlen ? flen[E0 + e] : maxlen;
        }
        // In Hex a reward and `terminal` only ever sit on the LAST node of a path (a descent stops at a terminal node), so every
        // interior node just passes its successor's value on, plus its own reward +0.0 (which turns a -0 into +0, once):
        // v_j = v_leaf' + 0.0f for j < len - 1, v_leaf' = (terminal ? 0 : v) + r at the last node.  That is two instructions
        // instead of `len` rounds (25 rounds on the deepest paths); the general scan remains for paths that do carry
        // something on an interior node.
        bool plain = true;
#pragma unroll
        for (int e = 0; e < EPW; e++) plain = plain && !__any(lane < flen[E0 + e] - 1 && (fterm[E0 + e] != 0 || frew[E0 + e] != 0u));
        if (plain) {
#pragma unroll
            for (int e = 0; e < EPW; e++) {
                if (flen[E0 + e] > 0) {
                    const float l0 = (fterm[E0 + e] ? 0.f : h2f(vb0[e])) + r0[e], l1 = (fterm[E0 + e] ? 0.f : h2f(vb1[e])) + r1[e];   // right in lane len - 1
                    const float b0 = readlane_f(l0, flen[E0 + e] - 1), b1 = readlane_f(l1, flen[E0 + e] - 1);
                    if (lane < flen[E0 + e] - 1) { x0[e] = b0 + 0.f; x1[e] = b1 + 0.f; }
                    else if (lane == flen[E0 + e] - 1) { x0[e] = b0; x1[e] = b1; }
                }
            }
        } else {
            for (int k = 0; k < maxlen; k++) {
#pragma unroll
                for (int e = 0; e < EPW; e++) {
                    const float n0 = dpp_next_lane(h2f(vb0[e]), x0[e]), n1 = dpp_next_lane(h2f(vb1[e]), x1[e]);
                    if (lane < flen[E0 + e]) { x0[e] = (fterm[E0 + e] ? 0.f : n0) + r0[e]; x1[e] = (fterm[E0 + e] ? 0.f : n1) + r1[e]; }
                }
            }
        }
        float w0[EPW], w1[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            w0[e] = h2f(f2h(h2f((uint16_t)fw[E0 + e]) + h2f(f2h(x0[e]))));
            w1[e] = h2f(f2h(h2f((uint16_t)(fw[E0 + e] >> 16)) + h2f(f2h(x1[e]))));
        }
        if (f.cpi) {
#pragma unroll
            for (int e = 0; e < EPW; e++) {
                const long rowbase = (envbase[e] + leaf[e]) * A;
                const bool k0 = in[e][0] && pi[e][0] != 0.f, k1 = in[e][1] && pi[e][1] != 0.f;
                const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
                const unsigned long long below = (1ull << lane) - 1ull;
                const int c0 = __builtin_popcountll(m0);
                if (k0) { const int j = __builtin_popcountll(m0 & below); f.cpi[rowbase + j] = pi[e][0]; f.cca[rowbase + j] = 0xffff0000u | (uint32_t)lane; }
                if (k1) { const int j = c0 + __builtin_popcountll(m1 & below); f.cpi[rowbase + j] = pi[e][1]; f.cca[rowbase + j] = 0xffff0000u | (uint32_t)(lane + Wsm); }
                if (fb[E0 + e] >= 0 && lane == 0) f.nk[envbase[e] + leaf[e]] = (int16_t)(c0 + __builtin_popcountll(m1));
            }
        }
        CLK(E0 ? 22 : 37)
        // stores, and the q range over all T slots of each env with the path's nodes replaced by their new statistics.  Lane t
        // still holds slot t's old (w, n) and lane j the j-th path node's new ones: a slot's old q counts unless the slot is on
        // the path (a 64-bit mask, OR-reduced over the lanes with DPP), a path node's new q always does.  No LDS involved.
        uint32_t wnew[EPW]; int nnew[EPW];
#pragma unroll
        for (int e = 0; e < EPW; e++) {
            wnew[e] = (uint32_t)f2h(w0[e]) | ((uint32_t)f2h(w1[e]) << 16);
            nnew[e] = (int)(int16_t)(fn[E0 + e] + 2);                            // n += 1 once per seat (cuda.cu:230), int16 wrap kept
            const bool onp = lane < flen[E0 + e];
            const int fnode_e = fnode[E0 + e];
            if (onp) {
                const long i = envbase[e] + fnode_e;
                *(uint32_t*)(f.w + i * 2) = wnew[e];
                f.n[i] = (int16_t)nnew[e];
            }
            const uint32_t lo = (onp && fnode_e < 32) ? (1u << fnode_e) : 0u, hi = (onp && fnode_e >= 32) ? (1u << (fnode_e - 32)) : 0u;
            const uint32_t mlo = wave_or_u32(lo), mhi = wave_or_u32(hi);
            const bool replaced = ((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u) != 0;
            if (fb[E0 + e] >= 0) {
                if (lane < T && !replaced) {
                    const float den = (float)fallN[E0 + e] + 1.e-4f;
                    const uint32_t e0 = enc(h2f((uint16_t)fallW[E0 + e]) / den), e1 = enc(h2f((uint16_t)(fallW[E0 + e] >> 16)) / den);
                    nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
                }
                if (onp) {
                    const float den = (float)nnew[e] + 1.e-4f;
                    const uint32_t e0 = enc(w0[e] / den), e1 = enc(w1[e] / den);
                    nmin = max(nmin, max(~e0, ~e1)); vmax = max(vmax, max(e0, e1));
                }
            }
        }
        CLK(E0 ? 23 : 38)
        };
        finish_envs(std::integral_constant<int, 0>{});
        if constexpr (FPASS > 1) finish_envs(std::integral_constant<int, 1>{});
        static_assert(RG <= 2, "finish_envs is instantiated for two passes");
        // max is associative: one reduction and one conditional atomic pair for all of the wave's envs
        nmin = wave_max_u32(nmin); vmax = wave_max_u32(vmax);
        if (lane == 0 && fb[0] >= 0) {
            uint32_t* q = f.qrange + BLM_QSTRIDE * ((blockIdx.x * WAVES + wave) % BLM_QSLOTS);
            // unconditional: with one pair per wave (1024 per launch over 64 slots) the atomics are cheap, and a checking
            // load first would put a round trip at the very end of every workgroup
            // words in memory = the unsigned codes XOR 0x80000000, compared SIGNED (bl_device.h: BL_QBIAS; include/boardlaw_amd.h)
            atomicMax((int*)q, (int)(nmin ^ 0x80000000u));
            atomicMax((int*)(q + 1), (int)(vmax ^ 0x80000000u));
        }
    }
    CLK(40)
}


// ------------------------------------------------------------------------------------------------------------------
// One Linear (+ ReZero tail) per launch, every layer split over the whole chip: the plan for networks whose weights are too
// many to stream through each workgroup's L1 (mlp_kernel's time is 2 * weights bytes / 64 B/clk per 32-row workgroup: 1024x8 is
// 17.9 MB = 130 us however small the batch), i.e. wide networks on small batches -- 13x13 / 1024 envs / 1024x8 (BASELINE
// config 4's per-GPU shape).  A workgroup of 4 waves takes 32 rows x 128 output features: the input rows (relu applied on the
// way in, as the next block's relu) are staged in LDS once, every wave streams the fragment-major weights of its 32 features
// (the same packing as above) into v_mfma_f32_32x32x16_f16, and the epilogue is the fused kernel's (rezero4: torch's
// rounding points).  The residual stream x lives in global memory between launches (two buffers, ping-pong: a workgroup
// writes features other workgroups still read as inputs).  Per launch a CU moves 64 KiB of activations + 256 KiB of weights
// instead of the whole network.
// ------------------------------------------------------------------------------------------------------------------
struct LayerArgs {
    const uint16_t* X; int ldx, Kvalid, Kpad; int relu_in;        // input rows (M, Kvalid) f16, row stride ldx; zero padded to Kpad
    const uint16_t* Wp; const uint16_t* bias; int N;              // packed (N/32 tiles x Kpad/64 blocks), bias (N); N % 32 == 0
    const uint16_t* Xres; const float* alpha;                     // residual rows (M, N) stride N and its ReZero alpha, or null: y itself
    uint16_t* Y; int ldy;                                         // body: x' (M, N)
    uint16_t* policy; uint16_t* value; int NH;                    // heads (Y == null): features 0..NH-2 -> policy (M, NH-1), NH-1 -> value (M)
    int M;
    int ncol;                                                     // column groups of 128 features per row tile (set by the launcher)
    int xcd;                                                      // 1: row tile r on XCD r % 8 (set by the launcher)
    int sc1_out;                                                  // 1: Y is read by other workgroups of the SAME launch (layers_persist_kernel)
};

struct NoWait { static constexpr bool early = false; __device__ __forceinline__ void operator()() const {} };

// 8-byte relaxed agent-scope atomics = `global_load/store_dwordx2 ... sc1`: the loads bypass the CU's L1, the stores go through to
// memory -- a valid payload form for a cross-workgroup hand-off WITHOUT fences (MI355X_MICROARCH.md, inter-workgroup visibility:
// "8-B agent atomics both sides"); an agent-scope release/acquire pair instead writes back / invalidates whole caches (3.4-8 us per
// hand-off, and polling with acquire loads cuts the chip's bandwidth -- the first version of the kernel below: 447 us per forward).
__device__ __forceinline__ uint2 ld_sc1(const void* p) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint2((uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void st_sc1(void* p, uint2 v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)v.x | ((unsigned long long)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One workgroup's share of one Linear: rows 32 * rowtile .., features 128 * colgroup ..  Contains one workgroup barrier; waves
// without a tile and lanes whose row is beyond M leave after it.
// LRG: row groups of 32 rows per workgroup (every weight fragment then feeds LRG MFMAs: see gemm_run).
template <int RD, int KBC, typename PRE, int LRG = 1>
__device__ __forceinline__ void layer_body(const LayerArgs& a, const int rowtile, const int colgroup, uint16_t* R, PRE pre) {
    const int ld = a.Kpad + 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = rowtile * 32 * LRG, tile = colgroup * 4 + wave, ntiles = a.N >> 5;
    const int brow = lane & 31, hf = lane >> 5;
    Ring<1, RD> rg;
    float16v acc[LRG];
    const bool active = tile < ntiles;
    // EARLY (the persistent kernel): the first weight fragments are requested BEFORE `pre` waits for the previous layer -- weights
    // do not depend on it -- and the input rows after
    constexpr bool EARLY = PRE::early;      // also: this layer's rows come from / go to other workgroups of THIS launch (sc1 payload)
    if (EARLY) { if (active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1); pre(); }
    // the input rows first, then the first weight fragments (vmcnt retires in order: see mlp_kernel)
    const half2v z2 = {(f16)0.f, (f16)0.f};
    if ((a.ldx & 7) == 0 && (a.Kvalid & 7) == 0) {
        // 8 threads per row, 16 bytes each: one 128-byte line per row and step; all of a thread's chunks in flight at once
        // (Kpad <= 1024: at most 16), the weight prefetch right behind them
        const int j = tid & 7;
        constexpr int NB = 16;
        uint4 v[LRG][NB];
#pragma unroll
        for (int rr = 0; rr < LRG; rr++) {
            const int r = (tid >> 3) + 32 * rr;
            const bool rok = row0 + r < a.M;
            const uint16_t* src = a.X + (long)(row0 + r) * a.ldx + 8 * j;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                v[rr][i] = make_uint4(0, 0, 0, 0);
                if (rok && 64 * i + 8 * j < a.Kvalid) {
                    if constexpr (EARLY) { const uint2 lo = ld_sc1(src + 64 * i), hi = ld_sc1(src + 64 * i + 4); v[rr][i] = make_uint4(lo.x, lo.y, hi.x, hi.y); }
                    else v[rr][i] = *(const uint4*)(src + 64 * i);
                }
            }
        }
        if (!EARLY && active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1);
#pragma unroll
        for (int rr = 0; rr < LRG; rr++) {
            uint16_t* dst = R + ((tid >> 3) + 32 * rr) * ld + 8 * j;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if (64 * i < a.Kpad) {
                    uint4 w = v[rr][i];
                    if (a.relu_in) {
                        w.x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.x), z2));
                        w.y = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.y), z2));
                        w.z = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.z), z2));
                        w.w = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, w.w), z2));
                    }
                    *(uint4*)(dst + 64 * i) = w;
                }
            }
        }
    } else {
        // rows that are only 4-byte aligned (the observation: 2 planes per cell): 32-bit words.  Wave w stages rows 8w .. 8w+7,
        // lane l their words l, l + 64, ...; a row's loads all in flight at once (Kpad <= 1024: at most 8 per lane and row)
        const int wpr = a.Kpad >> 1, wvalid = a.Kvalid >> 1;
        constexpr int WMAX = 8;
        uint32_t v[8 * LRG][WMAX];
#pragma unroll
        for (int i = 0; i < 8 * LRG; i++) {
            const int r = wave * 8 * LRG + i;
            const bool rok = row0 + r < a.M;
            const uint32_t* src = (const uint32_t*)a.X + ((long)(row0 + r) * a.ldx >> 1);
#pragma unroll
            for (int k = 0; k < WMAX; k++) {
                const int w = lane + 64 * k;
                v[i][k] = 0u;
                if (64 * k < wvalid) { if (rok && w < wvalid) v[i][k] = src[w]; }
            }
        }
        if (!EARLY && active) gemm_prefetch<1, RD>(rg, a.Wp, a.Kpad, tile, 1);      // behind the rows' loads: vmcnt retires in order
#pragma unroll
        for (int i = 0; i < 8 * LRG; i++) {
            const int r = wave * 8 * LRG + i;
#pragma unroll
            for (int k = 0; k < WMAX; k++) {
                const int w = lane + 64 * k;
                if (w < wpr) {
                    uint32_t x = v[i][k];
                    if (a.relu_in) x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(half2v, x), z2));
                    ((uint32_t*)R)[r * (ld >> 1) + w] = x;
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    const int n0 = tile * 32;
    uint2 biasr[4], xold[LRG][4];
#pragma unroll
    for (int g = 0; g < 4; g++) biasr[g] = *(const uint2*)(a.bias + n0 + 8 * g + 4 * hf);
#pragma unroll
    for (int rgi = 0; rgi < LRG; rgi++) {
        const long grow = row0 + 32 * rgi + brow;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            xold[rgi][g] = make_uint2(0, 0);
            if (a.Xres && grow < a.M) { if constexpr (EARLY) xold[rgi][g] = ld_sc1(a.Xres + grow * a.N + n0 + 8 * g + 4 * hf); else xold[rgi][g] = *(const uint2*)(a.Xres + grow * a.N + n0 + 8 * g + 4 * hf); }
        }
    }
    half2v al2 = {(f16)0.f, (f16)0.f};
    if (a.Xres) { const f16 al = (f16)((const __attribute__((address_space(4))) float*)a.alpha)[0]; al2[0] = al; al2[1] = al; }
    gemm_run<1, RD, KBC, false, LRG>(rg, R, ld, a.Wp, a.Kpad, tile, 1, acc);
#pragma unroll
    for (int rgi = 0; rgi < LRG; rgi++) {
        const long grow = row0 + 32 * rgi + brow;
        if (grow >= a.M) continue;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int f0 = n0 + 8 * g + 4 * hf;
            const float a4[4] = {acc[rgi][4 * g], acc[rgi][4 * g + 1], acc[rgi][4 * g + 2], acc[rgi][4 * g + 3]};
            uint2 xo, ro;
            rezero4(a4, biasr[g], xold[rgi][g], al2, a.Xres == nullptr, xo, ro);
            if (a.Y) { if (a.sc1_out) st_sc1(a.Y + grow * a.ldy + f0, xo); else *(uint2*)(a.Y + grow * a.ldy + f0) = xo; }
            else {
                const uint16_t o[4] = {(uint16_t)xo.x, (uint16_t)(xo.x >> 16), (uint16_t)xo.y, (uint16_t)(xo.y >> 16)};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (f0 + j < a.NH - 1) a.policy[grow * (a.NH - 1) + f0 + j] = o[j];
                    else if (f0 + j == a.NH - 1) a.value[grow] = o[j];
                }
            }
        }
    }
}

// RD - 1 k blocks of 4 KiB in flight per wave; KBC = Kpad / 64 when it is one of the body widths' (the block loop is then
// straight-line code and every MFMA waits for exactly its fragment -- with ONE wave per SIMD there is nobody to hide a
// drained weight stream behind, unlike in mlp_kernel), 0 = any (the intake).
#ifndef BLM_LAYER_RG
#define BLM_LAYER_RG 1          // row groups per workgroup of layer_kernel (measurement switch: 2 = 64-row tiles)
#endif
template <int RD, int KBC>
__global__ void __launch_bounds__(256) layer_kernel(LayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* R = (uint16_t*)smem;
    // Which 32 rows x 128 features this workgroup takes.  Workgroup i runs on XCD i % 8.  All column groups of a row tile are placed
    // on ONE XCD (row tile r on XCD r % 8), and the same way in every layer's launch: the rows a workgroup stages were written by
    // workgroups of its own XCD in the previous launch and are still in that XCD's L2, instead of coming from the seven others
    // through the fabric.  Placement is a speed matter only.
    // Measured (13x13, 1024x8, tools/ab_layers.sh): 1024 rows 94.6 -> 80.5 us per forward; 256 rows 71.3 -> 78.5 us -- there
    // every XCD then streams ALL the weights for its one row tile, where the plain mapping (column group x on XCD x, any row
    // tile) lets an XCD fetch only its eighth of them: the launcher picks by the row count.
    int rowtile, colgroup;
    if (a.xcd) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowtile = xcd + 8 * (slot / a.ncol); colgroup = slot % a.ncol;
        if (rowtile * 32 * BLM_LAYER_RG >= a.M) return;
    } else {
        rowtile = blockIdx.x / a.ncol; colgroup = blockIdx.x % a.ncol;
    }
    layer_body<RD, KBC, NoWait, BLM_LAYER_RG>(a, rowtile, colgroup, R, NoWait{});
}

// ------------------------------------------------------------------------------------------------------------------
// All Linears of the forward in ONE launch (round 4): the launch-per-Linear plan above sits on a floor of ~7.5 us per layer
// whatever the batch (a dependent launch boundary, a cold trip for the rows, the first weight fragments' trip), ten times per
// forward at 1024x8.  But a row tile's next layer depends on THAT ROW TILE's previous layer only: the eight workgroups that own
// its column groups (all on one XCD, see layer_kernel).  So a workgroup keeps its (row tile, column group) through all layers,
// and between two layers it
//   * requests the next layer's first weight fragments (they depend on nothing),
//   * publishes its rows -- every wave a release fence at agent scope, a workgroup barrier, one atomic add on the row tile's
//     counter for that layer -- and waits until the counter says all column groups of the row tile have published (one lane
//     polls with acquire loads, then the workgroup barrier: the CU's vector L1 is invalidated by the acquire, plain loads of
//     the rows follow; /opt/skills/guides recipe G16),
//   * stages the rows and runs the layer as before.
// Waiting only for one's own row tile is also what makes buffer re-use safe: layer l + 1 overwrites the rows layer l read, and
// only after every reader of that row tile has finished layer l.
// Co-residency: a workgroup spins for peers of its row tile, so those must get CUs.  The grid is at most 256 workgroups of 4
// waves and 66 KiB of LDS (two fit a CU), the peers of a row tile lie within 64 consecutive workgroup indices, and every spin is
// BOUNDED: if a wait runs out (another process hogging the chip) the kernel raises `error` in the counter block and finishes
// with whatever it has -- wrong results the host can see (networks.Inference checks the word), never a hung GPU.
// The last workgroup to finish zeroes the counters, so a captured forward replays without a reset launch.
// ------------------------------------------------------------------------------------------------------------------
#define BLM_MAX_LAYERS 12
#define BLM_SPIN_LIMIT (1 << 21)
struct PersistArgs {
    LayerArgs layer[BLM_MAX_LAYERS];
    int nlayers, rowtiles, ncol;      // ncol: column groups of the body layers = the grid's (the heads use fewer)
    int* counters;                    // [rowtiles][nlayers] arrivals + one word: workgroups finished
    int* error;                       // set to 1 when a bounded wait ran out
    int xcd;
    int local;                        // 1: the XCD-local protocol (layers_persist_kernel, round 5): counters = [rowtiles][nlayers][8] flags +
                                      // 8 tickets + one word: workgroups finished
};

__global__ void __launch_bounds__(256) zero_words_kernel(int* p, int n) {
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0;
}

// layers_persist_kernel's wait between two layers: all column groups of the row tile have published the previous layer.  ONE lane
// polls with relaxed loads (acquire loads in a poll loop invalidate the L1 every time), then the workgroup barrier.
struct RowTileWait {
    static constexpr bool early = true;
    const int* counter; int want; int* error;
    __device__ __forceinline__ void operator()() const {
        if (threadIdx.x == 0) {
            int polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > BLM_SPIN_LIMIT) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
};

// The XCD-local protocol's wait: lanes 0 .. want-1 of wave 0 each poll ONE flag -- the word column group c of this row tile stored
// (plain, after its rows) when it had finished the previous layer -- with L1-bypassing loads: producer and poller share an XCD, so
// the word and the rows behind it are in the L2 both sides use.
struct RowTileFlags {
    static constexpr bool early = true;
    const int* flags; int want; int* error;
    __device__ __forceinline__ void operator()() const {
        if ((int)threadIdx.x < want) {
            int polls = 0;
            while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                if (++polls > BLM_SPIN_LIMIT) { __hip_atomic_store(error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
};

// Round 5, `p.local`: the hand-off between two layers stays inside ONE XCD's L2.  Round 4's protocol above is placement-independent:
// rows as write-through (`sc1`) stores, a device-scope counter per row tile and layer, L1-bypassing loads -- which drops every row
// from the producer's L2 and brings it back at the cross-XCD rate even when, as always, the eight workgroups of a row tile DO share
// an XCD (91.9 us per forward against 80.8 for a launch per Linear).  Here a workgroup ASKS where it runs (s_getreg HW_REG_XCC_ID),
// takes a ticket from that XCD's counter and works on row tile xcd + 8 (ticket / ncol), column group ticket % ncol: the peers of
// a row tile share an L2 by construction, not by an assumed dispatch order.  So the rows are plain stores (they stay in that L2),
// `s_waitcnt vmcnt(0)` = acknowledged by it, a barrier, then the workgroup's flag word as a plain store; the peers poll the flags
// and read the rows with L1-bypassing loads, served by the same L2.  No device-scope atomic and no write-through on the path.
// What it needs of the dispatcher: every XCD receives (its row tiles) x ncol workgroups of the grid -- the hardware deals a grid's
// workgroups to the XCDs in turn, and the grid is 8 x ceil(rowtiles / 8) x ncol.  An XCD that received more leaves the surplus
// idle and one that received fewer cannot finish a row tile: its peers' bounded waits run out and raise `error` (wrong results
// the host sees, never a hang), exactly as when a peer is kept off the chip.
template <int RDB, int KBCB>
__global__ void __launch_bounds__(256) layers_persist_kernel(const PersistArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t* R = (uint16_t*)smem;
    const int tid = threadIdx.x;
    int rowtile, colgroup;
    if (p.local) {
        __shared__ int ticket[2];
        if (tid == 0) {
            int xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7;
            ticket[0] = xcc;
            ticket[1] = __hip_atomic_fetch_add(p.counters + (long)p.rowtiles * p.nlayers * 8 + xcc, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        rowtile = ticket[0] + 8 * (ticket[1] / p.ncol); colgroup = ticket[1] % p.ncol;
    } else if (p.xcd) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        rowtile = xcd + 8 * (slot / p.ncol); colgroup = slot % p.ncol;
    } else {
        rowtile = blockIdx.x / p.ncol; colgroup = blockIdx.x % p.ncol;
    }
    int* done = p.counters + (p.local ? (long)p.rowtiles * p.nlayers * 8 + 8 : (long)p.rowtiles * p.nlayers);
    if (p.local) {
        if (rowtile < p.rowtiles) {
            int* mine = p.counters + (long)rowtile * p.nlayers * 8;
            for (int l = 0; l < p.nlayers; l++) {
                const LayerArgs& a = p.layer[l];
                const RowTileFlags wait{mine + (l > 0 ? l - 1 : 0) * 8, l > 0 ? p.layer[l - 1].ncol : 0, p.error};
                if (colgroup < a.ncol) {
                    if (l == 0) layer_body<3, 0>(a, rowtile, colgroup, R, NoWait{});
                    else layer_body<RDB, KBCB>(a, rowtile, colgroup, R, wait);
                }
                if (l + 1 < p.nlayers) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's rows are in the XCD's L2 ...
                    __syncthreads();                                      // ... and so are the other waves' (and nobody still reads R)
                    if (tid == 0 && colgroup < a.ncol) __hip_atomic_store(mine + l * 8 + colgroup, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    } else if (rowtile < p.rowtiles) {
        int* mine = p.counters + (long)rowtile * p.nlayers;
        for (int l = 0; l < p.nlayers; l++) {
            const LayerArgs& a = p.layer[l];
            const RowTileWait wait{mine + (l > 0 ? l - 1 : 0), l > 0 ? p.layer[l - 1].ncol : 0, p.error};
            if (colgroup < a.ncol) {
                if (l == 0) layer_body<3, 0>(a, rowtile, colgroup, R, NoWait{});
                else layer_body<RDB, KBCB>(a, rowtile, colgroup, R, wait);
            }                                                         // (a workgroup beyond the layer's column groups -- the heads' -- has nothing to do)
            if (l + 1 < p.nlayers) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores have been acknowledged ...
                __syncthreads();                                      // ... and so have the other waves' (and nobody still reads R)
                if (tid == 0 && colgroup < a.ncol) __hip_atomic_fetch_add(mine + l, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // the last workgroup out zeroes the counters for the next forward (write-through stores: the words were last written by other
    // XCDs and are next read by them; all 256 threads: a write-through dword is one fabric write each)
    __shared__ int last;
    __syncthreads();
    if (tid == 0) last = __hip_atomic_fetch_add(done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (last) {
        const long words = p.local ? (long)p.rowtiles * p.nlayers * 8 + 8 : (long)p.rowtiles * p.nlayers;
        for (long i = tid; i < words; i += 256) __hip_atomic_store(p.counters + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(done, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace blmlp

#ifdef BL_MLP_CLK
extern "C" int bl_mlp_debug_clk(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(blmlp::g_debug_clk), 64 * 8) == hipSuccess ? 0 : -3; }
#endif
// rows: rows per workgroup -- 32, 64 (widths 256 and 512 only: two 64-row activation buffers of a wider network do not fit the
// LDS), or 0 = by the batch: a workgroup's time is its weight stream through the CU's L1, the same for 32 rows and for 64, so 64
// pay as soon as the 32-row tiles outnumber the chip's 256 CUs (every CU then takes several, one after the other: 32768 rows of
// 512x4 146 -> ... us, profiles/r06_mlp_rows.txt), and lose below that (half of the CUs would idle).  Same bits either way.
static int mlp_launch(const blmlp::Params& p, const blmlp::FinArgs* fin, bl_stream_t stream, int rows = 0) {
    using namespace blmlp;
    const int W = p.W, NHpad = p.NHpad, M = p.M;
    if (rows != 0 && rows != 32 && rows != 64) return BL_EINVAL;
    const bool can64 = W == 256 || W == 512;
    if (rows == 64 && !can64) return BL_ETOOBIG;
    // two activation buffers; the heads keep the neck in the first and stage split-K partials + outputs after it
    // (+ 2 KiB of slack per wave behind them: the finish epilogue's unconditional reads run past a row's end)
    // ... and behind both every layer's bias, (D + 1) x W halves
    auto acts_for = [&](int r) {
        const size_t buf = (size_t)r * (W + 8) * 2;
        const size_t staging = (size_t)(r / 32) * (NHpad / 32) * 16 * 64 * 4 + (size_t)r * NHpad * 2 + (fin ? 8 * 4 * 128 * 4 : 0);
        return buf + (staging > buf ? staging : buf);
    };
    const size_t biases = (size_t)(p.D + 1) * W * 2;
    if (rows == 0) rows = (can64 && (M + 31) / 32 > 256 && acts_for(64) + biases <= 160 * 1024) ? 64 : 32;
    const int RGn = rows / 32;
    const size_t lds = acts_for(rows) + biases;
    if (lds > 160 * 1024) return BL_ETOOBIG;
    Params pp = p;
    pp.bias_off = (int)(acts_for(rows) / 2);
    const dim3 grid(p.xcd_rows ? 8 * ((M + 8 * rows - 1) / (8 * rows)) : (M + rows - 1) / rows);
    hipStream_t hs = (hipStream_t)stream;
    const FinArgs f = fin ? *fin : FinArgs{};
    // above the 64 KiB default the limit has to be raised per kernel (gfx950 has 160 KiB per CU)
#define BL_MLP_LAUNCH1(NT, PASSES, WAVES, FIN, RD, RG)                                                                     \
    {                                                                                                                  \
        static size_t raised[64];                                                                                      \
        if (!bl_raise_lds_limit((const void*)mlp_kernel<NT, PASSES, WAVES, FIN, RD, RG>, lds, raised)) return BL_ELAUNCH;  \
        hipLaunchKernelGGL((mlp_kernel<NT, PASSES, WAVES, FIN, RD, RG>), grid, dim3(WAVES * 64), lds, hs, pp, f);          \
    }
#define BL_MLP_LAUNCH(NT, PASSES, WAVES, RD, RG) { if (fin) BL_MLP_LAUNCH1(NT, PASSES, WAVES, true, RD, RG) else BL_MLP_LAUNCH1(NT, PASSES, WAVES, false, RD, RG) }
    // 8 waves (two per SIMD) from W = 256 up: while one wave waits for its weight fragments the other issues MFMAs
    switch (W / 128) {
        case 1: if (fin) return BL_ETOOBIG; BL_MLP_LAUNCH1(1, 1, 4, false, 3, 1) break;      // the epilogue assumes 8 waves
        case 2: if (RGn == 2) BL_MLP_LAUNCH(1, 1, 8, 3, 2) else BL_MLP_LAUNCH(1, 1, 8, 3, 1) break;
        // 64 rows: accumulators 64 + residual 32 + activations 32 registers; the weight ring keeps two k blocks (one in flight per
        // wave is enough now that every block feeds twice the MFMAs)
        case 4: if (RGn == 2) BL_MLP_LAUNCH(2, 1, 8, BLM_RD64, 2) else BL_MLP_LAUNCH(2, 1, 8, BLM_RD32, 1) break;
        case 6: BL_MLP_LAUNCH(1, 3, 8, 3, 1) break;
        case 8: BL_MLP_LAUNCH(2, 2, 8, 3, 1) break;
        default: return BL_ETOOBIG;
    }
#undef BL_MLP_LAUNCH
#undef BL_MLP_LAUNCH1
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

static int mlp_check(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb, const void* bb,
                     const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad, int NH, int NHpad) {
    if (!obs || !w0 || !b0 || !wh || !bh || M <= 0 || K0 <= 0 || D < 0 || NH < 2) return BL_EINVAL;
    if (D > 0 && (!wb || !bb || !alphas)) return BL_EINVAL;
    if (W % 128 != 0 || W < 128 || W > 1024 || K0pad % 64 != 0 || K0pad < K0 || K0pad > W || NHpad % 32 != 0 || NHpad < NH) return BL_ETOOBIG;
    return BL_OK;
}

extern "C" int bl_mlp_forward_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                  const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                  int K0pad, int NH, int NHpad, void* policy_out, void* value_out, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    Params p{(const uint16_t*)obs, (const uint16_t*)w0, (const uint16_t*)b0, (const uint16_t*)wb, (const uint16_t*)bb, alphas,
             (const uint16_t*)wh, (const uint16_t*)bh, (uint16_t*)policy_out, (uint16_t*)value_out, M, K0, K0pad, W, D, NH, NHpad, 0, nullptr};
    return mlp_launch(p, nullptr, stream);
}

extern "C" int bl_mlp_layers_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                 const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                 int K0pad, int NH, int NHpad, void* scratch, void* policy_out, void* value_out, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out || !scratch) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    if ((K0 & 1) != 0) return BL_EINVAL;
    hipStream_t hs = (hipStream_t)stream;
    uint16_t* buf[2] = {(uint16_t*)scratch, (uint16_t*)scratch + (size_t)M * W};
    const dim3 rows((M + 32 * BLM_LAYER_RG - 1) / (32 * BLM_LAYER_RG));
    int rc = BL_OK;
    auto launch = [&](LayerArgs a, int Kpad) {
        a.ncol = (a.N / 32 + 3) / 4;
        a.xcd = M >= 512;               // row tiles pinned to XCDs once there are enough of them (see layer_kernel)
        const dim3 grid(a.xcd ? 8 * ((rows.x + 7) / 8) * a.ncol : rows.x * a.ncol);
        const size_t l = (size_t)32 * BLM_LAYER_RG * (Kpad + 8) * 2;
#define BL_LAYER_LAUNCH(RD, KBC)                                                                                                  \
        {                                                                                                                         \
            static size_t raised[64];                                                                                             \
            if (!bl_raise_lds_limit((const void*)layer_kernel<RD, KBC>, l, raised)) { rc = BL_ELAUNCH; return; }                   \
            hipLaunchKernelGGL((layer_kernel<RD, KBC>), grid, dim3(256), l, hs, a);                                               \
        }
        switch (Kpad) {
            case 1024: BL_LAYER_LAUNCH(8, 16) break;
            case 768: BL_LAYER_LAUNCH(8, 12) break;
            case 512: BL_LAYER_LAUNCH(6, 8) break;
            case 256: BL_LAYER_LAUNCH(4, 4) break;
            case 384: BL_LAYER_LAUNCH(6, 6) break;      // 13x13's intake
            case 192: BL_LAYER_LAUNCH(3, 3) break;      // 9x9's intake
            default: BL_LAYER_LAUNCH(3, 0) break;
        }
#undef BL_LAYER_LAUNCH
    };
    // intake: x = Linear(obs)
    launch(LayerArgs{(const uint16_t*)obs, K0, K0, K0pad, 0, (const uint16_t*)w0, (const uint16_t*)b0, W, nullptr, nullptr,
                     buf[0], W, nullptr, nullptr, 0, M}, K0pad);
    // ReZero blocks: x' = x + alpha * Linear(relu(x))
    for (int l = 0; l < D; l++)
        launch(LayerArgs{buf[l & 1], W, W, W, 1, (const uint16_t*)wb + (size_t)l * W * W, (const uint16_t*)bb + (size_t)l * W, W,
                         buf[l & 1], alphas + l, buf[(l + 1) & 1], W, nullptr, nullptr, 0, M}, W);
    // heads on the un-rectified neck
    launch(LayerArgs{buf[D & 1], W, W, W, 0, (const uint16_t*)wh, (const uint16_t*)bh, NHpad, nullptr, nullptr, nullptr, 0,
                     (uint16_t*)policy_out, (uint16_t*)value_out, NH, M}, W);
    if (rc != BL_OK) return rc;
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

static int layers_persist_launch(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                 const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                 int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                 void* policy_out, void* value_out, int local, bl_stream_t stream) {
    using namespace blmlp;
    if (!policy_out || !value_out || !scratch || !counters || !error) return BL_EINVAL;
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    if ((K0 & 1) != 0) return BL_EINVAL;
    if (D + 2 > BLM_MAX_LAYERS || (W != 256 && W != 512 && W != 768 && W != 1024)) return BL_ETOOBIG;
    PersistArgs p;
    p.nlayers = D + 2; p.rowtiles = (M + 31) / 32; p.ncol = W / 128; p.counters = counters; p.error = error; p.xcd = local || M >= 512; p.local = local;
    const unsigned grid = p.xcd ? 8u * ((p.rowtiles + 7) / 8) * p.ncol : (unsigned)p.rowtiles * p.ncol;
    if (grid > 256) return BL_ETOOBIG;          // every workgroup must find a CU while its row tile's peers run: see layers_persist_kernel
    uint16_t* buf[2] = {(uint16_t*)scratch, (uint16_t*)scratch + (size_t)M * W};
    auto set = [&](int i, LayerArgs a) { a.ncol = (a.N / 32 + 3) / 4; a.xcd = p.xcd; a.sc1_out = !local && a.Y != nullptr; p.layer[i] = a; };
    set(0, LayerArgs{(const uint16_t*)obs, K0, K0, K0pad, 0, (const uint16_t*)w0, (const uint16_t*)b0, W, nullptr, nullptr, buf[0], W, nullptr, nullptr, 0, M});
    for (int l = 0; l < D; l++)
        set(1 + l, LayerArgs{buf[l & 1], W, W, W, 1, (const uint16_t*)wb + (size_t)l * W * W, (const uint16_t*)bb + (size_t)l * W, W,
                             buf[l & 1], alphas + l, buf[(l + 1) & 1], W, nullptr, nullptr, 0, M});
    set(D + 1, LayerArgs{buf[D & 1], W, W, W, 0, (const uint16_t*)wh, (const uint16_t*)bh, NHpad, nullptr, nullptr, nullptr, 0,
                         (uint16_t*)policy_out, (uint16_t*)value_out, NH, M});
    const int kmax = K0pad > W ? K0pad : W;
    const size_t lds = (size_t)32 * (kmax + 8) * 2;
    hipStream_t hs = (hipStream_t)stream;
    // fresh memory: the counters are zeroed by a launch of their own (a kernel, not a memset node: those replay only once in a captured
    // graph on this ROCm); the kernel leaves them zero, so a caller that keeps the block passes zero_first = 0 from the second call on
    if (zero_first) hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(256), 0, hs, counters, local ? p.rowtiles * p.nlayers * 8 + 9 : p.rowtiles * p.nlayers + 1);
#define BL_PERSIST_LAUNCH(RD, KBC)                                                                                               \
    {                                                                                                                            \
        static size_t raised[64];                                                                                                \
        if (!bl_raise_lds_limit((const void*)layers_persist_kernel<RD, KBC>, lds, raised)) return BL_ELAUNCH;                     \
        hipLaunchKernelGGL((layers_persist_kernel<RD, KBC>), dim3(grid), dim3(256), lds, hs, p);                                 \
    }
    switch (W) {
        case 1024: BL_PERSIST_LAUNCH(8, 16) break;
        case 768: BL_PERSIST_LAUNCH(8, 12) break;
        case 512: BL_PERSIST_LAUNCH(6, 8) break;
        default: BL_PERSIST_LAUNCH(4, 4) break;
    }
#undef BL_PERSIST_LAUNCH
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}

extern "C" int bl_mlp_layers_persist_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                         const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                         int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                         void* policy_out, void* value_out, bl_stream_t stream) {
    return layers_persist_launch(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad, scratch, counters, zero_first, error,
                                 policy_out, value_out, 0, stream);
}

extern "C" int bl_mlp_layers_xcd_f16(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb,
                                     const void* bb, const float* alphas, const void* wh, const void* bh, int W, int D,
                                     int K0pad, int NH, int NHpad, void* scratch, int* counters, int zero_first, int* error,
                                     void* policy_out, void* value_out, bl_stream_t stream) {
    return layers_persist_launch(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad, scratch, counters, zero_first, error,
                                 policy_out, value_out, 1, stream);
}

extern "C" int bl_sim_infer_finish(const bl_search_t* s, int sim, const int16_t* leaves, const void* obs, const uint8_t* valid,
                                   const int32_t* leaf_seats, const void* w0, const void* b0, const void* wb, const void* bb,
                                   const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad, int NHpad,
                                   bl_stream_t stream) {
    using namespace blmlp;
    if (!s || !s->logits || !s->v || !s->w || !s->n || !s->rewards || !s->terminal || !s->qrange || !s->path || !leaves ||
        !valid || !leaf_seats || s->B <= 0 || s->T <= 0 || s->boardsize <= 0 || sim < 1 || sim >= s->T) return BL_EINVAL;
    const int A = s->boardsize * s->boardsize, M = s->B, K0 = 2 * A, NH = A + 1;
    if (s->T > 64 || A > 128 || W < 256) return BL_ETOOBIG;     // the epilogue keeps a whole env in one wave's registers
    if (int rc = mlp_check(obs, M, K0, w0, b0, wb, bb, alphas, wh, bh, W, D, K0pad, NH, NHpad)) return rc;
    const int xcd_rows = s->tune.mlp_no_xcd ? 0 : 1;     // tiles of same-XCD envs (see mlp_kernel)
    Params p{(const uint16_t*)obs, (const uint16_t*)w0, (const uint16_t*)b0, (const uint16_t*)wb, (const uint16_t*)bb, alphas,
             (const uint16_t*)wh, (const uint16_t*)bh, nullptr, nullptr, M, K0, K0pad, W, D, NH, NHpad, xcd_rows, s->n_active};
    int np2 = 1; while (np2 < A) np2 *= 2;
    const int Wsm = np2 < 64 ? np2 : 64;
    FinArgs f{(uint16_t*)s->logits, (uint16_t*)s->v, (uint16_t*)s->w, s->n, (const uint16_t*)s->rewards, s->terminal, s->path,
              s->qrange + (long)BLM_QSLOTS * BLM_QSTRIDE * (sim + 1), leaves, leaf_seats, valid, s->T, A, Wsm, np2 / Wsm,
              (s->cpi && s->cca && s->nk) ? s->cpi : nullptr, s->cca, s->nk, s->exp_table};
    return mlp_launch(p, &f, stream, s->tune.mlp_rows);
}
