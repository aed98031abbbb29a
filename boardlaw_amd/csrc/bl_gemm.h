// bl_gemm.h -- what the MFMA network kernels of bl_mlp.hip (all Linears in one kernel) and bl_layers.hip (a launch per Linear) share:
// the fragment-major weight ring and its loads (compiler-placed, and inline-asm with hand-placed waits), the block loop of one layer's
// GEMM on v_mfma_f32_32x32x16_f16 (one or several row groups per weight fragment), the ReZero epilogue with torch's rounding points, and
// the argument check of the C entry points.  Split out of bl_mlp.hip in round 6.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/boardlaw_amd.h"
#include "bl_host.h"

namespace blmlp {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef _Float16 f16;

__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (f16)f); }
__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(f16, b); }

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
    // the same in two halves for the hand-placed stream (-DBLM_A_BEFORE_WAIT, a measurement switch): a block's activation fragments
    // do not depend on its weights, so their LDS reads can go out BEFORE the wait for the weights (between two sched_barriers the
    // compiler cannot do that itself).  Measured neutral (profiles/r06_a_before_wait.txt: 4096 envs +0.2 %, 32768 envs -1 %) at 16 RG
    // more registers: off.
    auto load_a = [&](half8 (&a)[RG][4], int kb) {
#pragma unroll
        for (int g = 0; g < RG; g++) {
#pragma unroll
            for (int s = 0; s < 4; s++) a[g][s] = *(const half8*)(arow + g * 32 * ldin + kb * 64 + 8 * s);
        }
    };
    auto mma = [&](half8 (&b)[NT][4], half8 (&a)[RG][4]) {
        // k piece by k piece, all RG x NT accumulators in turn: an accumulator is touched every (RG NT)-th MFMA
#pragma unroll
        for (int s = 0; s < 4; s++) {
#pragma unroll
            for (int g = 0; g < RG; g++) {
                if constexpr (RELUR) a[g][s] = __builtin_bit_cast(half8, __builtin_elementwise_max(__builtin_bit_cast(short8, a[g][s]), floor8));
#pragma unroll
                for (int t = 0; t < NT; t++) acc[g * NT + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[t][s], a[g][s], acc[g * NT + t], 0, 0, 0);
            }
        }
    };
    if constexpr (KBC > 0 && ASMW) {
        static_assert(KBC >= RD - 1, "the prefetch requests RD - 1 blocks");
#pragma unroll
        for (int i = 0; i < KBC; i++) {
            constexpr int LOADS = KBC - (RD - 1);             // steps 0 .. LOADS - 1 request block i + RD - 1
            if (i < LOADS) ring_load_asm<NT>(rg.b[(i + RD - 1) % RD], Wp, KBC, tile0, blk(i + RD - 1));
#ifdef BLM_A_BEFORE_WAIT
            half8 afr[RG][4];
            load_a(afr, blk(i));
#endif
            // blocks requested after block i at this point: i + 1 .. min(i + RD - 1, KBC - 1)
            const int after = (i + RD - 1 < KBC ? i + RD - 1 : KBC - 1) - i;
            if (after >= 3) ring_wait<3, NT>(rg.b[i % RD]);
            else if (after == 2) ring_wait<2, NT>(rg.b[i % RD]);
            else if (after == 1) ring_wait<1, NT>(rg.b[i % RD]);
            else ring_wait<0, NT>(rg.b[i % RD]);
#ifndef BLM_A_BEFORE_WAIT
            compute(rg.b[i % RD], blk(i));
#else
            mma(rg.b[i % RD], afr);
#endif
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


}  // namespace blmlp

static inline int mlp_check(const void* obs, int M, int K0, const void* w0, const void* b0, const void* wb, const void* bb,
                     const float* alphas, const void* wh, const void* bh, int W, int D, int K0pad, int NH, int NHpad) {
    if (!obs || !w0 || !b0 || !wh || !bh || M <= 0 || K0 <= 0 || D < 0 || NH < 2) return BL_EINVAL;
    if (D > 0 && (!wb || !bb || !alphas)) return BL_EINVAL;
    if (W % 128 != 0 || W < 128 || W > 1024 || K0pad % 64 != 0 || K0pad < K0 || K0pad > W || NHpad % 32 != 0 || NHpad < NH) return BL_ETOOBIG;
    return BL_OK;
}

