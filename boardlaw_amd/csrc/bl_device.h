// bl_device.h -- device helpers shared by the translation units of libboardlaw_amd.so (bl_search.hip, bl_sim.hip, bl_hex.hip, bl_abi.hip, bl_expand.hip, bl_rows.hip,
// bl_mlp.hip): binary16 conversions, the order-preserving float<->u32 map of the q range, wave-wide DPP reductions,
// the search view `Search`, the Hex step on a board held in LDS, and the writer of compacted policy rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include "bl_powf.h"

#pragma clang fp contract(off)

#define BL_QSLOTS 64          // qrange state: 64 slots x {~enc(min), enc(max)} ...
#define BL_QSTRIDE 64         // ... one slot per 256 B (64 words): same-line atomics serialise in L2 (~12 ns each)
#define BL_QWORDS (BL_QSLOTS * BL_QSTRIDE)
#define BL_WAVE 64

namespace bl {

typedef _Float16 f16_t;
__device__ __forceinline__ float h2f(uint16_t b) { return (float)__builtin_bit_cast(f16_t, b); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (f16_t)f); }

// The q-range words IN MEMORY are the u32 codes below XOR BL_QBIAS, i.e. order-preserving as SIGNED int32 -- so that env shards on
// several GPUs get the batch-global range of transition_q (cuda.cu:101-105) from ONE in-place int32 all-reduce(MAX) of the row
// (parallel.allreduce_qrange: RCCL has no unsigned MAX through torch), with no widening launches around it.  The identity of the
// signed MAX is the word 0x80000000 (= code 0): what the reset kernels write.  Inside the kernels the codes stay unsigned.
#define BL_QBIAS 0x80000000u
__device__ __forceinline__ void q_atomic_max(uint32_t* p, uint32_t code) { atomicMax((int*)p, (int)(code ^ BL_QBIAS)); }
__device__ __forceinline__ void q_atomic_max_checked(uint32_t* p, uint32_t code) {       // a load first: skips the atomic when it cannot win
    const int v = (int)(code ^ BL_QBIAS);
    if (v > __hip_atomic_load((int*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax((int*)p, v);
}

// order-preserving float <-> u32
__host__ __device__ __forceinline__ uint32_t enc(float f) {
    uint32_t b = __builtin_bit_cast(uint32_t, f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float dec(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __builtin_bit_cast(float, b);
}

template <int G> __device__ __forceinline__ int gsum(int x) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) x += __shfl_xor(x, m, G);
    return x;
}
template <int G> __device__ __forceinline__ float gmaxf(float x) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) x = fmaxf(x, __shfl_xor(x, m, G));
    return x;
}
template <int G> __device__ __forceinline__ uint32_t gmaxu(uint32_t x) {
#pragma unroll
    for (int m = G / 2; m > 0; m >>= 1) { uint32_t y = __shfl_xor((int)x, m, G); x = x > y ? x : y; }
    return x;
}

// a / b, correctly rounded: instruction for instruction the sequence the compiler emits for `a / b` (LowerFDIV32 with fp32
// denormals on).  The compiler's own expansion hands the numerator's scale flag from v_div_scale to v_div_fmas in VCC, so two
// quotients can only follow each other; here the flag waits in an ordinary SGPR pair and VCC is loaded right before the
// v_div_fmas, so the FMA chains of independent quotients (the blocks of a Newton iteration, the two seats of a slot) interleave.
// (s_nop 3: the ISA's wait states between a write of VCC and v_div_fmas, which the assembler does not insert inside asm.)
__device__ __forceinline__ float ieee_div(float a, float b) {
    float ds, ns, q;
    unsigned long long fd, fn;
    asm("v_div_scale_f32 %0, %1, %3, %3, %2" : "=v"(ds), "=s"(fd) : "v"(a), "v"(b));     // denominator scaled
    asm("v_div_scale_f32 %0, %1, %2, %3, %2" : "=v"(ns), "=s"(fn) : "v"(a), "v"(b));     // numerator scaled; its flag steers v_div_fmas
    const float r = __builtin_amdgcn_rcpf(ds);
    const float nd = -ds;
    const float f0 = __builtin_fmaf(nd, r, 1.0f);
    const float f1 = __builtin_fmaf(f0, r, r);
    const float m = ns * f1;
    const float f2 = __builtin_fmaf(nd, m, ns);
    const float f3 = __builtin_fmaf(f2, f1, m);
    const float f4 = __builtin_fmaf(nd, f3, ns);
    asm("s_mov_b64 vcc, %1\n\ts_nop 3\n\tv_div_fmas_f32 %0, %2, %3, %4" : "=v"(q) : "s"(fn), "v"(f4), "v"(f1), "v"(f3) : "vcc");
    return __builtin_amdgcn_div_fixupf(q, b, a);
}

// N independent quotients with their dependent chains written side by side, step by step: left to itself the scheduler emits
// one quotient after the other (its latency model sees nothing to gain), and a wave that has the SIMD to itself then waits
// out every chain in turn.
template <int N>
__device__ __forceinline__ void ieee_div_n(const float (&a)[N], const float (&b)[N], float (&q)[N]) {
    float ds[N], ns[N], r[N], f0[N], f1[N], m[N], f2[N], f3[N], f4[N];
    unsigned long long fd[N], fn[N];
#pragma unroll
    for (int n = 0; n < N; n++) asm("v_div_scale_f32 %0, %1, %3, %3, %2" : "=v"(ds[n]), "=s"(fd[n]) : "v"(a[n]), "v"(b[n]));
#pragma unroll
    for (int n = 0; n < N; n++) asm("v_div_scale_f32 %0, %1, %2, %3, %2" : "=v"(ns[n]), "=s"(fn[n]) : "v"(a[n]), "v"(b[n]));
#pragma unroll
    for (int n = 0; n < N; n++) r[n] = __builtin_amdgcn_rcpf(ds[n]);
#pragma unroll
    for (int n = 0; n < N; n++) f0[n] = __builtin_fmaf(-ds[n], r[n], 1.0f);
#pragma unroll
    for (int n = 0; n < N; n++) f1[n] = __builtin_fmaf(f0[n], r[n], r[n]);
#pragma unroll
    for (int n = 0; n < N; n++) m[n] = ns[n] * f1[n];
#pragma unroll
    for (int n = 0; n < N; n++) f2[n] = __builtin_fmaf(-ds[n], m[n], ns[n]);
#pragma unroll
    for (int n = 0; n < N; n++) f3[n] = __builtin_fmaf(f2[n], f1[n], m[n]);
#pragma unroll
    for (int n = 0; n < N; n++) f4[n] = __builtin_fmaf(-ds[n], f3[n], ns[n]);
#pragma unroll
    for (int n = 0; n < N; n++) {
        float t;
        asm("s_mov_b64 vcc, %1\n\ts_nop 3\n\tv_div_fmas_f32 %0, %2, %3, %4" : "=v"(t) : "s"(fn[n]), "v"(f4[n]), "v"(f1[n]), "v"(f3[n]) : "vcc");
        q[n] = __builtin_amdgcn_div_fixupf(t, b[n], a[n]);
    }
}

__host__ __device__ __forceinline__ int al16(int x) { return (x + 15) & ~15; }

// The derivative term's denominator, cpu.cpp:60 `powf(bot, 2)`: g++ folds it to bot * bot from -O1 on (the default parity target,
// DESIGN.md section 2); the reference's own JIT build passes no -O flag and calls libm -- bl_tune_t.powf_libm selects that
// (bl_powf.h: glibc 2.35's powf restated, equal to the host libm's on all 2^32 floats; wave-uniform branch).
__device__ __forceinline__ float g_denominator(float bot, int powf_libm) {
    return powf_libm ? bl_powf2_glibc(bot, BLP_LOG2_TAB, BLP_EXP2_TAB) : bot * bot;
}

template <int CTRL, int RM>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, RM, 0xf, false); }
template <int CTRL, int RM>
__device__ __forceinline__ float dpp_f(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, RM, 0xf, false));
}
__device__ __forceinline__ int wave_sum_i32(int v) {     // integer: any order is exact
    v += dpp_i<0x111, 0xf>(0, v); v += dpp_i<0x112, 0xf>(0, v); v += dpp_i<0x114, 0xf>(0, v); v += dpp_i<0x118, 0xf>(0, v);
    v += dpp_i<0x142, 0xa>(0, v); v += dpp_i<0x143, 0xc>(0, v);
    return __builtin_amdgcn_readlane(v, 63);
}
// One v_max_f32_dpp per stage (the update_dpp + fmaxf form costs four VALU instructions per stage: a copy, the DPP move, fmaxf's
// canonicalising max and the max), ISA wait states in front of every DPP read.  Lanes without a source keep their value.  Operands are
// never NaN here, so v_max_f32 is fmaxf bit for bit.
__device__ __forceinline__ float wave_max_f32(float v) {   // max: any order is exact
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)dpp_i<0x111, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x112, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x114, 0xf>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x118, 0xf>(0, (int)v));
    v = max(v, (uint32_t)dpp_i<0x142, 0xa>(0, (int)v)); v = max(v, (uint32_t)dpp_i<0x143, 0xc>(0, (int)v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// Reduce the 64 qrange slots: every lane of the wave returns {lo, hi}.  transition_q, cuda.cu:101-105.
__device__ __forceinline__ void load_qrange(const uint32_t* qr, float& lo, float& hi) {
    const int lane = threadIdx.x & 63;
    uint32_t a = qr[BL_QSTRIDE * lane] ^ BL_QBIAS, b = qr[BL_QSTRIDE * lane + 1] ^ BL_QBIAS;
    a = wave_max_u32(a); b = wave_max_u32(b);       // DPP reductions + readlane (the shuffle butterflies were 12 LDS round trips)
    lo = dec(~a); hi = dec(b);
}
// The same in two halves, so that a kernel can ISSUE the slot loads together with its other prologue loads and reduce later (as one
// call behind other loads' uses the compiler issues it after their waits: a second memory round trip at the start of every workgroup).
__device__ __forceinline__ uint2 qrange_words(const uint32_t* qr) { return *(const uint2*)(qr + BL_QSTRIDE * (threadIdx.x & 63)); }
__device__ __forceinline__ void qrange_reduce(uint2 v, float& lo, float& hi) {
    const uint32_t a = wave_max_u32(v.x ^ BL_QBIAS), b = wave_max_u32(v.y ^ BL_QBIAS);
    lo = dec(~a); hi = dec(b);
}

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ------------------------------------------------------------------------------------------------------------------
// torch's reduce_kernel SUM over the last (contiguous) dimension of a (rows, A) f32 tensor, A < 128, as this torch build on ROCm
// orders it (ATen/native/cuda/Reduce.cuh: setReduceConfig / thread_reduce_impl / block_x_reduce): block width
// Wr = min(largest power of two <= A, 64); lane x < Wr adds v[x] and v[x + Wr] (when that exists) into two of its four
// accumulators, combines them ((a0 + a1) + 0) + 0, and the lanes are then summed by shfl_down with offsets 1, 2, 4, ... Wr/2 --
// the balanced tree ((v0+v1)+(v2+v3))+... that an XOR butterfly with the same offsets gives lanes 0 .. Wr-1 (float addition
// commutes bit for bit); lane 0's total is then handed to every lane (for Wr < 64 the lanes from Wr up only summed zeros: round 5's
// first version returned those, and every board below 8x8 drew action 0 -- found by the seeded-agent test at 5x5).  `own` = this lane's v[x] (0 for x >= Wr), `second` = v[x + Wr] or 0.  Lets a kernel that holds a row
// in registers reproduce `t.sum(-1)` bit for bit without the launch (A >= 128 takes torch's vectorised path, whose order depends
// on each row's address alignment: callers keep torch's own kernels there).
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float torch_row_sum(float own, float second, int Wr) {
    float v = (own + second) + 0.f + 0.f;
    for (int off = 1; off < Wr; off <<= 1) v = v + __shfl_xor(v, off, BL_WAVE);
    return __shfl(v, 0, BL_WAVE);
}
__host__ __device__ __forceinline__ int last_pow2_le(int a) { int w = 1; while (2 * w <= a) w *= 2; return w; }

// ------------------------------------------------------------------------------------------------------------------
// Hex.  Cell codes and rules: boardlaw/hex/cpp/cuda.cu:8-16,76-137; flood cuda.cu:18-74.
// ------------------------------------------------------------------------------------------------------------------
enum { EMPTY = 0, BLACK, WHITE, TOP, BOT, LEFT, RIGHT, MARK = 0xff };

// One group steps one board held in LDS `cells` (A bytes).  Returns the winner's sign in `win` (0 none, +1 black
// wins => rewards (+1,-1), -1 white wins => (-1,+1)).  Group-uniform control flow; `go` false groups idle.
// WAVE: the board belongs to ONE wave (G == 64) of a workgroup whose other waves are elsewhere -- the LDS round trips are ordered
// by a wave-scope fence instead of the workgroup barrier (a wave's DS operations execute in order).
template <bool WAVE> __device__ __forceinline__ void board_sync() {
    if constexpr (WAVE) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}
template <int G, bool WAVE = false>
__device__ __forceinline__ int hex_step_group(uint8_t* cells, int S, int seat, int action, bool go, int gl) {
    const int A = S * S;
    const float invS = 1.0f / (float)S;
    int label = 0, win = 0, start = 0;
    uint8_t plain = 0;
    if (go && gl == 0) {
        const int qd = (int)(((float)action + 0.5f) * invS), rm = action - qd * S;
        const int row = seat == 0 ? qd : rm, col = seat == 0 ? rm : qd;   // white plays transposed, cuda.cu:88-91
        unsigned adj = 0;
        const int dr[6] = {-1, -1, 0, 0, +1, +1}, dc[6] = {0, +1, -1, +1, -1, 0};
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int r = row + dr[k], c = col + dc[k];
            int code;
            if (r < 0) code = TOP; else if (r >= S) code = BOT; else if (c < 0) code = LEFT; else if (c >= S) code = RIGHT;
            else code = cells[r * S + c];
            adj |= 1u << code;
        }
        const bool aT = adj & (1u << TOP), aB = adj & (1u << BOT), aL = adj & (1u << LEFT), aR = adj & (1u << RIGHT);
        if (seat) { if (aL && aR) win = -1; label = aL ? LEFT : (aR ? RIGHT : WHITE); plain = WHITE; }
        else      { if (aT && aB) win = +1; label = aT ? TOP : (aB ? BOT : BLACK); plain = BLACK; }
        start = row * S + col;
        // the reference writes the plain colour then floods from it; a flood relabels the start cell too
        cells[start] = (label >= TOP) ? (uint8_t)MARK : plain;
    }
    label = __shfl(label, 0, G); win = __shfl(win, 0, G);
    plain = (uint8_t)__shfl((int)plain, 0, G);
    const bool flooding = go && label >= TOP;
    board_sync<WAVE>();
    // Relabel the 6-connected component of `plain` cells containing the start cell (== the BFS of cuda.cu:18-74):
    // sweep until no plain cell touches a MARKed one.
    while (true) {
        bool changed = false;
        if (flooding) {
            for (int a = gl; a < A; a += G) {
                if (cells[a] != plain) continue;
                const int r = (int)(((float)a + 0.5f) * invS), c = a - r * S;
                bool hit = false;
                if (r > 0) { hit |= cells[a - S] == MARK; if (c < S - 1) hit |= cells[a - S + 1] == MARK; }
                if (c > 0) hit |= cells[a - 1] == MARK;
                if (c < S - 1) hit |= cells[a + 1] == MARK;
                if (r < S - 1) { hit |= cells[a + S] == MARK; if (c > 0) hit |= cells[a + S - 1] == MARK; }
                if (hit) { cells[a] = MARK; changed = true; }
            }
        }
        board_sync<WAVE>();
        if (!__any(changed)) break;
    }
    if (flooding) for (int a = gl; a < A; a += G) if (cells[a] == MARK) cells[a] = (uint8_t)label;
    board_sync<WAVE>();
    return win;
}

// Hex's step by ONE WAVE on a board in LDS, the flood as a bit-board fill in wave-uniform registers (round 5; the board tiles of
// bl_hex.hip do the same with four lanes per env).  hex_step_group's sweeps cost an LDS round trip per cell and neighbour behind an
// EXEC branch each, and two or more sweeps with a barrier whenever the new stone touches an edge group -- 4 k cycles and up at the END of
// every descent's dependent chain.  Here: the six neighbours of the new stone are read by six lanes at once; the mover's plain-coloured
// cells become a bit set by one ballot per 64 cells; the component grows from the stone by 64-bit shifts on uniform values (scalar
// ALU) until it stops; its cells get the label.  Same component, same bytes (cuda.cu:18-74: the net effect is order-independent).
// NW64 = ceil(A / 64) words; all 64 lanes active; the caller has made the staged board visible to the wave.
template <int NW64>
__device__ __forceinline__ int hex_step_wave(uint8_t* cells, int S, int seat, int action, int lane) {
    const int A = S * S;
    const float invS = 1.0f / (float)S;
    const int qd = (int)(((float)action + 0.5f) * invS), rm = action - qd * S;
    const int row = seat == 0 ? qd : rm, col = seat == 0 ? rm : qd;       // white plays transposed, cuda.cu:88-91
    const int start = row * S + col;
    int code = EMPTY;
    if (lane < 6) {
        const int r = row + (lane < 2 ? -1 : (lane < 4 ? 0 : 1));
        const int c = col + (lane == 1 || lane == 3 ? 1 : (lane == 2 || lane == 4 ? -1 : 0));     // (-1,0) (-1,+1) (0,-1) (0,+1) (+1,-1) (+1,0)
        if (r < 0) code = TOP; else if (r >= S) code = BOT; else if (c < 0) code = LEFT; else if (c >= S) code = RIGHT;
        else code = cells[r * S + c];
    }
    const bool aT = __builtin_amdgcn_ballot_w64(code == TOP) != 0, aB = __builtin_amdgcn_ballot_w64(code == BOT) != 0;
    const bool aL = __builtin_amdgcn_ballot_w64(code == LEFT) != 0, aR = __builtin_amdgcn_ballot_w64(code == RIGHT) != 0;
    int label, win = 0, plain;
    if (seat) { if (aL && aR) win = -1; label = aL ? LEFT : (aR ? RIGHT : WHITE); plain = WHITE; }
    else      { if (aT && aB) win = +1; label = aT ? TOP : (aB ? BOT : BLACK); plain = BLACK; }
    if (label < TOP) {
        if (lane == 0) cells[start] = (uint8_t)plain;                     // no flood: the plain colour (cuda.cu:134)
    } else {
        unsigned long long P[NW64], M[NW64], NF[NW64], NL[NW64];
#pragma unroll
        for (int w = 0; w < NW64; w++) {
            const int a = 64 * w + lane;
            const int r = (int)(((float)a + 0.5f) * invS), c = a - r * S;
            const bool in = a < A;
            P[w] = __builtin_amdgcn_ballot_w64(in && cells[in ? a : 0] == (uint8_t)plain);
            NF[w] = __builtin_amdgcn_ballot_w64(in && c > 0);
            NL[w] = __builtin_amdgcn_ballot_w64(in && c < S - 1);
            M[w] = (start >> 6) == w ? 1ull << (start & 63) : 0ull;
        }
        auto shl = [&](const unsigned long long (&x)[NW64], int k, unsigned long long (&y)[NW64]) {      // 1 <= k <= 63
#pragma unroll
            for (int w = 0; w < NW64; w++) y[w] = (x[w] << k) | (w ? x[w - 1] >> (64 - k) : 0ull);
        };
        auto shr = [&](const unsigned long long (&x)[NW64], int k, unsigned long long (&y)[NW64]) {
#pragma unroll
            for (int w = 0; w < NW64; w++) y[w] = (x[w] >> k) | (w + 1 < NW64 ? x[w + 1] << (64 - k) : 0ull);
        };
        for (int it = 0; it < A; it++) {
            unsigned long long L[NW64], R[NW64], U[NW64], D[NW64], t[NW64];
#pragma unroll
            for (int w = 0; w < NW64; w++) { U[w] = M[w] & NL[w]; D[w] = M[w] & NF[w]; }
            shl(U, 1, L);
            shr(D, 1, R);
#pragma unroll
            for (int w = 0; w < NW64; w++) { U[w] = M[w] | L[w]; D[w] = M[w] | R[w]; }
            shr(U, S, t);
            shl(D, S, U);
            unsigned long long grew = 0;
#pragma unroll
            for (int w = 0; w < NW64; w++) { const unsigned long long nw = (L[w] | R[w] | t[w] | U[w]) & P[w] & ~M[w]; M[w] |= nw; grew |= nw; }
            if (grew == 0) break;
        }
#pragma unroll
        for (int w = 0; w < NW64; w++) if ((M[w] >> lane) & 1ull) cells[64 * w + lane] = (uint8_t)label;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier();
    return win;
}

__device__ __forceinline__ int color_of(int c) { return (c == BLACK || c == TOP || c == BOT) ? 0 : ((c == WHITE || c == LEFT || c == RIGHT) ? 1 : 2); }

struct Search {
    uint16_t* logits; uint16_t* v; uint16_t* w; int16_t* n; int16_t* children; int16_t* parents; int16_t* relation;
    uint16_t* rewards; uint8_t* terminal; uint8_t* boards; int32_t* seats; const uint16_t* c_puct; uint32_t* qrange;
    const float* exp_table; int B, T, S; int obs_f16; int16_t* path; const int32_t* order; int prio_thresh;
    float* cpi; uint32_t* cca; int16_t* nk;      // compacted policy rows, see compact_store()
    int16_t* fav;                                // (B,T) most visited child of a node, a hint for bl_expand.hip's speculative batches
    const int32_t* n_active;                     // device scalar or null: envs >= *n_active sit the simulation out
    int lazy;                                    // bl_tune_t.lazy_init: bl_sim_expand #sim gives slot `sim` its reset values
    int powf_libm;                               // bl_tune_t.powf_libm: the g term's denominator is glibc's powf(bot, 2), not bot * bot
};

// bl_tune_t.lazy_init, called by one wave of env b at the end of bl_sim_expand #sim: what MCTS.__init__ (mcts/__init__.py:43-67)
// put into slot `sim` of the big arrays, written now instead of by bl_sim_init -- children[b,sim,:] = -1 always (a new node has
// no children yet; an unused slot never gets any); when the simulation re-visited a terminal node instead of creating node `sim`,
// also logits[b,sim,:] = NaN and the root world in worlds[b,sim] (a created node gets its logits from the finish step and its
// board from the expansion).
__device__ __forceinline__ void lazy_slot_reset(const Search& s, long envbase, int sim, int A, bool created, int lane, int lanes = 64) {
    int16_t* ch = s.children + (envbase + sim) * A;
    for (int a = lane; a < A; a += lanes) ch[a] = (int16_t)-1;
    if (!created) {
        uint16_t* lg = s.logits + (envbase + sim) * A;
        const uint8_t* root = s.boards + envbase * A;
        uint8_t* brd = s.boards + (envbase + sim) * A;
        for (int a = lane; a < A; a += lanes) { lg[a] = 0x7e00u; brd[a] = root[a]; }
    }
}

// number of envs that take part (bl_search_t.n_active); a scalar load
__device__ __forceinline__ int active_envs(const Search& s) { return s.n_active ? __builtin_amdgcn_readfirstlane(*s.n_active) : s.B; }


// ------------------------------------------------------------------------------------------------------------------
// Compacted policy rows.  A descent level needs, per action with pi = expf(logit) != 0, {pi, the action, its child}.
// Actions with pi == 0 (illegal moves carry logit -inf) contribute s = +0 and g = -0 to the Newton sums -- x + (+-0) == x
// for every partial sum, so they change no rounding -- have probability 0, are never drawn and never get a child
// (cuda.cu:23-25,157-176).  Whoever stores a node's logits therefore also stores the row squeezed to the kept actions,
// in ascending action order:
//     cpi[b,t,j] f32 = exp_table[logit bits]          cca[b,t,j] u32 = child << 16 | action   (child = 0xffff: none yet)
//     nk[b,t]    i16 = number of kept actions
// bl_sim_expand reads these instead of logits[b,t,:] / children[b,t,:] (which stay maintained: they are the API).
// One wave per env; every lane passes its action's f16 logit bits (`a` ascending with the call order, `a < A` lanes
// only) and gets the running count back.  The exp-table gather is the only dependent load.
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int compact_store(const Search& s, long node, int A, int a, bool in, uint16_t logit_bits, int count) {
    const int lane = threadIdx.x & 63;
    float pi = 0.f;
    if (in) pi = s.exp_table[logit_bits];
    const bool keep = in && pi != 0.f;
    const unsigned long long mk = __ballot(keep);
    if (keep) {
        const int j = count + __builtin_popcountll(mk & ((1ull << lane) - 1ull));
        s.cpi[node * A + j] = pi;
        s.cca[node * A + j] = 0xffff0000u | (uint32_t)a;
    }
    return count + __builtin_popcountll(mk);
}

}  // namespace bl
