// Micro-benchmark + correctness gate for round 4's row-packed Newton iteration: one policy evaluation per
// 16-lane DPP row, FOUR evaluations per wave.  Lane l of a row holds kept actions P*l .. P*l+P-1 of its row's node; an iteration is
//   operands   bot = alpha - q, bot*bot                        (P each)
//   quotients  s = top/bot, g = -top/(bot*bot)                 (2P IEEE divisions per lane, chains side by side)
//   fold       the reference's serial sums: per 16-lane sweep step ONE cross-lane add per chain (v_add_f32_dpp row_shr:1, element
//              0 <- the neighbour's last element) and P-1 in-lane adds; the S and the g chain interleave, so a DPP read follows the
//              write of its source by one instruction (FAST: the wait state the kernel's self-test checks) or a nop more (SAFE)
//   readout    the row's totals (ds_bpermute from the lane of the last kept action; pad elements are top = 0, q = -inf, i.e.
//              s = +0 and g = -0, which leave every partial sum unchanged) and the Newton step per row (cuda.cu:48-65)
// Variants: PLAIN (every add its own instruction) and PK (S and g as the halves of a register pair: in-lane adds, operands and
// the quotients' FMAs as v_pk_*_f32, registers pinned so that the DPP adds can name the halves).
// Prints, per variant and P: cycles per wave-iteration at 1 / 1024 / 2048 / 4096 resident waves, and checks alpha, S, g and the
// iteration count of every row bit for bit against a serial float emulation of the reference on the host.
// Static instruction counts per iteration: llvm-objdump of the loop body (see profiles/r04_fold_rows.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../boardlaw_amd/csrc/bl_device.h"

#pragma clang fp contract(off)
using namespace bl;
typedef float v2f __attribute__((ext_vector_type(2)));

#define RS " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"

// ---- PLAIN fold: xs[p], xg[p] running totals, s[p], g[p] terms.  One sweep step for P elements per lane.
// bound_ctrl:0: lane 0 of a row reads 0.0 for its missing neighbour, i.e. 0.f + t0 -- the reference's `float S = 0.f` start.
template <int P, bool FAST> struct FoldPlain;
#define NOPF(FAST) (FAST ? "" : "s_nop 0\n\t")
template <bool FAST> struct FoldPlain<1, FAST> {
    static __device__ __forceinline__ void steps4(float (&xs)[1], float (&xg)[1], const float (&s)[1], const float (&g)[1]) {
#define ST1(N) N "v_add_f32_dpp %0, %0, %2" RS "v_add_f32_dpp %1, %1, %3" RS
        if constexpr (FAST) asm volatile("s_nop 1\n\t" ST1("") ST1("") ST1("") ST1("") : "+v"(xs[0]), "+v"(xg[0]) : "v"(s[0]), "v"(g[0]));
        else asm volatile("s_nop 1\n\t" ST1("s_nop 0\n\t") ST1("s_nop 0\n\t") ST1("s_nop 0\n\t") ST1("s_nop 0\n\t") : "+v"(xs[0]), "+v"(xg[0]) : "v"(s[0]), "v"(g[0]));
#undef ST1
    }
};
template <bool FAST> struct FoldPlain<2, FAST> {
    static __device__ __forceinline__ void steps4(float (&xs)[2], float (&xg)[2], const float (&s)[2], const float (&g)[2]) {
#define ST2(N) N "v_add_f32_dpp %0, %1, %4" RS "v_add_f32_dpp %2, %3, %6" RS "v_add_f32 %1, %0, %5\n\tv_add_f32 %3, %2, %7\n\t"
        if constexpr (FAST) asm volatile("s_nop 1\n\t" ST2("") ST2("") ST2("") ST2("") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xg[0]), "+v"(xg[1]) : "v"(s[0]), "v"(s[1]), "v"(g[0]), "v"(g[1]));
        else asm volatile("s_nop 1\n\t" ST2("s_nop 0\n\t") ST2("s_nop 0\n\t") ST2("s_nop 0\n\t") ST2("s_nop 0\n\t") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xg[0]), "+v"(xg[1]) : "v"(s[0]), "v"(s[1]), "v"(g[0]), "v"(g[1]));
#undef ST2
    }
};
template <bool FAST> struct FoldPlain<3, FAST> {
    static __device__ __forceinline__ void steps4(float (&xs)[3], float (&xg)[3], const float (&s)[3], const float (&g)[3]) {
#define ST3(N) N "v_add_f32_dpp %0, %2, %6" RS "v_add_f32_dpp %3, %5, %9" RS "v_add_f32 %1, %0, %7\n\tv_add_f32 %4, %3, %10\n\tv_add_f32 %2, %1, %8\n\tv_add_f32 %5, %4, %11\n\t"
        if constexpr (FAST) asm volatile("s_nop 1\n\t" ST3("") ST3("") ST3("") ST3("") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2])
                                         : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(g[0]), "v"(g[1]), "v"(g[2]));
        else asm volatile("s_nop 1\n\t" ST3("s_nop 0\n\t") ST3("s_nop 0\n\t") ST3("s_nop 0\n\t") ST3("s_nop 0\n\t") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2])
                          : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(g[0]), "v"(g[1]), "v"(g[2]));
#undef ST3
    }
};
template <bool FAST> struct FoldPlain<4, FAST> {
    static __device__ __forceinline__ void steps4(float (&xs)[4], float (&xg)[4], const float (&s)[4], const float (&g)[4]) {
#define ST4(N) N "v_add_f32_dpp %0, %3, %8" RS "v_add_f32_dpp %4, %7, %12" RS "v_add_f32 %1, %0, %9\n\tv_add_f32 %5, %4, %13\n\tv_add_f32 %2, %1, %10\n\tv_add_f32 %6, %5, %14\n\t" \
                 "v_add_f32 %3, %2, %11\n\tv_add_f32 %7, %6, %15\n\t"
        if constexpr (FAST) asm volatile("s_nop 1\n\t" ST4("") ST4("") ST4("") ST4("") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3])
                                         : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
        else asm volatile("s_nop 1\n\t" ST4("s_nop 0\n\t") ST4("s_nop 0\n\t") ST4("s_nop 0\n\t") ST4("s_nop 0\n\t") : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]), "+v"(xg[0]), "+v"(xg[1]), "+v"(xg[2]), "+v"(xg[3])
                          : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
#undef ST4
    }
};

// ---- PK fold: X[p] = {xs[p], xg[p]} in v[200+2p : 201+2p], T[p] = {s[p], g[p]} in v[220+2p : 221+2p]
template <int P, bool FAST> struct FoldPk;
#define PKN(FAST) (FAST ? "s_nop 0\n\t" : "s_nop 1\n\t")
template <bool FAST> struct FoldPk<4, FAST> {
    static __device__ __forceinline__ void steps4(v2f (&X)[4], const v2f (&T)[4]) {
#define SP4(N) N "v_add_f32_dpp v200, v206, v220" RS "v_add_f32_dpp v201, v207, v221" RS \
                 "v_pk_add_f32 v[202:203], v[200:201], v[222:223]\n\tv_pk_add_f32 v[204:205], v[202:203], v[224:225]\n\tv_pk_add_f32 v[206:207], v[204:205], v[226:227]\n\t"
        if constexpr (FAST) asm volatile(SP4("s_nop 0\n\t") SP4("s_nop 0\n\t") SP4("s_nop 0\n\t") SP4("s_nop 0\n\t")
                                         : "+{v[200:201]}"(X[0]), "+{v[202:203]}"(X[1]), "+{v[204:205]}"(X[2]), "+{v[206:207]}"(X[3])
                                         : "{v[220:221]}"(T[0]), "{v[222:223]}"(T[1]), "{v[224:225]}"(T[2]), "{v[226:227]}"(T[3]));
        else asm volatile(SP4("s_nop 1\n\t") SP4("s_nop 1\n\t") SP4("s_nop 1\n\t") SP4("s_nop 1\n\t")
                          : "+{v[200:201]}"(X[0]), "+{v[202:203]}"(X[1]), "+{v[204:205]}"(X[2]), "+{v[206:207]}"(X[3])
                          : "{v[220:221]}"(T[0]), "{v[222:223]}"(T[1]), "{v[224:225]}"(T[2]), "{v[226:227]}"(T[3]));
#undef SP4
    }
};
template <bool FAST> struct FoldPk<3, FAST> {
    static __device__ __forceinline__ void steps4(v2f (&X)[3], const v2f (&T)[3]) {
#define SP3(N) N "v_add_f32_dpp v200, v204, v220" RS "v_add_f32_dpp v201, v205, v221" RS \
                 "v_pk_add_f32 v[202:203], v[200:201], v[222:223]\n\tv_pk_add_f32 v[204:205], v[202:203], v[224:225]\n\t"
        if constexpr (FAST) asm volatile(SP3("s_nop 0\n\t") SP3("s_nop 0\n\t") SP3("s_nop 0\n\t") SP3("s_nop 0\n\t")
                                         : "+{v[200:201]}"(X[0]), "+{v[202:203]}"(X[1]), "+{v[204:205]}"(X[2])
                                         : "{v[220:221]}"(T[0]), "{v[222:223]}"(T[1]), "{v[224:225]}"(T[2]));
        else asm volatile(SP3("s_nop 1\n\t") SP3("s_nop 1\n\t") SP3("s_nop 1\n\t") SP3("s_nop 1\n\t")
                          : "+{v[200:201]}"(X[0]), "+{v[202:203]}"(X[1]), "+{v[204:205]}"(X[2])
                          : "{v[220:221]}"(T[0]), "{v[222:223]}"(T[1]), "{v[224:225]}"(T[2]));
#undef SP3
    }
};

// N pairs of IEEE quotients {a.x / b.x, a.y / b.y}: ieee_div_n's sequence with the seven FMA-class steps of two quotients in one
// v_pk_*_f32 each (IEEE fused multiply-add per half, same rounding: checked bit for bit against the host below)
template <int N>
__device__ __forceinline__ void ieee_div_pk(const v2f (&a)[N], const v2f (&b)[N], v2f (&q)[N]) {
    v2f ds[N], ns[N], r[N], f0[N], f1[N], m[N], f2[N], f3[N], f4[N];
    unsigned long long fdx[N], fdy[N], fnx[N], fny[N];
    const v2f one = {1.0f, 1.0f};
#pragma unroll
    for (int n = 0; n < N; n++) {
        asm("v_div_scale_f32 %0, %1, %3, %3, %2" : "=v"(ds[n].x), "=s"(fdx[n]) : "v"(a[n].x), "v"(b[n].x));
        asm("v_div_scale_f32 %0, %1, %3, %3, %2" : "=v"(ds[n].y), "=s"(fdy[n]) : "v"(a[n].y), "v"(b[n].y));
    }
#pragma unroll
    for (int n = 0; n < N; n++) {
        asm("v_div_scale_f32 %0, %1, %2, %3, %2" : "=v"(ns[n].x), "=s"(fnx[n]) : "v"(a[n].x), "v"(b[n].x));
        asm("v_div_scale_f32 %0, %1, %2, %3, %2" : "=v"(ns[n].y), "=s"(fny[n]) : "v"(a[n].y), "v"(b[n].y));
    }
#pragma unroll
    for (int n = 0; n < N; n++) { r[n].x = __builtin_amdgcn_rcpf(ds[n].x); r[n].y = __builtin_amdgcn_rcpf(ds[n].y); }
#pragma unroll
    for (int n = 0; n < N; n++) f0[n] = __builtin_elementwise_fma(-ds[n], r[n], one);
#pragma unroll
    for (int n = 0; n < N; n++) f1[n] = __builtin_elementwise_fma(f0[n], r[n], r[n]);
#pragma unroll
    for (int n = 0; n < N; n++) m[n] = ns[n] * f1[n];
#pragma unroll
    for (int n = 0; n < N; n++) f2[n] = __builtin_elementwise_fma(-ds[n], m[n], ns[n]);
#pragma unroll
    for (int n = 0; n < N; n++) f3[n] = __builtin_elementwise_fma(f2[n], f1[n], m[n]);
#pragma unroll
    for (int n = 0; n < N; n++) f4[n] = __builtin_elementwise_fma(-ds[n], f3[n], ns[n]);
#pragma unroll
    for (int n = 0; n < N; n++) {
        float tx, ty;
        asm("s_mov_b64 vcc, %1\n\ts_nop 3\n\tv_div_fmas_f32 %0, %2, %3, %4" : "=v"(tx) : "s"(fnx[n]), "v"(f4[n].x), "v"(f1[n].x), "v"(f3[n].x) : "vcc");
        asm("s_mov_b64 vcc, %1\n\ts_nop 3\n\tv_div_fmas_f32 %0, %2, %3, %4" : "=v"(ty) : "s"(fny[n]), "v"(f4[n].y), "v"(f1[n].y), "v"(f3[n].y) : "vcc");
        q[n].x = __builtin_amdgcn_div_fixupf(tx, b[n].x, a[n].x);
        q[n].y = __builtin_amdgcn_div_fixupf(ty, b[n].y, a[n].y);
    }
}

__device__ __forceinline__ float row_max_all(float v, int lane) {       // max over the lane's 16-lane row, in every lane of the row
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane | 15) << 2, __builtin_bit_cast(int, v)));
}

#define MAXE 96
// VAR 0: PLAIN, 1: PK.  out per row: alpha, S, g, iterations
template <int P, int VAR, bool FAST>
__global__ void __launch_bounds__(64) rows_kernel(const float* top_g, const float* q_g, const int* nk_g, float* out, long long* cyc, int reps) {
    const int lane = threadIdx.x, l = lane & 15;
    const long rowid = (long)blockIdx.x * 4 + (lane >> 4);
    const int nk = nk_g[rowid];
    float top[P], q[P];
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int e = P * l + p;
        top[p] = e < nk ? top_g[rowid * MAXE + e] : 0.f;
        q[p] = e < nk ? q_g[rowid * MAXE + e] : -INFINITY;
    }
    float a0 = 0.f;
#pragma unroll
    for (int p = 0; p < P; p++) if (P * l + p < nk) a0 = fmaxf(a0, q[p] + fmaxf(top[p], 1.e-4f));
    a0 = row_max_all(a0, lane);
    int nkmax = nk;
    nkmax = max(nkmax, __shfl_xor(nkmax, 16)); nkmax = max(nkmax, __shfl_xor(nkmax, 32));
    nkmax = __builtin_amdgcn_readfirstlane(nkmax);
    const int L = nkmax > 0 ? (nkmax - 1) / P : 0;             // the last lane of a row that holds a kept action
    const int rdaddr = ((lane & 48) | L) << 2;
    float alpha = a0, Sout = 0.f, Gout = 0.f;
    int iters = 0;
    const long long t0 = clock64();
    for (int rep = 0; rep < reps; rep++) {
        alpha = a0; iters = 0;
        float err = INFINITY;
        bool done = false;
        for (int it = 0; it < 101; it++) {
            float S, G;
            if constexpr (VAR == 0) {
                float num[2 * P], den[2 * P], quo[2 * P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float bot = alpha - q[p];
                    num[p] = top[p]; den[p] = bot; num[P + p] = -top[p]; den[P + p] = bot * bot;
                }
                ieee_div_n<2 * P>(num, den, quo);
                float s[P], g[P], xs[P], xg[P];
#pragma unroll
                for (int p = 0; p < P; p++) { s[p] = quo[p]; g[p] = quo[P + p]; }
                xs[0] = 0.f + s[0]; xg[0] = 0.f + g[0];
#pragma unroll
                for (int p = 1; p < P; p++) { xs[p] = xs[p - 1] + s[p]; xg[p] = xg[p - 1] + g[p]; }
                FoldPlain<P, FAST>::steps4(xs, xg, s, g);
                if (L > 4) FoldPlain<P, FAST>::steps4(xs, xg, s, g);
                if (L > 8) FoldPlain<P, FAST>::steps4(xs, xg, s, g);
                if (L > 12) FoldPlain<P, FAST>::steps4(xs, xg, s, g);
                S = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rdaddr, __builtin_bit_cast(int, xs[P - 1])));
                G = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rdaddr, __builtin_bit_cast(int, xg[P - 1])));
            } else {
                v2f num[P], den[P], T[P], X[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float bot = alpha - q[p];
                    num[p].x = top[p]; num[p].y = -top[p];
                    den[p].x = bot; den[p].y = bot * bot;
                }
                ieee_div_pk<P>(num, den, T);
                const v2f zero = {0.f, 0.f};
                X[0] = zero + T[0];
#pragma unroll
                for (int p = 1; p < P; p++) X[p] = X[p - 1] + T[p];
                FoldPk<P, FAST>::steps4(X, T);
                if (L > 4) FoldPk<P, FAST>::steps4(X, T);
                if (L > 8) FoldPk<P, FAST>::steps4(X, T);
                if (L > 12) FoldPk<P, FAST>::steps4(X, T);
                // (temporaries: __builtin_bit_cast of a vector ELEMENT reads element 0 whichever is named -- clang, ROCm 7.2)
                const float tS = X[P - 1].x, tG = X[P - 1].y;
                S = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rdaddr, __builtin_bit_cast(int, tS)));
                G = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(rdaddr, __builtin_bit_cast(int, tG)));
            }
            Sout = S; Gout = G;
            if (it == 100) break;
            const float ne = S - 1.f;
            const bool conv = (ne < 1e-3f) || (err == ne);
            if (!done) iters++;
            done = done || conv;
            const float step = ieee_div(ne, G);
            if (!done) { alpha -= step; err = ne; }
            if (__builtin_amdgcn_ballot_w64(!done) == 0) break;
        }
    }
    if (lane == 0) cyc[blockIdx.x] = clock64() - t0;
    if (l == 0) { out[rowid * 4 + 0] = alpha; out[rowid * 4 + 1] = Sout; out[rowid * 4 + 2] = Gout; out[rowid * 4 + 3] = (float)iters; }
}

// ---- host: the reference's newton_search (cuda.cu:35-68) on one row
static void host_row(const float* top, const float* q, int nk, float& alpha_o, float& S_o, float& G_o, int& iters_o) {
    volatile float alpha = 0.f;
    for (int a = 0; a < nk; a++) { float v = q[a] + fmaxf(top[a], 1.e-4f); if (v > alpha) alpha = v; }
    volatile float err = INFINITY;
    volatile float S = 0.f, G = 0.f;
    int iters = 0;
    for (int it = 0; it < 101; it++) {
        S = 0.f; G = 0.f;
        for (int a = 0; a < nk; a++) {
            volatile float bot = alpha - q[a];
            volatile float bb = bot * bot;
            volatile float s = top[a] / bot, g = (-top[a]) / bb;
            S = S + s; G = G + g;
        }
        if (it == 100) break;
        iters++;
        volatile float ne = S - 1.f;
        if ((ne < 1e-3f) || (err == ne)) break;
        volatile float step = ne / G;
        alpha = alpha - step; err = ne;
    }
    alpha_o = alpha; S_o = S; G_o = G; iters_o = iters;
}

template <int P, int VAR, bool FAST>
void run(const char* name, int W, int nklo, int nkhi, const float* dtop, const float* dq, const int* dnk, float* dout, long long* dcyc,
         const std::vector<float>& htop, const std::vector<float>& hq, std::vector<int>& hnk) {
    const int rows = W * 4, reps = 8;
    srand(7 + nklo);
    for (int r = 0; r < rows; r++) hnk[r] = nklo + rand() % (nkhi - nklo + 1);
    hipMemcpy((void*)dnk, hnk.data(), rows * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((rows_kernel<P, VAR, FAST>), dim3(W), dim3(64), 0, 0, dtop, dq, dnk, dout, dcyc, reps);
    hipLaunchKernelGGL((rows_kernel<P, VAR, FAST>), dim3(W), dim3(64), 0, 0, dtop, dq, dnk, dout, dcyc, reps);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
    std::vector<float> out(rows * 4); std::vector<long long> cyc(W);
    hipMemcpy(out.data(), dout, rows * 16, hipMemcpyDeviceToHost);
    hipMemcpy(cyc.data(), dcyc, W * 8, hipMemcpyDeviceToHost);
    long bad = 0, wave_iters = 0, row_iters = 0;
    const int check = rows < 4096 ? rows : 4096;
    for (int w = 0; w < W; w++) {
        int mx = 0;
        for (int r = 4 * w; r < 4 * w + 4; r++) {
            const int it = (int)out[r * 4 + 3];
            mx = it > mx ? it : mx; row_iters += it;
            if (r < check) {
                float a, S, G; int iters;
                host_row(&htop[(long)r * MAXE], &hq[(long)r * MAXE], hnk[r], a, S, G, iters);
                if (memcmp(&a, &out[r * 4], 4) || memcmp(&S, &out[r * 4 + 1], 4) || memcmp(&G, &out[r * 4 + 2], 4) || iters != it) {
                    if (bad < 3) printf("   row %d nk %d: alpha %.9g/%.9g S %.9g/%.9g g %.9g/%.9g iters %d/%d\n", r, hnk[r], out[r * 4], a, out[r * 4 + 1], S, out[r * 4 + 2], G, it, iters);
                    bad++;
                }
            }
        }
        wave_iters += mx;
    }
    double c = 0; for (int w = 0; w < W; w++) c += cyc[w];
    printf("%-22s P %d nk %2d..%2d waves %5d : %7.1f cycles per wave-iteration (%6.1f per row-iteration in use; rows converge after %.2f of the wave's %.2f iterations)   wrong rows %ld of %d\n",
           name, P, nklo, nkhi, W, c / reps / wave_iters, c / reps / row_iters, (double)row_iters / rows, (double)wave_iters / W, bad, check);
}

int main() {
    const int WMAX = 4096, ROWS = WMAX * 4;
    std::vector<float> htop((long)ROWS * MAXE), hq((long)ROWS * MAXE);
    std::vector<int> hnk(ROWS);
    srand(1);
    for (int r = 0; r < ROWS; r++) {
        // a node of a 64-simulation search: lam ~ 0.03, a softmax-like prior over ~54 legal moves, a handful of visited children
        float pi[MAXE], sum = 0.f;
        for (int e = 0; e < MAXE; e++) { pi[e] = expf(3.f * rand() / RAND_MAX); sum += pi[e]; }
        const float lam = 0.0625f * (54 + rand() % 60) / (54 + rand() % 60 + 81);
        const int visited = 1 + rand() % 10;
        for (int e = 0; e < MAXE; e++) {
            htop[(long)r * MAXE + e] = lam * pi[e] / sum * (MAXE / 54.f);
            hq[(long)r * MAXE + e] = 0.f;
        }
        for (int v = 0; v < visited; v++) hq[(long)r * MAXE + rand() % 40] = (float)(_Float16)(0.2f + 0.7f * rand() / RAND_MAX);
    }
    float *dtop, *dq, *dout; int* dnk; long long* dcyc;
    hipMalloc(&dtop, htop.size() * 4); hipMalloc(&dq, hq.size() * 4); hipMalloc(&dnk, ROWS * 4); hipMalloc(&dout, ROWS * 16); hipMalloc(&dcyc, WMAX * 8);
    hipMemcpy(dtop, htop.data(), htop.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    for (int W : {1, 1024, 2048, 4096}) {
#define RUN(P, V, F, NAME, LO, HI) run<P, V, F>(NAME, W, LO, HI, dtop, dq, dnk, dout, dcyc, htop, hq, hnk)
        RUN(4, 0, true, "PLAIN fast", 50, 58);
        RUN(4, 0, false, "PLAIN safe", 50, 58);
        RUN(4, 1, true, "PK fast", 50, 58);
        RUN(4, 1, false, "PK safe", 50, 58);
        RUN(3, 0, true, "PLAIN fast", 40, 48);
        RUN(3, 1, true, "PK fast", 40, 48);
        RUN(2, 0, true, "PLAIN fast", 20, 32);
        RUN(1, 0, true, "PLAIN fast", 4, 16);
        RUN(4, 0, true, "PLAIN fast, ragged", 1, 64);
        RUN(4, 1, true, "PK fast, ragged", 1, 64);
    }
    return 0;
}
