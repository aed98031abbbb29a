// Does gfx950 interlock "VALU writes VGPR -> DPP reads it" when fewer than 2 wait states separate them?
// Runs the serial-prefix fold with {2 chains + s_nop}, {2 chains, no nop}, {1 chain, back to back} and compares every
// lane of the result with the host's serial float sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#define S_NOP "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0\n"
#define S_NONOP "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define S_ONE "v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define R8(x) x x x x x x x x
template <int MODE> __global__ void fold(const float* s, const float* g, float* os, float* og) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    float ts = s[i], tg = g[i], x = ts, y = tg;
    for (int j = 0; j < 64; j += 8) {
        if (MODE == 0) asm volatile("s_nop 1\n" R8(S_NOP) : "+v"(x), "+v"(y) : "v"(ts), "v"(tg));
        if (MODE == 1) asm volatile("s_nop 1\n" R8(S_NONOP) : "+v"(x), "+v"(y) : "v"(ts), "v"(tg));
        if (MODE == 2) { asm volatile("s_nop 1\n" R8(S_ONE) : "+v"(x), "+v"(y) : "v"(ts), "v"(tg)); }
    }
    if (MODE == 2) { for (int j = 0; j < 64; j += 8) asm volatile("s_nop 1\n" R8(S_ONE) : "+v"(y), "+v"(x) : "v"(tg), "v"(ts)); }
    os[i] = x; og[i] = y;
}
int main() {
    const int W = 8192, N = W * 64;
    float *hs = new float[N], *hg = new float[N], *rs = new float[N], *rg = new float[N];
    srand(1);
    for (int i = 0; i < N; i++) { hs[i] = expf(-10.f * rand() / RAND_MAX); hg[i] = -expf(-5.f * rand() / RAND_MAX); }
    float *s, *g, *os, *og; hipMalloc(&s, N * 4); hipMalloc(&g, N * 4); hipMalloc(&os, N * 4); hipMalloc(&og, N * 4);
    hipMemcpy(s, hs, N * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg, N * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode++) {
        long bad = 0;
        for (int rep = 0; rep < 20; rep++) {
            if (mode == 0) hipLaunchKernelGGL(fold<0>, dim3(W), dim3(64), 0, 0, s, g, os, og);
            if (mode == 1) hipLaunchKernelGGL(fold<1>, dim3(W), dim3(64), 0, 0, s, g, os, og);
            if (mode == 2) hipLaunchKernelGGL(fold<2>, dim3(W), dim3(64), 0, 0, s, g, os, og);
            hipMemcpy(rs, os, N * 4, hipMemcpyDeviceToHost); hipMemcpy(rg, og, N * 4, hipMemcpyDeviceToHost);
            for (int w = 0; w < W; w++) { volatile float a = 0.f, b = 0.f; for (int l = 0; l < 64; l++) { a = a + hs[w * 64 + l]; b = b + hg[w * 64 + l]; if (a != rs[w * 64 + l] || b != rg[w * 64 + l]) bad++; } }
        }
        printf("mode %d (%s): %ld mismatching lanes of %ld\n", mode, mode == 0 ? "2 chains + s_nop" : mode == 1 ? "2 chains, no nop" : "1 chain back-to-back", bad, 20L * N * 1);
    }
}
