// Micro-benchmark + correctness check of serial-fold step variants on gfx950.
//   A  two chains (S and g registers), wave_shr:1, no nop            -- round 1's fold: 2 DPP instructions per step
//   B  one chain, row_shr:1, s_nop 1 after every step                -- the ISA's 2 wait states, 1 DPP per step
//   C  one chain, row_shr:1, s_nop 0
//   D  one chain, row_shr:1, nothing in between                      -- expected to give wrong sums (no interlock)
//   E  one chain, wave_shr:1, s_nop 1
//   F  two chains, row_shr:1, no nop
//   G  one chain, row_shr:1, one independent v_mov between steps
// Row variants fold every 16-lane row on its own (4 independent segments per instruction); wave variants fold 64 lanes.
// Prints cycles per STEP per wave at 1 / 1024 / 4096 / 8192 resident waves, and the number of wrong lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define WS "wave_shr:1 row_mask:0xf bank_mask:0xf"
#define RS "row_shr:1 row_mask:0xf bank_mask:0xf"

template <int MODE> __global__ void fold(const float* s, const float* g, float* os, float* og, long long* cyc, int reps) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float ts = s[i], tg = g[i];
    float x = ts, y = tg, z = 0.f;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        x = ts; y = tg;
        asm volatile("s_nop 1");
        for (int j = 0; j < 64; j += 16) {
            if (MODE == 0) asm volatile(R16("v_add_f32_dpp %0, %0, %2 " WS "\n v_add_f32_dpp %1, %1, %3 " WS "\n") : "+v"(x), "+v"(y) : "v"(ts), "v"(tg));
            if (MODE == 1) asm volatile(R16("v_add_f32_dpp %0, %0, %1 " RS "\n s_nop 1\n") : "+v"(x) : "v"(ts));
            if (MODE == 2) asm volatile(R16("v_add_f32_dpp %0, %0, %1 " RS "\n s_nop 0\n") : "+v"(x) : "v"(ts));
            if (MODE == 3) asm volatile(R16("v_add_f32_dpp %0, %0, %1 " RS "\n") : "+v"(x) : "v"(ts));
            if (MODE == 4) asm volatile(R16("v_add_f32_dpp %0, %0, %1 " WS "\n s_nop 1\n") : "+v"(x) : "v"(ts));
            if (MODE == 5) asm volatile(R16("v_add_f32_dpp %0, %0, %2 " RS "\n v_add_f32_dpp %1, %1, %3 " RS "\n") : "+v"(x), "+v"(y) : "v"(ts), "v"(tg));
            if (MODE == 6) asm volatile(R16("v_add_f32_dpp %0, %0, %2 " RS "\n v_mov_b32 %1, %2\n") : "+v"(x), "+v"(z) : "v"(ts));
            // lanes 0..31 only (EXEC upper half clear): does a half-empty wave64 VALU op retire sooner?
            if (MODE == 7) asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n" R16("v_add_f32_dpp %0, %0, %1 " RS "\n s_nop 0\n") "s_mov_b64 exec, s[20:21]\n" : "+v"(x) : "v"(ts) : "s20", "s21");
            if (MODE == 8) asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffffffff\n" R16("v_add_f32_dpp %0, %0, %1 " RS "\n") "s_mov_b64 exec, s[20:21]\n" : "+v"(x) : "v"(ts) : "s20", "s21");
            if (MODE == 9) asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0xffff\n" R16("v_add_f32_dpp %0, %0, %1 " RS "\n s_nop 0\n") "s_mov_b64 exec, s[20:21]\n" : "+v"(x) : "v"(ts) : "s20", "s21");
        }
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
    os[i] = x; og[i] = (MODE == 0 || MODE == 5) ? y : z;
}

static const char* NAMES[10] = {"A 2ch wave_shr        ", "B 1ch row_shr +nop1   ", "C 1ch row_shr +nop0   ", "D 1ch row_shr bare    ",
                               "E 1ch wave_shr +nop1  ", "F 2ch row_shr         ", "G 1ch row_shr +v_mov  ",
                               "H 1ch row_shr nop0 lo32", "I 1ch row_shr bare lo32", "J 1ch row_shr nop0 lo16"};

template <int MODE> void run(int W, const float* s, const float* g, float* os, float* og, long long* cyc, const std::vector<float>& hs,
                             const std::vector<float>& hg) {
    const int N = W * 64, reps = 64;
    hipLaunchKernelGGL(fold<MODE>, dim3(W), dim3(64), 0, 0, s, g, os, og, cyc, reps);
    hipLaunchKernelGGL(fold<MODE>, dim3(W), dim3(64), 0, 0, s, g, os, og, cyc, reps);
    hipDeviceSynchronize();
    std::vector<float> rs(N), rg(N); std::vector<long long> hc(W);
    hipMemcpy(rs.data(), os, N * 4, hipMemcpyDeviceToHost); hipMemcpy(rg.data(), og, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), cyc, W * 8, hipMemcpyDeviceToHost);
    const bool row = !(MODE == 0 || MODE == 4), two = (MODE == 0 || MODE == 5);
    const int lanes = MODE == 7 || MODE == 8 ? 32 : (MODE == 9 ? 16 : 64);
    long bad = 0;
    for (int w = 0; w < W; w++) {
        volatile float a = 0.f, b = 0.f;
        for (int l = 0; l < 64; l++) {
            if (row && (l % 16) == 0) { a = 0.f; b = 0.f; }
            a = a + hs[w * 64 + l]; b = b + hg[w * 64 + l];
            if (l < lanes && a != rs[w * 64 + l]) bad++;
            if (two && b != rg[w * 64 + l]) bad++;
        }
    }
    double c = 0; for (int w = 0; w < W; w++) c += hc[w];
    printf("%s waves %5d : %6.2f cycles/step   wrong lanes %ld of %ld\n", NAMES[MODE], W, c / W / (reps * 64.0), bad, (long)N * (two ? 2 : 1));
}

int main() {
    const int WMAX = 8192, N = WMAX * 64;
    std::vector<float> hs(N), hg(N);
    srand(1);
    for (int i = 0; i < N; i++) { hs[i] = expf(-10.f * rand() / RAND_MAX); hg[i] = -expf(-5.f * rand() / RAND_MAX); }
    float *s, *g, *os, *og; long long* cyc;
    hipMalloc(&s, N * 4); hipMalloc(&g, N * 4); hipMalloc(&os, N * 4); hipMalloc(&og, N * 4); hipMalloc(&cyc, WMAX * 8);
    hipMemcpy(s, hs.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(g, hg.data(), N * 4, hipMemcpyHostToDevice);
    for (int W : {1, 1024, 4096, 8192}) {
        run<0>(W, s, g, os, og, cyc, hs, hg); run<1>(W, s, g, os, og, cyc, hs, hg); run<2>(W, s, g, os, og, cyc, hs, hg);
        run<3>(W, s, g, os, og, cyc, hs, hg); run<4>(W, s, g, os, og, cyc, hs, hg); run<5>(W, s, g, os, og, cyc, hs, hg);
        run<6>(W, s, g, os, og, cyc, hs, hg); run<7>(W, s, g, os, og, cyc, hs, hg); run<8>(W, s, g, os, og, cyc, hs, hg); run<9>(W, s, g, os, og, cyc, hs, hg);
    }
    return 0;
}
