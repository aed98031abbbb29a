// Micro-benchmark + correctness check: the serial fold with P consecutive elements per lane.  A lane holds elements P*l .. P*l+P-1
// of its 16-lane row's chain; one step is ONE cross-lane add (v_add_f32_dpp row_shr:1: element 0 <- the neighbour's last element)
// followed by P-1 plain in-lane adds, so only one in P dependent additions pays the DPP read's wait states.
//   P1  one element per lane, row_shr:1 + s_nop 0 (the kernel's fold today)
//   P2 / P4  two / four elements per lane, s_nop 0 before the next DPP read;  P2n1: s_nop 1;  P2bare: no nop (expected wrong)
// Prints cycles per ELEMENT per wave at 1 / 1024 / 4096 / 8192 resident waves and the number of wrong prefix totals.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define R4(x) x x x x
#define R16(x) R4(R4(x))
#define RS "row_shr:1 row_mask:0xf bank_mask:0xf"

template <int MODE> __global__ void fold(const float* s, float* os, long long* cyc, int reps) {
    constexpr int P = MODE == 0 ? 1 : (MODE == 4 ? 4 : 2);
    const int i = (blockIdx.x * 64 + threadIdx.x) * P;
    float t[4] = {0, 0, 0, 0}, x[4] = {0, 0, 0, 0};
    for (int p = 0; p < P; p++) t[p] = s[i + p];
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
        x[0] = t[0];
        if (P >= 2) x[1] = x[0] + t[1];
        if (P >= 4) { x[2] = x[1] + t[2]; x[3] = x[2] + t[3]; }
        asm volatile("s_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
        if (MODE == 0) asm volatile(R16("v_add_f32_dpp %0, %0, %1 " RS "\n s_nop 0\n") : "+v"(x[0]) : "v"(t[0]));
        if (MODE == 1) asm volatile(R16("v_add_f32_dpp %0, %1, %2 " RS "\n v_add_f32 %1, %0, %3\n s_nop 0\n") : "+v"(x[0]), "+v"(x[1]) : "v"(t[0]), "v"(t[1]));
        if (MODE == 2) asm volatile(R16("v_add_f32_dpp %0, %1, %2 " RS "\n v_add_f32 %1, %0, %3\n s_nop 1\n") : "+v"(x[0]), "+v"(x[1]) : "v"(t[0]), "v"(t[1]));
        if (MODE == 3) asm volatile(R16("v_add_f32_dpp %0, %1, %2 " RS "\n v_add_f32 %1, %0, %3\n") : "+v"(x[0]), "+v"(x[1]) : "v"(t[0]), "v"(t[1]));
        if (MODE == 4) asm volatile(R16("v_add_f32_dpp %0, %3, %4 " RS "\n v_add_f32 %1, %0, %5\n v_add_f32 %2, %1, %6\n v_add_f32 %3, %2, %7\n s_nop 0\n")
                                    : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]));
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = clock64() - t0;
    for (int p = 0; p < P; p++) os[i + p] = x[p];
}

static const char* NAMES[5] = {"P1 row_shr +nop0      ", "P2 dpp+add +nop0      ", "P2 dpp+add +nop1      ", "P2 dpp+add bare       ", "P4 dpp+3 adds +nop0   "};

template <int MODE> void run(int W, const float* s, float* os, long long* cyc, const std::vector<float>& hs) {
    constexpr int P = MODE == 0 ? 1 : (MODE == 4 ? 4 : 2);
    const int N = W * 64 * P, reps = 64;
    hipLaunchKernelGGL(fold<MODE>, dim3(W), dim3(64), 0, 0, s, os, cyc, reps);
    hipLaunchKernelGGL(fold<MODE>, dim3(W), dim3(64), 0, 0, s, os, cyc, reps);
    hipDeviceSynchronize();
    std::vector<float> rs(N); std::vector<long long> hc(W);
    hipMemcpy(rs.data(), os, N * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), cyc, W * 8, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int w = 0; w < W; w++) {
        volatile float a = 0.f;
        for (int l = 0; l < 64 * P; l++) {
            if ((l % (16 * P)) == 0) a = 0.f;
            a = a + hs[w * 64 * P + l];
            if (a != rs[w * 64 * P + l]) bad++;
        }
    }
    double c = 0; for (int w = 0; w < W; w++) c += hc[w];
    printf("%s waves %5d : %6.2f cycles/element (%6.2f per 16-lane sweep step)   wrong totals %ld of %ld\n", NAMES[MODE], W, c / W / (reps * 16.0 * P),
           c / W / (reps * 16.0), bad, (long)N);
}

int main() {
    const int WMAX = 8192, N = WMAX * 64 * 4;
    std::vector<float> hs(N);
    srand(1);
    for (int i = 0; i < N; i++) hs[i] = (rand() & 1 ? 1.f : -1.f) * expf(-10.f * rand() / RAND_MAX);
    float *s, *os; long long* cyc;
    hipMalloc(&s, N * 4); hipMalloc(&os, N * 4); hipMalloc(&cyc, WMAX * 8);
    hipMemcpy(s, hs.data(), N * 4, hipMemcpyHostToDevice);
    for (int W : {1, 1024, 4096, 8192}) {
        run<0>(W, s, os, cyc, hs); run<1>(W, s, os, cyc, hs); run<2>(W, s, os, cyc, hs); run<3>(W, s, os, cyc, hs); run<4>(W, s, os, cyc, hs);
    }
    return 0;
}
