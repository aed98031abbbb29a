// Issue rate of the fp32 MFMAs on gfx950: cycles per instruction per SIMD with 1 and 2 waves per SIMD, 4 independent
// accumulators per wave (what bl_root_mlp_f32's inner loop issues).  hipcc --offload-arch=gfx950 -O3 -o mfma_f32_rate ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
template <int KIND> __global__ void k(float* out, long long* cyc, int reps, float a0, float b0) {
    f4 acc[4] = {}; f16v big[2] = {};
    float a = a0 + threadIdx.x, b = b0;
    long long t0 = clock64();
    for (int r = 0; r < reps; r++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i & 3], 0, 0, 0);
            else big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i & 1], 0, 0, 0);
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][3]; s += big[0][0] + big[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; long long* cyc; hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 1024 * 8);
    const int reps = 4096;
    for (int kind = 0; kind < 2; kind++) for (int threads : {256, 512}) for (int blocks : {1, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; w++) {
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps, 1.f, 2.f);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, reps, 1.f, 2.f);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double n = (double)reps * 16, flop = (kind == 0 ? 2048.0 : 4096.0) * n * (threads / 64) * blocks;
        printf("%s waves/SIMD %d blocks %3d: %.1f clock64-cycles per MFMA per wave, %.1f ns per MFMA per SIMD, %.1f TFLOP/s\n",
               kind == 0 ? "16x16x4f32" : "32x32x2f32", threads / 256, blocks, c / n, ms * 1e6 / n / (threads / 256), flop / ms / 1e9);
    }
    return 0;
}
