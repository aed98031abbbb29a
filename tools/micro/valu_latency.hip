// Micro-benchmark: cycles per instruction for the patterns the fold uses, one wave per SIMD vs several.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ void dep_add(float* out, float t, long long* cyc) {
    float x = threadIdx.x;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(t));) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x;
}
__global__ void indep_add(float* out, float t, long long* cyc) {
    float x = threadIdx.x, y = 1, z = 2, w = 3;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(t));) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x + y + z + w;
}
__global__ void dpp_pair(float* out, float t, long long* cyc) {
    float x = threadIdx.x, y = 1;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_f32_dpp %0, %0, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0" : "+v"(x), "+v"(y) : "v"(t));) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x + y;
}
__global__ void dpp_row_pair(float* out, float t, long long* cyc) {
    float x = threadIdx.x, y = 1;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_f32_dpp %0, %0, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n s_nop 0" : "+v"(x), "+v"(y) : "v"(t));) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x + y;
}
__global__ void dpp_quad(float* out, float t, long long* cyc) {   // 4 independent dpp chains, no nop
    float x = threadIdx.x, y = 1, z = 2, w = 3;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_add_f32_dpp %0, %0, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(t));) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x + y + z + w;
}
__global__ void readlane_add(float* out, float t, long long* cyc) {
    float x = threadIdx.x, acc = 0;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { REP16(asm volatile("v_readlane_b32 s20, %1, 5\n v_add_f32 %0, s20, %0" : "+v"(acc) : "v"(x) : "s20");) }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = acc;
}
__global__ void div_chain(float* out, float t, long long* cyc) {
    float x = threadIdx.x + 1.5f;
    long long t0 = clock64();
    for (int i = 0; i < 64; i++) { x = t / x; x = t / x; x = t / x; x = t / x; }
    cyc[blockIdx.x] = clock64() - t0; out[threadIdx.x] = x;
}
template <typename F> void run(const char* name, F k, int blocks, int n_instr) {
    float* out; long long* cyc; hipMalloc(&out, 64 * 4); hipMalloc(&cyc, blocks * 8);
    for (int r = 0; r < 2; r++) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, 1.0f, cyc); hipDeviceSynchronize(); }
    long long* h = new long long[blocks]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; i++) s += h[i];
    printf("%-14s blocks %5d : %.2f cycles per instruction (per wave)\n", name, blocks, s / blocks / n_instr);
}
int main() {
    for (int blocks : {1, 1024, 4096}) {
        run("dep_add", dep_add, blocks, 1024);
        run("indep_add x4", indep_add, blocks, 4096);
        run("dpp_pair+nop", dpp_pair, blocks, 2048);
        run("dpp_row_pair", dpp_row_pair, blocks, 2048);
        run("dpp_quad", dpp_quad, blocks, 4096);
        run("readlane+add", readlane_add, blocks, 2048);
        run("div_chain", div_chain, blocks, 256);
    }
}
