// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of the search kernels.
// Each kernel moves a KNOWN number of bytes over a buffer far larger than the 256 MiB Infinity Cache:
//   stream16   16 B per lane, fully coalesced read      (the guide's case: FETCH_SIZE reports half the bytes)
//   rows2      one wave reads rows of 81 x 2 B with a 10 KiB stride between rows  (children[b,t,:] / logits[b,t,:])
//   rows4      one wave reads 54 x 4 B rows with a stride                         (compacted cpi / cca rows)
//   gather4    64 lanes x 4 B at random 4-byte addresses (child statistics, exp table)
//   write2     one wave writes rows of 81 x 2 B
// Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass) and compare with the
// byte counts this program prints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void stream16(const uint4* p, size_t n, uint4* out) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = p[i]; acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
    if (acc.x == 0x12345678u) out[0] = acc;
}
__global__ void rows2(const uint16_t* p, size_t rows, size_t stride_halves, uint16_t* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (size_t r = wave; r < rows; r += nw) { const uint16_t* row = p + r * stride_halves; acc ^= row[lane]; if (lane + 64 < 81) acc ^= row[lane + 64]; }
    if (acc == 0x1234u) out[0] = (uint16_t)acc;
}
__global__ void rows4(const uint32_t* p, size_t rows, size_t stride_words, uint32_t* out) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    unsigned acc = 0;
    for (size_t r = wave; r < rows; r += nw) { if (lane < 54) acc ^= p[r * stride_words + lane]; }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void gather4(const uint32_t* p, size_t words, size_t n, uint32_t* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        size_t h = i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; acc ^= p[h % words];
    }
    if (acc == 0x12345u) out[0] = acc;
}
__global__ void write2(uint16_t* p, size_t rows, size_t stride_halves) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t r = wave; r < rows; r += nw) { uint16_t* row = p + r * stride_halves; row[lane] = (uint16_t)r; if (lane + 64 < 81) row[lane + 64] = (uint16_t)lane; }
}

int main() {
    const size_t bytes = 3ull << 30;                       // 3 GiB >> 256 MiB
    void* buf; void* out;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, bytes); hipDeviceSynchronize();
    const size_t n16 = bytes / 16;
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const uint4*)buf, n16, (uint4*)out);
    printf("stream16 read bytes %zu\n", n16 * 16);
    const size_t stride = 5120, rows = bytes / (stride * 2);                 // 10 KiB between rows
    hipLaunchKernelGGL(rows2, dim3(4096), dim3(256), 0, 0, (const uint16_t*)buf, rows, stride, (uint16_t*)out);
    printf("rows2    read bytes %zu  (%zu rows x 162 B; 64-B lines touched per row: 3-4 -> %zu..%zu)\n", rows * 162, rows, rows * 192, rows * 256);
    const size_t rows4n = bytes / (stride * 2);
    hipLaunchKernelGGL(rows4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, rows4n, stride / 2, (uint32_t*)out);
    printf("rows4    read bytes %zu  (%zu rows x 216 B; lines per row: 4 -> %zu)\n", rows4n * 216, rows4n, rows4n * 256);
    const size_t ng = 64ull << 20;
    hipLaunchKernelGGL(gather4, dim3(4096), dim3(256), 0, 0, (const uint32_t*)buf, bytes / 4, ng, (uint32_t*)out);
    printf("gather4  read bytes %zu  (%zu random words; one 64-B line each -> %zu)\n", ng * 4, ng, ng * 64);
    hipLaunchKernelGGL(write2, dim3(4096), dim3(256), 0, 0, (uint16_t*)buf, rows, stride);
    printf("write2   written bytes %zu  (lines touched %zu..%zu)\n", rows * 162, rows * 192, rows * 256);
    hipDeviceSynchronize();
    return 0;
}
