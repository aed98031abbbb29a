// How long the chip takes merely to START a grid: kernels whose workgroups do (almost) nothing, at the grid/block shapes
// bl_sim_expand could use for 4096 envs (one workgroup of two waves per env today).  Time per launch from HIP events over
// back-to-back launches on one stream; `work` > 0 adds that many dependent v_add per wave (a stand-in for a body).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void empty_kernel(float* out, int work) {
    float x = threadIdx.x;
    for (int i = 0; i < work; i++) x = x * 1.0001f + 1.f;
    if (work < 0) out[blockIdx.x * blockDim.x + threadIdx.x] = x;      // never: keeps x alive
}
__global__ void lds_kernel(float* out, int work) {
    extern __shared__ float sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    float x = sm[(threadIdx.x + 1) % blockDim.x];
    for (int i = 0; i < work; i++) x = x * 1.0001f + 1.f;
    if (work < 0) out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
static float time_launch(int grid, int block, int work, size_t lds, int reps = 200) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float* out = nullptr;
    for (int i = 0; i < 20; i++) { if (lds) hipLaunchKernelGGL(lds_kernel, dim3(grid), dim3(block), lds, 0, out, work); else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), 0, 0, out, work); }
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) { if (lds) hipLaunchKernelGGL(lds_kernel, dim3(grid), dim3(block), lds, 0, out, work); else hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(block), 0, 0, out, work); }
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / reps;
}
int main() {
    printf("us per launch (back-to-back launches on one stream, so this includes the launch-to-launch gap)\n");
    for (int work : {0, 2000, 20000}) {
        for (auto gb : std::vector<std::pair<int, int>>{{1, 64}, {256, 128}, {1024, 128}, {4096, 128}, {2048, 256}, {1024, 512}, {8192, 64}, {4096, 256}, {4096, 64}, {16384, 64}}) {
            printf("work %5d  grid %5d x block %3d (%5d waves): plain %7.2f   with 128 B LDS + barrier %7.2f\n", work, gb.first, gb.second, gb.first * gb.second / 64,
                   time_launch(gb.first, gb.second, work, 0), time_launch(gb.first, gb.second, work, 128));
        }
    }
    return 0;
}
