// bl_rand.hip -- bl_rand_block: the T-1 descend uniforms of a move as ONE launch, stream-identical to the reference's protocol.
//
// The reference's descend draws `at::rand_like(logits.select(2, 0))` -- a (B,T) f16 tensor -- from torch's generator once per
// simulation (boardlaw/mcts/cpp/cuda.cu:191).  On the device that is torch's grid-stride Philox kernel
// (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel + uniform_kernel):
//     thread idx < threads draws, in loop l, the Philox4x32-10 block of  key = seed,  counter = {offset/4 + l  (low 64 bits),
//     idx (high 64 bits)}  and writes component j to element idx + threads*(4l + j) (if < numel) as
//     f16(2^-32 + float(u) * 2^-32), with 1.0 mapped to 0 (curand's (0,1] turned into [0,1));
//     the generator's offset then advances by 4*loops.
// Call c of a move therefore sees offset + c*4*loops, and all n_calls calls are one grid here: same seed, same counters, same
// conversion -- the same bits (tests/test_rng_stream.py compares the block with n_calls stacked torch.rand_like tensors on the
// device, and a whole seeded search under MoveRng with the same search under TorchRng).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"
#include "bl_host.h"

namespace bl {

__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long m0 = (unsigned long long)0xD2511F53u * c.x, m1 = (unsigned long long)0xCD9E8D57u * c.z;
        c = uint4{(unsigned)(m1 >> 32) ^ c.y ^ k.x, (unsigned)m1, (unsigned)(m0 >> 32) ^ c.w ^ k.y, (unsigned)m0};
        k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
    }
    return c;
}

// hiprand_uniform (rocrand_uniform.h: uniform_distribution) then torch's uniform_kernel for Half with from = 0, to = 1
__device__ __forceinline__ uint16_t uniform_f16(unsigned int u) {
    const float r = 2.3283064365386963e-10f + (float)u * 2.3283064365386963e-10f;
    const uint16_t h = f2h(r);
    return h == 0x3c00u ? (uint16_t)0 : h;
}

// grid: x over a call's threads (idx), y over (call, loop) -- no integer division anywhere; the launch is bound by the Philox
// rounds' 32-bit multiplies (quarter rate), 40 per element at torch's one-element-per-thread geometry
__global__ void __launch_bounds__(256) rand_block_kernel(uint16_t* out, int n_calls, long numel, long threads, int loops,
                                                         unsigned long long seed_or_ptr, unsigned long long offset_or_ptr,
                                                         unsigned int intragraph, int captured) {
    // at::cuda::philox::unpack
    unsigned long long seed = seed_or_ptr, offset = offset_or_ptr;
    if (captured) {
        seed = (unsigned long long)*(const long long*)seed_or_ptr;
        offset = (unsigned long long)*(const long long*)offset_or_ptr + intragraph;
    }
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= threads) return;
    const uint2 key{(unsigned)seed, (unsigned)(seed >> 32)};
    for (int cl = blockIdx.y; cl < n_calls * loops; cl += gridDim.y) {          // cl = c * loops + l
        int c = cl, l = 0;
        if (loops > 1) { c = cl / loops; l = cl - c * loops; }
        const unsigned long long ctr = offset / 4 + (unsigned long long)cl;
        const uint4 r = philox4x32_10(uint4{(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)idx, (unsigned)((unsigned long long)idx >> 32)}, key);
        uint16_t* dst = out + (long)c * numel;
        const long li = idx + threads * 4 * l;
        if (li < numel) dst[li] = uniform_f16(r.x);
        if (li + threads < numel) dst[li + threads] = uniform_f16(r.y);
        if (li + 2 * threads < numel) dst[li + 2 * threads] = uniform_f16(r.z);
        if (li + 3 * threads < numel) dst[li + 3 * threads] = uniform_f16(r.w);
    }
}

// The same draws for a search's descents on (B,T) tensors, written only where a descent can read them: descend #c+1 looks at
// rands[b, t] for nodes t <= c (the ones that exist).  torch's geometry is one element per thread with element idx = b*T + t, so
// in the kernel above no wave could skip anything (a wave's 64 elements are one env's slots).  Here a wave takes ONE slot t of 64
// envs -- waves of slots > c never start their Philox rounds, half of a move's -- and the tile goes through LDS so that the
// stores are whole row segments.  Same counters, same conversion: the written elements carry the same bits.
// grid: x over groups of 64 envs, y = call c; 256 threads = 4 waves = 4 slots at a time.
__global__ void __launch_bounds__(256) rand_block_slots_kernel(uint16_t* out, int n_calls, int B, int T, unsigned long long seed_or_ptr,
                                                               unsigned long long offset_or_ptr, unsigned int intragraph, int captured) {
    extern __shared__ uint16_t tile[];              // [64][T + 2]
    unsigned long long seed = seed_or_ptr, offset = offset_or_ptr;
    if (captured) {
        seed = (unsigned long long)*(const long long*)seed_or_ptr;
        offset = (unsigned long long)*(const long long*)offset_or_ptr + intragraph;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, ld = T + 2;
    const int c = blockIdx.y, e = blockIdx.x * 64 + lane;
    const int nslots = c + 1 < T ? c + 1 : T;
    const uint2 key{(unsigned)seed, (unsigned)(seed >> 32)};
    const unsigned long long ctr = offset / 4 + (unsigned long long)c;            // loops == 1: one Philox block per element
    for (int t0 = 0; t0 < nslots; t0 += 4) {
        const int t = t0 + wave;
        if (t < nslots && e < B) {
            const unsigned long long idx = (unsigned long long)e * T + t;
            const uint4 r = philox4x32_10(uint4{(unsigned)ctr, (unsigned)(ctr >> 32), (unsigned)idx, (unsigned)(idx >> 32)}, key);
            tile[lane * ld + t] = uniform_f16(r.x);
        }
    }
    __syncthreads();
    uint16_t* dst = out + (long)c * B * T;
    for (int el = wave; el < 64; el += 4) {
        const int env = blockIdx.x * 64 + el;
        if (env >= B) break;
        for (int t = lane; t < nslots; t += 64) dst[(long)env * T + t] = tile[el * ld + t];
    }
}

}  // namespace bl

extern "C" int bl_rand_block(void* out, int n_calls, long numel, long threads, int loops, unsigned long long seed_or_ptr,
                             unsigned long long offset_or_ptr, unsigned int offset_intragraph, int captured, int only_slots_upto_call,
                             bl_stream_t stream) {
    if (!out || n_calls <= 0 || numel <= 0 || threads <= 0 || threads % 256 != 0 || loops <= 0) return BL_EINVAL;
    if ((long)loops * threads * 4 < numel || (long)(loops - 1) * threads * 4 >= numel) return BL_EINVAL;     // loops = (numel-1)/(4*threads)+1
    if (captured && (!seed_or_ptr || !offset_or_ptr)) return BL_EINVAL;
    if (only_slots_upto_call < 0 || (only_slots_upto_call > 0 && numel % only_slots_upto_call != 0)) return BL_EINVAL;
    if (only_slots_upto_call > 0 && loops == 1 && threads >= numel && only_slots_upto_call <= 1024 && n_calls <= 65535) {
        const int T = only_slots_upto_call, B = (int)(numel / T);
        // 64 x (T + 2) f16 of LDS: above 64 KiB (T >= 511) the kernel's dynamic-LDS limit has to be raised first
        static size_t raised[64] = {};
        if (!bl_raise_lds_limit((const void*)bl::rand_block_slots_kernel, (size_t)64 * (T + 2) * 2, raised)) return BL_ELAUNCH;
        hipLaunchKernelGGL(bl::rand_block_slots_kernel, dim3((unsigned)((B + 63) / 64), (unsigned)n_calls), dim3(256), (size_t)64 * (T + 2) * 2,
                           (hipStream_t)stream, (uint16_t*)out, n_calls, B, T, seed_or_ptr, offset_or_ptr, offset_intragraph, captured);
        return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
    }
    long gy = (long)n_calls * loops;
    if (gy > 65535) gy = 65535;
    hipLaunchKernelGGL(bl::rand_block_kernel, dim3((unsigned)(threads / 256), (unsigned)gy), dim3(256), 0, (hipStream_t)stream, (uint16_t*)out,
                       n_calls, numel, threads, loops, seed_or_ptr, offset_or_ptr, offset_intragraph, captured);
    return hipGetLastError() == hipSuccess ? BL_OK : BL_ELAUNCH;
}
