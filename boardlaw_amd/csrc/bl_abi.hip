// bl_abi.hip -- the library's identity and the small stand-alone operators of the C ABI (include/boardlaw_amd.h): version, error
// strings, the host-side exp table and q-range decode, the device self-test, powf2, the ReZero elementwise kernels of the torch-GEMM
// plan, the action draws, the batched copy.  Split out of bl_kernels.hip in round 6.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include "../../include/boardlaw_amd.h"
#include "bl_device.h"
#include "bl_dispatch.h"

#pragma clang fp contract(off)

int bl_fold_selftest(int use_fast, hipStream_t stream);      // bl_expand.hip

namespace bl {

// ReZero residual under fp16 autocast, fused (networks.py:17-18): x_out = x + alpha*y with torch's rounding points --
// alpha (an f32 0-dim parameter) is cast to the tensors' dtype f16, the product is rounded to f16, the sum is rounded
// to f16 -- plus relu(x_out) for the next block, 8 halves per thread.
__global__ void __launch_bounds__(256) rezero_relu_kernel(const uint16_t* x, const uint16_t* y, const float* alpha,
                                                         uint16_t* x_out, uint16_t* relu_out, long n8, long n) {
    const float al = h2f(f2h(*alpha));
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const uint4 xv = ((const uint4*)x)[i], yv = ((const uint4*)y)[i];
        const uint32_t xs[4] = {xv.x, xv.y, xv.z, xv.w}, ys[4] = {yv.x, yv.y, yv.z, yv.w};
        uint32_t o[4], r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t ow = 0, rw = 0;
#pragma unroll
            for (int hlf = 0; hlf < 2; hlf++) {
                const uint16_t xb = (uint16_t)(xs[j] >> (16 * hlf)), yb = (uint16_t)(ys[j] >> (16 * hlf));
                const uint16_t ob = f2h(h2f(xb) + h2f(f2h(al * h2f(yb))));
                const uint16_t rb = (ob & 0x8000u) ? (uint16_t)((ob & 0x7fffu) > 0x7c00u ? ob : 0) : ob;   // relu keeps NaN
                ow |= (uint32_t)ob << (16 * hlf); rw |= (uint32_t)rb << (16 * hlf);
            }
            o[j] = ow; r[j] = rw;
        }
        ((uint4*)x_out)[i] = make_uint4(o[0], o[1], o[2], o[3]);
        ((uint4*)relu_out)[i] = make_uint4(r[0], r[1], r[2], r[3]);
    }
    // tail (n not a multiple of 8)
    for (long i = n8 * 8 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const uint16_t ob = f2h(h2f(x[i]) + h2f(f2h(al * h2f(y[i]))));
        x_out[i] = ob;
        relu_out[i] = (ob & 0x8000u) ? (uint16_t)((ob & 0x7fffu) > 0x7c00u ? ob : 0) : ob;
    }
}

// The same ReZero tail in fp32 (the root evaluation runs outside autocast, mcts/__init__.py:72-76): x_out = x + alpha*y
// with the product and the sum rounded separately, as torch's mul and add kernels do, plus relu(x_out).
__global__ void __launch_bounds__(256) rezero_relu_f32_kernel(const float* x, const float* y, const float* alpha,
                                                             float* x_out, float* relu_out, long n) {
    const float al = *alpha;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float o = x[i] + al * y[i];
        x_out[i] = o;
        relu_out[i] = (o < 0.f) ? 0.f : o;          // keeps NaN, like torch's relu
    }
}


// actions ~ Categorical(probs / sum(probs)) by inverse CDF, one uniform per env: the first action whose running total
// (ascending a, f32) reaches u * total, among those with positive probability; the last such action if rounding leaves the
// running total short.  One wave per env.
// Up to BL_COPY_MAX device-to-device copies as one launch: blockIdx.y = the copy (rows x row_bytes, each end with its own
// pitch), its blocks stride over 16-byte words when both ends and pitches allow it, else over 2-byte or 1-byte units.
struct CopyBatch { bl_copy_t it[BL_COPY_MAX]; };
template <typename U>
__device__ __forceinline__ void copy_units(const bl_copy_t& c, size_t tid, size_t nth) {
    const size_t w = c.row_bytes / sizeof(U), total = w * c.rows;
    const uint8_t* src = (const uint8_t*)c.src; uint8_t* dst = (uint8_t*)c.dst;
    if (c.rows == 1) { for (size_t i = tid; i < w; i += nth) ((U*)dst)[i] = ((const U*)src)[i]; return; }
    for (size_t i = tid; i < total; i += nth) {
        const size_t r = i / w, k = i - r * w;
        ((U*)(dst + r * c.dst_pitch))[k] = ((const U*)(src + r * c.src_pitch))[k];
    }
}
__global__ void __launch_bounds__(256) copy_many_kernel(CopyBatch cb) {
    const bl_copy_t& c = cb.it[blockIdx.y];
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const unsigned long long bits = (unsigned long long)(uintptr_t)c.src | (unsigned long long)(uintptr_t)c.dst | c.row_bytes |
                                    (c.rows > 1 ? (c.src_pitch | c.dst_pitch) : 0ull);
    if ((bits & 15) == 0) copy_units<uint4>(c, tid, nth);
    else if ((bits & 1) == 0) copy_units<uint16_t>(c, tid, nth);
    else copy_units<uint8_t>(c, tid, nth);
}

__global__ void __launch_bounds__(BL_WAVE) draw_actions_kernel(const uint16_t* probs, const float* u, long long* actions, int A) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint16_t* p = probs + (long)b * A;
    float carry = 0.f, total = 0.f;
    for (int a0 = 0; a0 < A; a0 += BL_WAVE) total += (a0 + lane < A) ? h2f(p[a0 + lane]) : 0.f;
    for (int m = 32; m > 0; m >>= 1) total += __shfl_xor(total, m, BL_WAVE);
    const float target = u[b] * total;
    int pick = -1, lastpos = -1;
    for (int a0 = 0; a0 < A; a0 += BL_WAVE) {
        const int a = a0 + lane;
        const float v = a < A ? h2f(p[a]) : 0.f;
        float x = v;
        for (int d = 1; d < BL_WAVE; d <<= 1) { const float y = __shfl_up(x, d, BL_WAVE); if (lane >= d) x += y; }
        x += carry;
        const unsigned long long pos = __ballot(v > 0.f), hit = __ballot(v > 0.f && x >= target);
        if (pick < 0 && hit) pick = a0 + __builtin_ctzll(hit);
        if (pos) lastpos = a0 + 63 - __builtin_clzll(pos);
        carry = __shfl(x, BL_WAVE - 1, BL_WAVE);
    }
    if (lane == 0) actions[b] = pick >= 0 ? pick : (lastpos >= 0 ? lastpos : 0);
}

// ------------------------------------------------------------------------------------------------------------------
// MCTSAgent's action draw (mcts/__init__.py:221) as torch computes it, in ONE launch, one wave per env:
//     torch.distributions.Categorical(logits=x).sample(),  x = root logits .float()
//   = argmax(softmax(x - x.logsumexp(-1, keepdim=True)) / q),  q = empty_like(probs).exponential_(1)      [torch.multinomial, one draw]
// operation for operation with the launches it replaces (amax; |m| == inf -> 0; sub; exp; sum -- torch_row_sum; log; add; sub;
// the persistent softmax: lane l holds elements l, l + W, per-lane max and exp-sum in order, XOR butterflies with offsets W/2 .. 1;
// two IEEE divisions; argmax with the lower index on ties).  q is drawn by torch's own exponential_ kernel, so the generator is
// consumed exactly as by the reference's call.  A < 128.  tests/test_rng_stream.py: the actions equal torch's on the same q.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(BL_WAVE) categorical_kernel(const uint16_t* logits, const float* q, long long* actions, int A, int W, int iters, int Wr) {
    __shared__ float tbuf[128];
    const int b = blockIdx.x, lane = threadIdx.x;
    const uint16_t* lrow = logits + (long)b * A;
    const float* qrow = q + (long)b * A;
    float x[2], qq[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int a = lane + it * W;
        const bool in = lane < W && it < iters && a < A;
        x[it] = in ? h2f(lrow[a]) : -INFINITY;
        qq[it] = in ? qrow[a] : 1.f;
    }
    // logsumexp (ReduceOps.cpp: logsumexp_out_impl)
    float m = wave_max_f32((x[0] > x[1]) ? x[0] : x[1]);
    if (fabsf(m) == INFINITY) m = 0.f;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int a = lane + it * W;
        if (lane < W && it < iters && a < A) tbuf[a] = expf(x[it] - m);
    }
    __syncthreads();
    const float own = lane < Wr ? tbuf[lane] : 0.f, second = (lane < Wr && lane + Wr < A) ? tbuf[lane + Wr] : 0.f;
    const float lse = logf(torch_row_sum(own, second, Wr)) + m;
    // softmax(x - lse) (PersistentSoftmax.cuh: softmax_warp_forward, is_log_softmax = false)
    float y[2], e[2];
    y[0] = x[0] - lse; y[1] = x[1] - lse;
    float mx = y[0];
    if (iters > 1) mx = (mx > y[1]) ? mx : y[1];
    for (int off = W / 2; off > 0; off /= 2) { const float o = __shfl_xor(mx, off, BL_WAVE); mx = (mx < o) ? o : mx; }
    float sum = 0.f;
    e[0] = expf(y[0] - mx); sum += e[0];
    if (iters > 1) { e[1] = expf(y[1] - mx); sum += e[1]; } else e[1] = 0.f;
    for (int off = W / 2; off > 0; off /= 2) sum = sum + __shfl_xor(sum, off, BL_WAVE);
    // argmax(probs / q), lower index on ties (ArgMaxOps)
    float best = -INFINITY; int besta = 0x7fffffff;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int a = lane + it * W;
        if (lane < W && it < iters && a < A) {
            const float r = (e[it] / sum) / qq[it];
            if (r > best || (r == best && a < besta) || r != r) { best = r; besta = a; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, BL_WAVE); const int oa = __shfl_xor(besta, off, BL_WAVE);
        if (ob > best || (ob == best && oa < besta)) { best = ob; besta = oa; }
    }
    if (lane == 0) actions[b] = besta == 0x7fffffff ? 0 : besta;
}

__global__ void __launch_bounds__(256) powf2_kernel(const float* x, float* out, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = g_denominator(x[i], 1);
}

}  // namespace bl

using namespace bl;

extern "C" {

int bl_abi_version(void) { return 4; }

const char* bl_strerror(int code) {
    switch (code) {
        case BL_OK: return "ok";
        case BL_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case BL_ETOOBIG: return "size beyond kernel limits (A <= 1024, T <= 32767, S <= 8, boardsize <= 32)";
        case BL_ELAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}

int bl_exp_table_host(float* t) {
    if (!t) return BL_EINVAL;
    for (uint32_t i = 0; i < 65536; i++) {
        uint16_t h = (uint16_t)i;
        // binary16 -> binary32 (exact)
        uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1f, mnt = h & 0x3ffu, bits;
        if (e == 0) {
            if (mnt == 0) bits = sign;
            else { int sh = -1; do { sh++; mnt <<= 1; } while (!(mnt & 0x400u)); bits = sign | ((uint32_t)(112 - sh) << 23) | ((mnt & 0x3ffu) << 13); }
        } else if (e == 31) bits = sign | 0x7f800000u | (mnt << 13);
        else bits = sign | ((e + 112) << 23) | (mnt << 13);
        float x; memcpy(&x, &bits, 4);
        t[i] = expf(x);
    }
    return BL_OK;
}

int bl_qrange_decode(const uint32_t* st, float* mm) {
    if (!st || !mm) return BL_EINVAL;
    uint32_t a = 0, b = 0;
    for (int i = 0; i < BL_QSLOTS; i++) {
        const uint32_t x = st[BL_QSTRIDE * i] ^ BL_QBIAS, y = st[BL_QSTRIDE * i + 1] ^ BL_QBIAS;       // memory words -> unsigned codes
        if (x > a) a = x;
        if (y > b) b = y;
    }
    mm[0] = dec(~a); mm[1] = dec(b);
    return BL_OK;
}

int bl_powf2(const float* x, float* out, long n, bl_stream_t stream) {
    if (!x || !out || n <= 0) return BL_EINVAL;
    hipLaunchKernelGGL(powf2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out, n);
    return check_launch();
}

int bl_selftest(bl_stream_t stream) {
    const int wrong_safe = bl_fold_selftest(0, (hipStream_t)stream);
    if (wrong_safe != 0) return wrong_safe < 0 ? wrong_safe : BL_ELAUNCH;      // the ISA-compliant fold must be exact
    const int wrong_fast = bl_fold_selftest(1, (hipStream_t)stream);
    if (wrong_fast < 0) return wrong_fast;
    return wrong_fast;
}

int bl_rezero_relu_f16(const void* x, const void* y, const float* alpha, void* x_out, void* relu_out, long n,
                       bl_stream_t stream) {
    if (!x || !y || !alpha || !x_out || !relu_out || n <= 0) return BL_EINVAL;
    const long n8 = n / 8;
    long blocks = (n8 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(rezero_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                       (const uint16_t*)y, alpha, (uint16_t*)x_out, (uint16_t*)relu_out, n8, n);
    return check_launch();
}

int bl_rezero_relu_f32(const float* x, const float* y, const float* alpha, float* x_out, float* relu_out, long n,
                       bl_stream_t stream) {
    if (!x || !y || !alpha || !x_out || !relu_out || n <= 0) return BL_EINVAL;
    long blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rezero_relu_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, alpha, x_out, relu_out, n);
    return check_launch();
}

int bl_draw_actions(const void* probs, const float* uniforms, long long* actions, int B, int A, bl_stream_t stream) {
    if (!probs || !uniforms || !actions || B <= 0 || A <= 0) return BL_EINVAL;
    hipLaunchKernelGGL(draw_actions_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)probs, uniforms, actions, A);
    return check_launch();
}

int bl_categorical(const void* logits, const float* q, long long* actions, int B, int A, bl_stream_t stream) {
    if (!logits || !q || !actions || B <= 0 || A <= 0) return BL_EINVAL;
    if (A >= 128) return BL_ETOOBIG;           // torch's sum takes its vectorised path there (row-alignment-dependent order): the caller keeps torch's launches
    int np2 = 1; while (np2 < A) np2 *= 2;
    const int W = np2 < 64 ? np2 : 64, iters = np2 / W;
    const int Wr = last_pow2_le(A) < 64 ? last_pow2_le(A) : 64;
    hipLaunchKernelGGL(categorical_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)logits, q, actions, A, W, iters, Wr);
    return check_launch();
}

int bl_copy_many(const bl_copy_t* items, int n, bl_stream_t stream) {
    if (n < 0 || n > BL_COPY_MAX || (n > 0 && !items)) return BL_EINVAL;
    CopyBatch c{};
    unsigned long long most = 0;
    for (int k = 0; k < n; k++) {
        const unsigned long long bytes = items[k].row_bytes * items[k].rows;
        if (bytes && (!items[k].src || !items[k].dst)) return BL_EINVAL;
        if (items[k].rows > 1 && (items[k].src_pitch < items[k].row_bytes || items[k].dst_pitch < items[k].row_bytes)) return BL_EINVAL;
        c.it[k] = items[k];
        if (bytes > most) most = bytes;
    }
    if (most == 0) return BL_OK;
    unsigned long long blocks = (most / 16 + 255) / 256;           // one 16-byte word per thread of the largest copy ...
    if (blocks < 1) blocks = 1;
    if (blocks > 128) blocks = 128;                                 // ... up to 128 blocks per copy, then strided
    hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)blocks, n), dim3(256), 0, (hipStream_t)stream, c);
    return check_launch();
}


}  // extern "C"
