// bl_torchgen.cpp -- libbl_torchgen.so: the one place where this package touches a torch C++ type.
//
// The reference draws its descend uniforms with at::rand_like from torch's generator (boardlaw/mcts/cpp/cuda.cu:191), T-1 times
// per move.  bl_rand_block (libboardlaw_amd.so, torch-free) reproduces those T-1 draws with one launch; to stay on the
// generator's Philox stream -- eagerly and inside HIP-graph capture -- it needs what every torch random kernel gets from
// `gen->philox_cuda_state(increment)`: the seed and offset (values, or during capture the device pointers the graph's replay
// prologue refills) and the offset consumed so far inside the capture.  This shim makes exactly that call and hands the numbers
// back as plain integers; it also advances the generator by `increment`, like the torch kernels it stands in for.
//
// Plumbing only (host code, no kernels); built by boardlaw_amd/build.py with the host compiler against torch's headers.
#include <ATen/hip/HIPGeneratorImpl.h>
#include <cstdint>
#include <mutex>

extern "C" {

// generator_impl: torch.Generator._cdata (the at::GeneratorImpl* of a CUDA/HIP generator).
// out[0] = seed (value, or device pointer to an int64 when captured), out[1] = offset (value or device pointer),
// out[2] = offset_intragraph, out[3] = 1 when captured (out[0], out[1] are pointers) else 0.  Returns 0, or -1 on a torch exception.
int bl_torch_philox_state(void* generator_impl, uint64_t increment, int64_t out[4]) {
    try {
        auto* gen = static_cast<at::CUDAGeneratorImpl*>(static_cast<c10::GeneratorImpl*>(generator_impl));
        at::PhiloxCudaState st;
        {
            std::lock_guard<std::mutex> lock(gen->mutex_);      // "Acquire lock when using random generators"
            st = gen->philox_cuda_state(increment);
        }
        if (st.captured_) {
            out[0] = (int64_t)(intptr_t)st.seed_.ptr; out[1] = (int64_t)(intptr_t)st.offset_.ptr;
            out[2] = (int64_t)st.offset_intragraph_; out[3] = 1;
        } else {
            out[0] = (int64_t)st.seed_.val; out[1] = (int64_t)st.offset_.val; out[2] = 0; out[3] = 0;
        }
        return 0;
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
