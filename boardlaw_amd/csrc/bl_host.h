// bl_host.h -- host-side helpers shared by the translation units of libboardlaw_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

// Raises a kernel's dynamic-LDS limit above the 64 KiB default when a launch needs it (gfx950: 160 KiB per CU).  The attribute
// is per device, so what has been raised is remembered per device (an idempotent cache, not state a result depends on).
inline bool bl_raise_lds_limit(const void* kernel, size_t bytes, size_t (&raised)[64]) {
    if (bytes <= 65536) return true;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = 0; raised[0] = 0; }
    if (bytes > raised[dev]) {
        if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
        raised[dev] = bytes;
    }
    return true;
}
