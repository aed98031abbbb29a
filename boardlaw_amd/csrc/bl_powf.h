// bl_powf.h -- powf(x, 2.0f) as glibc 2.35 computes it on an x86-64 host with FMA (the reference's own JIT build passes no -O flag,
// boardlaw/cuda.py:29-45, so its CPU path calls libm's powf at boardlaw/mcts/cpp/cpu.cpp:60 instead of folding it to bot*bot).
//
// The algorithm lives in a third-party dependency that is not in /root/reference: glibc 2.35 (Ubuntu 2.35-0ubuntu3.11),
// sysdeps/ieee754/flt-32/e_powf.c with its tables powf_log2_data.c / exp2f_data.c (Szabolcs Nagy's ARM optimized-routines powf),
// built as the multiarch variant __powf_fma (sysdeps/x86_64/fpu/multiarch/e_powf-fma.c: -mfma -mavx2), which the ifunc resolver
// picks on every CPU with FMA.  Restated here operation by operation FROM THE INSTRUCTIONS OF THAT BUILD (which products are
// contracted into fused multiply-adds decides the last bit): log2(x) in double from a 16-entry table and a degree-4 polynomial,
// times y, exp2 in double from a 32-entry table and a degree-3 polynomial, one rounding to float at the end.  Its result differs
// from the correctly rounded x*x on 0.036 % of all floats.  tests/test_powf.py pins this restatement to the host libm's powf(x, 2)
// on every tested float (and oracle/liboracle_powf.so, which calls libm itself, stays the checker of the search results).
// Constants: read from that libm's tables (they are the published ones of powf_log2_data.c / exp2f_data.c for
// POWF_LOG2_TABLE_BITS = 4, EXP2F_TABLE_BITS = 5, no TOINT intrinsics).
#pragma once
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#define BLP_FN __host__ __device__ __forceinline__
#else
#define BLP_FN static inline
#endif

#define BLP_LOG2_WORDS 32       // {invc, logc} x 16
#define BLP_EXP2_WORDS 32
// the tables as plain arrays for whoever uploads them (device: constant memory or LDS)
static const double BLP_LOG2_TAB[BLP_LOG2_WORDS] = {
    0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2,
    0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
    0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2,
    0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
    0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2,
    0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3,
    0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3,
    0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
    0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5,
    0x1.0000000000000p+0, 0x0.0p+0,
    0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4,
    0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3,
    0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3,
    0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
    0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2,
    0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2,
};
static const uint64_t BLP_EXP2_TAB[BLP_EXP2_WORDS] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull,
};

BLP_FN double blp_asdouble(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }
BLP_FN uint64_t blp_asuint64(double d) { uint64_t u; memcpy(&u, &d, 8); return u; }
BLP_FN float blp_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
BLP_FN uint32_t blp_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

// powf(x, 2.0f).  log2tab / exp2tab: BLP_LOG2_TAB / BLP_EXP2_TAB wherever the caller keeps them.
BLP_FN float bl_powf2_glibc(float x, const double* log2tab, const uint64_t* exp2tab) {
    uint32_t ix = blp_asuint(x);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {           // e_powf.c:150: x < 0x1p-126, inf or nan (y = 2 is none of these)
        if (2u * ix - 1u >= 2u * 0x7f800000u - 1u) return x * x;   // :162 zeroinfnan(ix): x2 = x * x, y is an even integer and positive
        ix &= 0x7fffffffu;                                         // :174 finite x < 0: checkint(y) == 2, the sign is dropped
        if (ix < 0x00800000u) {                                    // :184 subnormal x: normalise
            ix = blp_asuint(blp_asfloat(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    // log2_inline, e_powf.c:48-78 (the products the FMA build contracts are written as fma)
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = log2tab[2 * i], logc = log2tab[2 * i + 1];
    const double z = (double)blp_asfloat(iz);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double ya = __builtin_fma(r, 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(r, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1);
    const double r2 = r * r;
    double q = __builtin_fma(r, 0x1.71547652ab82bp+0, y0);
    const double r4 = r2 * r2;
    q = __builtin_fma(r2, p, q);
    const double logx = __builtin_fma(ya, r4, q);
    const double ylogx = 2.0 * logx;                               // :190 y * logx
    if (((blp_asuint64(ylogx) >> 47) & 0xffffu) >= (0x405f800000000000ull >> 47)) {      // :191 |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return blp_asfloat(0x7f800000u);  // :194 overflow (__math_oflowf(0))
        if (ylogx <= -150.0) return 0.0f;                          // :196 underflow (__math_uflowf(0))
    }
    // exp2_inline, e_powf.c:95-118
    const double kd0 = ylogx + 0x1.8000000000000p+47;
    const uint64_t ki = blp_asuint64(kd0);
    const double kd = kd0 - 0x1.8000000000000p+47;
    const double rr = ylogx - kd;
    const uint64_t t = exp2tab[ki & 31u] + (ki << 47);
    const double s = blp_asdouble(t);
    const double zz = __builtin_fma(rr, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yy = __builtin_fma(rr, 0x1.62e42ff0c52d6p-1, 1.0);
    yy = __builtin_fma(zz, rr2, yy);
    yy = yy * s;
    return (float)yy;
}
