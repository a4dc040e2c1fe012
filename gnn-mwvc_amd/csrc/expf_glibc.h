// expf_glibc.h — single-precision exp evaluated the way glibc (>= 2.27) does on
// x86-64 hosts with FMA, so that the device-side sigmoid reproduces the
// reference's `1.0f / (1.0f + expf(-x))` (reference src/gnn_inference.cpp:51)
// bit for bit instead of "within an ulp".
//
// Published algorithm (Szabolcs Nagy's expf from ARM optimized-routines, adopted
// by glibc 2.27): exp(x) = 2^(k/32) * 2^(r/32) with k = round(x * 32/ln2) taken
// by the add-a-big-constant trick, a 32-entry table of 2^(i/32) and a cubic in
// r, all in double precision, one final rounding to float.  The operation order
// below — which multiply-adds are fused — is the one the FMA build of glibc
// executes (the `__expf_fma` ifunc variant picked on every AVX2+FMA CPU).
//
// tests/test_expf_restatement.py compiles this header for the host and compares
// it with the host libm over tens of millions of inputs; the GPU tests compare
// device scores with the oracle's host-libm scores.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GNNVC_HD __host__ __device__ __forceinline__
#else
#define GNNVC_HD static inline
#endif

namespace gnnvc {

// bits(2^(i/32)) - (i << 47), i = 0..31 (correctly rounded doubles)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__
#else
static
#endif
const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

GNNVC_HD uint64_t f64_bits(double d) {
    uint64_t u;
    memcpy(&u, &d, sizeof u);
    return u;
}
GNNVC_HD double bits_f64(uint64_t u) {
    double d;
    memcpy(&d, &u, sizeof d);
    return d;
}
GNNVC_HD uint32_t f32_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}

GNNVC_HD float expf_glibc(float x) {
    const double kShift = 0x1.8p+52;                  // 1.5 * 2^52
    const double kInvLn2N = 0x1.71547652b82fep+5;     // 32 / ln 2
    const double kC0 = 0x1.c6af84b912394p-20;         // poly / 32^3
    const double kC1 = 0x1.ebfce50fac4f3p-13;         // poly / 32^2
    const double kC2 = 0x1.62e42ff0c52d6p-6;          // poly / 32
    const uint32_t ix = f32_bits(x);
    const uint32_t abstop = (ix >> 20) & 0x7ff;
    if (abstop > 0x42a) {                             // |x| >= 88 or NaN/Inf
        if (ix == 0xff800000u) return 0.0f;           // exp(-inf)
        if (abstop >= 0x7f8) return x + x;            // NaN, +inf
        if (x > 0x1.62e42ep6f) return __builtin_inff();   // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;              // underflow
        // -103.97 <= x <= -88 and 88 <= x <= 88.72: the main path is still exact enough
    }
    const double xd = (double)x;
    double kd = __builtin_fma(kInvLn2N, xd, kShift);
    const uint64_t ki = f64_bits(kd);
    kd -= kShift;
    const double r = __builtin_fma(kInvLn2N, xd, -kd);
    const uint64_t t = kExp2fTab[ki & 31] + (ki << 47);
    const double s = bits_f64(t);
    const double z = __builtin_fma(kC0, r, kC1);
    const double r2 = r * r;
    double y = __builtin_fma(kC2, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

}  // namespace gnnvc
