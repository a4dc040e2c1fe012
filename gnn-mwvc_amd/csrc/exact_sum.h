// exact_sum.h — the sequential fp32 sum of the reference, evaluated in parallel, bit for bit.
//
// graph_layer::forward adds a row's neighbours one after the other into one fp32 accumulator
// (reference src/gnn_inference.cpp:33-36): acc = RN(acc + v_i), i in stored order.  A hub row of
// several hundred thousand neighbours is then one dependent add chain, milliseconds long whatever the
// hardware.  The chain can be cut without changing a bit:
//
//   While acc stays inside one binade, acc = m * u with u its unit in the last place and m an
//   integer in [2^23, 2^24) (or [0, 2^24) in the lowest binade, which also holds zero and the
//   denormals).  For v >= 0 write v = k * u + r, 0 <= r < u.  Round-to-nearest-even gives
//       m' = m + k          if r <  u/2
//       m' = m + k + 1      if r >  u/2
//       m' = m + k + ((m + k) & 1)   if r == u/2   (tie: to even)
//   so the add is the map  m -> m + d[m & 1]  with two small integers d[0], d[1]: it depends on acc
//   only through the PARITY of m.  Such maps compose associatively,
//       (A then B)[p] = A[p] + B[(p + A[p]) & 1],
//   hence a segment of the row is summarised by one pair of integers, segments combine with a parallel
//   scan, and the accumulator after the segment is m + D[m & 1] — exactly what the chain yields — as
//   long as m + D < 2^24 (no carry into the next binade; acc only grows, so checking the ends of the
//   pieces suffices).  Where a piece would carry, or holds a negative / infinite / NaN value, that piece
//   is added the plain way in fp32 from the exact accumulator in front of it, and the scan resumes with
//   the new binade.  Nothing is assumed about the data: any input takes one of the two routes and both
//   produce the chain's bits.
//
// The functions below are the scalar building blocks, usable on the host (tests/support) and on
// the device (k_giant_sum in gnnvc_kernels.hip).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define XS_HD __host__ __device__ __forceinline__
#else
#define XS_HD inline
#endif

namespace xsum {

constexpr uint32_t kSat = 1u << 26;      // increments saturate here: anything >= 2^24 already means "carried"
constexpr uint32_t kCarry = 1u << 24;
XS_HD uint32_t sat_(uint32_t x) { return x < kSat ? x : kSat; }

struct Map {
    uint32_t d0, d1;   // increment of m when the piece starts at even / odd m
};

// acc = m * 2^(E - 150), E = max(biased exponent, 1).  False: acc is negative, infinite or NaN and
// the integer route does not apply.
XS_HD bool decode_acc(uint32_t bits, uint32_t &E, uint32_t &m) {
    const uint32_t e = (bits >> 23) & 0xFFu;
    if ((bits >> 31) || e == 255u) return false;
    E = e ? e : 1u;
    m = (bits & 0x7FFFFFu) | (e ? 0x800000u : 0u);
    return true;
}

// the float whose significand is m in binade E (m < 2^24, and m >= 2^23 unless E == 1)
XS_HD uint32_t encode_acc(uint32_t E, uint32_t m) { return E == 1u ? m : ((E << 23) | (m & 0x7FFFFFu)); }

// One addend as a parity map relative to binade E.  bad: the value is negative (not -0.0), infinite
// or NaN — its piece must be added in fp32.  A value too large for the binade saturates the map
// (its add carries for certain).
XS_HD Map elem_map(uint32_t vbits, uint32_t E, bool &bad) {
    Map r = {0u, 0u};
    const uint32_t e = (vbits >> 23) & 0xFFu;
    if (vbits == 0x80000000u) return r;                  // -0.0f: acc + (-0) == acc (acc is never -0)
    if ((vbits >> 31) || e == 255u) { bad = true; return r; }
    const uint32_t Ev = e ? e : 1u;
    const uint32_t mv = (vbits & 0x7FFFFFu) | (e ? 0x800000u : 0u);
    if (mv == 0u) return r;
    if (Ev > E) { r.d0 = r.d1 = kSat; return r; }         // v alone exceeds acc's binade
    const uint32_t s = E - Ev;
    if (s == 0u) { r.d0 = r.d1 = mv; return r; }          // a multiple of u: exact
    if (s >= 25u) return r;                               // below u/2: acc unchanged
    const uint32_t k = mv >> s, rem = mv & ((1u << s) - 1u), half = 1u << (s - 1u);
    const uint32_t up = rem > half ? 1u : 0u, tie = rem == half ? 1u : 0u;
    r.d0 = k + up + (tie & k);                            // start even: m + k odd  <=>  k odd
    r.d1 = k + up + (tie & (k + 1u));                     // start odd
    return r;
}

// The same map from floating-point operations (what the kernel issues: v_ldexp, v_floor, v_cvt and two
// compares instead of a dozen integer instructions).  t = v / u is exact — a scaling by a power of two
// — whenever it matters (t >= 2^-126; anything smaller is far below u/2), floor(t) = k and
// t - floor(t) = r / u are exact for t < 2^24, and a t beyond that saturates: its add carries.
XS_HD Map elem_map_f(uint32_t vbits, uint32_t E, bool &bad) {
    // branch-free: a negative or non-finite value only raises `bad` (its map is then never used)
    bad |= (vbits > 0x80000000u) | ((vbits & 0x7F800000u) == 0x7F800000u);
    float v;
    __builtin_memcpy(&v, &vbits, 4);
    const float t = __builtin_ldexpf(v, 150 - (int)E);
    const float fl = __builtin_floorf(t);
    const uint32_t k = (uint32_t)__builtin_fminf(__builtin_fmaxf(fl, 0.0f), 67108864.0f);   // (NaN -> 0, huge -> kSat)
    const float fr = t - fl;
    const uint32_t up = fr > 0.5f ? 1u : 0u, tie = fr == 0.5f ? 1u : 0u;
    Map r;
    r.d0 = k + up + (tie & k);             // (<= kSat + 2: the caller saturates after a handful of appends)
    r.d1 = k + up + (tie & (k + 1u));
    return r;
}

XS_HD uint32_t sat(uint32_t x) { return x < kSat ? x : kSat; }

// A then B
XS_HD Map compose(const Map &a, const Map &b) {
    Map c;
    c.d0 = sat(a.d0 + ((a.d0 & 1u) ? b.d1 : b.d0));
    c.d1 = sat(a.d1 + ((a.d1 & 1u) ? b.d0 : b.d1));
    return c;
}

// Append one addend to a running map (compose(run, elem_map(v)) without the saturation).  An addend's increments are
// at most kSat + 2, so up to 31 appends fit 32 bits: the caller appends a lane's handful of addends, then calls
// saturate() once (the parity bits that steer the selects stay exact as long as nothing wraps).
constexpr int kMaxAppends = 31;
template <bool FLOAT_DECODE = false>
XS_HD void append(Map &run, uint32_t vbits, uint32_t E, bool &bad) {
    const Map e = FLOAT_DECODE ? elem_map_f(vbits, E, bad) : elem_map(vbits, E, bad);
    run.d0 = run.d0 + ((run.d0 & 1u) ? e.d1 : e.d0);
    run.d1 = run.d1 + ((run.d1 & 1u) ? e.d0 : e.d1);
}
XS_HD void saturate(Map &run) {
    run.d0 = sat(run.d0);
    run.d1 = sat(run.d1);
}

}  // namespace xsum
