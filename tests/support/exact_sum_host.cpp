// Host emulation of k_giant_sum (gnn-mwvc_amd/csrc/gnnvc_kernels.hip): the same window / lane / scan
// control flow on the scalar building blocks of csrc/exact_sum.h, with the 64 lanes of a wave as a
// loop.  Test infrastructure: lets the CPU suite check the parity-map algorithm against the plain
// sequential fp32 chain (reference src/gnn_inference.cpp:33-36) on adversarial streams.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "../../gnn-mwvc_amd/csrc/exact_sum.h"

namespace {

constexpr int kLanes = 64;

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float u2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct Stats { uint64_t steps, carries, plain_windows; };

// the window walk over elements [pos, n) of v, from accumulator `acc` on (pos a multiple of the window)
template <bool FD>
float stream_sum(const float *v, size_t n, int B, Stats *st, float acc = 0.0f, size_t pos = 0) {
    const size_t W = (size_t)kLanes * B;
    while (pos < n) {
        const size_t base = pos / W * W;
        if (st) ++st->steps;
        uint32_t E = 0, m = 0;
        const bool ok = xsum::decode_acc(f2u(acc), E, m);
        xsum::Map lane_map[kLanes];
        bool lane_bad[kLanes];
        int nbad = 0;
        for (int l = 0; l < kLanes; ++l) {
            xsum::Map run = {0u, 0u};
            bool bad = false;
            for (int i = 0; i < B; ++i) {
                const size_t idx = base + (size_t)l * B + i;
                if (idx < pos || idx >= n) continue;   // consumed already / past the end: no-op
                if (ok) xsum::append<FD>(run, f2u(v[idx]), E, bad);
            }
            xsum::saturate(run);
            lane_map[l] = run;
            lane_bad[l] = bad;
            nbad += bad;
        }
        if (!ok || nbad > 4) {   // the integer route does not apply: this window the plain way
            if (st) ++st->plain_windows;
            for (size_t idx = pos; idx < base + W && idx < n; ++idx) acc = acc + v[idx];
            pos = base + W;
            continue;
        }
        // inclusive scan over the lanes (Hillis-Steele, as the wave does it)
        for (int d = 1; d < kLanes; d <<= 1) {
            xsum::Map next[kLanes];
            for (int l = 0; l < kLanes; ++l) next[l] = l >= d ? xsum::compose(lane_map[l - d], lane_map[l]) : lane_map[l];
            for (int l = 0; l < kLanes; ++l) lane_map[l] = next[l];
        }
        const uint32_t p = m & 1u;
        int cross = -1;
        for (int l = 0; l < kLanes; ++l) {
            const uint32_t D = p ? lane_map[l].d1 : lane_map[l].d0;
            if (lane_bad[l] || m + D >= xsum::kCarry) { cross = l; break; }
        }
        if (cross < 0) {
            const uint32_t D = p ? lane_map[kLanes - 1].d1 : lane_map[kLanes - 1].d0;
            acc = u2f(xsum::encode_acc(E, m + D));
            pos = base + W;
            continue;
        }
        if (st) ++st->carries;
        const uint32_t Dex = cross ? (p ? lane_map[cross - 1].d1 : lane_map[cross - 1].d0) : 0u;
        acc = u2f(xsum::encode_acc(E, m + Dex));
        for (int i = 0; i < B; ++i) {
            const size_t idx = base + (size_t)cross * B + i;
            if (idx < pos || idx >= n) continue;
            acc = acc + v[idx];
        }
        pos = base + (size_t)(cross + 1) * B;
    }
    return acc;
}

// One SEGMENT of a stream (k_giant_segmap): the composed parity map of elements [a, b) relative to binade E — window by
// window, lanes folded and scanned as in the window walk, window totals composed in order.  bad: a negative / non-finite value.
template <bool FD>
xsum::Map segment_map(const float *v, size_t a, size_t b, int B, uint32_t E, bool &bad) {
    const size_t W = (size_t)kLanes * B;
    xsum::Map total = {0u, 0u};
    for (size_t base = a; base < b; base += W) {
        xsum::Map lane_map[kLanes];
        for (int l = 0; l < kLanes; ++l) {
            xsum::Map run = {0u, 0u};
            for (int i = 0; i < B; ++i) {
                const size_t idx = base + (size_t)l * B + i;
                if (idx < b) xsum::append<FD>(run, f2u(v[idx]), E, bad);
            }
            xsum::saturate(run);
            lane_map[l] = run;
        }
        for (int d = 1; d < kLanes; d <<= 1) {
            xsum::Map next[kLanes];
            for (int l = 0; l < kLanes; ++l) next[l] = l >= d ? xsum::compose(lane_map[l - d], lane_map[l]) : lane_map[l];
            for (int l = 0; l < kLanes; ++l) lane_map[l] = next[l];
        }
        total = xsum::compose(total, lane_map[kLanes - 1]);
    }
    return total;
}

// The segmented evaluation (k_giant_segsum / k_giant_segmap / k_giant_sum): every segment but the first gets a parity map
// under a GUESSED binade; the final walk uses a segment's map when the exact accumulator in front of it sits in that binade
// and the map does not carry, and walks the segment's windows otherwise.  guess: 0 = from a pairwise float sum of everything
// in front of the segment (what the kernels do), 1 = that guess + 1, 2 = - 1, 3 = a fixed wrong binade, 4 = pseudo-random.
template <bool FD>
float stream_sum_segmented(const float *v, size_t n, int B, int seg_windows, int guess, Stats *st) {
    const size_t W = (size_t)kLanes * B, S = W * (size_t)seg_windows;
    const size_t nseg = (n + S - 1) / S;
    float acc = 0.0f;
    double approx = 0.0;   // (any estimate will do: only speed depends on it)
    uint64_t rnd = 0x9E3779B97F4A7C15ull;
    for (size_t k = 0; k < nseg; ++k) {
        const size_t a = k * S, b = a + S < n ? a + S : n;
        bool used = false;
        if (k > 0) {
            uint32_t E = 0, m = 0, Eg = 1;
            const float pf = (float)approx;
            const uint32_t pe = (f2u(pf) >> 23) & 0xFFu;
            Eg = pe ? pe : 1u;
            if (guess == 1) Eg += 1;
            if (guess == 2) Eg = Eg > 1 ? Eg - 1 : 1;
            if (guess == 3) Eg = 100;
            if (guess == 4) { rnd = rnd * 6364136223846793005ull + 1442695040888963407ull; Eg = 1 + (uint32_t)((rnd >> 33) % 254); }
            bool bad = !(pf >= 0.0f) || pe == 255u;
            const xsum::Map mp = segment_map<FD>(v, a, b, B, Eg, bad);
            if (xsum::decode_acc(f2u(acc), E, m) && !bad && E == Eg) {
                const uint32_t D = (m & 1u) ? mp.d1 : mp.d0;
                if (m + D < xsum::kCarry) {
                    acc = u2f(xsum::encode_acc(E, m + D));
                    used = true;
                    if (st) ++st->carries;   // (counts the segments taken in one step)
                }
            }
        }
        if (!used) acc = stream_sum<FD>(v, b, B, st, acc, a);
        for (size_t i = a; i < b; ++i) approx += (double)v[i] == (double)v[i] ? (double)v[i] : 0.0;
    }
    return acc;
}

}  // namespace

extern "C" {

// segmented evaluation; stats3[1] = segments taken in one step
float xsum_stream_segmented(const float *v, size_t n, int B, int seg_windows, int guess, uint64_t *stats3) {
    Stats st = {0, 0, 0};
    const float r = B < 0 ? stream_sum_segmented<true>(v, n, -B, seg_windows, guess, &st)
                          : stream_sum_segmented<false>(v, n, B, seg_windows, guess, &st);
    if (stats3) { stats3[0] = st.steps; stats3[1] = st.carries; stats3[2] = st.plain_windows; }
    return r;
}

float xsum_stream(const float *v, size_t n, int B, uint64_t *stats3) {
    Stats st = {0, 0, 0};
    // B < 0: the floating-point decode of the addends (what the kernel issues)
    const float r = B < 0 ? stream_sum<true>(v, n, -B, &st) : stream_sum<false>(v, n, B, &st);
    if (stats3) { stats3[0] = st.steps; stats3[1] = st.carries; stats3[2] = st.plain_windows; }
    return r;
}

// the reference's chain: acc = acc + v[i], one rounded fp32 add per element
float xsum_sequential(const float *v, size_t n) {
    volatile float acc = 0.0f;
    for (size_t i = 0; i < n; ++i) acc = acc + v[i];
    return acc;
}

}  // extern "C"
