// synth.h — on-device synthetic corpus generator (BASELINE.json: "synthetic 384-d vectors").
// Counter-based so 153.6 GB of corpus never crosses PCIe.  Definition (bit-identical CPU twin in
// oracle/synth.c): Philox4x32-10, key = seed, counter = (row_lo, row_hi, f4, w); the 4 features of
// piece f4 are Irwin-Hall(4) sums of the 16-bit halves of the w=0 and w=1 outputs, centred and
// scaled to unit variance.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcv {

__device__ static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0;
        uint32_t n2 = h0 ^ c3 ^ k1;
        c0 = n0;
        c1 = l1;
        c2 = n2;
        c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

__device__ static inline float4 synth_piece(uint64_t seed, int64_t row, uint32_t f4) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t r0 = (uint32_t)(uint64_t)row, r1 = (uint32_t)((uint64_t)row >> 32);
    uint32_t a[4], b[4];
    philox4x32_10(r0, r1, f4, 0u, k0, k1, a);
    philox4x32_10(r0, r1, f4, 1u, k0, k1, b);
    const float scale = 1.7320508075688772f / 65536.0f;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int32_t s = (int32_t)((a[j] & 0xffffu) + (a[j] >> 16) + (b[j] & 0xffffu) + (b[j] >> 16));
        v[j] = (float)(s - 131070) * scale;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

// Clustered variant (bench.py --clustered, DESIGN.md §synthetic data): row = centroid(cluster(row)) / sqrt(D)
// + noise * u(row) * synth_row(seed, row), cluster(row) = hash(row) mod n_clusters, centroid(j) = synth
// row j of the stream `seed ^ kClusterSeedXor`, u(row) uniform in [0.5, 1.5) from a second hash.  In high
// dimension i.i.d. noise of one amplitude puts every member at the same angle from every other (cosines
// equal to 1e-4); the per-row amplitude spreads the cosines of a cluster over ~noise^2 * D, a continuum
// like the neighbourhood of a real sentence embedding, which a fixed-width screen has to survive.
constexpr uint64_t kClusterSeedXor = 0xC1057E25EED5ull;
__host__ __device__ static inline uint32_t synth_cluster_of(int64_t row, uint32_t n_clusters) {
    return (uint32_t)(((uint64_t)row * 0x9E3779B97F4A7C15ull) >> 33) % n_clusters;
}
__host__ __device__ static inline float synth_amplitude_of(int64_t row) {  // uniform in [0.5, 1.5), 24 bits
    return 0.5f + (float)(uint32_t)((((uint64_t)row * 0xD6E8FEB86659FD93ull) >> 40) & 0xffffffu) * (1.0f / 16777216.0f);
}
// Scaled variant (bench.py's 768-d dot-metric legs): row = a(row) * synth_row(seed, row), a(row) uniform in [lo, hi) from
// the same hash — un-normalised rows whose norms spread, what a dot-product model (MsMarcoBertBaseDotV5) stores.
__host__ __device__ static inline float synth_scale_of(int64_t row, float lo, float span) {
    return fmaf(span, (float)(uint32_t)((((uint64_t)row * 0xD6E8FEB86659FD93ull) >> 40) & 0xffffffu) * (1.0f / 16777216.0f), lo);
}
__device__ static inline float4 synth_piece_scaled(uint64_t seed, int64_t row, uint32_t f4, float lo, float span) {
    const float4 n = synth_piece(seed, row, f4);
    const float a = synth_scale_of(row, lo, span);
    return make_float4(a * n.x, a * n.y, a * n.z, a * n.w);
}
__device__ static inline float4 synth_piece_clustered(uint64_t seed, int64_t row, uint32_t f4, uint32_t n_clusters,
                                                      float noise, float inv_sqrt_d) {
    const float4 c = synth_piece(seed ^ kClusterSeedXor, (int64_t)synth_cluster_of(row, n_clusters), f4);
    const float4 n = synth_piece(seed, row, f4);
    const float a = noise * synth_amplitude_of(row);
    return make_float4(fmaf(a, n.x, c.x * inv_sqrt_d), fmaf(a, n.y, c.y * inv_sqrt_d),
                       fmaf(a, n.z, c.z * inv_sqrt_d), fmaf(a, n.w, c.w * inv_sqrt_d));
}

}  // namespace pcv
