/*
 * synth.c — CPU twin of the on-device synthetic corpus generator (perceive_amd/csrc/synth.h).
 * TEST INFRASTRUCTURE ONLY.  Not part of the reference: BASELINE.json asks for synthetic 384-d
 * vectors; the generator is defined so that CPU and GPU produce bit-identical f32 rows.
 *
 * Definition.  Philox4x32-10 (Salmon et al., SC'11), key = (seed_lo, seed_hi),
 * counter = (row_lo, row_hi, f4, w): for the 4 consecutive features 4*f4 .. 4*f4+3 of row `row`
 * take A = philox(w=0), B = philox(w=1); feature j gets the Irwin-Hall(4) value
 *     S_j = lo16(A_j) + hi16(A_j) + lo16(B_j) + hi16(B_j)            (integer, exact)
 *     v_j = (float)(S_j - 131070) * SCALE,   SCALE = sqrt(3)/65536    (unit variance)
 * Optional normalisation: nx = sum_i (double)v_i*(double)v_i in index order,
 * inv = (float)(1.0/sqrt(nx)), v_i <- v_i * inv.
 */
#include "oracle.h"

#include <math.h>

#define PHILOX_M0 0xD2511F53u
#define PHILOX_M1 0xCD9E8D57u
#define PHILOX_W0 0x9E3779B9u
#define PHILOX_W1 0xBB67AE85u
#define SYNTH_SCALE (1.7320508075688772f / 65536.0f)

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)PHILOX_M0 * c0;
        uint64_t p1 = (uint64_t)PHILOX_M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += PHILOX_W0;
        k1 += PHILOX_W1;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static void synth_piece(uint64_t seed, int64_t row, uint32_t f4, float v[4]) {
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t ctr[4] = {(uint32_t)(uint64_t)row, (uint32_t)((uint64_t)row >> 32), f4, 0u};
    uint32_t a[4], b[4];
    orc_philox4x32_10(ctr, key, a);
    ctr[3] = 1u;
    orc_philox4x32_10(ctr, key, b);
    for (int j = 0; j < 4; ++j) {
        int32_t s = (int32_t)((a[j] & 0xffffu) + (a[j] >> 16) + (b[j] & 0xffffu) + (b[j] >> 16));
        v[j] = (float)(s - 131070) * SYNTH_SCALE;
    }
}

void orc_synth_row(uint64_t seed, int64_t row, int D, int normalize, float* out) {
    for (int f4 = 0; f4 < D / 4; ++f4) synth_piece(seed, row, (uint32_t)f4, out + 4 * f4);
    if (normalize) {
        double nx = 0.0;
        for (int i = 0; i < D; ++i) nx += (double)out[i] * (double)out[i];
        float inv = (float)(1.0 / sqrt(nx));
        for (int i = 0; i < D; ++i) out[i] = out[i] * inv;
    }
}

void orc_synth_rows(uint64_t seed, int64_t first_row, int64_t n, int D, int normalize, float* out) {
    for (int64_t r = 0; r < n; ++r) orc_synth_row(seed, first_row + r, D, normalize, out + (size_t)r * D);
}

/* Clustered variant (twin of synth_piece_clustered in perceive_amd/csrc/synth.h). */
#define CLUSTER_SEED_XOR 0xC1057E25EED5ull
static uint32_t cluster_of(int64_t row, uint32_t n_clusters) {
    return (uint32_t)(((uint64_t)row * 0x9E3779B97F4A7C15ull) >> 33) % n_clusters;
}

static float amplitude_of(int64_t row) { /* uniform in [0.5, 1.5), 24 bits */
    return 0.5f + (float)(uint32_t)((((uint64_t)row * 0xD6E8FEB86659FD93ull) >> 40) & 0xffffffu) * (1.0f / 16777216.0f);
}

void orc_synth_rows_clustered(uint64_t seed, int64_t first_row, int64_t n, int D, int normalize, int n_clusters,
                              float noise, float* out) {
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    for (int64_t r = 0; r < n; ++r) {
        float* o = out + (size_t)r * D;
        const int64_t row = first_row + r;
        const int64_t cl = (int64_t)cluster_of(row, (uint32_t)n_clusters);
        const float a = noise * amplitude_of(row);
        for (int f4 = 0; f4 < D / 4; ++f4) {
            float c[4], v[4];
            synth_piece(seed ^ CLUSTER_SEED_XOR, cl, (uint32_t)f4, c);
            synth_piece(seed, row, (uint32_t)f4, v);
            for (int j = 0; j < 4; ++j) o[4 * f4 + j] = fmaf(a, v[j], c[j] * inv_sqrt_d);
        }
        if (normalize) {
            double nx = 0.0;
            for (int i = 0; i < D; ++i) nx += (double)o[i] * (double)o[i];
            float inv = (float)(1.0 / sqrt(nx));
            for (int i = 0; i < D; ++i) o[i] = o[i] * inv;
        }
    }
}

/* Scaled rows (twin of synth_piece_scaled): row = a(row) * synth_row, a uniform in [lo, hi) from the amplitude hash. */
void orc_synth_rows_scaled(uint64_t seed, int64_t first_row, int64_t n, int D, float amp_lo, float amp_hi, float* out) {
    const float span = amp_hi - amp_lo;
    for (int64_t r = 0; r < n; ++r) {
        float* o = out + (size_t)r * D;
        const int64_t row = first_row + r;
        const float a = fmaf(span, (float)(uint32_t)((((uint64_t)row * 0xD6E8FEB86659FD93ull) >> 40) & 0xffffffu) * (1.0f / 16777216.0f), amp_lo);
        for (int f4 = 0; f4 < D / 4; ++f4) {
            float v[4];
            synth_piece(seed, row, (uint32_t)f4, v);
            for (int j = 0; j < 4; ++j) o[4 * f4 + j] = a * v[j];
        }
    }
}
