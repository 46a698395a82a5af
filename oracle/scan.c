/*
 * scan.c — CPU restatement of the brute-force similarity + top-k path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h: parity unpinned by the reference; cross-checked against PyTorch-CPU fixtures).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* lib.rs:63-65 — `set1.matmul(&set2.transpose(0, 1))`. */
void orc_dot_product(const float* a, int B, const float* m, int64_t N, int D, float* out) {
    for (int b = 0; b < B; ++b) {
        const float* q = a + (size_t)b * D;
        for (int64_t n = 0; n < N; ++n) {
            const float* x = m + (size_t)n * D;
            float acc = 0.0f;
            for (int i = 0; i < D; ++i) acc += q[i] * x[i];
            out[(size_t)b * N + n] = acc;
        }
    }
}

/* `x / x.linalg_norm(2.0, [dim], true, Kind::Float)` of lib.rs:68-69,74-75, one row. */
static void normalise_row_f32(const float* x, int D, float* out) {
    float ss = 0.0f;
    for (int i = 0; i < D; ++i) ss += x[i] * x[i];
    float nrm = sqrtf(ss);
    for (int i = 0; i < D; ++i) out[i] = x[i] / nrm; /* no epsilon, like the reference */
}

/* lib.rs:73-77 */
void orc_cosine_similarity_multi_query(const float* a, int B, const float* m, int64_t N, int D, float* out) {
    float* an = (float*)malloc((size_t)B * D * sizeof(float));
    float* xn = (float*)malloc((size_t)D * sizeof(float));
    for (int b = 0; b < B; ++b) normalise_row_f32(a + (size_t)b * D, D, an + (size_t)b * D);
    for (int64_t n = 0; n < N; ++n) {
        normalise_row_f32(m + (size_t)n * D, D, xn);
        for (int b = 0; b < B; ++b) {
            const float* q = an + (size_t)b * D;
            float acc = 0.0f;
            for (int i = 0; i < D; ++i) acc += q[i] * xn[i];
            out[(size_t)b * N + n] = acc;
        }
    }
    free(an);
    free(xn);
}

/* lib.rs:67-71 — the query is a [D] vector normalised over dim 0; the result is [N]. */
void orc_cosine_similarity_single_query(const float* q, const float* m, int64_t N, int D, float* out) {
    orc_cosine_similarity_multi_query(q, 1, m, N, D, out);
}

/* Canonical score: the same quantity as lib.rs:67-77 evaluated without f32 rounding.
 * f32*f32 is exact in f64 (24+24 significand bits), sums run in index order. */
double orc_canonical_score(const float* q, const float* x, int D, int metric) {
    double dot = 0.0, nq = 0.0, nx = 0.0;
    for (int i = 0; i < D; ++i) {
        double a = (double)q[i], b = (double)x[i];
        dot += a * b;
        nq += a * a;
        nx += b * b;
    }
    if (metric == 1) return isfinite(dot) ? dot : NAN;
    /* cosine is defined for squared norms in [2^-126, inf): below that 1/|x| leaves the f32 range
     * the scan stores it in, and the reference's own f32 division has long produced inf/NaN */
    if (!(nq >= 0x1p-126) || !(nx >= 0x1p-126) || !isfinite(nq) || !isfinite(nx)) return NAN;
    double c = dot / (sqrt(nq) * sqrt(nx));
    return isfinite(c) ? c : NAN;
}

/* a ranks before b? (descending score, ties -> lower position) */
static int ranks_before(double sa, int64_t pa, double sb, int64_t pb) {
    if (sa != sb) return sa > sb;
    return pa < pb;
}

int orc_topk(const float* q, const float* m, int64_t N, int D, int metric, int k, int64_t* out_pos,
             double* out_score) {
    int cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        double c = orc_canonical_score(q, m + (size_t)n * D, D, metric);
        if (isnan(c)) continue;
        if (cnt == k && !ranks_before(c, n, out_score[k - 1], out_pos[k - 1])) continue;
        int j = (cnt < k) ? cnt : k - 1;
        while (j > 0 && ranks_before(c, n, out_score[j - 1], out_pos[j - 1])) {
            out_score[j] = out_score[j - 1];
            out_pos[j] = out_pos[j - 1];
            --j;
        }
        out_score[j] = c;
        out_pos[j] = n;
        if (cnt < k) ++cnt;
    }
    return cnt;
}

/* search.rs:269-278 */
float orc_ndarray_distance(const float* a, const float* b, int D) {
    float dot = 0.0f;
    for (int i = 0; i < D; ++i) dot += a[i] * b[i];
    float result = 1.0f - (dot / (float)D);
    return result > 0.0f ? result : 0.0f;
}

/* search.rs:157-182 with hnsw.search replaced by an exact scan over the selected sources.
 * Ranking uses the canonical f64 dot so that clamped (distance 0) entries still order by
 * similarity; the reported distance is the reference's f32 formula applied to that dot. */
int orc_search_vector(const float* q, const float* m, const int64_t* ids, const int64_t* source_of_row,
                      int64_t N, int D, const int64_t* sources, int n_sources, int k, int64_t* out_ids,
                      float* out_dist) {
    int64_t* pos = (int64_t*)malloc((size_t)k * sizeof(int64_t));
    double* sc = (double*)malloc((size_t)k * sizeof(double));
    int cnt = 0;
    for (int64_t n = 0; n < N; ++n) {
        int selected = (n_sources == 0);
        for (int s = 0; s < n_sources && !selected; ++s) selected = (sources[s] == source_of_row[n]);
        if (!selected) continue; /* search.rs:166 */
        double c = orc_canonical_score(q, m + (size_t)n * D, D, 1);
        if (isnan(c)) continue;
        if (cnt == k && !ranks_before(c, n, sc[k - 1], pos[k - 1])) continue;
        int j = (cnt < k) ? cnt : k - 1;
        while (j > 0 && ranks_before(c, n, sc[j - 1], pos[j - 1])) {
            sc[j] = sc[j - 1];
            pos[j] = pos[j - 1];
            --j;
        }
        sc[j] = c;
        pos[j] = n;
        if (cnt < k) ++cnt;
    }
    for (int j = 0; j < cnt; ++j) {
        out_ids[j] = ids ? ids[pos[j]] : pos[j];
        double d = 1.0 - sc[j] / (double)D; /* search.rs:275 */
        out_dist[j] = (float)(d > 0.0 ? d : 0.0); /* search.rs:277 */
    }
    free(pos);
    free(sc);
    return cnt;
}

/* search.rs:288-294 */
void orc_serialize_embedding(const float* v, size_t n, uint8_t* out) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t u;
        memcpy(&u, &v[i], 4);
        out[4 * i + 0] = (uint8_t)(u & 0xff);
        out[4 * i + 1] = (uint8_t)((u >> 8) & 0xff);
        out[4 * i + 2] = (uint8_t)((u >> 16) & 0xff);
        out[4 * i + 3] = (uint8_t)((u >> 24) & 0xff);
    }
}

/* search.rs:281-286 — `chunks(4)` then from_le_bytes; a short tail chunk panics in the reference,
 * here it is ignored and the caller sees the shorter count. */
size_t orc_deserialize_embedding(const uint8_t* blob, size_t n_bytes, float* out) {
    size_t n = n_bytes / 4;
    for (size_t i = 0; i < n; ++i) {
        uint32_t u = (uint32_t)blob[4 * i] | ((uint32_t)blob[4 * i + 1] << 8) |
                     ((uint32_t)blob[4 * i + 2] << 16) | ((uint32_t)blob[4 * i + 3] << 24);
        memcpy(&out[i], &u, 4);
    }
    return n;
}
