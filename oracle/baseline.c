/*
 * baseline.c — the timed CPU baseline of bench.py (cpu_baseline.kind = "port").
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY.  A multithreaded C port of the reference's brute-force
 * path lib.rs:67-77 followed by the sort+truncate of search.rs:179-180, in the reference's f32.
 * It is a reported number, not an optimisation target.
 */
#define _GNU_SOURCE
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_hardware_threads(void) {
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

/* 8 independent f32 accumulators so gcc can vectorise without -ffast-math. */
__attribute__((target_clones("avx2,fma", "default"))) static float dot_f32(const float* a, const float* b,
                                                                          int D) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= D; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
    float s = ((acc[0] + acc[4]) + (acc[1] + acc[5])) + ((acc[2] + acc[6]) + (acc[3] + acc[7]));
    for (; i < D; ++i) s += a[i] * b[i];
    return s;
}

typedef struct {
    int64_t pos;
    float score;
} ent_t;

/* keep the k best (descending score, ties -> lower position) in a small sorted array */
static void topk_push(ent_t* list, int* cnt, int k, float s, int64_t pos) {
    if (isnan(s)) return;
    if (*cnt == k) {
        const ent_t* w = &list[k - 1];
        if (!(s > w->score || (s == w->score && pos < w->pos))) return;
    }
    int j = (*cnt < k) ? *cnt : k - 1;
    while (j > 0 && (s > list[j - 1].score || (s == list[j - 1].score && pos < list[j - 1].pos))) {
        list[j] = list[j - 1];
        --j;
    }
    list[j].score = s;
    list[j].pos = pos;
    if (*cnt < k) ++*cnt;
}

typedef struct {
    const float* qn; /* normalised queries [B][D] */
    const float* m;
    float* mn;       /* reference-shaped: normalised corpus [N][D] */
    float* scores;   /* reference-shaped: [B][N] */
    int64_t n0, n1, N;
    int B, D, k, phase;
    ent_t* lists; /* [B][k] per thread */
    int* cnts;    /* [B] */
} job_t;

static void* fused_worker(void* arg) {
    job_t* j = (job_t*)arg;
    for (int b = 0; b < j->B; ++b) j->cnts[b] = 0;
    for (int64_t n = j->n0; n < j->n1; ++n) {
        const float* x = j->m + (size_t)n * j->D;
        float inv = 1.0f / sqrtf(dot_f32(x, x, j->D));
        for (int b = 0; b < j->B; ++b) {
            float s = dot_f32(j->qn + (size_t)b * j->D, x, j->D) * inv;
            topk_push(j->lists + (size_t)b * j->k, &j->cnts[b], j->k, s, n);
        }
    }
    return NULL;
}

static void* shaped_worker(void* arg) {
    job_t* j = (job_t*)arg;
    if (j->phase == 0) { /* matches / matches.linalg_norm(...)  (lib.rs:75) — materialised */
        for (int64_t n = j->n0; n < j->n1; ++n) {
            const float* x = j->m + (size_t)n * j->D;
            float* o = j->mn + (size_t)n * j->D;
            float nrm = sqrtf(dot_f32(x, x, j->D));
            for (int i = 0; i < j->D; ++i) o[i] = x[i] / nrm;
        }
    } else if (j->phase == 1) { /* matmul (lib.rs:64) — full [B,N] output */
        for (int64_t n = j->n0; n < j->n1; ++n)
            for (int b = 0; b < j->B; ++b)
                j->scores[(size_t)b * j->N + n] =
                    dot_f32(j->qn + (size_t)b * j->D, j->mn + (size_t)n * j->D, j->D);
    } else { /* sort + truncate (search.rs:179-180) as a selection over the score rows */
        for (int b = 0; b < j->B; ++b) j->cnts[b] = 0;
        for (int b = 0; b < j->B; ++b)
            for (int64_t n = j->n0; n < j->n1; ++n)
                topk_push(j->lists + (size_t)b * j->k, &j->cnts[b], j->k, j->scores[(size_t)b * j->N + n], n);
    }
    return NULL;
}

static void run_threads(job_t* jobs, int T, void* (*fn)(void*)) {
    pthread_t* th = (pthread_t*)malloc((size_t)T * sizeof(pthread_t));
    for (int t = 1; t < T; ++t) pthread_create(&th[t], NULL, fn, &jobs[t]);
    fn(&jobs[0]);
    for (int t = 1; t < T; ++t) pthread_join(th[t], NULL);
    free(th);
}

static void merge_lists(job_t* jobs, int T, int B, int k, int64_t* out_pos, float* out_score) {
    ent_t* acc = (ent_t*)malloc((size_t)k * sizeof(ent_t));
    for (int b = 0; b < B; ++b) {
        int cnt = 0;
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < jobs[t].cnts[b]; ++i) {
                ent_t* e = &jobs[t].lists[(size_t)b * k + i];
                topk_push(acc, &cnt, k, e->score, e->pos);
            }
        for (int i = 0; i < k; ++i) {
            out_pos[(size_t)b * k + i] = i < cnt ? acc[i].pos : -1;
            out_score[(size_t)b * k + i] = i < cnt ? acc[i].score : NAN;
        }
    }
    free(acc);
}

static float* normalise_queries(const float* queries, int B, int D) {
    float* qn = (float*)malloc((size_t)B * D * sizeof(float));
    for (int b = 0; b < B; ++b) {
        const float* q = queries + (size_t)b * D;
        float nrm = sqrtf(dot_f32(q, q, D));
        for (int i = 0; i < D; ++i) qn[(size_t)b * D + i] = q[i] / nrm;
    }
    return qn;
}

static job_t* make_jobs(int T, const float* qn, const float* m, int64_t N, int B, int D, int k) {
    job_t* jobs = (job_t*)calloc((size_t)T, sizeof(job_t));
    for (int t = 0; t < T; ++t) {
        jobs[t].qn = qn;
        jobs[t].m = m;
        jobs[t].N = N;
        jobs[t].n0 = N * t / T;
        jobs[t].n1 = N * (t + 1) / T;
        jobs[t].B = B;
        jobs[t].D = D;
        jobs[t].k = k;
        jobs[t].lists = (ent_t*)malloc((size_t)B * k * sizeof(ent_t));
        jobs[t].cnts = (int*)calloc((size_t)B, sizeof(int));
    }
    return jobs;
}

static void free_jobs(job_t* jobs, int T) {
    for (int t = 0; t < T; ++t) {
        free(jobs[t].lists);
        free(jobs[t].cnts);
    }
    free(jobs);
}

double orc_baseline_scan_fused(const float* queries, int B, const float* m, int64_t N, int D, int k,
                               int threads, int64_t* out_pos, float* out_score) {
    int T = threads > 0 ? threads : 1;
    double t0 = now_s();
    float* qn = normalise_queries(queries, B, D);
    job_t* jobs = make_jobs(T, qn, m, N, B, D, k);
    run_threads(jobs, T, fused_worker);
    merge_lists(jobs, T, B, k, out_pos, out_score);
    double t1 = now_s();
    free_jobs(jobs, T);
    free(qn);
    return t1 - t0;
}

double orc_baseline_scan_reference_shaped(const float* queries, int B, const float* m, int64_t N, int D,
                                          int k, int threads, int64_t* out_pos, float* out_score) {
    int T = threads > 0 ? threads : 1;
    double t0 = now_s();
    float* qn = normalise_queries(queries, B, D);
    float* mn = (float*)malloc((size_t)N * D * sizeof(float));
    float* scores = (float*)malloc((size_t)B * N * sizeof(float));
    job_t* jobs = make_jobs(T, qn, m, N, B, D, k);
    for (int phase = 0; phase < 3; ++phase) {
        for (int t = 0; t < T; ++t) {
            jobs[t].mn = mn;
            jobs[t].scores = scores;
            jobs[t].phase = phase;
        }
        run_threads(jobs, T, shaped_worker);
    }
    merge_lists(jobs, T, B, k, out_pos, out_score);
    double t1 = now_s();
    free_jobs(jobs, T);
    free(qn);
    free(mn);
    free(scores);
    return t1 - t0;
}
