// scan_kernels.hip — gfx950 kernels of the exact similarity scan (replaces lib.rs:63-77 +
// search.rs:157-182 of the reference).
//
// Pipeline of one search (all on the context stream, no host round trip in between):
//   prep_queries -> seed -> scan (wave | mfma) -> rescore -> select
// The scan is a *screening* pass: it streams the corpus once, computes an approximate score s per
// (query,row) with |s - c| <= eps of the canonical f64 score c, and emits every row that could
// still be in the top-k:   emit iff !(s < tau_q - 2*eps),   tau_q = running k-th best s.
// tau only grows, so the emitted set is a superset of the exact top-k; rescore+select then rank
// the few hundred survivors per query in exact f64.  HBM traffic = one pass over the rows.
#include <algorithm>
#include <cstdlib>

#include "scan.h"
#include "synth.h"

namespace pcv {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Every pointer the kernels follow comes out of a struct in memory, so the compiler only knows it
// as a generic ("flat") address.  flat_load/flat_atomic count on BOTH vmcnt and lgkmcnt and return
// out of order: every LDS wait then has to drain the corpus prefetch as well.  All global traffic
// therefore goes through these address-space(1) accessors (global_load / global_store / global_atomic).
#define PCV_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ T gld(const T* p) {
    return *(const PCV_GLOBAL T*)p;
}
template <class T>
__device__ __forceinline__ void gst(T* p, T v) {
    *(PCV_GLOBAL T*)p = v;
}
__device__ __forceinline__ float4 gld4(const float4* p) {
    const f32x4 v = *(const PCV_GLOBAL f32x4*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gld4(const float* p) { return gld4((const float4*)p); }
__device__ __forceinline__ uint32_t g_atomic_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add((PCV_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t g_atomic_max(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_max((PCV_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// returns the value found (== expected on success)
__device__ __forceinline__ uint32_t g_atomic_cas(uint32_t* p, uint32_t expected, uint32_t desired) {
    __hip_atomic_compare_exchange_strong((PCV_GLOBAL uint32_t*)p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}

// corpus rows are read exactly once per scan: optionally mark the loads non-temporal
template <bool NTL>
__device__ __forceinline__ float4 ld_row(const float4* p) {
    if constexpr (NTL) {
        const f32x4 v = __builtin_nontemporal_load((const PCV_GLOBAL f32x4*)p);
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return gld4(p);
    }
}

__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) {
    return __hip_atomic_load((const PCV_GLOBAL uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ int find_seg(const ScanParams& p, uint32_t gb) {
    int s = 0;
#pragma unroll
    for (int i = 1; i < kMaxSeg; ++i)
        if (i < p.nseg && gb >= p.seg[i].blk0) s = i;
    return s;
}

// slots[q][0..k) always hold approximate scores of k DISTINCT rows (or -inf), each slot only ever
// grows, so min(slots) is a valid lower bound of the final k-th best approximate score.
// offer_slot: try to replace the current minimum by the score `s` of a row not yet in the slots.
__device__ __forceinline__ void offer_slot(const ScanParams& p, int q, float s) {
    const uint32_t key = f32_key(s);
    uint32_t* sl = p.slots + (size_t)q * kMaxK;
    for (int attempt = 0; attempt < 8; ++attempt) {
        uint32_t mn = 0xffffffffu;
        int mi = 0;
        for (int i = 0; i < p.k; ++i) {
            uint32_t v = ld_relaxed(&sl[i]);
            if (v < mn) {
                mn = v;
                mi = i;
            }
        }
        if (key <= mn) return;
        if (g_atomic_cas(&sl[mi], mn, key) == mn) {
            uint32_t nm = 0xffffffffu;
            for (int i = 0; i < p.k; ++i) nm = min(nm, ld_relaxed(&sl[i]));
            g_atomic_max(&p.tau[q * kHot], nm);
            return;
        }
    }
}

// offer_slot for the NT queries a lane owns (query 32*t + c), all chains advancing in lock step so
// their memory round trips overlap.
template <int NT>
__device__ __forceinline__ void offer_slots(const ScanParams& p, int c, const bool (&want)[NT], const float (&s)[NT]) {
    bool live[NT];
    uint32_t key[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        live[t] = want[t];
        key[t] = f32_key(s[t]);
    }
    for (int attempt = 0; attempt < 8; ++attempt) {
        uint32_t mn[NT];
        int mi[NT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            mn[t] = 0xffffffffu;
            mi[t] = 0;
            if (live[t]) {
                const uint32_t* sl = p.slots + (size_t)(32 * t + c) * kMaxK;
                for (int i = 0; i < p.k; ++i) {
                    const uint32_t v = ld_relaxed(&sl[i]);
                    if (v < mn[t]) {
                        mn[t] = v;
                        mi[t] = i;
                    }
                }
                if (key[t] <= mn[t]) live[t] = false;
            }
            any |= live[t];
        }
        if (!any) return;
        uint32_t old[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            old[t] = live[t] ? g_atomic_cas(&p.slots[(size_t)(32 * t + c) * kMaxK + mi[t]], mn[t], key[t]) : 0u;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (live[t] && old[t] == mn[t]) {
                const uint32_t* sl = p.slots + (size_t)(32 * t + c) * kMaxK;
                uint32_t nm = 0xffffffffu;
                for (int i = 0; i < p.k; ++i) nm = min(nm, ld_relaxed(&sl[i]));
                g_atomic_max(&p.tau[(32 * t + c) * kHot], nm);
                live[t] = false;
            }
    }
}

// A surviving (query,row) pair: append to the query's candidate list and, unless the row was
// already ranked by the seed kernel, try to raise the running k-th best.
__device__ __noinline__ void emit_hit(const ScanParams& p, int q, int seg, uint32_t row, float s,
                                      bool feeds_slots) {
    uint32_t idx = g_atomic_add(&p.cand_cnt[q * kHot], 1u);
    if (idx < p.cand_cap) {
        gst(&p.cand[(size_t)q * p.cand_cap + idx], ((uint64_t)(uint32_t)seg << 32) | row);
        gst(&p.cand_s[(size_t)q * p.cand_cap + idx], s);
    }
    if (feeds_slots && isfinite(s)) offer_slot(p, q, s);
}

// ------------------------------------------------------------------------------------------------
// finalize-time kernels
// ------------------------------------------------------------------------------------------------

// row-major staging [n][D] -> blocked layout, rows row0.. of the segment (buffer pre-zeroed)
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ rows, int64_t n, int D, int D4,
                                                        float4* __restrict__ blk, uint32_t row0) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(t & 31);
    const int64_t u = t >> 5;
    const int f4 = (int)(u % D4);
    const int64_t lb = u / D4;  // block relative to the first touched block
    const uint32_t first_blk = row0 >> 5;
    const int64_t row = (lb + first_blk) * 32 + r;  // row inside the segment
    const int64_t src = row - row0;
    if (src < 0 || src >= n) return;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int f = f4 * 4 + j;
        v[j] = f < D ? rows[src * D + f] : 0.0f;
    }
    blk[((lb + first_blk) * D4 + f4) * 32 + r] = make_float4(v[0], v[1], v[2], v[3]);
}

// per-row scale = 1/|x| (cosine) or 1 (dot); 0 marks rows that can never be a result
// (padding, zero / non-finite norm).  |x|^2 accumulated in f64 in feature order.
__global__ __launch_bounds__(256) void row_scales_kernel(const float4* __restrict__ blk, uint32_t nblocks,
                                                         uint32_t nrows, int D4, int metric,
                                                         float* __restrict__ scale, uint32_t* max_norm_bits) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nblocks * 32) return;
    float out = 0.0f;
    if (row < nrows) {
        const float4* base = blk + (size_t)(row >> 5) * D4 * 32 + (row & 31);
        double nx = 0.0;
        for (int f4 = 0; f4 < D4; ++f4) {
            float4 v = base[(size_t)f4 * 32];
            nx += (double)v.x * (double)v.x;
            nx += (double)v.y * (double)v.y;
            nx += (double)v.z * (double)v.z;
            nx += (double)v.w * (double)v.w;
        }
        const bool finite = nx < __builtin_inf();  // false for inf and NaN
        if (metric == PCV_METRIC_DOT) {
            out = finite ? 1.0f : 0.0f;
        } else {
            out = (finite && nx >= 0x1p-126) ? (float)(1.0 / sqrt(nx)) : 0.0f;
        }
        if (out != 0.0f) {
            float nrm = (float)sqrt(nx) * 1.000001f;
            atomicMax(max_norm_bits, __builtin_bit_cast(uint32_t, nrm));
        }
    }
    scale[row] = out;
}

__global__ __launch_bounds__(256) void synth_inv_kernel(uint32_t nrows, int D4src, uint64_t seed, int64_t first_row,
                                                        float* __restrict__ inv) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    double nx = 0.0;
    for (int f4 = 0; f4 < D4src; ++f4) {
        float4 v = synth_piece(seed, first_row + row, (uint32_t)f4);
        nx += (double)v.x * (double)v.x;
        nx += (double)v.y * (double)v.y;
        nx += (double)v.z * (double)v.z;
        nx += (double)v.w * (double)v.w;
    }
    inv[row] = (float)(1.0 / sqrt(nx));
}

// thread per (block, piece, row-in-block), grid-stride (a 100M-row segment has 9.6e9 work items,
// more than one launch dimension can carry): segment rows row0..row0+nrows get synth rows
// first_row.. ; pieces beyond D stay zero
__global__ __launch_bounds__(256) void synth_fill_kernel(float4* __restrict__ blk, uint32_t nrows, uint32_t row0,
                                                         int D4src, int D4, uint64_t seed, int64_t first_row,
                                                         const float* __restrict__ inv, int64_t total) {
    const uint32_t first_blk = row0 >> 5;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int r = (int)(t & 31);
        const int64_t u = t >> 5;
        const int f4 = (int)(u % D4src);
        const int64_t lb = u / D4src;
        const int64_t row = (lb + first_blk) * 32 + r;
        const int64_t src = row - row0;
        if (src < 0 || src >= nrows) continue;
        float4 v = synth_piece(seed, first_row + src, (uint32_t)f4);
        if (inv) {
            float s = inv[src];
            v.x *= s;
            v.y *= s;
            v.z *= s;
            v.w *= s;
        }
        blk[((lb + first_blk) * D4 + f4) * 32 + r] = v;
    }
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const SegDesc* __restrict__ segs, int nseg,
                                                          const int64_t* __restrict__ pos, int64_t n, int D, int D4,
                                                          float* __restrict__ out_rows,
                                                          int64_t* __restrict__ out_ids) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int D4src = (D + 3) / 4;
    const int64_t i = t / D4src;
    const int f4 = (int)(t % D4src);
    if (i >= n) return;
    const int64_t gp = pos[i];
    int s = -1;
    for (int j = 0; j < nseg; ++j)
        if (gp >= segs[j].pos0 && gp < segs[j].pos0 + (int64_t)segs[j].nrows) s = j;
    float4 v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
    int64_t id = -1;
    if (s >= 0) {
        const uint32_t row = (uint32_t)(gp - segs[s].pos0);
        v = segs[s].blk[((size_t)(row >> 5) * D4 + f4) * 32 + (row & 31)];
        id = segs[s].ids ? segs[s].ids[row] : segs[s].id0 + row;
    }
    const float vv[4] = {v.x, v.y, v.z, v.w};
    for (int j = 0; j < 4; ++j)
        if (f4 * 4 + j < D) out_rows[i * D + f4 * 4 + j] = vv[j];
    if (f4 == 0 && out_ids) out_ids[i] = id;
}

// ------------------------------------------------------------------------------------------------
// per-search kernels
// ------------------------------------------------------------------------------------------------

// one workgroup per query slot (64 slots always, so the bf16 tile rows >= B are zeroed)
__global__ __launch_bounds__(64) void prep_queries_kernel(const float* __restrict__ queries, int B, int D, int Dp,
                                                          int metric, float eps_rel, float max_norm, int k,
                                                          float* __restrict__ qf32, uint16_t* __restrict__ qbf16,
                                                          float* __restrict__ qraw, double* __restrict__ qnorm2,
                                                          float* __restrict__ margin, uint32_t* __restrict__ tau,
                                                          uint32_t* __restrict__ slots,
                                                          uint32_t* __restrict__ cand_cnt) {
    extern __shared__ float sraw[];  // [Dp]
    const int q = blockIdx.x;
    const int tid = threadIdx.x;
    if (q >= B) {
        if (q < kMfmaQueries)
            for (int i = tid; i < Dp; i += 64) qbf16[(size_t)q * Dp + i] = 0;
        return;
    }
    __shared__ double s_nq;
    const float* src = queries + (size_t)q * D;
    for (int i = tid; i < Dp; i += 64) sraw[i] = i < D ? src[i] : 0.0f;
    __syncthreads();
    if (tid == 0) {  // f64, feature order: the canonical |q|^2 (oracle/scan.c:orc_canonical_score)
        double nq = 0.0;
        for (int i = 0; i < D; ++i) nq += (double)sraw[i] * (double)sraw[i];
        s_nq = nq;
    }
    __syncthreads();
    const double nq = s_nq;
    const bool ok = (nq < __builtin_inf()) && (metric == PCV_METRIC_DOT || nq >= 0x1p-126);
    const float inv = (metric == PCV_METRIC_DOT) ? 1.0f : (ok ? (float)(1.0 / sqrt(nq)) : 0.0f);
    for (int i = tid; i < Dp; i += 64) {
        const float raw = sraw[i];
        const float qh = ok ? raw * inv : 0.0f;
        qraw[(size_t)q * Dp + i] = raw;
        qf32[(size_t)q * Dp + i] = qh;
        const __bf16 hb = (__bf16)qh;
        qbf16[(size_t)q * Dp + i] = __builtin_bit_cast(uint16_t, hb);
    }
    for (int i = tid; i < kMaxK; i += 64) slots[(size_t)q * kMaxK + i] = kKeyNegInf;
    if (tid == 0) {
        qnorm2[q] = ok ? nq : __builtin_nan("");
        // cosine: scores are O(1); dot: |s - c| <= eps_rel * |q| * max|x|
        float m = 2.0f * eps_rel;
        if (metric == PCV_METRIC_DOT) m *= (float)sqrt(nq) * max_norm * 1.0001f;
        margin[q] = m;
        tau[q * kHot] = kKeyNegInf;
        cand_cnt[q * kHot] = 0;
    }
}

// k rounds of workgroup-wide argmax over `n` keys in LDS (0 = absent); round j's winner goes to
// out[j] (0 when exhausted).  All 256 threads call it.
__device__ __forceinline__ void topk_keys_lds(uint32_t* keys, uint32_t n, int k, unsigned long long* red4,
                                              uint32_t* out) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int j = 0; j < k; ++j) {
        unsigned long long best = 0;
        for (uint32_t i = tid; i < n; i += 256) {
            const unsigned long long c = ((unsigned long long)keys[i] << 32) | (0xffffffffu - i);
            best = c > best ? c : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_xor(best, off);
            best = o > best ? o : best;
        }
        if (lane == 0) red4[wave] = best;
        __syncthreads();
        unsigned long long w = red4[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) w = red4[i] > w ? red4[i] : w;
        const uint32_t wkey = (uint32_t)(w >> 32);
        if (tid == 0) {
            out[j] = wkey;
            if (wkey) keys[0xffffffffu - (uint32_t)w] = 0;
        }
        __syncthreads();
    }
}

// Seed, step 1: workgroup (part, query group) ranks rows [part*1024, +1024) of segment 0 against QG
// queries with f32 FMA chains and keeps the k best keys per query.  Gives the streaming kernels a
// useful threshold from the first block on: with W waves in flight the first round screens 32*W rows
// against the seed threshold.  QG queries share every row load (one query per workgroup made the
// 64-query seed L2-bandwidth-bound: 1.6 GB of row re-reads).
template <int QG>
__global__ __launch_bounds__(256) void seed_partial_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ float smem[];
    const int Dp = p.D4 * 4;
    float* sq = smem;                              // [QG][Dp]
    uint32_t* keys = (uint32_t*)(smem + QG * Dp);  // [QG][kSeedPartRows]
    const int part = blockIdx.x, q0 = blockIdx.y * QG, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const SegDesc& sg = p.seg[0];
    const uint32_t nseed = min(sg.nrows, p.seed_blocks * 32u);
    for (int i = tid; i < QG * Dp; i += 256) {
        const int q = q0 + i / Dp;
        sq[i] = q < p.B ? gld(&p.qf32[(size_t)q * Dp + (i % Dp)]) : 0.0f;
    }
    __syncthreads();
    // thread t owns rows base + t + 256*u, u = 0..RPT-1
    const uint32_t row0 = part * kSeedPartRows + tid;
    constexpr int RPT = kSeedPartRows / 256;
    const float4* base[RPT];
    float acc[RPT][QG];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t row = min(row0 + 256u * u, nseed ? nseed - 1 : 0u);
        base[u] = sg.blk + (size_t)(row >> 5) * p.D4 * 32 + (row & 31);
#pragma unroll
        for (int g = 0; g < QG; ++g) acc[u][g] = 0.0f;
    }
    for (int f0 = 0; f0 < p.D4; f0 += 4) {  // D4 is a multiple of 16
        // all 4*RPT row loads of this step are issued before any of them is used (left to itself the
        // compiler sinks each load next to its FMAs and the loop runs at one L2 latency per piece)
        float4 v[4][RPT];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int u = 0; u < RPT; ++u) v[j][u] = gld4(base[u] + (size_t)(f0 + j) * 32);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const float4 qv = *(const float4*)&sq[g * Dp + (f0 + j) * 4];
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    acc[u][g] = fmaf(qv.x, v[j][u].x, acc[u][g]);
                    acc[u][g] = fmaf(qv.y, v[j][u].y, acc[u][g]);
                    acc[u][g] = fmaf(qv.z, v[j][u].z, acc[u][g]);
                    acc[u][g] = fmaf(qv.w, v[j][u].w, acc[u][g]);
                }
            }
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const uint32_t row = row0 + 256u * u;
        const float sc = row < nseed ? gld(&sg.scale[row]) : 0.0f;
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float s = acc[u][g] * sc;
            keys[g * kSeedPartRows + tid + 256 * u] = (sc != 0.0f && isfinite(s)) ? f32_key(s) : 0u;  // 0 = absent
        }
    }
    __syncthreads();
    // selection: each wave ranks whole queries on its own (k rounds of a wave-wide max, no barriers)
    for (int g = wave; g < QG; g += 4) {
        const int q = q0 + g;
        if (q >= p.B) continue;
        uint32_t* kq = keys + g * kSeedPartRows;
        uint32_t* out = p.seed_part + ((size_t)q * kSeedParts + part) * kMaxK;
        for (int j = 0; j < p.k; ++j) {
            unsigned long long best = 0;
            for (uint32_t i = lane; i < (uint32_t)kSeedPartRows; i += 64) {
                const unsigned long long cnd = ((unsigned long long)kq[i] << 32) | (0xffffffffu - i);
                best = cnd > best ? cnd : best;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(best, off);
                best = o > best ? o : best;
            }
            const uint32_t wkey = (uint32_t)(best >> 32);
            if (lane == 0) {
                gst(&out[j], wkey);
                if (wkey) kq[0xffffffffu - (uint32_t)best] = 0;
            }
            __builtin_amdgcn_wave_barrier();  // LDS is in order within a wave: the next round sees the removal
        }
    }
}

// Seed, step 2: merge the per-part lists into the query's slots and threshold.
__global__ __launch_bounds__(256) void seed_merge_kernel(const ScanParams* __restrict__ pp, int nparts) {
    const ScanParams& p = *pp;
    __shared__ uint32_t keys[kSeedParts * kMaxK];
    __shared__ uint32_t outk[kMaxK];
    __shared__ unsigned long long red4[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const uint32_t n = (uint32_t)nparts * p.k;
    for (uint32_t i = tid; i < n; i += 256)
        keys[i] = gld(&p.seed_part[((size_t)q * kSeedParts + i / p.k) * kMaxK + i % p.k]);
    __syncthreads();
    topk_keys_lds(keys, n, p.k, red4, outk);
    for (int j = tid; j < p.k; j += 256) gst(&p.slots[(size_t)q * kMaxK + j], outk[j] ? outk[j] : kKeyNegInf);
    if (tid == 0) gst(&p.tau[q * kHot], outk[p.k - 1] ? outk[p.k - 1] : kKeyNegInf);  // k-th best seed row, -inf if fewer
}

// Wave-reduction scan for 1..4 queries (BASELINE config "10M x 384, batch=1"): pure HBM streaming.
// Lane (r = lane&31, h = lane>>5) owns row r of the block and the pieces f4 = 2j+h; the two halves
// of a row are combined with one cross-lane add.  f32 FMA chain -> eps ~ Dp * 2^-24.
template <int NB, bool NTL>
__global__ __launch_bounds__(256) void scan_wave_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ float sq[];  // [NB][Dp]
    const int Dp = p.D4 * 4;
    for (int i = threadIdx.x; i < NB * Dp; i += 256) sq[i] = gld(&p.qf32[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, h = lane >> 5;
    float mrg[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) mrg[b] = gld(&p.margin[b]);
    const uint32_t total_waves = gridDim.x * 4;
    const int NCH = p.D4 >> 4;  // chunks of 8 pieces per lane (64 features of the row, both halves)
    if (blockIdx.x * 4 + wave >= p.total_blocks) return;

    // Same flat (block, chunk) stream with register chunk buffers as the MFMA kernel: the loads of the
    // next chunk — also across block boundaries — are in flight while the current one is multiplied.
    struct Cur {
        uint32_t gb;
        int si;
        uint32_t lb;
        const float4* base;
        int ch;
    } cons, prod;
    auto enter = [&](Cur& k, uint32_t gb) {
        k.gb = gb;
        k.ch = 0;
        if (gb < p.total_blocks) {
            k.si = find_seg(p, gb);
            k.lb = gb - p.seg[k.si].blk0;
            k.base = p.seg[k.si].blk + (size_t)k.lb * p.D4 * 32 + h * 32 + r;
        }
    };
    enter(cons, blockIdx.x * 4 + wave);
    prod = cons;
    float sc_cur = gld(&p.seg[cons.si].scale[(size_t)cons.lb * 32 + r]), sc_next = 0.0f;
    uint32_t tk[NB];
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0f;

    float4 buf[2][8];
    auto produce = [&](float4 (&bf)[8]) {
        if (prod.gb >= p.total_blocks) return;
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[i] = ld_row<NTL>(prod.base + (size_t)(prod.ch * 8 + i) * 64);
        if (++prod.ch == NCH) {
            enter(prod, prod.gb + total_waves);
            if (prod.gb < p.total_blocks) sc_next = gld(&p.seg[prod.si].scale[(size_t)prod.lb * 32 + r]);
        }
    };
    auto consume = [&](const float4 (&bf)[8]) {
        if (cons.ch == (NCH >= 2 ? NCH - 2 : 0)) {  // thresholds one chunk ahead of the epilogue
#pragma unroll
            for (int b = 0; b < NB; ++b) tk[b] = ld_relaxed(&p.tau[b * kHot]);
        }
        const float* qb = sq + h * 4 + cons.ch * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 v = bf[i];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const float4 qv = *(const float4*)&qb[b * Dp + i * 8];
                acc[b] = fmaf(qv.x, v.x, acc[b]);
                acc[b] = fmaf(qv.y, v.y, acc[b]);
                acc[b] = fmaf(qv.z, v.z, acc[b]);
                acc[b] = fmaf(qv.w, v.w, acc[b]);
            }
        }
        if (++cons.ch == NCH) {  // block done: lane r (h = 0) owns row r
            const uint32_t row = cons.lb * 32 + r;
            bool any = false;
            float s[NB], thr[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc[b] += __shfl_xor(acc[b], 32);
                s[b] = acc[b] * sc_cur;
                thr[b] = key_f32(tk[b]) - mrg[b];
                any |= (h == 0) && (sc_cur != 0.0f) && !(s[b] < thr[b]);
                acc[b] = 0.0f;
            }
            if (__any(any)) {
                const bool feeds = !(cons.si == 0 && cons.lb < p.seed_blocks);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if ((h == 0) && (sc_cur != 0.0f) && !(s[b] < thr[b])) emit_hit(p, b, cons.si, row, s[b], feeds);
            }
            enter(cons, cons.gb + total_waves);
            sc_cur = sc_next;
        }
    };
    produce(buf[0]);
    while (true) {
        produce(buf[1]);
        consume(buf[0]);
        if (cons.gb >= p.total_blocks) return;
        produce(buf[0]);
        consume(buf[1]);
        if (cons.gb >= p.total_blocks) return;
    }
}

// MFMA tile scan for up to 64 queries (BASELINE config "100M x 384, batch=64").
// D[row][query] = sum_k A[row][k] * B[k][query] on v_mfma_f32_32x32x16_bf16: A = 32 corpus rows of a
// block (f32 from HBM, rounded to bf16 in registers), B = the query tile (bf16, LDS, XOR-swizzled so
// the 16-lane ds_read_b128 groups are conflict-free).  One bf16 product term -> eps = 2^-8: ~2.4x
// more survivors than an f32 screen on random data, for 1/16 of the f32 matrix cost; survivors are
// re-ranked exactly anyway.  Each wave streams its own blocks straight into registers: the blocked
// HBM layout makes every load two contiguous 512 B runs, so there is no LDS round trip for the corpus.
// Variants: NT = 32-query tiles per wave (1, 2 -> B <= 64; 4 -> B <= 128), WPB = waves per
// workgroup, NBUF = chunk buffers per wave (NBUF-1 chunks of loads in flight while one is consumed).
//   B <= 64 : 256 threads, 3 workgroups/CU (155 VGPR, 3 waves/SIMD), NBUF 2 (NBUF 3 at 2 waves/SIMD
//             measured no faster)
//   B <= 128: 512 threads sharing one 96 KB query tile, 1 workgroup/CU, 2 waves/SIMD, NBUF 3: the
//             scan stays HBM-bound (MFMA ~25 % busy), so 128 queries cost the same 23 ms as 64
// (A 512-thread / 4-waves-per-SIMD build of the B <= 64 kernel was tried: the 128-VGPR cap spills 23
// registers into the chunk loop and runs 15 % slower.)
struct BlockCursor {  // position of a wave in its flat (block, chunk) stream
    uint32_t gb;      // launch-wide block index (>= total_blocks: exhausted)
    int si;           // segment
    uint32_t lb;      // block inside the segment
    const float4* base;
    int ch;           // chunk inside the block
};

template <int NT, bool NTL, int WPB, int NBUF>
__global__ __launch_bounds__(WPB * 64, WPB == 4 ? (NBUF == 2 ? 3 : 2) : 2) void scan_mfma_kernel(
    const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ uint4 lq[];  // [NT*32][Dp/8] 16-byte pieces of 8 bf16, swizzled
    const int D4 = p.D4;
    const int P8 = D4 >> 1;   // 16-B pieces per query row
    const int NCH = D4 >> 4;  // chunks of 64 features
    for (int i = threadIdx.x; i < NT * 32 * P8; i += WPB * 64) {
        const int q = i / P8, pc = i - q * P8;
        lq[q * P8 + ((pc & ~15) | ((pc ^ q) & 15))] = __builtin_bit_cast(uint4, gld4((const float4*)p.qbf16 + i));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    float mrg[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) mrg[t] = (32 * t + c < p.B) ? gld(&p.margin[32 * t + c]) : 0.0f;

    const uint32_t total_waves = gridDim.x * WPB;
    if (blockIdx.x * WPB + wave >= p.total_blocks) return;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;

    auto enter_block = [&](BlockCursor& k, uint32_t gb) {
        k.gb = gb;
        k.ch = 0;
        if (gb < p.total_blocks) {
            k.si = find_seg(p, gb);
            k.lb = gb - p.seg[k.si].blk0;
            k.base = p.seg[k.si].blk + (size_t)k.lb * D4 * 32 + h * 64 + c;
        }
    };
    BlockCursor cons, prod;  // consumer (MFMA) and producer (loads) positions; prod runs NBUF-1 chunks ahead
    enter_block(cons, blockIdx.x * WPB + wave);
    prod = cons;
    // scale of this lane's row (1/|x|, 1 or 0): multiplied into the A operand before the bf16
    // rounding, so the accumulators are final screening scores.  sc_next belongs to the block the
    // producer has entered but the consumer has not.
    float sc_cur = gld(&p.seg[cons.si].scale[(size_t)cons.lb * 32 + c]), sc_next = 0.0f;

    float4 buf[NBUF][8];
    // lane's pieces of k-step ks of a chunk: f4 = chunk*16 + ks*4 + 2h + e  (2h folded into base)
    auto produce = [&](float4 (&b)[8]) {
        if (prod.gb >= p.total_blocks) return;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            b[i] = ld_row<NTL>(prod.base + (size_t)(prod.ch * 16 + (i >> 1) * 4 + (i & 1)) * 32);
        if (++prod.ch == NCH) {
            enter_block(prod, prod.gb + total_waves);
            if (prod.gb < p.total_blocks) sc_next = gld(&p.seg[prod.si].scale[(size_t)prod.lb * 32 + c]);
        }
    };

    // thresholds of the block being finished: issued one chunk ahead of the epilogue so their
    // latency is not exposed
    uint32_t tauk[NT];
    auto tau_prefetch = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) tauk[t] = (32 * t + c < p.B) ? ld_relaxed(&p.tau[(32 * t + c) * kHot]) : 0u;
    };

    auto epilogue = [&](int esi, uint32_t elb) {
        if (NCH < 2) tau_prefetch();
        float thr[NT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q = 32 * t + c;
            thr[t] = (q < p.B) ? key_f32(tauk[t]) - mrg[t] : __builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= !(acc[t][i] < thr[t]);
        }
        if (__any(any)) {
            // Rare path — but every wave takes it a handful of times while the thresholds are still
            // loose, and under a saturated memory system each dependent round trip costs ~4 us, so it
            // is organised in phases that keep all tiles' requests in flight together:
            //   A  per lane (= one query per tile): hit mask, count, best score          (registers)
            //   B  one list-space reservation per tile                                   (1 round trip)
            //   C  store the hits (fire and forget)
            //   D  offer the best hit to the running top-k, only if it beats the threshold the lane
            //      already holds (hits inside the 2*eps margin cannot raise it)          (<= 3 round trips)
            const bool feeds = !(esi == 0 && elb < p.seed_blocks);
            const float* scp = p.seg[esi].scale + (size_t)elb * 32;
            uint32_t hitmask[NT], idx[NT];
            float best[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = 32 * t + c;
                hitmask[t] = 0;
                best[t] = -__builtin_inff();
                if (q < p.B) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float s = acc[t][i];  // the row scale is already folded into the A operand
                        if (!(s < thr[t])) {
                            // a zero / non-finite score may belong to a padding or invalid row (scale 0)
                            const bool suspect = (s == 0.0f) || !isfinite(s);
                            if (!suspect || gld(&scp[(i & 3) + 8 * (i >> 2) + 4 * h]) != 0.0f) {
                                hitmask[t] |= 1u << i;
                                if (isfinite(s)) best[t] = fmaxf(best[t], s);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)
                idx[t] = hitmask[t] ? g_atomic_add(&p.cand_cnt[(32 * t + c) * kHot], (uint32_t)__builtin_popcount(hitmask[t])) : 0u;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int q = 32 * t + c;
                uint32_t at = idx[t];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (hitmask[t] & (1u << i)) {
                        if (at < p.cand_cap) {
                            const uint32_t row = elb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
                            gst(&p.cand[(size_t)q * p.cand_cap + at], ((uint64_t)(uint32_t)esi << 32) | row);
                            gst(&p.cand_s[(size_t)q * p.cand_cap + at], acc[t][i]);
                        }
                        ++at;
                    }
            }
            if (feeds) {
                bool want[NT];
                bool anyw = false;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    want[t] = hitmask[t] && best[t] > key_f32(tauk[t]);  // tau only grows: below it, no effect
                    anyw |= want[t];
                }
                if (anyw) offer_slots<NT>(p, c, want, best);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    };

    auto consume = [&](const float4 (&b)[8]) {
        if (NCH >= 2 && cons.ch == NCH - 2) tau_prefetch();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            f32x8 v = {b[2 * ks].x,     b[2 * ks].y,     b[2 * ks].z,     b[2 * ks].w,
                       b[2 * ks + 1].x, b[2 * ks + 1].y, b[2 * ks + 1].z, b[2 * ks + 1].w};
            const bf16x8 a = __builtin_convertvector(v * sc_cur, bf16x8);
            const int pc = 2 * (cons.ch * 4 + ks) + h;
            const int ph = (pc & ~15) | ((pc ^ c) & 15);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 q8 = *(const bf16x8*)&lq[(32 * t + c) * P8 + ph];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q8, acc[t], 0, 0, 0);
            }
        }
        if (++cons.ch == NCH) {
            epilogue(cons.si, cons.lb);
            enter_block(cons, cons.gb + total_waves);
            sc_cur = sc_next;
        }
    };

    // NBUF-1 chunks of loads are always in flight while one chunk feeds the matrix cores
    // (written out per buffer: every buf[] index must be a literal, or the array moves to scratch)
#define PCV_STEP(REFILL, CONS)          \
    produce(buf[REFILL]);               \
    consume(buf[CONS]);                 \
    if (cons.gb >= p.total_blocks) return;
    produce(buf[0]);
    if constexpr (NBUF == 2) {
        while (true) {
            PCV_STEP(1, 0)
            PCV_STEP(0, 1)
        }
    } else {
        static_assert(NBUF == 3, "2 or 3 chunk buffers");
        produce(buf[1]);
        while (true) {
            PCV_STEP(2, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
        }
    }
#undef PCV_STEP
}

// Exact canonical score of every surviving (query,row) pair: f64, products exact, sums in feature
// order — the same definition as oracle/scan.c:orc_canonical_score.
__global__ __launch_bounds__(256) void rescore_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ float sqr[];  // [Dp] raw query
    const int q = blockIdx.y;
    const int Dp = p.D4 * 4;
    const uint32_t cnt = min(p.cand_cnt[q * kHot], p.cand_cap);
    if (blockIdx.x * 256u >= cnt) return;
    for (int i = threadIdx.x; i < Dp; i += 256) sqr[i] = p.qraw[(size_t)q * Dp + i];
    __syncthreads();
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    if (j >= cnt) return;
    // rows emitted while tau was still low: the final threshold already excludes most of them
    // (same rule as the scan: a row with s < tau - 2*eps cannot reach the final k-th best)
    const float thr_final = key_f32(p.tau[q * kHot]) - p.margin[q];
    if (p.cand_s[(size_t)q * p.cand_cap + j] < thr_final) {
        p.cand_score[(size_t)q * p.cand_cap + j] = __builtin_nan("");
        return;
    }
    const uint64_t e = p.cand[(size_t)q * p.cand_cap + j];
    const SegDesc& sg = p.seg[(int)(e >> 32)];
    const uint32_t row = (uint32_t)e;
    const float4* base = sg.blk + (size_t)(row >> 5) * p.D4 * 32 + (row & 31);
    double dot = 0.0, nx = 0.0;
#pragma unroll 8
    for (int f4 = 0; f4 < p.D4; ++f4) {
        const float4 v = base[(size_t)f4 * 32];
        const float4 qv = *(const float4*)&sqr[f4 * 4];
        dot += (double)qv.x * (double)v.x;
        nx += (double)v.x * (double)v.x;
        dot += (double)qv.y * (double)v.y;
        nx += (double)v.y * (double)v.y;
        dot += (double)qv.z * (double)v.z;
        nx += (double)v.z * (double)v.z;
        dot += (double)qv.w * (double)v.w;
        nx += (double)v.w * (double)v.w;
    }
    const double nq = p.qnorm2[q];
    const double inf = __builtin_inf();
    double score = __builtin_nan("");
    if (p.metric == PCV_METRIC_DOT) {
        if (dot < inf && dot > -inf && nq == nq) score = dot;
    } else if (nq >= 0x1p-126 && nq < inf && nx >= 0x1p-126 && nx < inf) {
        const double cc = dot / (sqrt(nq) * sqrt(nx));
        if (cc < inf && cc > -inf) score = cc;
    }
    p.cand_score[(size_t)q * p.cand_cap + j] = score;
}

// Cooperative form of the rescoring for dim <= 512: a workgroup takes a slice of 1024 survivors of
// one query, compacts the ones the final threshold still admits, then every wave handles 8 of them
// at a time — 8 lanes per row fetch its pieces together (all loads in flight at once) into LDS, and
// one lane per row runs the canonical feature-order f64 sums from there.  Same arithmetic as
// rescore_kernel, ~10x less latency.
constexpr int kCoopMaxD4 = 112;  // LDS: (D4 + 32*(D4+1)) * 16 B <= 64 KB
__global__ __launch_bounds__(256) void rescore_coop_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ float4 lds4[];  // [D4] raw query | 4 waves x 8 slots x (D4+1) row pieces
    __shared__ uint32_t surv[1024];
    __shared__ uint32_t nsurv;
    const int q = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D4 = p.D4;
    const uint32_t cnt = min(p.cand_cnt[q * kHot], p.cand_cap);
    const uint32_t base = blockIdx.x * 1024u;
    if (base >= cnt) return;
    float4* sq = lds4;
    float4* rows = lds4 + D4 + (size_t)wave * 8 * (D4 + 1);
    for (int i = tid; i < D4; i += 256) sq[i] = ((const float4*)(p.qraw + (size_t)q * D4 * 4))[i];
    if (tid == 0) nsurv = 0;
    __syncthreads();
    const float thr_final = key_f32(p.tau[q * kHot]) - p.margin[q];
    for (uint32_t j = base + tid; j < min(cnt, base + 1024u); j += 256) {
        if (p.cand_s[(size_t)q * p.cand_cap + j] < thr_final)
            p.cand_score[(size_t)q * p.cand_cap + j] = __builtin_nan("");
        else
            surv[atomicAdd(&nsurv, 1u)] = j;
    }
    __syncthreads();
    const uint32_t ns = nsurv;
    const int slot = lane >> 3, part = lane & 7;
    const double nq = p.qnorm2[q];
    const double inf = __builtin_inf();
    for (uint32_t g = wave * 8; g < ns; g += 32) {
        const uint32_t si = g + slot;
        const bool live = si < ns;
        const uint32_t j = live ? surv[si] : 0;
        if (live) {
            const uint64_t e = p.cand[(size_t)q * p.cand_cap + j];
            const SegDesc& sg = p.seg[(int)(e >> 32)];
            const uint32_t row = (uint32_t)e;
            const float4* src = sg.blk + (size_t)(row >> 5) * D4 * 32 + (row & 31);
            for (int f4 = part; f4 < D4; f4 += 8) rows[slot * (D4 + 1) + f4] = src[(size_t)f4 * 32];
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0);  // the row pieces of this wave are in LDS
        if (live && part == 0) {
            const float4* r = rows + slot * (D4 + 1);
            double dot = 0.0, nx = 0.0;
            for (int f4 = 0; f4 < D4; ++f4) {
                const float4 v = r[f4];
                const float4 qv = sq[f4];
                dot += (double)qv.x * (double)v.x;
                nx += (double)v.x * (double)v.x;
                dot += (double)qv.y * (double)v.y;
                nx += (double)v.y * (double)v.y;
                dot += (double)qv.z * (double)v.z;
                nx += (double)v.z * (double)v.z;
                dot += (double)qv.w * (double)v.w;
                nx += (double)v.w * (double)v.w;
            }
            double score = __builtin_nan("");
            if (p.metric == PCV_METRIC_DOT) {
                if (dot < inf && dot > -inf && nq == nq) score = dot;
            } else if (nq >= 0x1p-126 && nq < inf && nx >= 0x1p-126 && nx < inf) {
                const double cc = dot / (sqrt(nq) * sqrt(nx));
                if (cc < inf && cc > -inf) score = cc;
            }
            p.cand_score[(size_t)q * p.cand_cap + j] = score;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

struct Best {
    double score;
    int64_t pos;
    uint32_t idx;
};
__device__ __forceinline__ bool better(double sa, int64_t pa, double sb, int64_t pb) {
    return sa > sb || (sa == sb && pa < pb);
}

// Final ranking of the rescored survivors: descending canonical score, ties -> lower global
// position.  One workgroup per query.  The valid survivors (normally a few dozen) are compacted
// into LDS, then k rounds of argmax run there; lists that do not fit fall back to global memory.
constexpr int kSelCap = 1024;
__global__ __launch_bounds__(256) void select_kernel(const ScanParams* __restrict__ pp, pcv_hit_dev* __restrict__ out) {
    const ScanParams& p = *pp;
    __shared__ double c_s[kSelCap];
    __shared__ int64_t c_p[kSelCap];
    __shared__ uint32_t c_i[kSelCap];
    __shared__ double r_s[4];
    __shared__ int64_t r_p[4];
    __shared__ uint32_t r_i[4];
    __shared__ uint32_t n_valid;
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t cnt = min(p.cand_cnt[q * kHot], p.cand_cap);
    const uint64_t* cand = p.cand + (size_t)q * p.cand_cap;
    double* sc = p.cand_score + (size_t)q * p.cand_cap;
    if (tid == 0) {
        n_valid = 0;
        p.cand_cnt_out[q] = p.cand_cnt[q * kHot];  // uncapped: the host sizes a rerun from it
    }
    __syncthreads();
    for (uint32_t i = tid; i < cnt; i += 256) {
        const double s = sc[i];
        if (!(s == s)) continue;  // NaN: undefined score or ruled out
        const uint32_t slot = atomicAdd(&n_valid, 1u);
        if (slot < (uint32_t)kSelCap) {
            const uint64_t e = cand[i];
            c_s[slot] = s;
            c_p[slot] = p.seg[(int)(e >> 32)].pos0 + (int64_t)(uint32_t)e;
            c_i[slot] = i;
        }
    }
    __syncthreads();
    const uint32_t nv = n_valid;
    const bool in_lds = nv <= (uint32_t)kSelCap;
    const uint32_t n = in_lds ? nv : cnt;
    for (int j = 0; j < p.k; ++j) {
        double bs = -__builtin_inf();
        int64_t bp = INT64_MAX;
        uint32_t bi = 0xffffffffu;  // index into the LDS list (in_lds) or the global list
        for (uint32_t i = tid; i < n; i += 256) {
            double s;
            int64_t pos;
            if (in_lds) {
                s = c_s[i];
                pos = c_p[i];
            } else {
                s = sc[i];
                const uint64_t e = cand[i];
                pos = p.seg[(int)(e >> 32)].pos0 + (int64_t)(uint32_t)e;
            }
            if (!(s == s)) continue;
            if (bi == 0xffffffffu || better(s, pos, bs, bp)) {
                bs = s;
                bp = pos;
                bi = i;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double os = __shfl_xor(bs, off);
            const int64_t op = __shfl_xor(bp, off);
            const uint32_t oi = __shfl_xor(bi, off);
            if (oi != 0xffffffffu && (bi == 0xffffffffu || better(os, op, bs, bp))) {
                bs = os;
                bp = op;
                bi = oi;
            }
        }
        if (lane == 0) {
            r_s[wave] = bs;
            r_p[wave] = bp;
            r_i[wave] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (r_i[w] != 0xffffffffu && (bi == 0xffffffffu || better(r_s[w], r_p[w], bs, bp))) {
                    bs = r_s[w];
                    bp = r_p[w];
                    bi = r_i[w];
                }
            pcv_hit_dev hit;
            if (bi != 0xffffffffu) {
                const uint32_t gi = in_lds ? c_i[bi] : bi;
                const uint64_t e = cand[gi];
                const SegDesc& sg = p.seg[(int)(e >> 32)];
                const uint32_t row = (uint32_t)e;
                hit.score = bs;
                hit.pos = bp;
                hit.id = sg.ids ? sg.ids[row] : sg.id0 + (int64_t)row;
                if (in_lds)
                    c_s[bi] = __builtin_nan("");
                else
                    sc[bi] = __builtin_nan("");
            } else {
                hit.score = __builtin_nan("");
                hit.pos = -1;
                hit.id = -1;
            }
            out[(size_t)q * p.k + j] = hit;
        }
        __syncthreads();
    }
}

// One record after a shard's [B][k] hits says whether any of its candidate lists overflowed (the
// pass then has to be repeated with larger lists): pos = 1 | 0.  It travels with the hits through the
// all-gather so that every rank takes the same decision without a second collective.
__global__ __launch_bounds__(64) void overflow_flag_kernel(const uint32_t* __restrict__ cnt, int B, uint32_t cap,
                                                           pcv_hit_dev* __restrict__ rec) {
    bool over = false;
    for (int q = threadIdx.x; q < B; q += 64) over |= cnt[q] > cap;
    over = __any(over);
    if (threadIdx.x == 0) {
        rec->score = 0.0;
        rec->pos = over ? 1 : 0;
        rec->id = 0;
    }
}

// merge of per-shard top-k lists after the all-gather: [n_shards][B][k] -> [B][k]; shards are `stride`
// records apart (B*k, or B*k+1 with the overflow record, whose OR then lands in out[B*k])
__global__ __launch_bounds__(64) void merge_kernel(const pcv_hit_dev* __restrict__ lists_all, int n_shards, int B, int k,
                                                   size_t stride, int flagged, pcv_hit_dev* __restrict__ out) {
    extern __shared__ unsigned char taken[];  // [n_shards*k]
    const int q = blockIdx.x, lane = threadIdx.x;
    const int total = n_shards * k;
    if (flagged && q == 0 && lane == 0) {
        int64_t any = 0;
        for (int sh = 0; sh < n_shards; ++sh) any |= lists_all[sh * stride + (size_t)B * k].pos;
        pcv_hit_dev f;
        f.score = 0.0;
        f.pos = any ? 1 : 0;
        f.id = 0;
        out[(size_t)B * k] = f;
    }
    for (int i = lane; i < total; i += 64) taken[i] = 0;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        double bs = 0;
        int64_t bp = 0;
        int bi = -1;
        for (int i = lane; i < total; i += 64) {
            if (taken[i]) continue;
            const pcv_hit_dev& e = lists_all[(size_t)(i / k) * stride + (size_t)q * k + (i % k)];
            if (e.pos < 0 || !(e.score == e.score)) continue;
            if (bi < 0 || better(e.score, e.pos, bs, bp)) {
                bs = e.score;
                bp = e.pos;
                bi = i;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double os = __shfl_xor(bs, off);
            const int64_t op = __shfl_xor(bp, off);
            const int oi = __shfl_xor(bi, off);
            if (oi >= 0 && (bi < 0 || better(os, op, bs, bp))) {
                bs = os;
                bp = op;
                bi = oi;
            }
        }
        if (lane == 0) {
            pcv_hit_dev hit;
            if (bi >= 0) {
                hit = lists_all[(size_t)(bi / k) * stride + (size_t)q * k + (bi % k)];
                taken[bi] = 1;
            } else {
                hit.score = __builtin_nan("");
                hit.pos = -1;
                hit.id = -1;
            }
            out[(size_t)q * k + j] = hit;
        }
        __syncthreads();
    }
}

// lib.rs:63-77 as a plain [B][N] matrix for small inputs (highlight.rs:109, tests): f32.
__global__ __launch_bounds__(256) void similarity_matrix_kernel(const float* __restrict__ a, int B,
                                                                const float* __restrict__ m, int64_t N, int D,
                                                                int cosine, float* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float* q = a + (size_t)b * D;
    const float* x = m + (size_t)n * D;
    float dot = 0.0f, nq = 0.0f, nx = 0.0f;
    for (int i = 0; i < D; ++i) {
        dot = fmaf(q[i], x[i], dot);
        nq = fmaf(q[i], q[i], nq);
        nx = fmaf(x[i], x[i], nx);
    }
    out[(size_t)b * N + n] = cosine ? dot / (sqrtf(nq) * sqrtf(nx)) : dot;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------

static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

void launch_pack_rows(hipStream_t st, const float* rows, int64_t n, int D, int D4, float4* blk, uint32_t nblocks,
                      uint32_t row0) {
    if (n <= 0) return;
    const uint32_t first_blk = row0 >> 5;
    const uint32_t last_blk = (uint32_t)((row0 + n - 1) >> 5);
    const int64_t threads = (int64_t)(last_blk - first_blk + 1) * D4 * 32;
    if (threads > (int64_t)0xffffff00u) abort();  // callers stage <= 2^18 rows per call
    pack_rows_kernel<<<cdiv64(threads, 256), 256, 0, st>>>(rows, n, D, D4, blk, row0);
}

void launch_row_scales(hipStream_t st, const float4* blk, uint32_t nblocks, uint32_t nrows, int D4, int metric,
                       float* scale, uint32_t* max_norm_bits) {
    if (nblocks == 0) return;
    row_scales_kernel<<<cdiv64((int64_t)nblocks * 32, 256), 256, 0, st>>>(blk, nblocks, nrows, D4, metric, scale,
                                                                           max_norm_bits);
}

void launch_synth_fill(hipStream_t st, float4* blk, uint32_t nblocks, uint32_t nrows, uint32_t row0, int D, int D4,
                       uint64_t seed, int64_t first_row, int normalize) {
    if (nrows == 0) return;
    const int D4src = D / 4;
    float* inv = nullptr;
    if (normalize) {
        hipMallocAsync((void**)&inv, (size_t)nrows * sizeof(float), st);
        synth_inv_kernel<<<cdiv64(nrows, 256), 256, 0, st>>>(nrows, D4src, seed, first_row, inv);
    }
    const uint32_t first_blk = row0 >> 5;
    const uint32_t last_blk = (uint32_t)(((uint64_t)row0 + nrows - 1) >> 5);
    const int64_t threads = (int64_t)(last_blk - first_blk + 1) * D4src * 32;
    const unsigned grid = (unsigned)std::min<int64_t>((threads + 255) / 256, 1 << 20);
    synth_fill_kernel<<<grid, 256, 0, st>>>(blk, nrows, row0, D4src, D4, seed, first_row, inv, threads);
    if (inv) hipFreeAsync(inv, st);
}

void launch_gather_rows(hipStream_t st, const SegDesc* d_segs, int nseg, const int64_t* d_pos, int64_t n, int D,
                        int D4, float* out_rows, int64_t* out_ids) {
    if (n <= 0) return;
    const int64_t threads = n * ((D + 3) / 4);
    gather_rows_kernel<<<cdiv64(threads, 256), 256, 0, st>>>(d_segs, nseg, d_pos, n, D, D4, out_rows, out_ids);
}

void launch_prep_queries(hipStream_t st, const float* d_queries, int B, int D, int Dp, int metric, float eps_rel,
                         float max_norm, int k, float* qf32, uint16_t* qbf16, float* qraw, double* qnorm2,
                         float* margin, uint32_t* tau, uint32_t* slots, uint32_t* cand_cnt) {
    const int grid = B > kMfmaQueries ? B : kMfmaQueries;
    prep_queries_kernel<<<grid, 64, (size_t)Dp * sizeof(float), st>>>(d_queries, B, D, Dp, metric, eps_rel, max_norm, k, qf32, qbf16, qraw,
                                             qnorm2, margin, tau, slots, cand_cnt);
}

void launch_seed(hipStream_t st, const ScanParams& p, const ScanParams* dp) {
    if (p.seed_blocks == 0 || p.nseg == 0) return;
    const int nparts = (int)((p.seed_blocks * 32u + kSeedPartRows - 1) / kSeedPartRows);
    if (p.B <= 2) {
        const size_t lds = (size_t)p.D4 * 4 * sizeof(float) + kSeedPartRows * sizeof(uint32_t);
        seed_partial_kernel<1><<<dim3(nparts, p.B), 256, lds, st>>>(dp);
    } else {
        const size_t lds = 8 * ((size_t)p.D4 * 4 * sizeof(float) + kSeedPartRows * sizeof(uint32_t));
        seed_partial_kernel<8><<<dim3(nparts, (p.B + 7) / 8), 256, lds, st>>>(dp);
    }
    seed_merge_kernel<<<p.B, 256, 0, st>>>(dp, nparts);
}

void launch_scan_wave(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus) {
    if (p.total_blocks == 0) return;
    const size_t lds = (size_t)p.B * p.D4 * 4 * sizeof(float);
    const unsigned gm = (p.flags >> 8) & 0xff;
    unsigned grid = (unsigned)num_cus * (gm ? gm : 8);
    const unsigned need = (p.total_blocks + 3) / 4;
    if (grid > need) grid = need;
    const bool ntl = (p.flags & 1) == 0;  // non-temporal corpus loads unless flag bit 0 is set
#define PCV_WAVE(NB)                                                   \
    if (ntl)                                                           \
        scan_wave_kernel<NB, true><<<grid, 256, lds, st>>>(dp);         \
    else                                                               \
        scan_wave_kernel<NB, false><<<grid, 256, lds, st>>>(dp);
    switch (p.B) {
        case 1: PCV_WAVE(1); break;
        case 2: PCV_WAVE(2); break;
        case 3: PCV_WAVE(3); break;
        default: PCV_WAVE(4); break;
    }
#undef PCV_WAVE
}

// Largest query count one MFMA pass can take at this padded dim: the bf16 query tile
// (32*NT rows x Dp) must fit the 160 KB LDS of a CU.  0: the tile does not fit at all.
int mfma_pass_queries(int Dp) {
    for (int nt : {4, 2, 1})
        if ((size_t)nt * 32 * Dp * 2 <= 156 * 1024) return nt * 32;
    return 0;
}

template <int NT, bool NTL, int WPB, int NBUF>
static void launch_mfma_variant(hipStream_t st, const ScanParams* dp, unsigned grid, size_t lds) {
    if (lds > 64 * 1024) {
        static bool attr_set = false;  // one flag per instantiation
        if (!attr_set) {
            hipFuncSetAttribute((const void*)scan_mfma_kernel<NT, NTL, WPB, NBUF>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
    }
    scan_mfma_kernel<NT, NTL, WPB, NBUF><<<grid, WPB * 64, lds, st>>>(dp);
}

void launch_scan_mfma(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus) {
    if (p.total_blocks == 0) return;
    const int NT = p.B <= 32 ? 1 : (p.B <= 64 ? 2 : 4);
    const size_t lds = (size_t)NT * 32 * p.D4 * 4 * sizeof(uint16_t);
    const unsigned gm = (p.flags >> 8) & 0xff;
    const bool ntl = (p.flags & 1) == 0;   // non-temporal corpus loads unless flag bit 0 is set
    // small tiles: 256-thread workgroups, 3 per CU.  Tiles too big for that (B > 64, or dim > ~440):
    // 512-thread workgroups sharing one tile, as many per CU as the LDS holds.
    const bool wide = NT == 4 || lds * 3 > 156 * 1024;
    const unsigned wpb = wide ? 8 : 4;
    const unsigned per_cu = wide ? (unsigned)std::max<size_t>(1, std::min<size_t>(2, (156 * 1024) / lds)) : 3;
    unsigned grid = (unsigned)num_cus * (gm ? gm : per_cu);
    const unsigned need = (p.total_blocks + wpb - 1) / wpb;
    if (grid > need) grid = need;
#define PCV_MFMA(NT_, NTL_)                                                      \
    if (wide)                                                                    \
        launch_mfma_variant<NT_, NTL_, 8, 3>(st, dp, grid, lds);                 \
    else                                                                         \
        launch_mfma_variant<NT_, NTL_, 4, 2>(st, dp, grid, lds);
    if (NT == 1) {
        if (ntl) { PCV_MFMA(1, true) } else { PCV_MFMA(1, false) }
    } else if (NT == 2) {
        if (ntl) { PCV_MFMA(2, true) } else { PCV_MFMA(2, false) }
    } else {
        if (ntl) { launch_mfma_variant<4, true, 8, 3>(st, dp, grid, lds); } else { launch_mfma_variant<4, false, 8, 3>(st, dp, grid, lds); }
    }
#undef PCV_MFMA
}

void launch_rescore(hipStream_t st, const ScanParams& p, const ScanParams* dp) {
    if (p.D4 <= kCoopMaxD4) {
        const size_t lds = ((size_t)p.D4 + 4 * 8 * ((size_t)p.D4 + 1)) * sizeof(float4);
        dim3 grid((p.cand_cap + 1023) / 1024, p.B);
        rescore_coop_kernel<<<grid, 256, lds, st>>>(dp);
        return;
    }
    const size_t lds = (size_t)p.D4 * 4 * sizeof(float);
    dim3 grid((p.cand_cap + 255) / 256, p.B);
    rescore_kernel<<<grid, 256, lds, st>>>(dp);
}

void launch_select(hipStream_t st, const ScanParams& p, const ScanParams* dp, pcv_hit_dev* out) {
    select_kernel<<<p.B, 256, 0, st>>>(dp, out);
}

void launch_merge(hipStream_t st, const pcv_hit_dev* lists, int n_shards, int B, int k, pcv_hit_dev* out, int flagged) {
    const size_t stride = (size_t)B * k + (flagged ? 1 : 0);
    merge_kernel<<<B, 64, (size_t)n_shards * k, st>>>(lists, n_shards, B, k, stride, flagged, out);
}

void launch_overflow_flag(hipStream_t st, const uint32_t* cnt, int B, uint32_t cap, pcv_hit_dev* rec) {
    overflow_flag_kernel<<<1, 64, 0, st>>>(cnt, B, cap, rec);
}

void launch_similarity_matrix(hipStream_t st, const float* a, int B, const float* m, int64_t N, int D, int cosine,
                              float* out) {
    if (N <= 0 || B <= 0) return;
    dim3 grid(cdiv64(N, 256), B);
    similarity_matrix_kernel<<<grid, 256, 0, st>>>(a, B, m, N, D, cosine, out);
}

}  // namespace pcv
