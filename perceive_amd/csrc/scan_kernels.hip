// scan_kernels.hip — gfx950 kernels of the exact similarity scan (replaces lib.rs:63-77 +
// search.rs:157-182 of the reference).
//
// Pipeline of one search: ONE host->device copy (parameters, segment table, queries) and three
// kernels on the context stream, results written straight into pinned host memory:
//   prep_seed  ->  scan (wave | mfma)  ->  rescore_select
// The scan is a *screening* pass: it streams the corpus once and keeps, per query, the running k-th
// best f32 score tau (k DISTINCT rows' f32 scores live in slots[q][0..k), tau <= min(slots) and only
// grows).  With |s32 - c| <= eps32 and |s16 - c| <= eps16 bounding the f32 / bf16 screening scores
// against the canonical f64 score c:
//   coarse test (MFMA kernel, every row):   drop iff s16 < tau - (eps16 + eps32)
//   fine test   (f32 score of a survivor):  drop iff s32 < tau - 2*eps32
// Either way c < tau - eps32 <= c_j for each of the k slot rows j, so a dropped row cannot be in the
// exact top-k however stale tau is.  rescore_select ranks the few survivors per query in exact f64.
// HBM traffic = one pass over the rows (+ one extra row read per coarse survivor).
#include <algorithm>
#include <cstdlib>
#include <numeric>
#include <type_traits>

#include "common.h"
#include "scan.h"
#include "synth.h"

// Timing experiments of round 4 (tools/exp_build.sh <n> on the GPU box; results are WRONG in builds 7-9): 7 = the rows of 256 blocks over
// and over (they come out of the L2), 8 = every block ends at its test (no survivor is handled), 9 = both, 12 = the thresholds
// fetched by every wave beside the U from LDS (what the drain wave's lag costs in survivors).
#ifndef PCV_EXP
#define PCV_EXP 0
#endif
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
__device__ __forceinline__ uint32_t ld_relaxed(const uint32_t* p) {
    return __hip_atomic_load((const PCV_GLOBAL uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_relaxed(uint32_t* p, uint32_t v) {
    __hip_atomic_store((PCV_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Streamed row chunks are read through a buffer descriptor (four scalar registers: the block's base, wave-uniform) with the
// lane's own byte offset — one vector register that never changes — and a scalar chunk offset: no 64-bit address arithmetic
// in vector registers.  (The compiler did that arithmetic in the registers of a chunk buffer; overwriting a register that a
// load may still be writing costs an s_waitcnt vmcnt(0), i.e. every chunk in flight, per block.)  Loads past `bytes` return 0.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void* ubase, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)ubase, 0, bytes, 0x00020000);
}
template <bool NTL>
__device__ __forceinline__ float4 ld_piece(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_bytes, uint32_t chunk_bytes) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_bytes, chunk_bytes, NTL ? 2 : 0);  // aux 2 = nt
    return __builtin_bit_cast(float4, v);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int off) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off), hi = __shfl_xor((uint32_t)(v >> 32), off);
    return ((unsigned long long)hi << 32) | lo;
}

// Position of a wave in the launch's segment table.  A wave visits launch-wide block indices in
// ascending order, so the segment only ever moves forward: the common step (same segment) costs a
// compare against `end`; a segment change is one binary search over the blk0 column of the table.
struct SegCursor {
    int si = -1;
    uint32_t begin = 0, end = 0;  // launch-wide block range of segment si
    const float4* blk = nullptr;
    const float* scale = nullptr;
    const uint4* blk16 = nullptr;
    const uint4* blk8 = nullptr;
    const float* scale8 = nullptr;
    const uint4* mid16 = nullptr;
    const float* scale16 = nullptr;
};
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ const T* uniform_ptr(const T* ptr) {  // a wave-uniform pointer, moved to scalar registers
    const uint64_t v = (uint64_t)ptr;
    return (const T*)(((uint64_t)uniform((uint32_t)(v >> 32)) << 32) | uniform((uint32_t)v));
}
__device__ __forceinline__ void seek_seg(const ScanParams& p, SegCursor& c, uint32_t gb) {
    if (gb < c.end) return;
    int lo = c.si + 1, hi = p.nseg - 1;  // last table entry with blk0 <= gb
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (gld(&p.seg[mid].blk0) <= gb)
            lo = mid;
        else
            hi = mid - 1;
    }
    // The fields come back through v_readfirstlane: the wait for these loads then sits HERE, on the rare path.  Left in vector
    // registers, the compiler has to wait at the point where the two paths meet — an s_waitcnt vmcnt(0) in front of the first
    // use of a pointer, at every block, which drains the row chunks in flight (and the cursor took ten vector registers).
    c.si = lo;
    c.begin = uniform(gld(&p.seg[lo].blk0));
    c.end = c.begin + uniform(gld(&p.seg[lo].nblocks));
    c.blk = uniform_ptr(gld(&p.seg[lo].blk));
    c.scale = uniform_ptr(gld(&p.seg[lo].scale));
    c.blk16 = uniform_ptr(gld(&p.seg[lo].blk16));
    c.blk8 = uniform_ptr(gld(&p.seg[lo].blk8));
    c.scale8 = uniform_ptr(gld(&p.seg[lo].scale8));
    c.mid16 = uniform_ptr(gld(&p.seg[lo].mid16));
    c.scale16 = uniform_ptr(gld(&p.seg[lo].scale16));
}

// slots[q][0..k) always hold f32 scores of k DISTINCT rows (or -inf), each slot only ever grows, so
// min(slots) is a valid lower bound of the final k-th best f32 score.
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
            g_atomic_max(&p.tau_c[q], nm);
            return;
        }
    }
}

// The same with the whole wave working on ONE (query, score): the k slots are read lane-parallel, so an
// offer is three dependent round trips whatever k is.  All 64 lanes must call it with the same arguments.
__device__ __forceinline__ void offer_slot_wave(const ScanParams& p, int q, float s, int lane) {
    const uint32_t key = f32_key(s);
    uint32_t* sl = p.slots + (size_t)q * kMaxK;
    for (int attempt = 0; attempt < 8; ++attempt) {
        unsigned long long best = ~0ull;  // (key << 32 | slot), minimum
        for (int i = lane; i < p.k; i += 64) {
            const unsigned long long c = ((unsigned long long)ld_relaxed(&sl[i]) << 32) | (uint32_t)i;
            best = c < best ? c : best;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = shfl_xor_u64(best, off);
            best = o < best ? o : best;
        }
        const uint32_t mn = (uint32_t)(best >> 32), mi = (uint32_t)best;
        if (key <= mn) return;
        uint32_t old = 0;
        if (lane == 0) old = g_atomic_cas(&sl[mi], mn, key);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == mn) {
            uint32_t nm = 0xffffffffu;
            for (int i = lane; i < p.k; i += 64) nm = min(nm, ld_relaxed(&sl[i]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nm = min(nm, (uint32_t)__shfl_xor(nm, off));
            if (lane == 0) {
                g_atomic_max(&p.tau[q * kHot], nm);
                g_atomic_max(&p.tau_c[q], nm);
            }
            return;
        }
    }
}

// Where a row with screening score s stands to the pass's ceiling for query q (scan.h, CeilRec): 0 = it counts and may raise
// the thresholds, 1 = it counts but must not raise them, 2 = it does not count.  No ceiling: 0.
__device__ __forceinline__ int ceil_class(const ScanParams& p, int q, float s) {
    if (!p.ceil) return 0;
    const float lo = gld(&p.ceil[q].lo), hi = gld(&p.ceil[q].hi);
    return s < lo ? 0 : (s > hi ? 2 : 1);
}

// The same by a group of eight lanes (lane `sub` of 8; the eight groups of a wave offer eight different (query, score) pairs at
// once): the drain wave of scan_mfma8_kernel's DRAIN form raises the thresholds of up to eight queries in the three round trips
// one offer takes.  `active`: this group has an offer (all eight lanes agree); lanes of idle groups do nothing.
__device__ __forceinline__ void offer_slot_g8(const ScanParams& p, bool active, int q, float s, int sub) {
    const uint32_t key = f32_key(s);
    uint32_t* sl = p.slots + (size_t)q * kMaxK;
    for (int attempt = 0; attempt < 8 && __any(active); ++attempt) {
        unsigned long long best = ~0ull;  // (key << 32 | slot), minimum
        if (active)
            for (int i = sub; i < p.k; i += 8) {
                const unsigned long long c = ((unsigned long long)ld_relaxed(&sl[i]) << 32) | (uint32_t)i;
                best = c < best ? c : best;
            }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            const unsigned long long o = shfl_xor_u64(best, off);
            best = o < best ? o : best;
        }
        const uint32_t mn = (uint32_t)(best >> 32), mi = (uint32_t)best;
        if (active && key <= mn) active = false;  // k rows at or above it are there already
        uint32_t old = 0;
        if (active && sub == 0) old = g_atomic_cas(&sl[mi], mn, key);
        old = (uint32_t)__shfl((int)old, (threadIdx.x & 63) & ~7);
        if (active && old == mn) {  // the slot is ours: the new minimum is the threshold
            uint32_t nm = 0xffffffffu;
            for (int i = sub; i < p.k; i += 8) nm = min(nm, ld_relaxed(&sl[i]));
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) nm = min(nm, (uint32_t)__shfl_xor((int)nm, off));
            if (sub == 0) {
                g_atomic_max(&p.tau[q * kHot], nm);
                g_atomic_max(&p.tau_c[q], nm);
            }
            active = false;
        }
    }
}

// A surviving (query,row) pair of the wave kernel: append to the query's candidate list and, unless the
// row was already ranked by the seed kernel, try to raise the running k-th best.
__device__ __noinline__ void emit_hit(const ScanParams& p, int q, int seg, uint32_t row, float s,
                                      bool feeds_slots) {
    const int cc = ceil_class(p, q, s);
    if (cc == 2) return;
    feeds_slots = feeds_slots && cc == 0;
    uint32_t idx = g_atomic_add(&p.cand_cnt[q * kHot], 1u);
    if (idx < p.cand_cap) {
        gst(&p.cand[(size_t)q * p.cand_cap + idx], ((uint64_t)(uint32_t)seg << 32) | row);
        gst(&p.cand_s[(size_t)q * p.cand_cap + idx], s);
    }
    if (feeds_slots && isfinite(s)) offer_slot(p, q, s);
}

// ------------------------------------------------------------------------------------------------
// ingestion-time kernels
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

__global__ __launch_bounds__(256) void iota_ids_kernel(int64_t* __restrict__ ids, int64_t first, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) ids[i] = first + i;
}

// per-row scale = 1/|x| (cosine) or 1 (dot); 0 marks rows that can never be a result
// (padding, zero / non-finite norm).  |x|^2 accumulated in f64 in feature order.
__global__ __launch_bounds__(256) void row_scales_kernel(const float4* __restrict__ blk, uint32_t row_begin,
                                                         uint32_t row_end, uint32_t nrows, int D4, int metric,
                                                         float* __restrict__ scale, uint32_t* max_norm_bits) {
    const uint32_t row = row_begin + blockIdx.x * 256 + threadIdx.x;
    if (row >= row_end) return;
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

// Screening copy of blocks [first_block, nblocks): piece f8 of row r = bf16(RNE) of features 8*f8..8*f8+7 times the
// row's scale; rows that are not searchable (scale 0: padding, bad norm) become exact zeros.
__global__ __launch_bounds__(256) void coarse_pack_kernel(const float4* __restrict__ blk, const float* __restrict__ scale,
                                                          uint4* __restrict__ blk16, uint32_t first_block, uint32_t nblocks, int D4) {
    const int D8 = D4 >> 1;
    const size_t per_block = (size_t)D8 * 32;
    const size_t total = (size_t)(nblocks - first_block) * per_block;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const uint32_t b = first_block + (uint32_t)(i / per_block);
        const uint32_t rem = (uint32_t)(i % per_block);
        const uint32_t f8 = rem >> 5, r = rem & 31;
        const float sc = scale[(size_t)b * 32 + r];
        uint4 out = make_uint4(0, 0, 0, 0);
        if (sc != 0.0f) {
            const float4 lo = blk[((size_t)b * D4 + 2 * f8) * 32 + r], hi = blk[((size_t)b * D4 + 2 * f8 + 1) * 32 + r];
            const f32x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            out = __builtin_bit_cast(uint4, __builtin_convertvector(v * sc, bf16x8));  // the conversion the f32 scan kernel does
        }
        blk16[((size_t)b * D8 + f8) * 32 + r] = out;
    }
}


// Int8 screening copy of blocks [first_block, nblocks), one wave per block: lane (r, h) owns row r and every
// other 16-feature piece.  Pass 1: max |y_i| over the BLOCK's searchable rows (y = x * scale); pass 2: x^_i = rint(y_i * s_blk),
// s_blk = 127 / max: one scale for the 32 rows, so the scan's test of a block needs one float and its right-hand side is the same
// for every row (per row scales made the block pre-test loose — the smallest scale of 16 rows stood for all of them, a third
// of the blocks of a 12.5M-row pass went on to the row-by-row test at ~2 us each — and cost 144 B per block).  A row
// quantised with a smaller scale than its own 127 / max|y_i| keeps |x^_i| <= 127 and |y_i - x^_i / s| <= 0.5 / s: the bound of
// scan.h holds with s = s_blk; on Gaussian rows the margin grows by ~7 %.  Rows that are not searchable are stored as zeros (they
// reach the fine screen only under a non-positive right-hand side and end there: scale 0); a block without a searchable row
// gets s_blk = NaN (no comparison succeeds); all-zero searchable rows (dot metric) alone: s_blk = 1.
__global__ __launch_bounds__(256) void coarse_pack8_kernel(const float4* __restrict__ blk, const float* __restrict__ scale,
                                                           uint4* __restrict__ blk8, float* __restrict__ scale8, uint32_t first_block,
                                                           uint32_t nblocks, int D4) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int D16 = ((D4 * 4 + 127) & ~127) >> 4;
    for (uint32_t b = first_block + blockIdx.x * 4 + (threadIdx.x >> 6); b < nblocks; b += gridDim.x * 4) {
        const float sc = scale[(size_t)b * 32 + r];
        const float4* src = blk + (size_t)b * D4 * 32 + r;
        float mx = 0.0f;
        for (int g = h; g < D16; g += 2)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * g + e < D4) {
                    const float4 v = src[(size_t)(4 * g + e) * 32];
                    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(v.x * sc), fabsf(v.y * sc)), fmaxf(fabsf(v.z * sc), fabsf(v.w * sc))));
                }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const bool searchable = sc != 0.0f && mx < __builtin_inff();  // (NaN features: fmaxf ignores them; such rows have scale 0)
        float bm = searchable ? mx : 0.0f;
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) bm = fmaxf(bm, __shfl_xor(bm, off));
        const bool any_row = __any(searchable);
        const float s_blk = !any_row ? __builtin_nanf("") : (bm > 0.0f ? 127.0f / bm : 1.0f);
        for (int g = h; g < D16; g += 2) {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (searchable && 4 * g + e < D4) {
                    const float4 v = src[(size_t)(4 * g + e) * 32];
                    const float y[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int qv = max(-127, min(127, (int)rintf(y[j] * s_blk)));
                        w[e] |= (uint32_t)(qv & 0xff) << (8 * j);
                    }
                }
            blk8[((size_t)b * D16 + g) * 32 + r] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (lane == 0) scale8[b] = s_blk;
    }
}


// Mid copy of rows [first_row, nrows) (scan.h): one wave per row at a time; lane j < Dp/8 owns the 8 features 8j..8j+7 (two
// pieces of the blocked row), the maximum goes round the wave, the 16 bytes go out as part of the row's Dp * 2 contiguous ones.
__global__ __launch_bounds__(256) void mid_pack_kernel(const float4* __restrict__ blk, const float* __restrict__ scale, uint4* __restrict__ mid16,
                                                       float* __restrict__ scale16, uint32_t first_row, uint32_t nrows, int D4) {
    const int lane = threadIdx.x & 63;
    const int P8 = D4 >> 1;  // 16-byte pieces of a mid row
    for (uint32_t row = first_row + blockIdx.x * 4 + (threadIdx.x >> 6); row < nrows; row += gridDim.x * 4) {
        const float sc = scale[row];
        const float4* src = blk + (size_t)(row >> 5) * D4 * 32 + (row & 31);
        float mx = 0.0f;
        for (int j = lane; j < P8; j += 64) {
            const float4 a = src[(size_t)(2 * j) * 32], b = src[(size_t)(2 * j + 1) * 32];
            mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(a.x * sc), fabsf(a.y * sc)), fmaxf(fabsf(a.z * sc), fabsf(a.w * sc))),
                                 fmaxf(fmaxf(fabsf(b.x * sc), fabsf(b.y * sc)), fmaxf(fabsf(b.z * sc), fabsf(b.w * sc)))));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const bool searchable = sc != 0.0f && mx < __builtin_inff();
        const float s2 = !searchable ? 0.0f : (mx > 0.0f ? 32766.0f / mx : 1.0f);
        for (int j = lane; j < P8; j += 64) {
            const float4 a = src[(size_t)(2 * j) * 32], b = src[(size_t)(2 * j + 1) * 32];
            const float y[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lo = searchable ? max(-32767, min(32767, (int)rintf(y[2 * e] * s2))) : 0;
                const int hi = searchable ? max(-32767, min(32767, (int)rintf(y[2 * e + 1] * s2))) : 0;
                w[e] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
            }
            mid16[(size_t)row * P8 + j] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        if (lane == 0) scale16[row] = searchable ? s2 : __builtin_nanf("");
    }
}

// The same copy for segments that have their int8 screening copy (the normal case), one wave per 32-row BLOCK: the block's
// pieces are read as the scan reads them — two contiguous 512-byte runs per instruction, every row once — quantised with ONE
// scale for the block, s2 = scale8[b] * 32766 / 127 (the int8 copy's block maximum: any s2 <= 32766 / max|y_i| of a row keeps
// |y_i - Y_i / s2| <= 0.5 / s2, which is all the mid screen's bound uses; no pass for the maximum), turned row-major in LDS
// and written as the block's Dp * 64 contiguous bytes.  mid_pack_kernel above gathers 16-byte pieces 512 bytes apart, twice per
// row: 1.0 TB/s, 228 ms per 100M x 384 rows (profiles/r03_batch256_kernel_stats.csv).
__global__ __launch_bounds__(64) void mid_pack_block_kernel(const float4* __restrict__ blk, const float* __restrict__ scale,
                                                            const float* __restrict__ scale8, uint4* __restrict__ mid16,
                                                            float* __restrict__ scale16, uint32_t first_block, uint32_t nblocks, int D4) {
    extern __shared__ uint2 mtile[];  // [32][D4 + 2]: a row's 8-byte pieces, two of padding (rows stay 16-byte aligned)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int LD = D4 + 2, P2 = D4 >> 1;
    for (uint32_t b = first_block + blockIdx.x; b < nblocks; b += gridDim.x) {
        const float sc = scale[(size_t)b * 32 + r];
        const float s2 = scale8[b] * (32766.0f / 127.0f);  // NaN: a block without a searchable row
        const bool searchable = sc != 0.0f && s2 == s2;
        const float4* src = blk + (size_t)b * D4 * 32 + h * 32 + r;  // piece 2j + h of row r: src + 64 j
        for (int j0 = 0; j0 < P2; j0 += 16) {  // 16 KB of a block requested at a time (six waves a CU at 384-d: ~100 KB in flight)
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + u < P2) v[u] = ld_row<true>(src + (size_t)(j0 + u) * 64);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (j0 + u < P2) {
                    const float y[4] = {v[u].x * sc, v[u].y * sc, v[u].z * sc, v[u].w * sc};
                    int q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = searchable ? max(-32767, min(32767, (int)rintf(y[e] * s2))) : 0;
                    mtile[r * LD + 2 * (j0 + u) + h] =
                        make_uint2(((uint32_t)q[0] & 0xffffu) | ((uint32_t)q[1] << 16), ((uint32_t)q[2] & 0xffffu) | ((uint32_t)q[3] << 16));
                }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is written (one wave: nobody else touches it)
        __builtin_amdgcn_wave_barrier();
        uint4* dst = mid16 + (size_t)b * 32 * P2;
        for (int i = lane; i < 32 * P2; i += 64) {
            const int row = i / P2, pc = i - row * P2;
            dst[i] = *(const uint4*)&mtile[row * LD + 2 * pc];
        }
        if (lane < 32) scale16[(size_t)b * 32 + lane] = searchable ? s2 : __builtin_nanf("");
        __builtin_amdgcn_s_waitcnt(0xc07f);  // the reads are done before the next block overwrites the tile
        __builtin_amdgcn_wave_barrier();
    }
}

struct SynthShape {  // n_clusters == 0: plain i.i.d. rows, times a per-row amplitude in [amp_lo, amp_lo + amp_span) if amp_span >= 0
    uint32_t n_clusters;
    float noise, inv_sqrt_d;
    float amp_lo, amp_span;  // amp_span < 0: no amplitude
};
__device__ __forceinline__ float4 synth_value(uint64_t seed, int64_t row, uint32_t f4, const SynthShape& sh) {
    if (sh.n_clusters) return synth_piece_clustered(seed, row, f4, sh.n_clusters, sh.noise, sh.inv_sqrt_d);
    return sh.amp_span >= 0.0f ? synth_piece_scaled(seed, row, f4, sh.amp_lo, sh.amp_span) : synth_piece(seed, row, f4);
}

__global__ __launch_bounds__(256) void synth_inv_kernel(uint32_t nrows, int D4src, uint64_t seed, int64_t first_row,
                                                        SynthShape sh, float* __restrict__ inv) {
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    double nx = 0.0;
    for (int f4 = 0; f4 < D4src; ++f4) {
        float4 v = synth_value(seed, first_row + row, (uint32_t)f4, sh);
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
                                                         SynthShape sh, const float* __restrict__ inv, int64_t total) {
    const uint32_t first_blk = row0 >> 5;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int r = (int)(t & 31);
        const int64_t u = t >> 5;
        const int f4 = (int)(u % D4src);
        const int64_t lb = u / D4src;
        const int64_t row = (lb + first_blk) * 32 + r;
        const int64_t src = row - row0;
        if (src < 0 || src >= nrows) continue;
        float4 v = synth_value(seed, first_row + src, (uint32_t)f4, sh);
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

// The clean state every pass starts from and rescore_select_kernel leaves behind.
__global__ __launch_bounds__(256) void reset_scan_state_kernel(uint32_t* __restrict__ tau, uint32_t* __restrict__ slots,
                                                               uint32_t* __restrict__ cand_cnt) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < kMfmaQueries * kMaxK) slots[t] = kKeyNegInf;
    if (t < kMfmaQueries) {
        tau[t * kHot] = kKeyNegInf;
        tau[kMfmaQueries * kHot + t] = kKeyNegInf;  // (ScanParams::tau_c)
        cand_cnt[t * kHot] = 0;
        cand_cnt[t * kHot + 32] = 0;
        cand_cnt[t * kHot + 33] = 0;
    }
}

// is block `lb` of segment `si` one the seed kernel ranked?  (its rows are in the slots already: no second offer)
__device__ __forceinline__ bool is_seed_block(const ScanParams& p, int si, uint32_t lb) {
    return si == 0 && (lb & ((1u << p.seed_shift) - 1u)) == 0 && (lb >> p.seed_shift) < p.seed_blocks;
}

// min over the k slots of query q = the threshold the seed rows alone justify.  Four threads per query.
// Every consumer of a threshold takes max(this, tau): the seed kernel only fills the slots (no hand-off
// inside a launch), and tau is raised by the scan's offers.
__device__ __forceinline__ uint32_t seed_threshold_key(const ScanParams& p, int q, int sub /*0..3*/) {
    uint32_t mn = 0xffffffffu;
    if (q < p.B)
        for (int j = sub; j < p.k; j += 4) mn = min(mn, ld_relaxed(&p.slots[(size_t)q * kMaxK + j]));
    mn = min(mn, (uint32_t)__shfl_xor(mn, 1));
    mn = min(mn, (uint32_t)__shfl_xor(mn, 2));
    return q < p.B ? mn : kKeyNegInf;
}

// Host block -> device mirror (parameters, segment table, queries) by the compute queue itself: a
// hipMemcpyAsync of this size goes through the SDMA engine and costs two queue hand-offs (~15 us).
__global__ __launch_bounds__(256) void upload_kernel(const uint4* __restrict__ src_pinned, uint4* __restrict__ dst, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n16) dst[i] = src_pinned[i];
}

// prep + seed in one launch.  Workgroup (part, query group) first brings its QG queries into scan form
// (x / |x| for cosine; |q|^2 by a lane-parallel f64 sum — the canonical feature-order sum is only needed
// for the exact scores and is made by rescore_select_kernel) — the `part == 0` workgroups also publish
// them (f32, bf16, raw, margins) for the scan — then ranks rows [part*256, +256) of segment 0 against
// them with f32 FMA chains.  Seed rows are split in k disjoint groups (position in the sample mod k); the best score of
// each group goes to slot j: k distinct rows, so min(slots) is a valid running k-th best, without any
// selection step and without any hand-off between workgroups (the atomics are fire-and-forget).  The
// streaming kernels thus start with a useful threshold: with W waves in flight the first round screens
// 32*W rows against it.
template <int QG>
__global__ __launch_bounds__(256) void prep_seed_kernel(const ScanParams* __restrict__ pp, const float4* __restrict__ seg0_blk,
                                                        const float* __restrict__ seg0_scale, uint32_t nseed) {
    const ScanParams& p = *pp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Dp = p.D4 * 4, D = p.D;
    float* sq = smem;                                // [QG][Dp]
    uint32_t* gmax = (uint32_t*)(smem + QG * Dp);    // [QG][kMaxK]
    const int part = blockIdx.x, q0 = blockIdx.y * QG, tid = threadIdx.x;
    const bool writer = part == 0;
    // thread t owns seed row part*256 + t.  16 pieces (256 B of the row) are requested per step before any
    // of them is used: the phase is latency-bound (a workgroup reads 384 KB once), so what counts is bytes
    // in flight — 64 KB per workgroup.  The first step's loads go out before the queries are prepared.
    static_assert(kSeedPartRows == 256, "one seed row per thread");
    const uint32_t sr = part * kSeedPartRows + tid;  // seed row: row sr & 31 of seed block sr >> 5
    const uint32_t row = (((sr >> 5) << p.seed_shift) << 5) | (sr & 31);
    const bool valid = (sr >> 5) < p.seed_blocks && row < nseed;
    const uint32_t rowc = valid ? row : 0u;
    const float4* base = seg0_blk + (size_t)(rowc >> 5) * p.D4 * 32 + (rowc & 31);
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = gld4(base + (size_t)j * 32);
    const float sc = valid ? gld(&seg0_scale[row]) : 0.0f;
    __builtin_amdgcn_sched_barrier(0);

    for (int i = tid; i < QG * kMaxK; i += 256) gmax[i] = kKeyNegInf;
    {
        // all QG queries at once: LPQ lanes per query, each with every LPQ-th element, loads issued eight at a
        // time (one query per wave after the other cost ~10 us each in dependent round trips)
        constexpr int LPQ = QG >= 4 ? 256 / QG : 64;
        const int g = tid / LPQ, sub = tid % LPQ;
        const int q = q0 + g;
        const bool mine = g < QG;
        const bool have = mine && q < p.B;
        float* dst = sq + (mine ? g : 0) * Dp;
        const float* src = p.queries + (size_t)(have ? q : 0) * D;
        double sum = 0.0;
        for (int i0 = sub; i0 < Dp; i0 += 8 * LPQ) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * LPQ;
                x[j] = (have && i < D) ? gld(&src[i]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * LPQ;
                if (mine && i < Dp) dst[i] = x[j];
                sum += (double)x[j] * (double)x[j];
            }
        }
#pragma unroll
        for (int off = LPQ / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const bool ok = sum >= 0x1p-126 && sum < __builtin_inf();
        const float inv = (p.metric == PCV_METRIC_DOT) ? 1.0f : (ok ? (float)(1.0 / sqrt(sum)) : 0.0f);
        const bool live = (p.metric == PCV_METRIC_DOT) ? (sum < __builtin_inf()) : ok;
        if (mine) {
            for (int i = sub; i < Dp; i += LPQ) {  // each lane rewrites the elements it wrote
                const float raw = dst[i];
                const float qh = (have && live) ? raw * inv : 0.0f;
                dst[i] = qh;
                if (writer && have) {
                    gst(&p.qraw[(size_t)q * Dp + i], raw);
                    gst(&p.qf32[(size_t)q * Dp + i], qh);
                    const __bf16 hb = (__bf16)qh;
                    gst(&p.qbf16[(size_t)q * Dp + i], __builtin_bit_cast(uint16_t, hb));
                }
            }
            if (writer && have && sub == 0) {
                // cosine: scores are O(1); dot: |s - c| <= eps * |q| * max|x|
                float unit = 1.0f;
                if (p.metric == PCV_METRIC_DOT) unit = (float)sqrt(sum) * p.max_norm * 1.0001f;
                gst(&p.margin[q], (p.eps16 + p.eps32) * unit);
                gst(&p.margin32[q], 2.0f * p.eps32 * unit);
            }
        }
    }
    if (writer && blockIdx.y == 0) {
        // tile rows the scan kernel stages but no query fills
        uint16_t* z = p.qbf16 + (size_t)p.B * Dp;
        const int nz = ((int)p.tile_rows > p.B) ? ((int)p.tile_rows - p.B) * Dp : 0;
        for (int i = tid; i < nz; i += 256) gst(&z[i], (uint16_t)0);
        if (tid == 0 && p.flag_rec) {
            pcv_hit_dev f;
            f.score = 0.0;
            f.pos = 0;
            f.id = 0;
            *p.flag_rec = f;
        }
    }
    __syncthreads();

    float acc[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) acc[g] = 0.0f;
    for (int f0 = 0; f0 < p.D4; f0 += 16) {  // D4 is a multiple of 16
        if (f0 > 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = gld4(base + (size_t)(f0 + j) * 32);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const float4 qv = *(const float4*)&sq[g * Dp + (f0 + j) * 4];
                acc[g] = fmaf(qv.x, v[j].x, acc[g]);
                acc[g] = fmaf(qv.y, v[j].y, acc[g]);
                acc[g] = fmaf(qv.z, v[j].z, acc[g]);
                acc[g] = fmaf(qv.w, v[j].w, acc[g]);
            }
    }
    {
        const uint32_t grp = sr % (uint32_t)p.k;  // groups by position in the sample: every group gets rows whatever the stride
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const float s = acc[g] * sc;
            if (sc != 0.0f && isfinite(s) && (q0 + g >= p.B || ceil_class(p, q0 + g, s) == 0)) atomicMax(&gmax[g * kMaxK + grp], f32_key(s));
        }
    }
    __syncthreads();
    for (int i = tid; i < QG * p.k; i += 256) {
        const int g = i / p.k, j = i - g * p.k, q = q0 + g;
        const uint32_t key = gmax[g * kMaxK + j];
        if (q < p.B && key != kKeyNegInf) g_atomic_max(&p.slots[(size_t)q * kMaxK + j], key);
    }
}

// The same prep + seed with the contraction on the exact-f32 matrix instruction (dim <= 1024, the normal
// case): a workgroup takes 128 seed rows (four corpus blocks, one per wave) against a group of 32 queries;
// v_mfma_f32_32x32x2_f32 is bitwise an fmaf chain, so the slots still hold f32-accurate scores.  The VALU
// form above reads every query value from LDS once per row and is bound by that (12 us of LDS cycles + the
// exposed load latency); here a query value is read once per 32 rows and the next 16 row pieces are in
// flight while 64 MFMAs run.  Lane (r = lane&31, h = lane>>5) holds the pieces 2m+h of row r — the wave
// kernel's register layout — so one 16-byte load feeds four MFMAs (k-pair j: features 8m+j | 8m+4+j).
__global__ __launch_bounds__(256) void prep_seed_mfma_kernel(const ScanParams* __restrict__ pp, const float4* __restrict__ seg0_blk,
                                                             const float* __restrict__ seg0_scale, uint32_t nseed) {
    const ScanParams& p = *pp;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Dp = p.D4 * 4, D = p.D, LD = Dp + 4;  // padded query rows: the 16-lane ds_read_b128 groups hit 16 bank quads
    float* sq = smem;                            // [32][LD]
    uint32_t* gmax = (uint32_t*)(smem + 32 * LD);  // [32][k]
    const int part = blockIdx.x, q0 = blockIdx.y * 32, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const bool writer = part == 0;
    const uint32_t sb = part * 4 + wave;           // this wave's seed block = corpus block sb << seed_shift of segment 0
    const uint32_t lb = sb << p.seed_shift;
    const bool active = sb < p.seed_blocks && lb * 32u < nseed;
    const int P = p.D4 >> 1;                       // pieces per lane
    const float4* base = seg0_blk + (size_t)(active ? lb : 0) * p.D4 * 32 + h * 32 + r;
    float4 va[16], vb[16];
    auto load16 = [&](float4 (&v)[16], int m0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (active && m0 + j < P) ? gld4(base + (size_t)(m0 + j) * 64) : make_float4(0, 0, 0, 0);
    };
    load16(va, 0);  // in flight while the queries are prepared
    __builtin_amdgcn_sched_barrier(0);

    for (int i = tid; i < 32 * p.k; i += 256) gmax[i] = kKeyNegInf;
    {
        // 32 queries at once, 8 lanes per query, loads issued eight at a time
        const int g = tid >> 3, sub = tid & 7;
        const int q = q0 + g;
        const bool have = q < p.B;
        float* dst = sq + g * LD;
        const float* src = p.queries + (size_t)(have ? q : 0) * D;
        double sum = 0.0;
        for (int i0 = sub; i0 < Dp; i0 += 64) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + 8 * j;
                x[j] = (have && i < D) ? gld(&src[i]) : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + 8 * j;
                if (i < Dp) dst[i] = x[j];
                sum += (double)x[j] * (double)x[j];
            }
        }
        sum += __shfl_xor(sum, 4);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 1);
        const bool ok = sum >= 0x1p-126 && sum < __builtin_inf();
        const float inv = (p.metric == PCV_METRIC_DOT) ? 1.0f : (ok ? (float)(1.0 / sqrt(sum)) : 0.0f);
        const bool live = have && ((p.metric == PCV_METRIC_DOT) ? (sum < __builtin_inf()) : ok);
        for (int i = sub; i < Dp; i += 8) {  // each lane rewrites the elements it wrote
            const float raw = dst[i];
            const float qh = live ? raw * inv : 0.0f;
            dst[i] = qh;
            if (writer && have) {
                gst(&p.qraw[(size_t)q * Dp + i], raw);
                gst(&p.qf32[(size_t)q * Dp + i], qh);
                const __bf16 hb = (__bf16)qh;
                gst(&p.qbf16[(size_t)q * Dp + i], __builtin_bit_cast(uint16_t, hb));
            }
        }
        if (writer && have && sub == 0) {
            // cosine: scores are O(1); dot: |s - c| <= eps * |q| * max|x|
            float unit = 1.0f;
            if (p.metric == PCV_METRIC_DOT) unit = (float)sqrt(sum) * p.max_norm * 1.0001f;
            gst(&p.margin[q], (p.eps16 + p.eps32) * unit);
            gst(&p.margin32[q], 2.0f * p.eps32 * unit);
        }
    }
    if (writer && blockIdx.y == 0) {
        // tile rows the scan kernel stages but no query fills
        uint16_t* z = p.qbf16 + (size_t)p.B * Dp;
        const int nz = ((int)p.tile_rows > p.B) ? ((int)p.tile_rows - p.B) * Dp : 0;
        for (int i = tid; i < nz; i += 256) gst(&z[i], (uint16_t)0);
        if (tid == 0 && p.flag_rec) {
            pcv_hit_dev f;
            f.score = 0.0;
            f.pos = 0;
            f.id = 0;
            *p.flag_rec = f;
        }
    }
    __syncthreads();

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    const float* qrow = sq + r * LD + h * 4;  // lane (c = r, h): B operand of k-half h
    auto mma16 = [&](const float4 (&v)[16], int m0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (m0 + j < P) {  // wave-uniform
                const float4 qv = *(const float4*)&qrow[(m0 + j) * 8];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].x, qv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].y, qv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].z, qv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j].w, qv.w, acc, 0, 0, 0);
            }
        }
    };
    for (int m0 = 0; m0 < P; m0 += 32) {
        if (m0 + 16 < P) load16(vb, m0 + 16);
        __builtin_amdgcn_sched_barrier(0);
        mma16(va, m0);
        if (m0 + 32 < P) load16(va, m0 + 32);
        __builtin_amdgcn_sched_barrier(0);
        if (m0 + 16 < P) mma16(vb, m0 + 16);
    }
    // lane (c = r, h) holds query c's scores of the block rows (i&3) + 8(i>>2) + 4h
    if (active) {
        float sc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t row = lb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            sc[i] = row < nseed ? gld(&seg0_scale[row]) : 0.0f;
        }
        const float clo = (p.ceil && q0 + r < p.B) ? gld(&p.ceil[q0 + r].lo) : __builtin_inff();  // (scan.h, CeilRec: only rows that count)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float s = acc[i] * sc[i];
            const uint32_t sr = sb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;  // position in the sample: its group (every group gets rows whatever the stride)
            if (sc[i] != 0.0f && isfinite(s) && s < clo) atomicMax(&gmax[r * p.k + sr % (uint32_t)p.k], f32_key(s));
        }
    }
    __syncthreads();
    for (int i = tid; i < 32 * p.k; i += 256) {
        const int g = i / p.k, j = i - g * p.k, q = q0 + g;
        const uint32_t key = gmax[i];
        if (q < p.B && key != kKeyNegInf) g_atomic_max(&p.slots[(size_t)q * kMaxK + j], key);
    }
}

struct BlockCursor {  // position of a wave in its flat (block, chunk) stream; all of it wave-uniform
    uint32_t gb;      // launch-wide block index (>= total_blocks: exhausted)
    SegCursor sc;     // segment
    uint32_t lb;      // block inside the segment
    __amdgpu_buffer_rsrc_t rows;  // descriptor of the block's bytes in what the scan streams
    int ch;           // chunk inside the block
};

// Wave-reduction scan for 1..4 queries (BASELINE config "10M x 384, batch=1"): pure HBM streaming.
// Lane (r = lane&31, h = lane>>5) owns row r of the block and the pieces f4 = 2j+h; the two halves
// of a row are combined with one cross-lane add.  f32 FMA chain -> the fine test applies directly.
template <int NB, bool NTL>
__global__ __launch_bounds__(256) void scan_wave_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ __attribute__((aligned(16))) float sq[];  // [NB][Dp]
    const int Dp = p.D4 * 4;
    for (int i = threadIdx.x; i < NB * Dp; i += 256) sq[i] = gld(&p.qf32[i]);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, h = lane >> 5;
    float mrg[NB];
    uint32_t tau0[NB];  // what the seed rows alone justify (see seed_threshold_key)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        mrg[b] = gld(&p.margin32[b]);
        tau0[b] = seed_threshold_key(p, b, lane & 3);
    }
    const uint32_t total_waves = gridDim.x * 4;
    const int NCH = p.D4 >> 4;  // chunks of 8 pieces per lane (64 features of the row, both halves)
    if (blockIdx.x * 4 + wave >= p.total_blocks) return;

    // Same flat (block, chunk) stream with register chunk buffers as the MFMA kernel: the loads of the
    // next chunk — also across block boundaries — are in flight while the current one is multiplied.
    BlockCursor cons, prod;
    const uint32_t lane_off = (uint32_t)(h * 32 + r) * 16u;
    auto enter = [&](BlockCursor& k, uint32_t gb) {
        k.gb = gb;
        k.ch = 0;
        if (gb < p.total_blocks) {
            seek_seg(p, k.sc, gb);
            k.lb = gb - k.sc.begin;
            k.rows = row_rsrc(k.sc.blk + (size_t)k.lb * p.D4 * 32, (uint32_t)p.D4 * 512u);
        }
    };
    enter(cons, blockIdx.x * 4 + wave);
    prod = cons;
    float sc_cur = gld(&cons.sc.scale[(size_t)cons.lb * 32 + r]), sc_next = 0.0f;
    uint32_t tk[NB];
    float acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = 0.0f;

    float4 buf[2][8];
    auto produce = [&](float4 (&bf)[8]) {
        // ALWAYS issues its loads — past the end of the wave's stream they re-read a chunk of its last block (two or three
        // chunks per wave and launch).  With an early return here the compiler has to place every s_waitcnt for the case that
        // the younger chunks were never requested: each multiply then waited for (nearly) all loads in flight, the ones just
        // issued included, and the chunk buffers hid nothing.
#pragma unroll
        for (int i = 0; i < 8; ++i) bf[i] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)i * 1024u, (uint32_t)prod.ch * 8192u);
        if (prod.gb < p.total_blocks && ++prod.ch == NCH) {
            enter(prod, prod.gb + total_waves);  // (leaves descriptor and lb where they are when the stream is over)
            sc_next = gld(&prod.sc.scale[(size_t)prod.lb * 32 + r]);
        }
    };
    auto pre = [&]() {  // thresholds one chunk ahead of the block's end, requested before that step's row loads and not
        if (cons.ch == (NCH >= 2 ? NCH - 2 : 0)) {  // touched until then (see scan_mfma_kernel)
#pragma unroll
            for (int b = 0; b < NB; ++b) tk[b] = ld_relaxed(&p.tau[b * kHot]);
        }
    };
    auto consume = [&](const float4 (&bf)[8]) {
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
                thr[b] = key_f32(max(tau0[b], tk[b])) - mrg[b];
                any |= (h == 0) && (sc_cur != 0.0f) && !(s[b] < thr[b]);
                acc[b] = 0.0f;
            }
            if (__any(any)) {
                const bool feeds = !is_seed_block(p, cons.sc.si, cons.lb);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    if ((h == 0) && (sc_cur != 0.0f) && !(s[b] < thr[b])) emit_hit(p, b, cons.sc.si, row, s[b], feeds);
            }
            enter(cons, cons.gb + total_waves);
            sc_cur = sc_next;
        }
    };
    produce(buf[0]);
    while (true) {
        pre();
        produce(buf[1]);
        consume(buf[0]);
        if (cons.gb >= p.total_blocks) return;
        pre();
        produce(buf[0]);
        consume(buf[1]);
        if (cons.gb >= p.total_blocks) return;
    }
}

// MFMA tile scan for up to 128 queries (BASELINE config "100M x 384, batch=64").
// D[row][query] = sum_k A[row][k] * B[k][query] on v_mfma_f32_32x32x16_bf16: A = 32 corpus rows of a
// block (f32 from HBM, rounded to bf16 in registers), B = the query tile (bf16, LDS, XOR-swizzled so
// the 16-lane ds_read_b128 groups are conflict-free).  One bf16 product term -> eps16 = 2^-8: the
// coarse test passes a few hundred rows per query out of 10^8, for 1/16 of the f32 matrix cost; each
// of those is scored again in exact f32 by the whole wave (one more read of that row) and has to pass
// the fine test before it is emitted, so the candidate lists stay a few dozen long even when thousands
// of rows sit inside the bf16 margin of the k-th best (clustered embeddings).  Each wave streams its
// own blocks straight into registers: the blocked HBM layout makes every load two contiguous 512 B
// runs, so there is no LDS round trip for the corpus.
// Variants: NT = 32-query tiles per wave (1, 2 -> B <= 64; 4 -> B <= 128), WPB = waves per
// workgroup, NBUF = chunk buffers per wave (NBUF-1 chunks of loads in flight while one is consumed).
//   B <= 64 : 256 threads, 3 workgroups/CU (3 waves/SIMD), NBUF 2 (NBUF 3 at 2 waves/SIMD measured no faster)
//   B <= 128: 512 threads sharing one 96 KB query tile, 1 workgroup/CU, 2 waves/SIMD, NBUF 3: the
//             scan stays HBM-bound (MFMA ~25 % busy), so 128 queries cost the same 23 ms as 64
// (A 512-thread / 4-waves-per-SIMD build of the B <= 64 kernel was tried: the 128-VGPR cap spills 23
// registers into the chunk loop and runs 15 % slower.)

// exact-f32 dot of query row `qf` with corpus row (`rowbase` = its piece 0; pieces are 32 float4 apart),
// computed by the whole wave: every lane returns the same bits
__device__ __forceinline__ float wave_dot_f32(const float* qf, const float4* rowbase, int D4, int lane) {
    float part = 0.0f;
    for (int f4 = lane; f4 < D4; f4 += 64) {
        const float4 v = gld4(rowbase + (size_t)f4 * 32);
        const float4 qv = gld4(qf + 4 * f4);
        part = fmaf(qv.x, v.x, part);
        part = fmaf(qv.y, v.y, part);
        part = fmaf(qv.z, v.z, part);
        part = fmaf(qv.w, v.w, part);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    return part;
}

// The coarse survivors of one (row block, 32-query tile) of an MFMA scan: `mask` = this lane's surviving
// accumulators (bit i = row (i&3) + 8*(i>>2) + 4*(lane>>5) of the block, query 32*t + (lane&31)).
// Rare path (a few hundred coarse survivors per query over 10^8 rows; every wave also takes it a handful of
// times while the thresholds are still loose).  Each survivor is handled by the whole wave: exact-f32 score
// from a read of that f32 row, fine test against the current threshold, and only then the list append and
// the offer to the running top-k.
__device__ __forceinline__ void fine_survivors(const ScanParams& p, uint32_t mask, int t, const SegCursor& esc, uint32_t elb,
                                               const uint32_t* ltau0, int lane, int D4) {
    unsigned long long ball = __ballot(mask != 0);
    if (!ball) return;
    const bool feeds = !is_seed_block(p, esc.si, elb);
    const float* scp = esc.scale + (size_t)elb * 32;
    const float4* bbase = esc.blk + (size_t)elb * D4 * 32;
    const int Dp = D4 * 4;
    while (ball) {
        const int src = __builtin_ctzll(ball);
        ball &= ball - 1;
        uint32_t m = __builtin_amdgcn_readlane(mask, src);
        const int q = 32 * t + (src & 31), hh = src >> 5;
        const float m32 = gld(&p.margin32[q]);
        if (lane == 0) g_atomic_add(&p.cand_cnt[q * kHot + 32], (uint32_t)__builtin_popcount(m));  // statistics (no return value is used)
        while (m) {
            const int i = __builtin_ctz(m);
            m &= m - 1;
            const int rib = (i & 3) + 8 * (i >> 2) + 4 * hh;  // row of the block this accumulator holds
            // row scale, threshold and the row itself are requested together: one memory round trip
            // per survivor (tested one after the other they were three, ~6 us under a saturated stream)
            const float sc = gld(&scp[rib]);
            const uint32_t tkey = ld_relaxed(&p.tau[q * kHot]);
            if (esc.mid16) {
                // mid screen (scan.h): the row's 16-bit copy, Dp * 2 contiguous bytes, one 16-byte piece per lane; only what it
                // cannot rule out goes on to the f32 row.  (Requested with scale and threshold: one round trip.)
                const uint32_t row = elb * 32 + (uint32_t)rib;
                const float s2 = gld(&esc.scale16[row]);
                float part = 0.0f;
                for (int pc = lane; pc < (Dp >> 3); pc += 64) {  // (one trip up to 512-d, two up to 1024-d)
                    const uint4 pv = __builtin_bit_cast(uint4, gld4((const float4*)esc.mid16 + (size_t)row * (Dp >> 3) + pc));
                    const float4 qa = gld4(p.qf32 + (size_t)q * Dp + 8 * pc), qb = gld4(p.qf32 + (size_t)q * Dp + 8 * pc + 4);
                    part = fmaf(qa.x, (float)(int16_t)(pv.x & 0xffff), part);
                    part = fmaf(qa.y, (float)((int32_t)pv.x >> 16), part);
                    part = fmaf(qa.z, (float)(int16_t)(pv.y & 0xffff), part);
                    part = fmaf(qa.w, (float)((int32_t)pv.y >> 16), part);
                    part = fmaf(qb.x, (float)(int16_t)(pv.z & 0xffff), part);
                    part = fmaf(qb.y, (float)((int32_t)pv.z >> 16), part);
                    part = fmaf(qb.z, (float)(int16_t)(pv.w & 0xffff), part);
                    part = fmaf(qb.w, (float)((int32_t)pv.w >> 16), part);
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
                if (sc == 0.0f) continue;
                const float taum = key_f32(max(ltau0[q], tkey));
                const float quant = gld(&p.q8c[4 * q + 2]) * 0.5003f / s2;  // |q'|_1 * 0.5002 / s2, rounded up
                if (!(part / s2 >= taum - (quant + 1.5f * m32))) continue;  // (NaN scale: dropped)
                if (lane == 0) g_atomic_add(&p.cand_cnt[q * kHot + 33], 1u);  // statistics: pairs the mid screen let through
                // (measured and dropped: coarse survivors noted in a 16-entry LDS queue per wave and worked off four at a time,
                // 16 lanes per mid row, one round trip for the four: clustered corpus 6.86 against 6.84 ms — the waves of a SIMD
                // cover each other's round trips already; Gaussian rows slower, 0.973 against 0.941 ms at 12.5M rows and 7.43
                // against 7.03 at 768-d / 128 queries: the deferred rows are the ones whose offers raise the thresholds,
                // coarse survivors per query went from 500 to 850)
            }
            const float dot = wave_dot_f32(p.qf32 + (size_t)q * Dp, bbase + rib, D4, lane);
            if (sc == 0.0f) continue;  // padding / unsearchable row
            const float s32 = dot * sc;
            const float taun = key_f32(max(ltau0[q], tkey));
            if (s32 < taun - m32) continue;
            const int cc = ceil_class(p, q, s32);
            if (cc == 2) continue;
            if (lane == 0) {
                const uint32_t at = g_atomic_add(&p.cand_cnt[q * kHot], 1u);
                if (at < p.cand_cap) {
                    gst(&p.cand[(size_t)q * p.cand_cap + at], ((uint64_t)(uint32_t)esc.si << 32) | (elb * 32 + (uint32_t)rib));
                    gst(&p.cand_s[(size_t)q * p.cand_cap + at], s32);
                }
            }
            if (feeds && cc == 0 && isfinite(s32) && s32 > taun) offer_slot_wave(p, q, s32, lane);
        }
    }
}

// Place of 16-byte piece `pc` of query row `q` inside the row's P8 pieces of the LDS tile: XOR with the row number
// inside groups of 16 pieces, so that the 16 lanes of a ds_read_b128 group (16 consecutive rows, one piece index) hit 16
// distinct bank quads.  P8 is a multiple of 8 (dimension padded to 64), not always of 16: a trailing group of 8 pieces is
// swizzled inside itself (2-way conflicts there; XOR with four bits would leave the row).
__device__ __forceinline__ int swizzle_piece(int pc, int q, int P8) {
    return pc < (P8 & ~15) ? ((pc & ~15) | ((pc ^ q) & 15)) : ((pc & ~7) | ((pc ^ q) & 7));
}

// SRC16: stream the segments' screening copies (bf16, scale folded in: scan.h) instead of converting the f32 rows —
// half the bytes per row and no conversion work; a chunk is then 4 pieces per lane instead of 8.
template <int NT, bool NTL, int WPB, int NBUF, bool SRC16>
__global__ __launch_bounds__(WPB * 64, WPB == 4 ? ((SRC16 ? NBUF <= 4 : NBUF == 2) ? 3 : 2) : 2) void scan_mfma_kernel(
    const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ uint4 lq[];  // [NT*32][Dp/8] 16-byte pieces of 8 bf16, swizzled
    const int D4 = p.D4;
    const int P8 = D4 >> 1;   // 16-B pieces per query row
    const int NCH = D4 >> 4;  // chunks of 64 features
    for (int i = threadIdx.x; i < NT * 32 * P8; i += WPB * 64) {
        const int q = i / P8, pc = i - q * P8;
        lq[q * P8 + swizzle_piece(pc, q, P8)] = __builtin_bit_cast(uint4, gld4((const float4*)p.qbf16 + i));
    }
    __shared__ uint32_t ltau0[NT * 32];  // what the seed rows alone justify, per query (see seed_threshold_key)
    for (int q = threadIdx.x >> 2; q < NT * 32; q += WPB * 16) {
        const uint32_t key = seed_threshold_key(p, q, threadIdx.x & 3);
        if ((threadIdx.x & 3) == 0) ltau0[q] = key;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    float mrg[NT];
    uint32_t tau0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        mrg[t] = (32 * t + c < p.B) ? gld(&p.margin[32 * t + c]) : 0.0f;
        tau0[t] = ltau0[32 * t + c];
    }

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
            seek_seg(p, k.sc, gb);
            k.lb = gb - k.sc.begin;
            if constexpr (SRC16)
                k.rows = row_rsrc((const float4*)k.sc.blk16 + (size_t)k.lb * (D4 >> 1) * 32, (uint32_t)(D4 >> 1) * 512u);
            else
                k.rows = row_rsrc(k.sc.blk + (size_t)k.lb * D4 * 32, (uint32_t)D4 * 512u);
        }
    };
    const uint32_t lane_off = (uint32_t)(SRC16 ? h * 32 + c : h * 64 + c) * 16u;  // the lane's bytes inside a block
    BlockCursor cons, prod;  // consumer (MFMA) and producer (loads) positions; prod runs NBUF-1 chunks ahead
    enter_block(cons, blockIdx.x * WPB + wave);
    prod = cons;
    // scale of this lane's row (1/|x|, 1 or 0): multiplied into the A operand before the bf16
    // rounding, so the accumulators are final screening scores.  sc_next belongs to the block the
    // producer has entered but the consumer has not.
    float sc_cur = SRC16 ? 1.0f : gld(&cons.sc.scale[(size_t)cons.lb * 32 + c]), sc_next = 0.0f;

    constexpr int PCS = SRC16 ? 4 : 8;  // 16-byte pieces a lane loads per chunk of 64 features
    float4 buf[NBUF][PCS];
    // lane's pieces of k-step ks of a chunk: f4 = chunk*16 + ks*4 + 2h + e  (2h folded into base);
    // screening copy: f8 = chunk*8 + ks*2 + h  (h folded into base)
    auto produce = [&](float4 (&b)[PCS]) {
        // ALWAYS issues its loads — past the end of the wave's stream they re-read a chunk of its last block (two or three
        // chunks per wave and launch).  With an early return here the compiler has to place every s_waitcnt for the case that
        // the younger chunks were never requested: each multiply then waited for (nearly) all loads in flight, the ones just
        // issued included, and the chunk buffers hid nothing.
#pragma unroll
        for (int i = 0; i < PCS; ++i) {
            if constexpr (SRC16)
                b[i] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)i * 1024u, (uint32_t)prod.ch * 4096u);
            else
                b[i] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)((i >> 1) * 4 + (i & 1)) * 512u, (uint32_t)prod.ch * 8192u);
        }
        if (prod.gb < p.total_blocks && ++prod.ch == NCH) {
            enter_block(prod, prod.gb + total_waves);  // (leaves descriptor and lb where they are when the stream is over)
            if constexpr (!SRC16) sc_next = gld(&prod.sc.scale[(size_t)prod.lb * 32 + c]);
        }
    };

    // thresholds of the block being finished: requested one chunk ahead of the epilogue and BEFORE that step's row loads
    // (PCV_STEP), and not touched until the epilogue: vmcnt retires in order, so the first use of a threshold waits for every
    // load issued before it — used at once (a max with the seed threshold) it drained the row chunks in flight at every block.
    // Loaded for every lane, also those of tile rows beyond the batch (their words exist and are never raised): a load under
    // a lane-dependent condition is compiled as a branch with an s_waitcnt vmcnt(0) inside.
    uint32_t tauk[NT];
    auto tau_prefetch = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) tauk[t] = ld_relaxed(&p.tau[(32 * t + c) * kHot]);
    };

    auto epilogue = [&](const SegCursor& esc, uint32_t elb) {
        if (NCH < 2) tau_prefetch();
        float thr[NT];
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q = 32 * t + c;
            thr[t] = (q < p.B) ? key_f32(max(tau0[t], tauk[t])) - mrg[t] : __builtin_inff();
#pragma unroll
            for (int i = 0; i < 16; ++i) any |= !(acc[t][i] < thr[t]);
        }
        if (__any(any)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                uint32_t mask = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) mask |= (!(acc[t][i] < thr[t])) ? (1u << i) : 0u;
                fine_survivors(p, mask, t, esc, elb, ltau0, lane, D4);
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    };

    auto consume = [&](const float4 (&b)[PCS]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 a;
            if constexpr (SRC16) {
                a = __builtin_bit_cast(bf16x8, b[ks]);
            } else {
                f32x8 v = {b[2 * ks].x,     b[2 * ks].y,     b[2 * ks].z,     b[2 * ks].w,
                           b[2 * ks + 1].x, b[2 * ks + 1].y, b[2 * ks + 1].z, b[2 * ks + 1].w};
                a = __builtin_convertvector(v * sc_cur, bf16x8);
            }
            const int pc = 2 * (cons.ch * 4 + ks) + h;
            const int ph = swizzle_piece(pc, c, P8);  // (query row 32*t + c: the same low four bits as c)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16x8 q8 = *(const bf16x8*)&lq[(32 * t + c) * P8 + ph];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q8, acc[t], 0, 0, 0);
            }
        }
        if (++cons.ch == NCH) {
            epilogue(cons.sc, cons.lb);
            enter_block(cons, cons.gb + total_waves);
            sc_cur = sc_next;
        }
    };

    // NBUF-1 chunks of loads are always in flight while one chunk feeds the matrix cores
    // (written out per buffer: every buf[] index must be a literal, or the array moves to scratch)
#define PCV_STEP(REFILL, CONS)                              \
    if (NCH >= 2 && cons.ch == NCH - 2) tau_prefetch();     \
    produce(buf[REFILL]);                                   \
    consume(buf[CONS]);                                     \
    if (cons.gb >= p.total_blocks) return;
    produce(buf[0]);
    if constexpr (NBUF == 2) {
        while (true) {
            PCV_STEP(1, 0)
            PCV_STEP(0, 1)
        }
    } else if constexpr (NBUF == 3) {
        produce(buf[1]);
        while (true) {
            PCV_STEP(2, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
        }
    } else if constexpr (NBUF == 4) {
        produce(buf[1]);
        produce(buf[2]);
        while (true) {
            PCV_STEP(3, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
            PCV_STEP(2, 3)
        }
    } else {
        static_assert(NBUF == 6, "2, 3, 4 or 6 chunk buffers");
        produce(buf[1]);
        produce(buf[2]);
        produce(buf[3]);
        produce(buf[4]);
        while (true) {
            PCV_STEP(5, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
            PCV_STEP(2, 3)
            PCV_STEP(3, 4)
            PCV_STEP(4, 5)
        }
    }
#undef PCV_STEP
}

// MFMA scan over the int8 screening copies (scan.h): the same flat (block, chunk) stream as scan_mfma_kernel with
// chunks of 128 features (4 k-steps of v_mfma_i32_32x32x32_i8, 4 pieces of 16 bytes per lane), 384 B per 384-d row.
// The query tile is quantised by every workgroup itself while it is staged (q^_i = rint(q'_i * s_q), s_q = 127 / max|q'_i|),
// rows padded to an odd number of 16-byte pieces so the 16-lane ds_read_b128 groups are conflict-free without a swizzle.
// Coarse test on the exact integer dot product `acc` of row r and query q (proof: scan.h):
//     keep  iff  acc >= s_row * U_q - V_q,     U_q = (tau_q - eps32') s_q - 0.5 sqrt(D) |y|,   V_q = 0.5 |q'|_1 s_q + 0.25 D
// (|y| = 1 for cosine, max row norm for dot; constants carry rounding slack).  Survivors take the same wave-cooperative
// exact-f32 fine screen as in scan_mfma_kernel.
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

// The speculative start threshold of query q (scan.h), by one wave: the spec_rank-th largest seed slot, or the median
// slot + the learned gap if that is higher; ranks by counting.  Raises tau, records the guess for the check at the end
// of the pass and sends the query's median / best seed slot home for the learning.
__device__ __forceinline__ void set_guess(const ScanParams& p, int q, int lane, bool have) {
    uint32_t guess = kKeyNegInf, med = kKeyNegInf, top = kKeyNegInf;
    const bool learned = p.spec_gap == p.spec_gap;
    if ((p.spec_rank > 0 || learned) && have) {
        const uint32_t* sl = p.slots + (size_t)q * kMaxK;
        const int mid = (p.k - 1) / 2;
        for (int a = lane; a < p.k; a += 64) {
            const uint32_t va = ld_relaxed(&sl[a]);
            int ahead = 0;
            for (int i = 0; i < p.k; ++i) {
                const uint32_t vi = ld_relaxed(&sl[i]);
                ahead += (vi > va || (vi == va && i < a)) ? 1 : 0;
            }
            if (ahead == p.spec_rank - 1) guess = va;
            if (ahead == mid) med = va;
            if (ahead == 0) top = va;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            guess = max(guess, (uint32_t)__shfl_xor(guess, off));
            med = max(med, (uint32_t)__shfl_xor(med, off));
            top = max(top, (uint32_t)__shfl_xor(top, off));
        }
        if (learned && med != kKeyNegInf) {
            // never further above the best seed score than that one is above the median seed score: a query whose
            // seed scores are bunched (its neighbourhood was sampled) is not given the gap of queries whose are not
            const float fm = key_f32(med), ft = key_f32(top);
            const float g = fminf(fm + p.spec_gap, ft + (ft - fm));
            // ... and only a query whose seed scores spread like those the gap was learned from takes it at all
            const bool alike = fabsf((ft - fm) - p.spec_spread) <= 0.5f * p.spec_spread;
            if (alike && isfinite(g)) guess = max(guess, f32_key(g));
        }
        if (lane == 0 && guess != kKeyNegInf) {
            g_atomic_max(&p.tau[q * kHot], guess);
            g_atomic_max(&p.tau_c[q], guess);
        }
    }
    if (lane == 0 && have && p.spec_base_host) {
        p.spec_base_host[q] = med != kKeyNegInf ? key_f32(med) : __builtin_nanf("");
        p.spec_top_host[q] = top != kKeyNegInf ? key_f32(top) : __builtin_nanf("");
    }
    if (lane == 0) p.spec[q] = guess;
}

// ... as a kernel of its own in front of the scans that have no query-quantisation step (bf16 copy / f32 rows)
__global__ __launch_bounds__(256) void set_guess_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    set_guess(p, q, threadIdx.x & 63, q < p.B);
}

// Queries of the pass -> int8 tile of the int8 screen, one wave per tile row: q^_i = rint(q'_i * s_q), s_q = 127 / max|q'_i|
// (q' = the scan-side query prep_seed wrote), and the per-query constants of the test (scan_mfma8_kernel).  Tile rows
// beyond the batch are zeros with s_q = 0.
__global__ __launch_bounds__(256) void quantize_queries_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Dp = p.D4 * 4, Dp8 = (Dp + 127) & ~127;
    const bool have = q < p.B;
    const float* src = p.qf32 + (size_t)q * Dp;
    float x[16];  // Dp <= 1024: |acc| <= 1024 * 127^2 < 2^24, so the integer dot product converts to f32 exactly
    float mx = 0.0f, l1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        x[j] = (have && i < Dp) ? gld(&src[i]) : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        mx = fmaxf(mx, fabsf(x[j]));
        l1 += fabsf(x[j]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        l1 += __shfl_xor(l1, off);
    }
    const float s_q = (mx > 0.0f && mx < __builtin_inff() && l1 < __builtin_inff()) ? 127.0f / mx : 0.0f;
    int8_t* row = p.q8 + (size_t)q * Dp8;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        if (i < Dp8) row[i] = (int8_t)max(-127, min(127, (int)rintf((s_q != 0.0f ? x[j] : 0.0f) * s_q)));
    }
    if (lane == 0) {
        p.q8c[4 * q] = s_q;
        p.q8c[4 * q + 1] = 0.5002f * l1 * s_q + 0.2501f * (float)Dp8 + 4.0f;
        p.q8c[4 * q + 2] = l1;
    }
    set_guess(p, q, lane, have);
}

// (Where the waves' time goes, tools/build_stamps.sh, 100M x 384, 64 queries: with every wave taking the same number of blocks the
// first workgroup to arrive on a CU is done after 5.71 ms, the second after 5.98, the third after 6.25, the launch after 6.38:
// the CU's arbitration favours its older waves.  That is no loss: with the blocks handed out at run time — claims of 16
// consecutive blocks from per-XCD counters, requested one claim ahead — every wave ends between 6.20 and 6.37 ms and the launch
// takes the same 6.3-6.4 ms; so does one 12-wave workgroup per CU.  The memory system delivers ~6.2 TB/s to this access
// pattern whether 8 or 12 waves per CU are asking.  The fine screen takes 1.2 % of a wave's time at this size.)
// (Measured, 100M x 384, 64 queries, one box: this form 6.241 ms; with a producer that returns early once the wave's stream is
// over — the compiler then places every wait for the case that nothing younger was requested, so that each multiply waits for
// nearly all loads in flight — 6.239; with plain global loads and 64-bit vector addresses 6.268; both 6.281.  At 3 waves per
// SIMD and 12 per CU the other waves hide what one wave's chunk buffers would; the pass runs at the read rate the memory
// system gives this pattern.)
// f(integral_constant<int, I>) for I = FROM .. TO-1, until one returns true (a loop written out at compile time)
template <int FROM, int TO, class F>
__device__ __forceinline__ bool static_for_until(F&& f) {
    if constexpr (FROM < TO) {
        if (f(std::integral_constant<int, FROM>{})) return true;
        return static_for_until<FROM + 1, TO>(f);
    } else {
        return false;
    }
}

// ---- the survivor ring of scan_mfma8_kernel's DRAIN form -------------------------------------------------------------------
// A coarse survivor used to be handled on the spot by the wave that found it: mid row, f32 row, list append, slot offer — up to
// four dependent memory round trips during which the wave's chunk pipeline stood still, and an s_waitcnt vmcnt(0) in front of
// them that emptied it (profiles/r03_stamps_12p5m_b64_blockscale.txt: 18.7 us for a block with a survivor against 6.2 us; 8 of a
// wave's 127 blocks at 12.5M rows).  In the DRAIN form a streaming wave only WRITES the survivor — two LDS words, no global
// memory access, nothing for vmcnt to wait on — into a ring in LDS, and the last wave of the workgroup (the drain wave) does
// nothing but work the ring off, four survivors at a time with all their loads in flight together.  It owns the list appends
// and the slot offers, so the thresholds keep rising during the pass as before.
//   entry = lo: block inside the segment;  hi: bit 31 valid | segment << 13 | query << 5 | row inside the block
//   producer: reserve with one LDS atomic (ring.tail), wait — the whole wave together — until the ring has room up to its last
//             entry (ring.head, written by the drain wave), write lo, then hi;
//   drain wave: wait for hi of the entry at its head to turn valid, read, clear hi, advance head.
// No deadlock: the oldest unwritten entry belongs to a wave that waits for room for at most kRing / 2 entries beyond it (one
// reservation spans at most 64 lanes x 16 rows = 1024 entries), and the drain wave consumes everything in front of it without
// needing anyone.  The drain wave ends when every streaming wave has signed off (ring.done) and the ring is empty.
constexpr uint32_t kRing = 2048;
constexpr int kRingSegBits = 18;  // segments a launch in this form can address
struct SurvRing {
    uint32_t lo[kRing], hi[kRing];
    int acc[kRing];      // the survivor's integer dot product and
    float sblk[kRing];   // its block's quantisation scale: the drain wave repeats the coarse test against the thresholds of ITS time
    uint32_t tail, head, done, pad;
};
__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lds_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS operations of one wave execute in order; this keeps the compiler from reordering them
#define PCV_LDS_ORDER() asm volatile("" ::: "memory")

// The survivors of one lane (`m`: its surviving accumulators `acc` of query q, see fine_survivors) go into the ring.  Called by
// all lanes of a streaming wave together.
__device__ __forceinline__ void ring_push(SurvRing& ring, uint32_t m, const i32x16& acc, float sblk, uint32_t q, uint32_t hh, uint32_t si,
                                          uint32_t elb) {
    const uint32_t n = (uint32_t)__builtin_popcount(m);
    uint32_t pos = 0;
    if (m) pos = lds_add(&ring.tail, n);
    while (__any(m != 0 && (pos + n - 1u) - lds_ld(&ring.head) >= kRing)) __builtin_amdgcn_s_sleep(2);
    const uint32_t hib = 0x80000000u | (si << 13) | (q << 5) | (4u * hh);
#pragma unroll
    for (int i = 0; i < 16; ++i) {  // (unrolled: acc is indexed by constants and stays in registers)
        if (m & (1u << i)) {
            const uint32_t slot = pos & (kRing - 1);
            lds_st(&ring.lo[slot], elb);
            ring.acc[slot] = acc[i];
            ring.sblk[slot] = sblk;
            PCV_LDS_ORDER();
            lds_st(&ring.hi[slot], hib + (uint32_t)((i & 3) + 8 * (i >> 2)));  // row of the block this accumulator holds
            ++pos;
        }
    }
}

// The drain wave.  Up to 64 entries leave the ring at a time, one per lane, and meet the coarse test again — same formula, but
// against the thresholds as they stand now (tau of the 64 queries sits on the 64 lanes, renewed every round): what queued up
// while the thresholds rose is shed here at LDS speed, without a memory access.  What is left is worked off eight at a time,
// eight lanes each, all loads of the eight in flight together: mid screen (if the segment has the mid copy), exact-f32 score
// from the f32 row, fine test, list append; then the offers to the running top-k, one after the other by the whole wave.
// Same tests, same constants as fine_survivors.
// Order: the thresholds rise through the offers of rows that score above them, and while they are loose (the first blocks of
// every wave) survivors arrive faster than one wave can look at them.  The int8 estimate acc / (s_blk s_q) is within ~1e-3 of
// the score, the certified margin of the coarse test is 0.024 (384-d): an entry whose estimate clears the threshold by less
// than a quarter of the margin is a LIKELY candidate and is worked on at once; the others — four in five, nearly none of
// which ends up a candidate — wait on a stack of the drain wave's own (LDS, nobody else touches it) until the ring is empty,
// and meet the coarse test a third time then: most never cost a memory access.
// `streaming` = waves that sign off in ring.done; NQ = queries of the tile.
constexpr uint32_t kStack = 2048;
struct DrainStack {
    uint32_t lo[kStack], hi[kStack];
    int acc[kStack];
    float sblk[kStack];
};
template <int NQ>
__device__ __forceinline__ void drain_survivors(const ScanParams& p, SurvRing& ring, DrainStack& stk, const uint32_t* ltau0, const float* lsq,
                                                const float* lvq, float* lU, int lane, int D4, uint32_t streaming, unsigned long long* stamp) {
    static_assert(NQ <= 64, "one query per lane");
    // diagnostic build (-DPCV_STAMPS; stamp = this wave's 8 words): 0 entry, 3 end, 4 entries out of the ring, 5 ticks at work,
    // 6 rounds of eight worked, 7 entries shed by a repeated coarse test, 2 most entries waiting (ring: high word, stack: low word)
#ifdef PCV_STAMPS
#define PCV_DSTAMP(slot) if (stamp && lane == 0) stamp[slot] = __builtin_amdgcn_s_memrealtime();
#define PCV_DCOUNT(slot, n) if (stamp && lane == 0) stamp[slot] += (n);
#define PCV_DMAX(hi, lo_) if (stamp && lane == 0) { const unsigned long long o = stamp[2]; \
        stamp[2] = (max(o >> 32, (unsigned long long)(hi)) << 32) | max(o & 0xffffffffull, (unsigned long long)(lo_)); }
#else
#define PCV_DSTAMP(slot)
#define PCV_DCOUNT(slot, n)
#define PCV_DMAX(hi, lo_)
#endif
    PCV_DSTAMP(0)
#ifdef PCV_STAMPS
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();  // shader clock: slot 1 = its ticks over the wave's life
#endif
    const int Dp = D4 * 4, Dp8 = (Dp + 127) & ~127;
    const uint32_t g = (uint32_t)lane >> 3;
    const int sub = lane & 7;
    // this lane's query (lane < NQ): constants of the coarse test (scan_mfma8_kernel's epilogue)
    const int ql = lane < NQ ? lane : NQ - 1;
    const float my_sq = lsq[ql], my_vq = lvq[ql];
    const uint32_t my_tau0 = ltau0[ql];
    const float my_e32 = ql < p.B ? 0.5f * gld(&p.margin32[ql]) : 0.0f;
    const float nrm = (p.metric == PCV_METRIC_DOT) ? p.max_norm : 1.0f;
    const float c1 = 0.5002f * sqrtf((float)Dp8) * nrm;
    const float dead = (p.metric == PCV_METRIC_DOT) ? -__builtin_inff() : __builtin_inff();
    uint32_t my_tau = ld_relaxed(&p.tau_c[ql]);
    uint32_t head = 0, sp = 0, idle = 0;
    int c_si = -1;  // the segment this lane last worked on (its table entry stays in registers)
    const float4* c_blk = nullptr;
    const float* c_scale = nullptr;
    const uint4* c_mid = nullptr;
    const float* c_s16 = nullptr;
    __builtin_amdgcn_s_setprio(3);

    // one entry per lane group of eight (`on`: the group has one): everything after the coarse test
    auto work = [&](bool on, uint32_t hi, uint32_t lo) {
#ifdef PCV_STAMPS
        const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
#endif
        const int q = (int)((hi >> 5) & 0xffu), rib = (int)(hi & 31u), si = (int)((hi >> 13) & ((1u << kRingSegBits) - 1u));
        const uint32_t lb = lo, row = lb * 32u + (uint32_t)rib;
        if (on && si != c_si) {
            c_si = si;
            c_blk = gld(&p.seg[si].blk);
            c_scale = gld(&p.seg[si].scale);
            c_mid = gld(&p.seg[si].mid16);
            c_s16 = gld(&p.seg[si].scale16);
        }
        float sc = 0.0f, m32 = 0.0f, s2 = 1.0f, l1 = 0.0f, part = 0.0f;
        uint32_t tkey = kKeyNegInf;
        const bool has_mid = on && c_mid != nullptr;
        const float* qf = p.qf32 + (size_t)q * Dp;
        if (on) {
            sc = gld(&c_scale[row]);
            tkey = ld_relaxed(&p.tau[q * kHot]);
            m32 = gld(&p.margin32[q]);
        }
        if (has_mid) {  // mid screen (scan.h): Dp * 2 contiguous bytes, 16-byte pieces dealt round the 8 lanes
            s2 = gld(&c_s16[row]);
            l1 = gld(&p.q8c[4 * q + 2]);
            const float4* mrow = (const float4*)c_mid + (size_t)row * (Dp >> 3);
            for (int pc0 = sub; pc0 < (Dp >> 3); pc0 += 24) {  // (three pieces a round: the kernel's register count is the larger
                uint4 pv[3];                                    //  of this wave's and the streaming waves')
                float4 qa[3], qb[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int pc = pc0 + 8 * j;
                    if (pc < (Dp >> 3)) {
                        pv[j] = __builtin_bit_cast(uint4, gld4(mrow + pc));
                        qa[j] = gld4(qf + 8 * pc);
                        qb[j] = gld4(qf + 8 * pc + 4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (pc0 + 8 * j < (Dp >> 3)) {
                        part = fmaf(qa[j].x, (float)(int16_t)(pv[j].x & 0xffff), part);
                        part = fmaf(qa[j].y, (float)((int32_t)pv[j].x >> 16), part);
                        part = fmaf(qa[j].z, (float)(int16_t)(pv[j].y & 0xffff), part);
                        part = fmaf(qa[j].w, (float)((int32_t)pv[j].y >> 16), part);
                        part = fmaf(qb[j].x, (float)(int16_t)(pv[j].z & 0xffff), part);
                        part = fmaf(qb[j].y, (float)((int32_t)pv[j].z >> 16), part);
                        part = fmaf(qb[j].z, (float)(int16_t)(pv[j].w & 0xffff), part);
                        part = fmaf(qb[j].w, (float)((int32_t)pv[j].w >> 16), part);
                    }
                }
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) part += __shfl_xor(part, off);
        bool go = on && sc != 0.0f;  // (0: padding / unsearchable row)
        if (has_mid) {
            const float taum = key_f32(max(ltau0[q], tkey));
            const float quant = l1 * 0.5003f / s2;  // |q'|_1 * 0.5002 / s2, rounded up
            go = go && (part / s2 >= taum - (quant + 1.5f * m32));  // (NaN scale: dropped)
            if (go && sub == 0) g_atomic_add(&p.cand_cnt[q * kHot + 33], 1u);  // statistics: pairs the mid screen let through
        }
        // exact-f32 score of what is left: the row's pieces are 32 float4 apart in the blocked layout
        float dot = 0.0f;
        if (go) {
            const float4* rowbase = c_blk + (size_t)lb * D4 * 32 + rib;
            for (int f0 = sub; f0 < D4; f0 += 32) {
                float4 v[4], qv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f4 = f0 + 8 * j;
                    if (f4 < D4) {
                        v[j] = gld4(rowbase + (size_t)f4 * 32);
                        qv[j] = gld4(qf + 4 * f4);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (f0 + 8 * j < D4) {
                        dot = fmaf(qv[j].x, v[j].x, dot);
                        dot = fmaf(qv[j].y, v[j].y, dot);
                        dot = fmaf(qv[j].z, v[j].z, dot);
                        dot = fmaf(qv[j].w, v[j].w, dot);
                    }
                }
            }
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
        const float s32 = dot * sc;
        const float taun = key_f32(max(ltau0[q], tkey));
        const int cc = on ? ceil_class(p, q, s32) : 2;
        const bool cand = go && !(s32 < taun - m32) && cc != 2;
        if (cand && sub == 0) {
            const uint32_t at = g_atomic_add(&p.cand_cnt[q * kHot], 1u);
            if (at < p.cand_cap) {
                gst(&p.cand[(size_t)q * p.cand_cap + at], ((uint64_t)(uint32_t)si << 32) | row);
                gst(&p.cand_s[(size_t)q * p.cand_cap + at], s32);
            }
        }
        // the offers to the running top-k, all groups at once (one after the other by the whole wave they were three round trips
        // each, and the thresholds of a pass rise through exactly these)
        const bool offer = cand && cc == 0 && !is_seed_block(p, si, lb) && isfinite(s32) && s32 > taun;
        if (__any(offer)) offer_slot_g8(p, offer, q, s32, sub);
        PCV_DCOUNT(5, __builtin_amdgcn_s_memrealtime() - ts)
        PCV_DCOUNT(6, 1)
    };
    // U of this lane's query from the threshold as it stands, for this wave's repeat of the coarse test and — through lU — for
    // the streaming waves' tests.  The thresholds are fetched before every round of work and every eighth idle poll (~1 us),
    // from their side-by-side copy: one 256-byte load, used at the NEXT call (nobody waits for it here).  (Every streaming
    // wave used to fetch them for every block, a 64-line gather each — agent-scope loads that go past the L2 to the memory
    // side: 2816 waves x 64 requests per ~6 us beside the row stream.  That, not the survivors' round trips, was what set a
    // 64-query pass apart from a one-query pass: 0.94 -> 0.78 ms at 12.5M rows.)
    float myU = __builtin_inff(), myT = -__builtin_inff();
    auto refresh = [&](bool fetch) {
        const uint32_t tau_now = my_tau;
        if (fetch) my_tau = ld_relaxed(&p.tau_c[ql]);
        const float T = (key_f32(max(my_tau0, tau_now)) - my_e32) * my_sq;
        myU = ql < p.B ? (my_sq != 0.0f ? (T - fabsf(T) * 2e-6f) - c1 : dead) : __builtin_inff();
        myT = ql < p.B && my_sq != 0.0f ? T : -__builtin_inff();
        if (lane < NQ) __hip_atomic_store(&lU[lane], myU, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // the entries of the lanes in `set` (one per lane: ehi, elo), eight at a time
    auto work_set = [&](unsigned long long set, uint32_t ehi, uint32_t elo) {
        bool first = true;
        while (set) {
            if (!first) refresh(true);  // (a round takes microseconds: the streaming waves get the thresholds it raised)
            first = false;
            int src = -1;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                if (set) {
                    const int s0 = __builtin_ctzll(set);
                    set &= set - 1;
                    if (g == j) src = s0;
                }
            }
            const bool on = src >= 0;
            work(on, (uint32_t)__shfl((int)ehi, on ? src : 0), (uint32_t)__shfl((int)elo, on ? src : 0));
        }
    };

    for (;;) {
        refresh((idle & 7u) == 0);

        uint32_t tail = lds_ld(&ring.tail);
        bool over = false;
        if (tail == head) {
            over = lds_ld(&ring.done) == streaming;
            if (over) {
                PCV_LDS_ORDER();
                tail = lds_ld(&ring.tail);  // (a wave signs off after its last reservation)
            }
        }
        if (tail != head) {
            idle = 0;
            const uint32_t n = min(tail - head, 64u);
            const bool have = (uint32_t)lane < n;
            uint32_t ehi = 0, elo = 0;
            int eacc = 0;
            float esb = 0.0f;
            if (have) {
                const uint32_t slot = (head + (uint32_t)lane) & (kRing - 1);
                do ehi = lds_ld(&ring.hi[slot]);
                while (!(ehi & 0x80000000u));
                PCV_LDS_ORDER();
                elo = lds_ld(&ring.lo[slot]);
                eacc = ring.acc[slot];
                esb = ring.sblk[slot];
                PCV_LDS_ORDER();
                lds_st(&ring.hi[slot], 0u);
            }
            head += n;
            PCV_LDS_ORDER();
            if (lane == 0) lds_st(&ring.head, head);  // the producers may go on while these entries are worked on
            const int eq = (int)((ehi >> 5) & 0xffu);
            const float U = __shfl(myU, eq), vq = __shfl(my_vq, eq), Tq = __shfl(myT, eq);
            const float rhs = fmaf(esb, U, -vq);
            const bool keep = have && (float)eacc >= rhs;
            // likely: the estimate clears the threshold by less than a quarter of the margin (margin = s_blk T - rhs)
            const bool likely = keep && (float)eacc >= fmaf(0.25f, rhs, 0.75f * esb * Tq);
            if (have) g_atomic_add(&p.cand_cnt[eq * kHot + 32], 1u);  // statistics: coarse survivors
            const unsigned long long wait = __ballot(keep && !likely);
            const uint32_t nw = (uint32_t)__builtin_popcountll(wait);
            PCV_DCOUNT(4, n)
            PCV_DCOUNT(7, n - (uint32_t)__builtin_popcountll(__ballot(keep)))
            PCV_DMAX(tail - (head - n), sp + nw)
            if (sp + nw <= kStack) {
                if (keep && !likely) {
                    const uint32_t at = sp + (uint32_t)__builtin_popcountll(wait & ((1ull << lane) - 1ull));
                    stk.lo[at] = elo;
                    stk.hi[at] = ehi;
                    stk.acc[at] = eacc;
                    stk.sblk[at] = esb;
                }
                sp += nw;
                work_set(__ballot(likely), ehi, elo);
            } else {
                work_set(__ballot(keep), ehi, elo);
            }
        } else if (sp > 0) {
            // nothing new: the eight youngest waiting entries, against the thresholds of now
            idle = 0;
            const uint32_t n = min(sp, 8u);
            sp -= n;
            const bool have = g < n;
            const uint32_t at = sp + (have ? g : 0u);
            const uint32_t ehi = stk.hi[at], elo = stk.lo[at];
            const int eacc = stk.acc[at];
            const float esb = stk.sblk[at];
            const int eq = (int)((ehi >> 5) & 0xffu);
            const float U = __shfl(myU, eq), vq = __shfl(my_vq, eq);
            const bool keep = have && (float)eacc >= fmaf(esb, U, -vq);
            PCV_DCOUNT(7, n - (uint32_t)__builtin_popcountll(__ballot(keep && sub == 0)))
            if (__any(keep)) work(keep, ehi, elo);
        } else if (over) {
            break;
        } else {
            ++idle;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    PCV_DSTAMP(3)
    PCV_DCOUNT(1, __builtin_amdgcn_s_memtime() - clk0)
#undef PCV_DSTAMP
#undef PCV_DCOUNT
#undef PCV_DMAX
}

#ifdef PCV_STAMPS  // diagnostic build (tools/build_stamps.sh): where a wave's time goes; the 100 MHz constant clock
#define PCV_STAMP(slot)                                                                                            \
    if (p.stamps && lane == 0) p.stamps[(size_t)(blockIdx.x * WPB + wave) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime();
#ifdef PCV_STAMPS_TIMELINE  // slots 5, 6, 7 = time at the end of the wave's 8th, 32nd, 96th block instead of the fine-screen counts
#define PCV_COUNT(slot, n) \
    if ((slot) < 5 && p.stamps && lane == 0) p.stamps[(size_t)(blockIdx.x * WPB + wave) * 8 + (slot)] += (n);
#else
#define PCV_COUNT(slot, n) \
    if (p.stamps && lane == 0) p.stamps[(size_t)(blockIdx.x * WPB + wave) * 8 + (slot)] += (n);
#endif
#else
#define PCV_STAMP(slot)
#define PCV_COUNT(slot, n)
#endif
// DRAIN: the last of the workgroup's WPB waves is the drain wave of the survivor ring above, the others stream.
// NCHT: chunks per block as a compile-time constant (0: taken from the pass's dimension at run time).  With the chunk count known
// the loop below is written out over lcm(NCHT, NBUF) steps, and every request has its place in program order: the thresholds
// and the scale of a block are asked for two steps before its test, 8 row loads behind them — the test waits with
// s_waitcnt vmcnt(8).  In the run-time form the compiler cannot relate `cons.ch == NCH - 2` (where the request is made) to
// `++cons.ch == NCH` (where it is used), has to allow for the one-chunk case that asks inside the test, and ends EVERY block
// with s_waitcnt vmcnt(0): each block then waited for the chunk requested a moment before.
// (Measured and dropped, round 4: the thresholds through LDS for the forms WITHOUT a drain wave as well — the 128-query tile at
// 385..1024 features — with wave 0 of the workgroup as their keeper: it fetched the side-by-side copy at every chunk step through a
// buffer descriptor that is empty for the other waves (an out-of-range load makes no memory request, and all waves execute the same
// instructions) and renewed U for all queries.  Same survivors as with every wave fetching for itself, 3.20 against 3.01 ms at
// 30M x 512 / 128 queries, 8.4 against 7.0 ms at 50M x 768 while U was still read two steps before the test: there a block is
// 16-24 KB and takes tens of microseconds, the 128-line gather per block is not what bounds it, and a keeper that is itself
// busy in the fine screen keeps everybody's thresholds waiting.)
template <int NT, bool NTL, int WPB, int NBUF, bool DRAIN, int NCHT>
__global__ __launch_bounds__(WPB * 64, NT == 4 ? 2 : 3) void scan_mfma8_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    constexpr uint32_t SW = DRAIN ? WPB - 1 : WPB;  // streaming waves of a workgroup
    constexpr bool ULDS = DRAIN;                    // the test's U_q comes from LDS
    extern __shared__ uint4 lq8[];  // [NT*32][LDQ] pieces of 16 int8
    const int D4 = p.D4;
    const int Dp = D4 * 4;
    const int Dp8 = (Dp + 127) & ~127;
    const int P16 = NCHT ? NCHT * 8 : Dp8 >> 4;   // pieces per row
    const int LDQ = P16 + 1;    // odd
    const int NCH = NCHT ? NCHT : Dp8 >> 7;   // chunks of 128 features
    __shared__ uint32_t ltau0[NT * 32];
    __shared__ float lsq[NT * 32], lvq[NT * 32];
    __shared__ uint32_t ring_words[DRAIN ? sizeof(SurvRing) / 4 : 4];
    SurvRing& ring = *(SurvRing*)ring_words;  // (touched in the DRAIN form only)
    __shared__ uint32_t stack_words[DRAIN ? sizeof(DrainStack) / 4 : 4];
    DrainStack& stack = *(DrainStack*)stack_words;
    __shared__ float lU[NT * 32];  // DRAIN form: U_q of the test (below) from the thresholds as the drain wave last saw them
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    PCV_STAMP(0)
    if constexpr (DRAIN) {
        for (uint32_t i = threadIdx.x; i < kRing; i += WPB * 64) ring.hi[i] = 0u;
        if (threadIdx.x < 4) (&ring.tail)[threadIdx.x] = 0u;
    }
    for (int i = threadIdx.x; i < NT * 32 * P16; i += WPB * 64) {  // the tile quantize_queries_kernel prepared
        const int q = i / P16, pc = i - q * P16;
        lq8[(size_t)q * LDQ + pc] = __builtin_bit_cast(uint4, gld4((const float4*)p.q8 + i));
    }
    for (int q = threadIdx.x; q < NT * 32; q += WPB * 64) {
        lsq[q] = gld(&p.q8c[4 * q]);
        lvq[q] = gld(&p.q8c[4 * q + 1]);
    }
    for (int q = threadIdx.x >> 2; q < NT * 32; q += WPB * 16) {
        const uint32_t key = seed_threshold_key(p, q, threadIdx.x & 3);
        if ((threadIdx.x & 3) == 0) ltau0[q] = key;
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const float nrm = (p.metric == PCV_METRIC_DOT) ? p.max_norm : 1.0f;
    const float c1 = 0.5002f * sqrtf((float)Dp8) * nrm;
    const float dead_u = (p.metric == PCV_METRIC_DOT) ? -__builtin_inff() : __builtin_inff();
    // U_q of the test for the running threshold `tau_key` (derivation: the epilogue below)
    auto u_of = [&](int q, uint32_t tau_key, float e32q) -> float {
        const float sqq = lsq[q];
        const float T = (key_f32(max(ltau0[q], tau_key)) - e32q) * sqq;
        return (q < p.B) ? (sqq != 0.0f ? (T - fabsf(T) * 2e-6f) - c1 : dead_u) : __builtin_inff();
    };
    if constexpr (ULDS) {  // the first U of every query (the drain wave keeps them fresh from here on)
        for (int q = threadIdx.x; q < NT * 32; q += WPB * 64) lU[q] = u_of(q, ld_relaxed(&p.tau_c[q]), q < p.B ? 0.5f * gld(&p.margin32[q]) : 0.0f);
        __syncthreads();
    }
    float sq[NT], vq[NT], e32[NT];
    uint32_t tau0[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const bool live = 32 * t + c < p.B;
        sq[t] = lsq[32 * t + c];
        vq[t] = lvq[32 * t + c];
        e32[t] = live ? 0.5f * gld(&p.margin32[32 * t + c]) : 0.0f;
        tau0[t] = ltau0[32 * t + c];
    }

    if constexpr (DRAIN) {
        if (wave == SW) {
            drain_survivors<NT * 32>(p, ring, stack, ltau0, lsq, lvq, lU, lane, D4, SW,
                                     p.stamps ? p.stamps + (size_t)(blockIdx.x * WPB + wave) * 8 : nullptr);
            return;
        }
    }
    // a streaming wave's last act in the DRAIN form: sign off (after its last ring entry: LDS operations stay in order)
#define PCV_WAVE_DONE()                               \
    {                                                 \
        if constexpr (DRAIN) {                        \
            PCV_LDS_ORDER();                          \
            if (lane == 0) lds_add(&ring.done, 1u);   \
        }                                             \
        return;                                       \
    }
    const uint32_t total_waves = gridDim.x * SW;
    if (blockIdx.x * SW + wave >= p.total_blocks) PCV_WAVE_DONE()
    PCV_STAMP(1)

    i32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0;

    auto enter_block = [&](BlockCursor& k, uint32_t gb) __attribute__((always_inline)) {
        k.gb = gb;
        k.ch = 0;
        if (gb < p.total_blocks) {
            seek_seg(p, k.sc, gb);
            k.lb = gb - k.sc.begin;
#if PCV_EXP == 7 || PCV_EXP == 9  // timing experiment (wrong results): the rows of 256 blocks over and over — they come out of the L2
            k.rows = row_rsrc((const float4*)k.sc.blk8 + (size_t)(k.lb & 255u) * P16 * 32, (uint32_t)P16 * 512u);
#else
            k.rows = row_rsrc((const float4*)k.sc.blk8 + (size_t)k.lb * P16 * 32, (uint32_t)P16 * 512u);
#endif
        }
    };
    const uint32_t lane_off = (uint32_t)(h * 32 + c) * 16u;
    BlockCursor cons, prod;
    enter_block(cons, blockIdx.x * SW + wave);
    prod = cons;

    float4 buf[NBUF][4];
    // lane's piece of k-step ks of a chunk: f16 = chunk*8 + ks*2 + h  (h folded into base)
    auto produce = [&](float4 (&b)[4]) __attribute__((always_inline)) {
        // ALWAYS issues its loads — past the end of the wave's stream they re-read a chunk of its last block (two or three
        // chunks per wave and launch).  With an early return here the compiler has to place every s_waitcnt for the case that
        // the younger chunks were never requested: each multiply then waited for (nearly) all loads in flight, the ones just
        // issued included, and the chunk buffers hid nothing.
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)i * 1024u, (uint32_t)prod.ch * 4096u);
        if (prod.gb < p.total_blocks && ++prod.ch == NCH) enter_block(prod, prod.gb + total_waves);  // (the descriptor stays when the stream is over)
    };

    // thresholds and the quantisation scale of the block being finished, requested one chunk ahead of the epilogue:
    // lane (c, h) tests rows 4h + {0..3, 8..11, 16..19, 24..27} of the block
    uint32_t tauk[NT];
    float sblk = 0.0f;
    auto prefetch = [&]() __attribute__((always_inline)) {  // (before the step's row loads and untouched until the epilogue: see scan_mfma_kernel)
        if constexpr (ULDS) {
            // (U itself is read in the epilogue, at the moment of the test: an LDS read costs the block ~100 cycles, and while the
            // thresholds rise fast — the first blocks of a pass — a value read two chunk steps earlier let twice the rows through)
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) tauk[t] = ld_relaxed(&p.tau[(32 * t + c) * kHot]);
        }
        sblk = gld(cons.sc.scale8 + cons.lb);
    };

#ifdef PCV_STAMPS
    unsigned long long stamp_prev = __builtin_amdgcn_s_memrealtime();
    uint32_t stamp_nblk = 0;
#endif
    auto epilogue = [&](const SegCursor& esc, uint32_t elb) __attribute__((always_inline)) {
        if constexpr (NCHT == 0) {
            if (NCH < 2) prefetch();
        }
        float U[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (ULDS) {
                U[t] = __hip_atomic_load(&lU[32 * t + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#if PCV_EXP == 12  // experiment: is the U from LDS behind the thresholds?  (the fresh value beside it)
                {
                    const int q = 32 * t + c;
                    const float Tf = (key_f32(max(tau0[t], ld_relaxed(&p.tau[q * kHot]))) - e32[t]) * sq[t];
                    const float Uf = (q < p.B) ? (sq[t] != 0.0f ? (Tf - fabsf(Tf) * 2e-6f) - c1 : dead_u) : __builtin_inff();
                    U[t] = fmaxf(U[t], Uf);
                }
#endif
                continue;
            }
            const int q = 32 * t + c;
            const float T = (key_f32(max(tau0[t], tauk[t])) - e32[t]) * sq[t];
            // lowered by 2e-6 relative (f32 rounding of T and of the product with s_row) and by the |x^|_1 term
            // s_q = 0: a dead query.  Cosine: no row has a score, none is kept; dot (an all-zero query): every row scores 0,
            // all are kept and the fine screen sorts it out.
            const float dead = (p.metric == PCV_METRIC_DOT) ? -__builtin_inff() : __builtin_inff();
            U[t] = (q < p.B) ? (sq[t] != 0.0f ? (T - fabsf(T) * 2e-6f) - c1 : dead) : __builtin_inff();
        }
        // One scale per block: the right-hand side s_blk U - V is the same for the 16 rows a lane tests, so the largest
        // accumulator decides whether the block has a survivor at all.  Nearly every block ends here.
        float rhs[NT];
        bool hot = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            int m = acc[t][0];
#pragma unroll
            for (int i = 1; i < 16; ++i) m = max(m, acc[t][i]);
            rhs[t] = fmaf(sblk, U[t], -vq[t]);  // (NaN scale: a block without a searchable row, no comparison succeeds)
            hot |= (float)m >= rhs[t];
        }
#if PCV_EXP == 8 || PCV_EXP == 9  // timing experiment (wrong results): every block ends at its test
        asm volatile("" ::"s"(__ballot(hot)));
        if (false) {
#else
        if (__any(hot)) {
#endif
        uint32_t mask[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            mask[t] = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) mask[t] |= ((float)acc[t][i] >= rhs[t]) ? (1u << i) : 0u;
        }
        bool any = false;
#pragma unroll
        for (int t = 0; t < NT; ++t) any |= mask[t] != 0;
        if (__any(any)) {
#ifdef PCV_STAMPS
            const unsigned long long ts = __builtin_amdgcn_s_memrealtime();
#endif
            if constexpr (DRAIN) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (__any(mask[t] != 0)) ring_push(ring, mask[t], acc[t], sblk, (uint32_t)(32 * t + c), (uint32_t)h, (uint32_t)esc.si, elb);
            } else {
#pragma unroll
                for (int t = 0; t < NT; ++t) fine_survivors(p, mask[t], t, esc, elb, ltau0, lane, D4);
            }
#ifdef PCV_STAMPS
            PCV_COUNT(5, __builtin_amdgcn_s_memrealtime() - ts)  // time inside the fine screen
            PCV_COUNT(6, 1)                                      // blocks that reached it
#endif
        }
        PCV_COUNT(7, 1)  // blocks that passed the pre-test
        }
        PCV_COUNT(4, 1)  // blocks
#ifdef PCV_STAMPS
        {   // time since the previous block's end, booked by what this block did: low word of slot 2 = blocks that ended at
            // the block test, high word = blocks with a survivor
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            const unsigned long long dt = now - stamp_prev;
            stamp_prev = now;
            const bool past = __any(hot);  // (all lanes: PCV_COUNT runs on lane 0 only)
            PCV_COUNT(2, past ? (dt << 32) : dt)
#ifdef PCV_STAMPS_TIMELINE
            ++stamp_nblk;  // (low word: the 100 MHz clock; high word: the low 32 bits of the shader clock)
            if (stamp_nblk == 1 || stamp_nblk == 8 || stamp_nblk == 32 || stamp_nblk == 96) {
                const unsigned long long both = (now & 0xffffffffull) | (__builtin_amdgcn_s_memtime() << 32);
                if (p.stamps && lane == 0) p.stamps[(size_t)(blockIdx.x * WPB + wave) * 8 + (stamp_nblk == 1 ? 1 : stamp_nblk == 8 ? 5 : stamp_nblk == 32 ? 6 : 7)] = both;
            }
#endif
        }
#endif
        PCV_STAMP(3)     // (the last one stays: the wave's end)
        // (measured and dropped, against the older workgroups of a CU finishing 4 % before the younger ones: s_setprio rotating
        // block by block through the three workgroups of a CU, 0.946 -> 0.939 ms at 12.5M rows, 6.251 -> 6.227 at 100M; the last
        // 3.6 % of the blocks dealt out 2 : 1 : 0 to the thirds of the launch, 0.941 against 0.936 ms and 6.291 against 6.289.
        // Nor does it matter how much a wave reads in one piece: 2, 4 or 8 consecutive blocks per visit instead of one, 6.30 ms each.)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0;
    };

    auto consume = [&](const float4 (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const i32x4 a = __builtin_bit_cast(i32x4, b[ks]);
            const int pc = 2 * (cons.ch * 4 + ks) + h;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const i32x4 q8 = *(const i32x4*)&lq8[(size_t)(32 * t + c) * LDQ + pc];
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, q8, acc[t], 0, 0, 0);
            }
        }
        if (++cons.ch == NCH) {
            epilogue(cons.sc, cons.lb);
            enter_block(cons, cons.gb + total_waves);
        }
    };

    if constexpr (NCHT > 0) {
        // chunk count known: lcm(NCHT, NBUF) steps written out; step s consumes chunk s % NCHT of its block out of buffer
        // s % NBUF while chunk (s + NBUF - 1) % NCHT of a later block is requested into the buffer consumed a step ago
        constexpr int PERIOD = NCHT * NBUF / std::gcd(NCHT, NBUF);
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[i][j] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)j * 1024u, (uint32_t)(i % NCHT) * 4096u);
            if ((i % NCHT) == NCHT - 1 && prod.gb < p.total_blocks) enter_block(prod, prod.gb + total_waves);
        }
        auto step = [&](auto S) __attribute__((always_inline)) -> bool {  // true: the wave's stream is over
                constexpr int s = decltype(S)::value;
                constexpr int ch = s % NCHT, pch = (s + NBUF - 1) % NCHT;
                if constexpr (NCHT >= 2 ? ch == NCHT - 2 : true) prefetch();
                {   // produce (always: see `produce`)
                    float4(&b)[4] = buf[(s + NBUF - 1) % NBUF];
#pragma unroll
                    for (int j = 0; j < 4; ++j) b[j] = ld_piece<NTL>(prod.rows, lane_off + (uint32_t)j * 1024u, (uint32_t)pch * 4096u);
                    if constexpr (pch == NCHT - 1) {
                        if (prod.gb < p.total_blocks) enter_block(prod, prod.gb + total_waves);
                    }
                }
                {   // consume
                    const float4(&b)[4] = buf[s % NBUF];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const i32x4 a = __builtin_bit_cast(i32x4, b[ks]);
                        const int pc = 2 * (ch * 4 + ks) + h;
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const i32x4 q8 = *(const i32x4*)&lq8[(size_t)(32 * t + c) * LDQ + pc];
                            acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, q8, acc[t], 0, 0, 0);
                        }
                    }
                    if constexpr (ch == NCHT - 1) {
                        epilogue(cons.sc, cons.lb);
                        enter_block(cons, cons.gb + total_waves);
                        if (cons.gb >= p.total_blocks) return true;
                    }
                }
                return false;
        };
        while (true) {
            if (static_for_until<0, PERIOD>(step)) PCV_WAVE_DONE()
        }
    }
#define PCV_STEP(REFILL, CONS)                          \
    if (NCH >= 2 && cons.ch == NCH - 2) prefetch();     \
    produce(buf[REFILL]);                               \
    consume(buf[CONS]);                                 \
    if (cons.gb >= p.total_blocks) PCV_WAVE_DONE()
    produce(buf[0]);
    if constexpr (NBUF == 3) {
        produce(buf[1]);
        while (true) {
            PCV_STEP(2, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
        }
    } else {
        static_assert(NBUF == 4, "three or four chunk buffers");
        produce(buf[1]);
        produce(buf[2]);
        while (true) {
            PCV_STEP(3, 0)
            PCV_STEP(0, 1)
            PCV_STEP(1, 2)
            PCV_STEP(2, 3)
        }
    }
#undef PCV_STEP
#undef PCV_WAVE_DONE
}

// 65..256 queries over rows of at most 384 features: the 128-query tile above needs 232 registers (64 accumulators, four
// chunk buffers) and runs at 2 waves per SIMD.  Here a wave HOLDS a block (NCH chunks = all its 12 KB, in the registers the
// chunk buffers took) and multiplies it with the 64-query halves of the tile one after the other, so only 32 accumulators are
// live: 3 waves per SIMD.  NH halves of 64 queries: 2 (up to 128 queries: 256-thread workgroups, three 51 KB tiles per CU) or
// 4 (up to 256 queries: one 768-thread workgroup and one 102 KB tile per CU) — one pass over the rows for 256 queries.
//
// The test of a half reads its right-hand side U_q from LDS: at every block a wave renews the 32 values of ONE tile from the
// running thresholds (tile = (its block count + its number) mod tiles: every tile comes round every block), a half reads two
// words per tile.  A stale U is a threshold that was valid earlier, i.e. a lower one: never wrong, at most a survivor more.
// (Every wave deriving both U of every half from freshly loaded thresholds — ~45 vector instructions per tile and half, and
// the loads under a lane-dependent condition, which the compiler turns into a branch with an s_waitcnt vmcnt(0) inside — took
// 12.2 ms per 100M rows at 256 queries; the loads made unconditional 11.2; U from LDS 9.9; with the requests below 9.6.)
//
// Measured and not kept (100M x 384, 256 queries, same box each): the query pieces read from LDS three or five multiplies
// ahead, order pinned with scheduling barriers: 12.18 against 12.28 ms (the waves do not wait on LDS); two held blocks per
// wave at 2 waves per SIMD, the whole next block in flight: 11.27 against 11.10 ms (a third wave per SIMD hides as much);
// the waves of a SIMD started a third of a half apart: 9.83 against 9.80 ms.  (385..768 features at 65..128 queries in this
// form — one 768-thread workgroup per CU around the 98 KB tile, a 24 KB block held in 96 registers — does not fit 168 registers:
// 42 spilled, the chunk buffers among them.  Those shapes stay on scan_mfma8_kernel<4, .., 8>.)  With no test at all the pass takes 8.7 ms,
// and 7.3 ms with the rows coming from L2: the multiplies themselves, at the clock the chip holds under them (1.6 GHz).
template <bool NTL, int NCH, int NH>
__global__ __launch_bounds__(NH == 2 ? 256 : 768, 3) void scan_mfma8_hold_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    constexpr int TQ = NH * 64;       // queries of the tile
    constexpr int WPB = NH == 2 ? 4 : 12;
    extern __shared__ uint4 lq8[];  // [TQ][LDQ] pieces of 16 int8
    constexpr int P16 = NCH * 8;    // pieces per row
    constexpr int LDQ = P16 + 1;    // odd
    const int D4 = p.D4;
    __shared__ uint32_t ltau0[TQ];
    __shared__ float lsq[TQ], lvq[TQ], le32[TQ];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < TQ * P16; i += WPB * 64) {  // the tile quantize_queries_kernel prepared
        const int q = i / P16, pc = i - q * P16;
        lq8[(size_t)q * LDQ + pc] = __builtin_bit_cast(uint4, gld4((const float4*)p.q8 + i));
    }
    for (int q = threadIdx.x; q < TQ; q += WPB * 64) {
        lsq[q] = gld(&p.q8c[4 * q]);
        lvq[q] = gld(&p.q8c[4 * q + 1]);
        le32[q] = q < p.B ? 0.5f * gld(&p.margin32[q]) : 0.0f;
    }
    for (int q = threadIdx.x >> 2; q < TQ; q += WPB * 16) {
        const uint32_t key = seed_threshold_key(p, q, threadIdx.x & 3);
        if ((threadIdx.x & 3) == 0) ltau0[q] = key;
    }
    __syncthreads();
    const int c = lane & 31, h = lane >> 5;
    const float nrm = (p.metric == PCV_METRIC_DOT) ? p.max_norm : 1.0f;
    const float c1 = 0.5002f * sqrtf((float)(NCH * 128)) * nrm;
    const float dead = (p.metric == PCV_METRIC_DOT) ? -__builtin_inff() : __builtin_inff();
    __shared__ float lU[TQ];
    // U_q of the test for the running threshold `tau_key` (derivation: scan_mfma8_kernel)
    auto u_of = [&](int q, uint32_t tau_key) -> float {
        const float sq = lsq[q];
        const float T = (key_f32(max(ltau0[q], tau_key)) - le32[q]) * sq;
        const float live = (T - fabsf(T) * 2e-6f) - c1;
        return q < p.B ? (sq != 0.0f ? live : dead) : __builtin_inff();
    };
    for (int q = threadIdx.x; q < TQ; q += WPB * 64) lU[q] = u_of(q, ld_relaxed(&p.tau[q * kHot]));
    __syncthreads();

    const uint32_t total_waves = gridDim.x * WPB;
    struct Cur {
        uint32_t gb;
        SegCursor sc;
        uint32_t lb;
    } cur;
    cur.gb = blockIdx.x * WPB + wave;
    if (cur.gb >= p.total_blocks) return;
    uint32_t renew = wave % (uint32_t)(TQ / 32);  // tile whose U this wave renews at its next block

    float4 buf[NCH][4];
    float sblk = 0.0f;  // quantisation scale of the block held (scan.h)
    __amdgpu_buffer_rsrc_t rows;  // the block's bytes (see ld_piece)
    const uint32_t lane_off = (uint32_t)(h * 32 + c) * 16u;
    auto enter = [&]() {  // the block `cur` points at
        seek_seg(p, cur.sc, cur.gb);
        cur.lb = cur.gb - cur.sc.begin;
        rows = row_rsrc((const float4*)cur.sc.blk8 + (size_t)cur.lb * P16 * 32, (uint32_t)P16 * 512u);
    };
    auto load_chunk = [&](int ch) {
#pragma unroll
        for (int i = 0; i < 4; ++i) buf[ch][i] = ld_piece<NTL>(rows, lane_off + (uint32_t)i * 1024u, (uint32_t)ch * 4096u);
    };
    auto load_scales = [&]() {  // of the block (see scan_mfma8_kernel)
        sblk = gld(cur.sc.scale8 + cur.lb);
    };
    enter();
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) load_chunk(ch);
    load_scales();
    // The first NA chunks of the NEXT block are asked for a whole block ahead, the others in the last half, each as soon as the
    // multiplies of the chunk it replaces are out (the last half takes those chunks first).  A new block used to wait for its
    // first chunk, requested only two thirds of a half before its first use.  Where a chunk waits for a block: in four
    // registers (nb0, nb1: the 128-query form has them), or — 256 queries, where the registers do not reach — chunk 0 moves to
    // 4 KB of LDS per wave in the middle of the block and nb0 takes chunk 1.  All loads of the loop are unconditional — past
    // the end of the wave's stream they read its last block again — so that the compiler can count them (scan_mfma_kernel).
    constexpr int NA = NCH >= 2 ? 2 : 1;
    constexpr bool VIA_LDS = NA == 2 && NH == 4;
    float4 nb0[4], nb1[4];
    uint4* const stage = lq8 + (size_t)TQ * LDQ + (size_t)wave * 256 + lane;  // VIA_LDS: this wave's 4 KB behind the tile

    while (true) {
        const SegCursor esc = cur.sc;  // the block held: where its survivors live, and the scales of its rows
        const uint32_t elb = cur.lb;
        const float sblk_e = sblk;
        cur.gb += total_waves;  // the next block: its cursor and descriptor now
        const bool more = cur.gb < p.total_blocks;
        if (more) enter();
#pragma unroll
        for (int i = 0; i < 4; ++i) nb0[i] = ld_piece<NTL>(rows, lane_off + (uint32_t)i * 1024u, 0u);
        if constexpr (NA == 2 && !VIA_LDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) nb1[i] = ld_piece<NTL>(rows, lane_off + (uint32_t)i * 1024u, 4096u);
        }
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            const int q0 = 64 * half + c;  // this lane's queries: q0, q0 + 32
            uint32_t rtau = 0u;  // (every lane loads, and the value is not touched before the end of the half: see scan_mfma_kernel)
            if (half == 0) rtau = ld_relaxed(&p.tau[(32 * renew + c) * kHot]);
            i32x16 acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][i] = 0;
            if (VIA_LDS && half == NH / 2) {  // chunk 0 of the next block has long arrived: to LDS with it, its registers ask for chunk 1
#pragma unroll
                for (int i = 0; i < 4; ++i) stage[i * 64] = __builtin_bit_cast(uint4, nb0[i]);
#pragma unroll
                for (int i = 0; i < 4; ++i) nb0[i] = ld_piece<NTL>(rows, lane_off + (uint32_t)i * 1024u, 4096u);
            }
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                // the last half takes the chunks whose successors are only requested now first (integer sums: any order)
                const int ch = half == NH - 1 ? (cc + NA < NCH ? cc + NA : cc + NA - NCH) : cc;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const i32x4 a = __builtin_bit_cast(i32x4, buf[ch][ks]);
                    const int pc = 2 * (ch * 4 + ks) + h;
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const i32x4 q8 = *(const i32x4*)&lq8[(size_t)(q0 + 32 * t) * LDQ + pc];
                        acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, q8, acc[t], 0, 0, 0);
                    }
                }
                if (half == NH - 1) {  // last half: the chunk's registers are free, the next block's chunk follows at once
                    if (ch >= NA) {
                        load_chunk(ch);
                    } else if (ch == 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) buf[0][i] = VIA_LDS ? __builtin_bit_cast(float4, stage[i * 64]) : nb0[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) buf[NCH > 1 ? 1 : 0][i] = VIA_LDS ? nb0[i] : nb1[i];  // (NCH == 1: ch < NA = 1 means ch == 0, never here)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler would fetch all 2 * 4 * NCH query pieces first: 190 registers)
            }
            if (half == NH - 1) load_scales();
            // epilogue of this half (scan_mfma8_kernel has the derivation)
            float U[2], vq[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {  // (every LDS word is read whatever the lane's query is: selects, not branches around reads)
                const int q = q0 + 32 * t;
                vq[t] = lvq[q];
                U[t] = lU[q];
            }
            if (half == 0) {  // (after the reads of this half: the renewed values serve the halves to come)
                lU[32 * renew + c] = u_of(32 * (int)renew + c, rtau);
                renew = renew + 1 == (uint32_t)(TQ / 32) ? 0u : renew + 1;
            }
            float rhs[2];
            bool hot = false;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                int m = acc[t][0];
#pragma unroll
                for (int i = 1; i < 16; ++i) m = max(m, acc[t][i]);
                rhs[t] = fmaf(sblk_e, U[t], -vq[t]);
                hot |= (float)m >= rhs[t];
            }
#if PCV_EXP == 8 || PCV_EXP == 9
            asm volatile("" ::"s"(__ballot(hot)));
            if (false) {
#else
            if (__any(hot)) {  // (one scale per block: a block that gets here has a survivor)
#endif
                uint32_t mask[2] = {0u, 0u};
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int i = 0; i < 16; ++i) mask[t] |= ((float)acc[t][i] >= rhs[t]) ? (1u << i) : 0u;
#pragma unroll
                for (int t = 0; t < 2; ++t) fine_survivors(p, mask[t], 2 * half + t, esc, elb, ltau0, lane, D4);
            }
        }
        if (!more) return;
    }
}

__device__ __forceinline__ bool better(double sa, int64_t pa, double sb, int64_t pb) {
    return sa > sb || (sa == sb && pa < pb);
}

// Exact ranking of one query's survivors, one workgroup per query:
//   1. keep the survivors the FINAL threshold still admits (rows emitted while tau was low);
//   2. canonical score of each: f64, products exact, sums in feature order — the definition of
//      oracle/scan.c:orc_canonical_score.  COOP (dim <= 448): every wave takes 8 survivors at a time,
//      8 lanes fetch a row's pieces together into LDS (all loads in flight at once) and one lane per
//      row runs the feature-order sums from there; otherwise a thread per survivor reads its row
//      from global memory;
//   3. rank: descending score, ties -> lower global position.  Scored survivors collect in an LDS buffer
//      of kSelCap entries (the normal case is a few dozen) and are ranked there by counting, each thread
//      the rows that beat its own.  A list longer than the buffer (thousands of rows within 1e-4 of the
//      k-th best, or an adversarial row order) is consumed in slices of 1024: when the next slice might
//      not fit, the buffer is cut down to its k best, whose last entry becomes an exact admission
//      threshold for everything scored later;
//   4. write the k hits to the device list and, if asked, straight into pinned host memory; report the
//      uncapped survivor count; leave the per-query scan state clean for the next pass.
constexpr int kSelCap = 2048;
constexpr int kCoopMaxD4 = 112;  // LDS: (D4 + 32*(D4+1)) * 16 B <= 58 KB
template <bool COOP>
__global__ __launch_bounds__(256) void rescore_select_kernel(const ScanParams* __restrict__ pp) {
    const ScanParams& p = *pp;
    extern __shared__ float4 lds4[];  // [D4] raw query | COOP: 4 waves x 8 slots x (D4+1) row pieces
    __shared__ double c_s[kSelCap];
    __shared__ int64_t c_p[kSelCap];
    __shared__ uint64_t c_e[kSelCap];   // (segment << 32) | row of the entry
    __shared__ double k_s[kMaxK];
    __shared__ int64_t k_p[kMaxK];
    __shared__ uint64_t k_e[kMaxK];
    __shared__ uint64_t surv[1024];  // candidates of the current slice that pass the final threshold
    __shared__ uint32_t nsurv, n_valid, n_spec;  // n_spec: survivors certainly at or above the speculative threshold
    __shared__ double s_nq, t_s;  // canonical |q|^2; score of the admission threshold
    __shared__ int64_t t_p;       // ... and its position
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D4 = p.D4;
    const uint32_t raw_cnt = ld_relaxed(&p.cand_cnt[q * kHot]);
    const uint32_t cnt = min(raw_cnt, p.cand_cap);
    const uint64_t* cand = p.cand + (size_t)q * p.cand_cap;
    const float* cs = p.cand_s + (size_t)q * p.cand_cap;
    float4* sq = lds4;
    for (int i = tid; i < D4; i += 256) sq[i] = ((const float4*)(p.qraw + (size_t)q * D4 * 4))[i];
    __shared__ uint32_t s_tau0;
    if (tid >= 64 && tid < 68) {  // (not wave 0: it initialises, not wave 3: it sums the query norm)
        const uint32_t key = seed_threshold_key(p, q, tid & 3);
        if (tid == 64) s_tau0 = key;
    }
    if (tid == 0) {
        n_valid = 0;
        nsurv = 0;
        n_spec = 0;
        t_s = -__builtin_inf();
        t_p = INT64_MAX;
    }
    __syncthreads();
    const float thr_final = key_f32(max(s_tau0, ld_relaxed(&p.tau[q * kHot]))) - p.margin32[q];
    const uint32_t guess = (p.spec_rank > 0 || p.spec_gap == p.spec_gap) ? ld_relaxed(&p.spec[q]) : kKeyNegInf;
    const bool guessed = guess != kKeyNegInf;
    const float thr_spec = guessed ? key_f32(guess) + p.margin32[q] : __builtin_inff();  // f32 score >= this: exact score >= the guess
    if (tid == 255) {  // canonical |q|^2: f64, feature order (the other waves go on to the filter meanwhile)
        double nq = 0.0;
        const float* f = (const float*)sq;
#pragma unroll 16
        for (int i = 0; i < p.D; ++i) nq += (double)f[i] * (double)f[i];
        s_nq = nq;
    }
    const double inf = __builtin_inf();
    auto finish_score = [&](double dot, double nx, double nq) -> double {
        if (p.metric == PCV_METRIC_DOT) return (dot < inf && dot > -inf && nq < inf) ? dot : __builtin_nan("");
        if (nq >= 0x1p-126 && nq < inf && nx >= 0x1p-126 && nx < inf) {
            const double cc = dot / (sqrt(nq) * sqrt(nx));
            if (cc < inf && cc > -inf) return cc;
        }
        return __builtin_nan("");
    };
    // the pass's ceiling for this query (scan.h, CeilRec): only rows that rank strictly after it count
    const double ceil_s = p.ceil ? p.ceil[q].score : __builtin_inf();
    const int64_t ceil_p = p.ceil ? p.ceil[q].pos : -1;
    auto keep = [&](uint64_t e, double score) {  // one survivor's canonical score
        if (!(score == score)) return;  // NaN: undefined score
        const int64_t pos = p.seg[(int)(e >> 32)].pos0 + (int64_t)(uint32_t)e;
        if (!better(ceil_s, ceil_p, score, pos)) return;  // at or before the ceiling
        if (!better(score, pos, t_s, t_p)) return;  // k rows seen earlier are all ahead of it
        const uint32_t slot = atomicAdd(&n_valid, 1u);  // < kSelCap: the buffer had room for a whole slice
        c_s[slot] = score;
        c_p[slot] = pos;
        c_e[slot] = e;
    };
    // rank of buffer entry i among the nv entries (positions are unique: the ranks are a permutation)
    auto rank_of = [&](uint32_t i, uint32_t nv) {
        const double s = c_s[i];
        const int64_t pos = c_p[i];
        int rank = 0;
        for (uint32_t j = 0; j < nv; ++j) rank += better(c_s[j], c_p[j], s, pos) ? 1 : 0;
        return rank;
    };
    for (uint32_t base = 0; base < cnt; base += 1024) {
        if (n_valid + 1024u > (uint32_t)kSelCap) {  // same value in every thread: read after a barrier
            const uint32_t nv = n_valid;
            for (uint32_t i = tid; i < nv; i += 256) {
                const int r = rank_of(i, nv);
                if (r < p.k) {
                    k_s[r] = c_s[i];
                    k_p[r] = c_p[i];
                    k_e[r] = c_e[i];
                }
            }
            __syncthreads();
            for (int i = tid; i < p.k; i += 256) {  // nv > kSelCap - 1024 >= k
                c_s[i] = k_s[i];
                c_p[i] = k_p[i];
                c_e[i] = k_e[i];
            }
            if (tid == 0) {
                n_valid = (uint32_t)p.k;
                t_s = k_s[p.k - 1];
                t_p = k_p[p.k - 1];
            }
            __syncthreads();
        }
        for (uint32_t j = base + tid; j < min(cnt, base + 1024u); j += 256) {
            const float sj = cs[j];
            const uint64_t ej = cand[j];  // fetched with the score, not after the test
            if (!(sj < thr_final)) surv[atomicAdd(&nsurv, 1u)] = ej;
            if (guessed && sj >= thr_spec) atomicAdd(&n_spec, 1u);
        }
        __syncthreads();  // nsurv is final; s_nq is there
        const uint32_t ns = nsurv;
        const double nq = s_nq;
        if constexpr (COOP) {
            float4* rows = lds4 + D4 + (size_t)wave * 8 * (D4 + 1);
            const int slot = lane >> 3, part = lane & 7;
            for (uint32_t g = wave * 8; g < ns; g += 32) {
                const uint32_t si = g + slot;
                const bool live = si < ns;
                const uint64_t e = live ? surv[si] : 0;
                if (live) {
                    const SegDesc& sg = p.seg[(int)(e >> 32)];
                    const uint32_t row = (uint32_t)e;
                    const float4* src = sg.blk + (size_t)(row >> 5) * D4 * 32 + (row & 31);
                    // eight pieces per lane requested before any is stored (a load-store loop runs at one
                    // memory round trip per piece)
                    for (int f0 = part; f0 < D4; f0 += 64) {
                        float4 t[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) t[u] = f0 + 8 * u < D4 ? gld4(src + (size_t)(f0 + 8 * u) * 32) : make_float4(0, 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (f0 + 8 * u < D4) rows[slot * (D4 + 1) + f0 + 8 * u] = t[u];
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_s_waitcnt(0);  // the row pieces of this wave are in LDS
                // the two feature-order chains of a row run on two lanes: part 0 sums q*x, part 1 sums x*x
                // (one instruction stream: the multiplicand is selected, not branched on)
                double chain = 0.0;
                if (live && part < 2) {
                    const float4* r = rows + slot * (D4 + 1);
#pragma unroll 8
                    for (int f4 = 0; f4 < D4; ++f4) {  // unrolled: the LDS reads run ahead of the dependent f64 chain
                        const float4 v = r[f4];
                        const float4 m = part == 0 ? sq[f4] : v;
                        chain += (double)m.x * (double)v.x;
                        chain += (double)m.y * (double)v.y;
                        chain += (double)m.z * (double)v.z;
                        chain += (double)m.w * (double)v.w;
                    }
                }
                const double nx = __shfl(chain, (lane & ~7) | 1);
                if (live && part == 0) keep(e, finish_score(chain, nx, nq));
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            for (uint32_t si = tid; si < ns; si += 256) {
                const uint64_t e = surv[si];
                const SegDesc& sg = p.seg[(int)(e >> 32)];
                const uint32_t row = (uint32_t)e;
                const float4* src = sg.blk + (size_t)(row >> 5) * D4 * 32 + (row & 31);
                double dot = 0.0, nx = 0.0;
#pragma unroll 8
                for (int f4 = 0; f4 < D4; ++f4) {
                    const float4 v = src[(size_t)f4 * 32];
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
                keep(e, finish_score(dot, nx, nq));
            }
        }
        __syncthreads();
        if (tid == 0) nsurv = 0;
        __syncthreads();
    }
    const uint32_t nv = n_valid;
    pcv_hit_dev* out = p.out + (size_t)q * p.k;
    pcv_hit_dev* outh = p.out_host ? p.out_host + (size_t)q * p.k : nullptr;
    auto put = [&](int j, const pcv_hit_dev& hit) {
        out[j] = hit;
        if (outh) outh[j] = hit;
    };
    for (uint32_t i = tid; i < nv; i += 256) {
        const int r = rank_of(i, nv);
        if (r < p.k) {
            const uint64_t e = c_e[i];
            const SegDesc& sg = p.seg[(int)(e >> 32)];
            const uint32_t row = (uint32_t)e;
            pcv_hit_dev hit;
            hit.score = c_s[i];
            hit.pos = c_p[i];
            hit.id = sg.ids ? sg.ids[row] : sg.id0 + (int64_t)row;
            put(r, hit);
            if (r == p.k - 1 && p.kth_host) p.kth_host[q] = (float)c_s[i];
        }
    }
    if (tid == 0 && nv < (uint32_t)p.k && p.kth_host) p.kth_host[q] = __builtin_nanf("");
    pcv_hit_dev none;
    none.score = __builtin_nan("");
    none.pos = -1;
    none.id = -1;
    for (int j = (int)nv + tid; j < p.k; j += 256) put(j, none);
    // survivor count (uncapped: the host sizes a rerun from it), overflow record, clean state
    if (tid == 0) {
        const bool failed = guessed && n_spec < (uint32_t)p.k;  // (after the last slice's barrier)
        if (p.cnt_host) p.cnt_host[q] = failed ? kSpecFailed : raw_cnt;
        if (p.coarse_host) {
            p.coarse_host[q] = ld_relaxed(&p.cand_cnt[q * kHot + 32]);
            p.coarse_host[kMfmaQueries + q] = ld_relaxed(&p.cand_cnt[q * kHot + 33]);
        }
        st_relaxed(&p.cand_cnt[q * kHot + 32], 0u);
        st_relaxed(&p.cand_cnt[q * kHot + 33], 0u);
        if ((raw_cnt > p.cand_cap || failed) && p.flag_rec) p.flag_rec->pos = 1;
        st_relaxed(&p.tau[q * kHot], kKeyNegInf);
        st_relaxed(&p.tau_c[q], kKeyNegInf);
        st_relaxed(&p.cand_cnt[q * kHot], 0u);
    }
    for (int j = tid; j < p.k; j += 256) st_relaxed(&p.slots[(size_t)q * kMaxK + j], kKeyNegInf);
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
#define PCV_LAUNCHED() PCV_HIP(hipGetLastError())

void launch_pack_rows(hipStream_t st, const float* rows, int64_t n, int D, int D4, float4* blk, uint32_t row0) {
    if (n <= 0) return;
    const uint32_t first_blk = row0 >> 5;
    const uint32_t last_blk = (uint32_t)((row0 + n - 1) >> 5);
    const int64_t threads = (int64_t)(last_blk - first_blk + 1) * D4 * 32;
    if (threads > (int64_t)0xffffff00u)
        PCV_FAIL(PCV_ERR_INTERNAL, "pack_rows: %lld rows in one staging step (callers stage <= 2^18 rows per call)", (long long)n);
    pack_rows_kernel<<<cdiv64(threads, 256), 256, 0, st>>>(rows, n, D, D4, blk, row0);
    PCV_LAUNCHED();
}

void launch_iota_ids(hipStream_t st, int64_t* ids, int64_t first, int64_t n) {
    if (n <= 0) return;
    iota_ids_kernel<<<cdiv64(n, 256), 256, 0, st>>>(ids, first, n);
    PCV_LAUNCHED();
}

void launch_row_scales(hipStream_t st, const float4* blk, uint32_t first_block, uint32_t nblocks, uint32_t nrows, int D4,
                       int metric, float* scale, uint32_t* max_norm_bits) {
    if (nblocks <= first_block) return;
    const uint32_t r0 = first_block * 32, r1 = nblocks * 32;
    row_scales_kernel<<<cdiv64((int64_t)r1 - r0, 256), 256, 0, st>>>(blk, r0, r1, nrows, D4, metric, scale, max_norm_bits);
    PCV_LAUNCHED();
}

void launch_synth_fill(hipStream_t st, float4* blk, uint32_t nrows, uint32_t row0, int D, int D4, uint64_t seed,
                       int64_t first_row, int normalize, uint32_t n_clusters, float noise, float amp_lo, float amp_hi) {
    if (nrows == 0) return;
    const int D4src = D / 4;
    const SynthShape sh{n_clusters, noise, 1.0f / sqrtf((float)D), amp_lo, amp_hi > amp_lo ? amp_hi - amp_lo : (amp_hi == amp_lo && amp_lo > 0.0f ? 0.0f : -1.0f)};
    float* inv = nullptr;
    if (normalize) {
        PCV_HIP(hipMallocAsync((void**)&inv, (size_t)nrows * sizeof(float), st));
        synth_inv_kernel<<<cdiv64(nrows, 256), 256, 0, st>>>(nrows, D4src, seed, first_row, sh, inv);
    }
    const uint32_t first_blk = row0 >> 5;
    const uint32_t last_blk = (uint32_t)(((uint64_t)row0 + nrows - 1) >> 5);
    const int64_t threads = (int64_t)(last_blk - first_blk + 1) * D4src * 32;
    const unsigned grid = (unsigned)std::min<int64_t>((threads + 255) / 256, 1 << 20);
    synth_fill_kernel<<<grid, 256, 0, st>>>(blk, nrows, row0, D4src, D4, seed, first_row, sh, inv, threads);
    const hipError_t e = hipGetLastError();
    if (inv) (void)hipFreeAsync(inv, st);
    PCV_HIP(e);
}

void launch_gather_rows(hipStream_t st, const SegDesc* d_segs, int nseg, const int64_t* d_pos, int64_t n, int D,
                        int D4, float* out_rows, int64_t* out_ids) {
    if (n <= 0) return;
    const int64_t threads = n * ((D + 3) / 4);
    gather_rows_kernel<<<cdiv64(threads, 256), 256, 0, st>>>(d_segs, nseg, d_pos, n, D, D4, out_rows, out_ids);
    PCV_LAUNCHED();
}

void launch_reset_scan_state(hipStream_t st, uint32_t* tau, uint32_t* slots, uint32_t* cand_cnt) {
    reset_scan_state_kernel<<<kMfmaQueries * kMaxK / 256, 256, 0, st>>>(tau, slots, cand_cnt);
    PCV_LAUNCHED();
}

void launch_upload(hipStream_t st, const void* src_pinned, void* dst, size_t bytes) {
    const uint32_t n16 = (uint32_t)((bytes + 15) / 16);
    if (n16 == 0) return;
    upload_kernel<<<(n16 + 255) / 256, 256, 0, st>>>((const uint4*)src_pinned, (uint4*)dst, n16);
    PCV_LAUNCHED();
}

void launch_prep_seed(hipStream_t st, const ScanParams& p, const ScanParams* dp, const SegDesc& seg0) {
    const int nparts = std::max(1, (int)((p.seed_blocks * 32u + kSeedPartRows - 1) / kSeedPartRows));
    const uint32_t nseed = seg0.nrows;  // rows of the segment: the seed blocks are spread over it (seed_shift)
    const size_t Dp = (size_t)p.D4 * 4;
    const size_t lds1 = Dp * sizeof(float) + kMaxK * sizeof(uint32_t);
    const size_t lds_mfma = 32 * (Dp + 4) * sizeof(float) + 32 * (size_t)p.k * sizeof(uint32_t);
    if (lds_mfma <= 156 * 1024 && !(p.flags & 4)) {  // flag bit 2: the VALU form (tuning / comparison)
        allow_dynamic_lds((const void*)prep_seed_mfma_kernel, lds_mfma);
        const unsigned parts = (p.seed_blocks + 3) / 4;
        prep_seed_mfma_kernel<<<dim3(parts ? parts : 1, (p.B + 31) / 32), 256, lds_mfma, st>>>(dp, seg0.blk, seg0.scale, nseed);
    } else if (p.B <= 2 || 8 * lds1 > 64 * 1024) {  // few queries, or very wide rows: one query per workgroup
        prep_seed_kernel<1><<<dim3(nparts, p.B), 256, lds1, st>>>(dp, seg0.blk, seg0.scale, nseed);
    } else if (p.B <= 32 || 16 * lds1 > 64 * 1024 || !(p.flags & 2)) {  // 8 queries share every seed row load
        prep_seed_kernel<8><<<dim3(nparts, (p.B + 7) / 8), 256, 8 * lds1, st>>>(dp, seg0.blk, seg0.scale, nseed);
    } else {  // tuning (flag bit 1): 16 do
        prep_seed_kernel<16><<<dim3(nparts, (p.B + 15) / 16), 256, 16 * lds1, st>>>(dp, seg0.blk, seg0.scale, nseed);
    }
    PCV_LAUNCHED();
}

void launch_scan_wave(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus) {
    if (p.total_blocks == 0) return;
    const size_t lds = (size_t)p.B * p.D4 * 4 * sizeof(float);
    const unsigned gm = (p.flags >> 8) & 0xff;
    unsigned grid = (unsigned)num_cus * (gm ? gm : 8);
    const unsigned need = (p.total_blocks + 3) / 4;
    if (grid > need) grid = need;
    const bool ntl = (p.flags & 1) == 0;  // non-temporal corpus loads unless flag bit 0 is set
    for (const void* f : {(const void*)scan_wave_kernel<1, true>, (const void*)scan_wave_kernel<2, true>, (const void*)scan_wave_kernel<3, true>,
                          (const void*)scan_wave_kernel<4, true>, (const void*)scan_wave_kernel<1, false>, (const void*)scan_wave_kernel<2, false>,
                          (const void*)scan_wave_kernel<3, false>, (const void*)scan_wave_kernel<4, false>})
        allow_dynamic_lds(f, lds);
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
    PCV_LAUNCHED();
}

// Largest query count one MFMA pass can take at this padded dim: the bf16 query tile
// (32*NT rows x Dp) must fit the 160 KB LDS of a CU.  0: the tile does not fit at all.
int mfma_pass_queries(int Dp) {
    for (int nt : {4, 2, 1})
        if ((size_t)nt * 32 * Dp * 2 <= 156 * 1024) return nt * 32;
    return 0;
}

uint32_t mfma_tile_rows(int B) { return B <= 32 ? 32u : (B <= 64 ? 64u : 128u); }

template <int NT, bool NTL, int WPB, int NBUF, bool SRC16>
static void launch_mfma_variant(hipStream_t st, const ScanParams* dp, unsigned grid, size_t lds) {
    allow_dynamic_lds((const void*)scan_mfma_kernel<NT, NTL, WPB, NBUF, SRC16>, lds);  // (the static LDS of the kernel comes on top)
    scan_mfma_kernel<NT, NTL, WPB, NBUF, SRC16><<<grid, WPB * 64, lds, st>>>(dp);
}

template <int NT, bool NTL>
static void launch_mfma_shape(hipStream_t st, const ScanParams* dp, unsigned grid, size_t lds, bool wide, bool src16, unsigned nbuf,
                              unsigned wpb) {
    if (src16) {
        // the screening copy moves half the bytes per chunk: twice the chunks in flight for the same bytes in flight
        if (NT == 4 || wide) {
            if (nbuf == 4) launch_mfma_variant<NT, NTL, 8, 4, true>(st, dp, grid, lds);
            else launch_mfma_variant<NT, NTL, 8, 6, true>(st, dp, grid, lds);
        } else if constexpr (NT < 4) {
            if (nbuf == 6) launch_mfma_variant<NT, NTL, 4, 6, true>(st, dp, grid, lds);
            else launch_mfma_variant<NT, NTL, 4, 4, true>(st, dp, grid, lds);
        }
        return;
    }
    if (NT == 4 || wide) launch_mfma_variant<NT, NTL, 8, 3, false>(st, dp, grid, lds);
    else if constexpr (NT < 4) launch_mfma_variant<NT, NTL, 4, 2, false>(st, dp, grid, lds);
}

void launch_scan_mfma(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus) {
    if (p.total_blocks == 0) return;
    const int NT = p.B <= 32 ? 1 : (p.B <= 64 ? 2 : 4);
    const size_t lds = (size_t)NT * 32 * p.D4 * 4 * sizeof(uint16_t);
    const unsigned gm = (p.flags >> 8) & 0xff;
    const unsigned nbuf = (p.flags >> 24) & 0xf;
    const bool ntl = (p.flags & 1) == 0;   // non-temporal corpus loads unless flag bit 0 is set
    const bool src16 = (p.flags & 16) != 0;
    // small tiles: 256-thread workgroups, 3 per CU.  Tiles too big for that (B > 64, or dim > ~440):
    // 512-thread workgroups sharing one tile, as many per CU as the LDS holds.
    const bool wide = NT == 4 || lds * 3 > 156 * 1024;
    const unsigned per_cu = wide ? (unsigned)std::max<size_t>(1, std::min<size_t>(2, (156 * 1024) / lds)) : 3;
    // (12 waves sharing a tile that fits once per CU — 3 per SIMD — were tried for 128 queries: the 168-VGPR cap spills
    // 113 registers and the pass takes 11.3 ms instead of 6.8 ms for 50M rows)
    const unsigned wpb = wide ? 8 : 4;
    unsigned grid = (unsigned)num_cus * (gm ? gm : per_cu);
    const unsigned need = (p.total_blocks + wpb - 1) / wpb;
    if (grid > need) grid = need;
    if (p.spec_rank > 0 || p.spec_gap == p.spec_gap) {
        set_guess_kernel<<<(p.B + 3) / 4, 256, 0, st>>>(dp);
        PCV_LAUNCHED();
    }
    if (NT == 1) {
        if (ntl) launch_mfma_shape<1, true>(st, dp, grid, lds, wide, src16, nbuf, wpb);
        else launch_mfma_shape<1, false>(st, dp, grid, lds, wide, src16, nbuf, wpb);
    } else if (NT == 2) {
        if (ntl) launch_mfma_shape<2, true>(st, dp, grid, lds, wide, src16, nbuf, wpb);
        else launch_mfma_shape<2, false>(st, dp, grid, lds, wide, src16, nbuf, wpb);
    } else {
        if (ntl) launch_mfma_shape<4, true>(st, dp, grid, lds, wide, src16, nbuf, wpb);
        else launch_mfma_shape<4, false>(st, dp, grid, lds, wide, src16, nbuf, wpb);
    }
    PCV_LAUNCHED();
}

// (32*NT rows x (P16+1) pieces) of int8 query tile per workgroup
static size_t mfma8_lds(int nt, int Dp) { return (size_t)nt * 32 * ((((Dp + 127) & ~127) >> 4) + 1) * 16; }
int mfma8_pass_queries(int Dp) {
    if (((Dp + 127) & ~127) <= 384) return 256;  // the block-holding form, four halves of 64 queries
    for (int nt : {4, 2, 1})
        if (mfma8_lds(nt, Dp) <= 156 * 1024) return nt * 32;
    return 0;
}

// 64 queries: three chunk buffers (150 registers).  Since the row loads go through a buffer descriptor a fourth fits without
// spilling (166 registers) and changes nothing: 6.259 against 6.251 ms at 100M x 384, 0.933 / 0.943 at 12.5M, 5.999 / 5.989 at 768-d.
template <int NT, bool NTL, int WPB = 4, int NBUF = (NT == 2 ? 3 : 4), bool DRAIN = false, int NCHT = 0>
static void launch_mfma8_variant(hipStream_t st, const ScanParams* dp, unsigned grid, size_t lds) {
    allow_dynamic_lds((const void*)scan_mfma8_kernel<NT, NTL, WPB, NBUF, DRAIN, NCHT>, lds);
    scan_mfma8_kernel<NT, NTL, WPB, NBUF, DRAIN, NCHT><<<grid, WPB * 64, lds, st>>>(dp);
}
// the DRAIN form, chunk count (dimension / 128, rounded up) as a template argument
template <int NT, bool NTL, int NBUF>
static void launch_mfma8_drain(hipStream_t st, const ScanParams* dp, unsigned grid, size_t lds, int nch) {
    switch (nch) {
        case 1: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 1>(st, dp, grid, lds); break;
        case 2: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 2>(st, dp, grid, lds); break;
        case 3: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 3>(st, dp, grid, lds); break;
        case 4: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 4>(st, dp, grid, lds); break;
        case 6: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 6>(st, dp, grid, lds); break;
        case 8: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 8>(st, dp, grid, lds); break;
        default: launch_mfma8_variant<NT, NTL, 12, NBUF, true, 0>(st, dp, grid, lds); break;  // (640-d, 896-d: the run-time form)
    }
}

void launch_scan_mfma8(hipStream_t st, const ScanParams& p, const ScanParams* dp, int num_cus) {
    if (p.total_blocks == 0) return;
    const int NT = p.B <= 32 ? 1 : (p.B <= 64 ? 2 : (p.B <= 128 ? 4 : 8));  // 8: the 256-query tile of the block-holding form
    const size_t lds = mfma8_lds(NT, p.D4 * 4);
    const unsigned gm = (p.flags >> 8) & 0xff;
    const bool ntl = (p.flags & 1) == 0;
    const unsigned most = NT == 4 ? 2 : 3;  // waves per SIMD the register budget allows = 256-thread workgroups per CU
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(most, (156 * 1024) / lds));
    // a 128-query tile that fits a CU only once (rows wider than 576 features: 100 KB at 768-d) is shared by eight waves,
    // or the CU would run one wave per SIMD
    const bool wide8 = NT == 4 && per_cu == 1;
    const unsigned wpbt = wide8 ? 8 : 4;
    unsigned grid = (unsigned)num_cus * (gm ? gm : per_cu);
    const unsigned need = (p.total_blocks + wpbt - 1) / wpbt;
    if (grid > need) grid = need;
    if (p.D4 * 4 > 1024) PCV_FAIL(PCV_ERR_UNSUPPORTED, "int8 screen: dimension %d is too large", p.D);
    quantize_queries_kernel<<<NT * 32 / 4, 256, 0, st>>>(dp);
    PCV_LAUNCHED();
    const int nch = ((p.D4 * 4 + 127) & ~127) >> 7;
    if (p.B > 64 && nch <= 3 && (p.B > 128 || !(p.flags & 8u))) {  // (flag bit 3: the 128-query tile, for comparison)
        const bool four = p.B > 128;
        // (+ 4 KB per wave where a chunk of the next block waits in LDS: the 256-query form at more than one chunk per block)
        const size_t ldsh = mfma8_lds(four ? 8 : 4, p.D4 * 4) + (four && nch >= 2 ? 12 * 4096 : 0);
        const unsigned wpb = four ? 12 : 4;
        const unsigned g3 = std::min<unsigned>((unsigned)num_cus * (gm ? gm : (four ? 1u : 3u)), (p.total_blocks + wpb - 1) / wpb);
#define PCV_HOLD(NCHV, NHV)                                                                           \
    {                                                                                                 \
        allow_dynamic_lds(ntl ? (const void*)scan_mfma8_hold_kernel<true, NCHV, NHV> : (const void*)scan_mfma8_hold_kernel<false, NCHV, NHV>, ldsh); \
        if (ntl) scan_mfma8_hold_kernel<true, NCHV, NHV><<<g3, NHV == 2 ? 256 : 768, ldsh, st>>>(dp); \
        else scan_mfma8_hold_kernel<false, NCHV, NHV><<<g3, NHV == 2 ? 256 : 768, ldsh, st>>>(dp);    \
    }
        if (four) {
            if (nch == 3) PCV_HOLD(3, 4) else if (nch == 2) PCV_HOLD(2, 4) else PCV_HOLD(1, 4)
        } else {
            if (nch == 3) PCV_HOLD(3, 2) else if (nch == 2) PCV_HOLD(2, 2) else PCV_HOLD(1, 2)
        }
#undef PCV_HOLD
        PCV_LAUNCHED();
        return;
    }
    if (NT == 8) PCV_FAIL(PCV_ERR_UNSUPPORTED, "int8 screen: %d queries in one pass need rows of at most 384 features", p.B);
    // Up to 64 queries: ONE 12-wave workgroup per CU, eleven waves stream and the twelfth works off their coarse survivors
    // (the survivor ring above).  Flag bit 28: the older form, three 4-wave workgroups per CU, every wave handling its own.
    // (One to four queries keep the older form: their survivors are few, twelve streaming waves beat eleven — 10M x 768, one query:
    // 1.16 against 1.20 ms; from eight queries on the drain form is ahead, 0.63 against 0.67 ms at 10M x 384 / 16 queries.)
    if (NT <= 2 && p.B > 4 && !(p.flags & (1u << 28)) && p.nseg < (1 << kRingSegBits) && lds + 68 * 1024 <= 156 * 1024) {
        constexpr unsigned kW = 12;
        unsigned g12 = (unsigned)num_cus * (gm ? gm : 1u);
        g12 = std::min(g12, (p.total_blocks + (kW - 2)) / (kW - 1));
        if (NT == 1) {
            if (ntl) launch_mfma8_drain<1, true, 4>(st, dp, g12, lds, nch);
            else launch_mfma8_drain<1, false, 4>(st, dp, g12, lds, nch);
        } else {
            if ((p.flags >> 24 & 0xf) == 3) {
                if (ntl) launch_mfma8_drain<2, true, 3>(st, dp, g12, lds, nch);
                else launch_mfma8_drain<2, false, 3>(st, dp, g12, lds, nch);
            } else {
                if (ntl) launch_mfma8_drain<2, true, 4>(st, dp, g12, lds, nch);
                else launch_mfma8_drain<2, false, 4>(st, dp, g12, lds, nch);
            }
        }
        PCV_LAUNCHED();
        return;
    }
    if (NT == 1) {
        if (ntl) launch_mfma8_variant<1, true>(st, dp, grid, lds);
        else launch_mfma8_variant<1, false>(st, dp, grid, lds);
    } else if (NT == 2) {
        if (ntl) launch_mfma8_variant<2, true>(st, dp, grid, lds);
        else launch_mfma8_variant<2, false>(st, dp, grid, lds);
    } else if (wide8) {
        if (ntl) launch_mfma8_variant<4, true, 8>(st, dp, grid, lds);
        else launch_mfma8_variant<4, false, 8>(st, dp, grid, lds);
    } else {
        if (ntl) launch_mfma8_variant<4, true>(st, dp, grid, lds);
        else launch_mfma8_variant<4, false>(st, dp, grid, lds);
    }
    PCV_LAUNCHED();
}

void launch_coarse_pack8(hipStream_t st, const float4* blk, const float* scale, uint4* blk8, float* scale8, uint32_t first_block,
                         uint32_t nblocks, int D4) {
    if (first_block >= nblocks) return;
    const unsigned grid = (unsigned)std::min<uint32_t>((nblocks - first_block + 3) / 4, 1u << 16);
    coarse_pack8_kernel<<<grid, 256, 0, st>>>(blk, scale, blk8, scale8, first_block, nblocks, D4);
    PCV_LAUNCHED();
}

void launch_mid_pack(hipStream_t st, const float4* blk, const float* scale, const float* scale8, uint4* mid16, float* scale16,
                     uint32_t first_row, uint32_t nrows, int D4) {
    if (first_row >= nrows) return;
    if (scale8) {  // the segment has its int8 copy (and with it the blocks' maxima): block by block, every row read once
        const uint32_t b0 = first_row / kBlockRows, nb = (nrows + kBlockRows - 1) / kBlockRows;
        const size_t lds = (size_t)32 * (D4 + 2) * sizeof(uint2);
        allow_dynamic_lds((const void*)mid_pack_block_kernel, lds);
        const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(8, (150 * 1024) / lds));
        const unsigned gridb = std::min<uint32_t>(nb - b0, (unsigned)current_device_cus() * per_cu);
        mid_pack_block_kernel<<<gridb, 64, lds, st>>>(blk, scale, scale8, mid16, scale16, b0, nb, D4);
        PCV_LAUNCHED();
        return;
    }
    const unsigned grid = (unsigned)std::min<uint32_t>((nrows - first_row + 3) / 4, 256u * 8 * 4);
    mid_pack_kernel<<<grid, 256, 0, st>>>(blk, scale, mid16, scale16, first_row, nrows, D4);
    PCV_LAUNCHED();
}

void launch_coarse_pack(hipStream_t st, const float4* blk, const float* scale, uint4* blk16, uint32_t first_block, uint32_t nblocks,
                        int D4) {
    if (first_block >= nblocks) return;
    const size_t total = (size_t)(nblocks - first_block) * (D4 >> 1) * 32;
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 1u << 16);
    coarse_pack_kernel<<<grid, 256, 0, st>>>(blk, scale, blk16, first_block, nblocks, D4);
    PCV_LAUNCHED();
}

void launch_rescore_select(hipStream_t st, const ScanParams& p, const ScanParams* dp) {
    if (p.D4 <= kCoopMaxD4) {
        const size_t lds = ((size_t)p.D4 + 4 * 8 * ((size_t)p.D4 + 1)) * sizeof(float4);
        allow_dynamic_lds((const void*)rescore_select_kernel<true>, 100 * 1024);  // static LDS (28 KB) + dynamic may pass 64 KB
        rescore_select_kernel<true><<<p.B, 256, lds, st>>>(dp);
    } else {
        const size_t lds = (size_t)p.D4 * sizeof(float4);
        rescore_select_kernel<false><<<p.B, 256, lds, st>>>(dp);
    }
    PCV_LAUNCHED();
}

void launch_merge(hipStream_t st, const pcv_hit_dev* lists, int n_shards, int B, int k, pcv_hit_dev* out, int flagged) {
    const size_t stride = (size_t)B * k + (flagged ? 1 : 0);
    merge_kernel<<<B, 64, (size_t)n_shards * k, st>>>(lists, n_shards, B, k, stride, flagged, out);
    PCV_LAUNCHED();
}

void launch_similarity_matrix(hipStream_t st, const float* a, int B, const float* m, int64_t N, int D, int cosine,
                              float* out) {
    if (N <= 0 || B <= 0) return;
    dim3 grid(cdiv64(N, 256), B);
    similarity_matrix_kernel<<<grid, 256, 0, st>>>(a, B, m, N, D, cosine, out);
    PCV_LAUNCHED();
}

}  // namespace pcv
