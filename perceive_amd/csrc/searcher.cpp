// searcher.cpp — host side of the Searcher C ABI (replaces search.rs:29-260 of the reference).
//
// A searcher owns, per source, a list of device-resident corpus segments in the blocked HBM layout
// (scan.h).  Searching streams the selected segments once; see scan_kernels.hip for the pipeline.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"
#include "scan.h"

using namespace pcv;

namespace {

// One device allocation of corpus rows.  Rows are appended in place while there is room (cap_rows), so a
// source that grows by many small adds stays a handful of segments; rows [scaled_rows, nrows) have been
// uploaded but not yet given their scale (they become searchable at the next finalize).
struct Segment {
    float4* blk = nullptr;
    float* scale = nullptr;
    int64_t* ids = nullptr;  // nullptr -> implicit ids id0 + row (synthetic rows)
    uint4* blk16 = nullptr;  // bf16 screening copy (scan.h), built at finalize for the rows that have their scale
    uint4* blk8 = nullptr;   // int8 screening copy + its per-row quantisation scales
    float* scale8 = nullptr;
    uint4* mid16 = nullptr;  // row-major 16-bit copy (scan.h) + its per-row scales
    float* scale16 = nullptr;
    uint32_t mid_rows = 0;   // rows the mid copy covers
    int64_t id0 = 0;
    int64_t pos0 = 0;
    uint32_t nrows = 0, cap_rows = 0, scaled_rows = 0;
    uint32_t copied_rows = 0;  // rows the screening copy covers
    uint32_t nblocks() const { return (nrows + kBlockRows - 1) / kBlockRows; }
};

struct Source {
    int64_t id = 0;
    std::vector<Segment> segs;
    int64_t next_implicit_id = 0;
    int64_t reserve = 0;  // rows the host announced it is going to add (pcv_searcher_reserve)
    int64_t rows() const {
        int64_t n = 0;
        for (const Segment& g : segs) n += g.nrows;
        return n;
    }
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    void ensure(size_t want) {
        if (want <= n) return;
        if (p) PCV_HIP(hipFree(p));
        p = nullptr;
        n = 0;
        PCV_HIP(hipMalloc((void**)&p, want * sizeof(T)));
        n = want;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

constexpr int64_t kStageRows = 1 << 18;                  // rows per H2D staging step (384-d: 400 MB)
constexpr int64_t kMaxSegRows = (int64_t)0xffffffc0u;    // a candidate names its row in 32 bits
constexpr int64_t kGrowCapRows = (int64_t)1 << 22;       // spare room a new segment gets at most (6 GB at 384-d)
constexpr int64_t kMinSpareRows = 1024;                  // ... and at least
constexpr int64_t kGraphRows = (int64_t)4 << 20;        // passes over more rows than this are not replayed as graphs
constexpr size_t kPassAlign = 256;
inline size_t align_up(size_t v) { return (v + kPassAlign - 1) / kPassAlign * kPassAlign; }

}  // namespace

struct pcv_searcher {
    pcv_ctx* ctx = nullptr;
    int D = 0, Dp = 0, D4 = 0;
    int metric = PCV_METRIC_COSINE;
    int kernel = PCV_KERNEL_AUTO;
    int64_t shard_offset = 0;
    std::vector<Source> sources;
    uint32_t* d_max_norm_bits = nullptr;
    float max_norm = 0.0f;
    bool dirty = false;
    std::mutex mu;
    pcv_scan_stats stats{};
    DevBuf<float> d_stage;  // ingestion staging (released by finalize)

    // per-search workspace (sized for one pass of <= 128 queries)
    DevBuf<float> d_qf32, d_qraw, d_margin, d_margin32;
    DevBuf<uint16_t> d_qbf16;
    DevBuf<int8_t> d_q8;
    DevBuf<float> d_q8c;
#ifdef PCV_STAMPS
    DevBuf<unsigned long long> d_stamps;     // diagnostic build: per-wave time stamps of the scan launch (scan_mfma8_kernel)
#endif
    DevBuf<uint32_t> d_spec;                 // speculative start thresholds of a pass (scan.h)
    bool spec_hold = false;                  // a guess failed: the repeat of that pass runs without one
    int spec_rest = 0;                       // ... and so do the next passes: 16 after a first failure, doubling up to 1024
    int spec_penalty = 0;                    //     while failures keep coming, forgotten after 4096 passes without one
    int spec_clean = 0;
    // learned part of the guess (scan.h): statistics of (k-th best - median seed slot) over the queries of one pass shape
    struct GapStats {
        int64_t n = 0;
        double mean = 0.0, m2 = 0.0;  // Welford
        double spread = 0.0;          // mean (best - median) seed slot of the same queries
        float smallest = INFINITY;
        int holdoff = 0;              // passes still to run without a learned guess after one failed
        void add(float d, float sp) {
            if (n >= 8192) {  // keep following the queries: older ones count half
                n /= 2;
                m2 /= 2.0;
            }
            n += 1;
            const double dl = d - mean;
            mean += dl / (double)n;
            m2 += dl * (d - mean);
            spread += (sp - spread) / (double)n;
            smallest = std::min(smallest, d);
        }
        void reset() {
            n = 0;
            mean = m2 = spread = 0.0;
            smallest = INFINITY;
        }
        // used only where the gaps seen are tightly concentrated (queries alike, as far as this statistic goes): then six
        // standard deviations below their mean, and no more than 0.8 of the smallest seen; otherwise no learned guess
        float gap() const {
            if (n < 128 || holdoff > 0) return NAN;
            const double sd = std::sqrt(m2 / (double)(n - 1));
            if (!(mean > 0.0) || sd > 0.15 * mean) return NAN;
            const double g = std::min(mean - 6.0 * sd, 0.8 * (double)smallest);
            return g > 0.0 ? (float)g : NAN;
        }
    } gaps;
    int64_t gap_rows = -1;
    int gap_k = 0, gap_nseg = 0;
    DevBuf<float> d_cand_s;
    DevBuf<uint32_t> d_tau, d_slots, d_cnt;
    DevBuf<uint64_t> d_cand;
    DevBuf<pcv_hit_dev> d_hits;
    // what one pass takes up: ScanParams | SegDesc[nseg] | queries[B][D], built in pinned memory and
    // sent with ONE copy into its device mirror
    uint8_t* pin_pass = nullptr;
    uint8_t* d_pass = nullptr;
    size_t pass_cap = 0;
    // what one pass brings back: written by rescore_select_kernel straight into pinned memory
    struct Pinned {
        uint32_t cnt[kMfmaQueries];
        uint32_t coarse[2 * kMfmaQueries];  // rows per query that passed the coarse screen, then those that also passed the mid screen (statistics)
        float spec_base[kMfmaQueries];  // median / best seed slot per query, k-th best exact score per query (scan.h: spec_gap)
        float spec_top[kMfmaQueries];
        float kth[kMfmaQueries];
        pcv_hit_dev hits[kMfmaQueries * kMaxK];
    };
    Pinned* pin = nullptr;
    bool state_clean = false;  // tau / slots / counters are in the state a pass starts from
    uint32_t cand_cap = 8192;
    uint32_t scan_flags = 0;  // tuning knobs: PCV_SCAN_FLAGS at creation, pcv_searcher_set_tuning
    bool fail_copy_alloc = false;  // PCV_TUNE_FAIL_COPY_ALLOC
    int mid_copy = PCV_MID_COPY_AUTO;        // pcv_searcher_set_mid_copy
    bool mid_gave_way = false;               // AUTO: building it failed, or it was dropped to make room for rows: not tried again
    bool mids_present = false;               // every row of every segment is covered by a mid copy
    int mid_hot_passes = 0;                  // AUTO: passes in a row whose coarse screen let more than kMidTrigger rows per query through
    // AUTO builds the mid copy beside the searches, on a stream of its own: the call that decides to build it queues the build
    // and goes on without the copy, and so do the calls after it until the build's event has come (a 100M-row build is tens of
    // milliseconds; round 3 ran it inside the deciding search call, 228 ms with the older kernel).  Whatever changes rows or
    // copies waits for it first (settle_mid_build).
    // The allocations of such a build are host time too (132 ms for 6 GB on the test box): a helper thread makes them and
    // queues the kernels; it works on a list of its own (MidTask) and touches no segment — settle_mid_build joins it and hands
    // the buffers to the segments.
    hipStream_t side = nullptr;
    hipEvent_t side_go = nullptr, mid_done = nullptr;
    bool mid_building = false;
    struct MidTask {
        struct Item {
            Segment* g;  // (for settle_mid_build; the helper thread does not follow it)
            const float4* blk;
            const float* scale;
            const float* scale8;
            uint32_t cap_rows, first_row, rows;
            uint4* mid16;
            float* scale16;
            bool own;  // the helper thread allocated mid16 / scale16
        };
        std::vector<Item> items;
        std::thread th;
        std::atomic<bool> queued{false};  // the helper thread is through (the kernels may still run: mid_done)
        bool joined = false, failed = false;
    };
    std::unique_ptr<MidTask> mid_task;
    int screen_copy = PCV_SCREEN_COPY_AUTO;  // pcv_searcher_set_screening_copy
    bool screen_copy_gave_way = false;       // AUTO: the copies were dropped to make room for rows
    int copies_kind = 0;                     // 0: not every row of every segment is covered by a screening copy; 1 bf16; 2 int8
    bool wide_sharded = false;               // pcv_searcher_allow_wide_sharded_pass: the host vouches for int8 copies on every rank
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // A pass is five short launches and four event records: on small corpora its cost is the queueing.  The second time
    // a pass of the same shape comes by, its launch sequence is captured into a hipGraph and replayed from then on (what
    // changes between passes — parameters, segment table, queries — travels in the pinned block the first kernel reads).
    struct PassShape {
        int B = -1, k = 0, kernel = 0, src_kind = 0, nseg = 0, guess = 0;  // guess: the pass has the kernel that sets a speculative threshold
        uint32_t total_blocks = 0, seed_blocks = 0, flags = 0, seg0_rows = 0;
        size_t bytes = 0;
        const void *pin = nullptr, *dev = nullptr, *seg0_blk = nullptr, *seg0_scale = nullptr;
        bool operator==(const PassShape& o) const {
            return B == o.B && k == o.k && kernel == o.kernel && src_kind == o.src_kind && nseg == o.nseg && guess == o.guess && total_blocks == o.total_blocks &&
                   seed_blocks == o.seed_blocks && flags == o.flags && seg0_rows == o.seg0_rows && bytes == o.bytes && pin == o.pin &&
                   dev == o.dev && seg0_blk == o.seg0_blk && seg0_scale == o.seg0_scale;
        }
    };
    PassShape graph_shape, last_shape;
    hipGraphExec_t graph_exec = nullptr;
    int shape_seen = 0;
    float shape_fixed_ms = -1.0f;  // scan kernel's share of the pass time, of the last plainly launched pass of last_shape
    float graph_fixed_ms = 0.0f;   // ... of the shape the graph was captured for
    bool use_graph = true;
    // a pass queued by enqueue_pass and not yet collected by finish_pass
    struct Pending {
        bool active = false;
        bool done = false;  // nothing was launched (no selected rows): only the stream has to drain
        int B = 0;
        int64_t rows = 0;
        bool mid = false;  // every selected segment had its mid copy
        int src = 0;  // what the scan streamed: 0 f32 rows, 1 bf16 copies, 2 int8 copies
        int64_t stream_bytes = 0;  // ... and how many bytes of it the scan kernel has to read, padding included
        bool replayed = false;  // launched as a graph: only the pass as a whole was timed
        bool learned = false;   // the speculative threshold had a learned part
        bool guessing = false;  // the pass ran with a speculative threshold (and sent the seed statistics home)
    } pending;

    Source* find_source(int64_t id) {
        for (auto& s : sources)
            if (s.id == id) return &s;
        return nullptr;
    }
    Source& get_or_add_source(int64_t id) {
        if (Source* s = find_source(id)) return *s;
        sources.emplace_back();
        sources.back().id = id;
        return sources.back();
    }
};

namespace {

void free_segment(Segment& g) {
    if (g.blk) (void)hipFree(g.blk);
    if (g.scale) (void)hipFree(g.scale);
    if (g.ids) (void)hipFree(g.ids);
    if (g.blk16) (void)hipFree(g.blk16);
    if (g.blk8) (void)hipFree(g.blk8);
    if (g.scale8) (void)hipFree(g.scale8);
    if (g.mid16) (void)hipFree(g.mid16);
    if (g.scale16) (void)hipFree(g.scale16);
    g = Segment();
}

constexpr int64_t kMidTrigger = 4096;  // coarse survivors per query and pass above which AUTO builds the mid copy ...
constexpr int64_t kMidShare = 25;      // ... or whose f32 rows (32 B per feature: 16-byte pieces in 128-byte lines) come to more than 1/25 of the bytes streamed ...
constexpr int kMidPasses = 2;          // ... once that many passes in a row were above it

// The AUTO build under way beside the searches, if any: `wait` for it, or only look whether it is done; done = its rows count.
void settle_mid_build(pcv_searcher* s, bool wait) {
    if (!s->mid_building) return;
    pcv_searcher::MidTask& t = *s->mid_task;
    if (!t.joined) {
        if (!wait && !t.queued.load(std::memory_order_acquire)) return;
        t.th.join();
        t.joined = true;
    }
    if (!t.failed) {
        if (wait)
            PCV_HIP(hipEventSynchronize(s->mid_done));
        else if (hipEventQuery(s->mid_done) != hipSuccess) {
            (void)hipGetLastError();  // (hipErrorNotReady)
            return;
        }
    }
    s->mid_building = false;
    if (t.failed) {  // (the helper thread has given back what it had allocated)
        s->mid_gave_way = true;
        s->mid_task.reset();
        return;
    }
    for (auto& it : t.items) {
        it.g->mid16 = it.mid16;
        it.g->scale16 = it.scale16;
        it.g->mid_rows = it.rows;
    }
    s->mid_task.reset();
    s->mids_present = true;
    for (auto& src : s->sources)
        for (auto& g : src.segs)
            if (g.nrows > 0 && (!g.mid16 || g.mid_rows < g.nrows)) s->mids_present = false;
}

// Queue the build of the mid copies of every segment beside the searches (AUTO): a helper thread allocates and launches on
// the side stream, behind everything the context stream holds now.
void start_mid_build(pcv_searcher* s) {
    settle_mid_build(s, true);
    if (!s->side) {
        PCV_HIP(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
        PCV_HIP(hipEventCreateWithFlags(&s->side_go, hipEventDisableTiming));
        PCV_HIP(hipEventCreateWithFlags(&s->mid_done, hipEventDisableTiming));
    }
    PCV_HIP(hipEventRecord(s->side_go, s->ctx->stream));
    PCV_HIP(hipStreamWaitEvent(s->side, s->side_go, 0));
    s->mid_task.reset(new pcv_searcher::MidTask());
    pcv_searcher::MidTask& t = *s->mid_task;
    for (auto& src : s->sources)
        for (auto& g : src.segs) {
            if (g.nrows == 0 || (g.mid16 && g.mid_rows >= g.scaled_rows)) continue;
            t.items.push_back({&g, g.blk, g.scale, (g.blk8 && g.copied_rows >= g.scaled_rows) ? g.scale8 : nullptr, g.cap_rows,
                               g.mid16 ? g.mid_rows : 0u, g.scaled_rows, g.mid16, g.scale16, false});
        }
    const int device = s->ctx->device, D4 = s->D4, Dp = s->Dp;
    const bool fail = s->fail_copy_alloc;
    hipStream_t side = s->side;
    hipEvent_t done = s->mid_done;
    pcv_searcher::MidTask* tp = &t;
    t.th = std::thread([tp, device, D4, Dp, fail, side, done] {
        bool ok = hipSetDevice(device) == hipSuccess;
        try {
            for (auto& it : tp->items) {
                if (!ok) break;
                if (!it.mid16) {
                    ok = !fail && hipMalloc((void**)&it.mid16, (size_t)it.cap_rows * Dp * 2) == hipSuccess;
                    if (ok) {
                        it.own = true;
                        ok = hipMalloc((void**)&it.scale16, (size_t)it.cap_rows * sizeof(float)) == hipSuccess;
                    }
                    if (!ok) break;
                }
                launch_mid_pack(side, it.blk, it.scale, it.scale8, it.mid16, it.scale16, it.first_row, it.rows, D4);
            }
            if (ok) ok = hipEventRecord(done, side) == hipSuccess;
        } catch (...) {
            ok = false;
        }
        if (!ok) {
            (void)hipGetLastError();
            (void)hipStreamSynchronize(side);
            for (auto& it : tp->items)
                if (it.own) {
                    if (it.mid16) (void)hipFree(it.mid16);
                    if (it.scale16) (void)hipFree(it.scale16);
                }
            tp->failed = true;
        }
        tp->queued.store(true, std::memory_order_release);
    });
    s->mid_building = true;  // (mids_present stays false: the passes go on without the copy until settle_mid_build sees it done)
}

void drop_mid_copies(pcv_searcher* s) {
    settle_mid_build(s, true);
    s->mids_present = false;
    for (auto& src : s->sources)
        for (auto& g : src.segs) {
            if (g.mid16) (void)hipFree(g.mid16);
            if (g.scale16) (void)hipFree(g.scale16);
            g.mid16 = nullptr;
            g.scale16 = nullptr;
            g.mid_rows = 0;
        }
}

// Mid copies (scan.h) of the rows that have their scale and no copy yet.  `must`: an allocation failure is an error
// (PCV_MID_COPY_ON); otherwise it ends the attempt for good (mid_gave_way).
void build_mid_copies(pcv_searcher* s, bool must) {
    settle_mid_build(s, true);
    hipStream_t st = s->ctx->stream;
    for (auto& src : s->sources)
        for (auto& g : src.segs) {
            if (g.nrows == 0 || (g.mid16 && g.mid_rows >= g.scaled_rows)) continue;
            if (!g.mid16) {
                const size_t bytes = (size_t)g.cap_rows * s->Dp * 2;
                hipError_t e = s->fail_copy_alloc ? hipErrorOutOfMemory : hipMalloc((void**)&g.mid16, bytes);
                if (e == hipSuccess) {
                    e = hipMalloc((void**)&g.scale16, (size_t)g.cap_rows * sizeof(float));
                    if (e != hipSuccess) {
                        (void)hipFree(g.mid16);
                        g.mid16 = nullptr;
                    }
                }
                if (e != hipSuccess) {
                    (void)hipGetLastError();
                    g.mid16 = nullptr;
                    g.scale16 = nullptr;
                    if (must)
                        PCV_FAIL(PCV_ERR_DEVICE, "hipMalloc of %.2f GB for the mid copy of %u rows failed: %s", bytes / 1e9, g.cap_rows, hipGetErrorString(e));
                    PCV_HIP(hipStreamSynchronize(st));
                    drop_mid_copies(s);
                    s->mid_gave_way = true;
                    return;
                }
                g.mid_rows = 0;
            }
            launch_mid_pack(st, g.blk, g.scale, (g.blk8 && g.copied_rows >= g.scaled_rows) ? g.scale8 : nullptr, g.mid16, g.scale16, g.mid_rows,
                            g.scaled_rows, s->D4);
            g.mid_rows = g.scaled_rows;
        }
    s->mids_present = true;
    for (const auto& src : s->sources)
        for (const auto& g : src.segs)
            if (g.nrows > 0 && (!g.mid16 || g.mid_rows < g.nrows)) s->mids_present = false;
}

// AUTO mid copy: called by every search entry point in front of its passes
void maybe_build_mid_copies(pcv_searcher* s) {
    settle_mid_build(s, false);
    if (!(s->mid_copy == PCV_MID_COPY_AUTO && !s->mids_present && !s->mid_building && !s->mid_gave_way && s->mid_hot_passes >= kMidPasses)) return;
    // (only if the memory is plainly there: the copy is a convenience — a few per cent of a pass on Gaussian rows, a quarter on
    // clustered ones — and the headroom is for rows, for other searchers and models of the process, for the candidate lists of
    // an overflow rerun: a tenth of the device, at least 4 GB, stays free)
    size_t free_b = 0, total_b = 0, need = 0;
    for (const auto& src : s->sources)
        for (const auto& g : src.segs) need += (size_t)g.cap_rows * ((size_t)s->Dp * 2 + 4);
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need + std::max<size_t>((size_t)4 << 30, total_b / 10))
        start_mid_build(s);  // beside the searches: this call and the next ones go on without it meanwhile
    else
        s->mid_gave_way = true;
    s->mid_hot_passes = 0;
}

void drop_screening_copies(pcv_searcher* s) {
    settle_mid_build(s, true);  // (it reads the int8 copy's block scales)
    s->copies_kind = 0;
    for (auto& src : s->sources)
        for (auto& g : src.segs) {
            if (g.blk16) (void)hipFree(g.blk16);
            if (g.blk8) (void)hipFree(g.blk8);
            if (g.scale8) (void)hipFree(g.scale8);
            g.blk16 = g.blk8 = nullptr;
            g.scale8 = nullptr;
            g.copied_rows = 0;
        }
}

// the form AUTO stands for
inline int copy_kind_wanted(const pcv_searcher* s) {
    if (s->screen_copy == PCV_SCREEN_COPY_OFF || s->screen_copy_gave_way) return 0;
    if (s->screen_copy == PCV_SCREEN_COPY_BF16) return 1;
    // the int8 screen goes up to 1024 padded features (scan_mfma8_kernel); AUTO keeps the bf16 form for wider rows, as
    // long as the MFMA scan can stage their query tile at all
    if (s->Dp > 1024) return (s->screen_copy == PCV_SCREEN_COPY_AUTO && mfma_pass_queries(s->Dp) > 0) ? 1 : (s->screen_copy == PCV_SCREEN_COPY_AUTO ? 0 : 2);
    return 2;
}

// Allocate a segment with room for `cap_rows` rows; zero-filled so padding rows / features are exact
// zeros and no row is searchable before its scale has been computed.
Segment alloc_segment(pcv_searcher* s, int64_t cap_rows, bool with_ids) {
    PCV_REQUIRE(cap_rows > 0 && cap_rows <= kMaxSegRows, "segment of %lld rows is out of range", (long long)cap_rows);
    Segment g;
    const uint32_t nblk = (uint32_t)((cap_rows + kBlockRows - 1) / kBlockRows);
    g.cap_rows = nblk * kBlockRows;
    const size_t bytes = (size_t)nblk * s->D4 * 32 * sizeof(float4);
    hipError_t e = hipMalloc((void**)&g.blk, bytes);
    if (e != hipSuccess && s->mids_present && s->mid_copy == PCV_MID_COPY_AUTO) {
        // the rows themselves come first: the mid copies go before the screening copies do
        (void)hipGetLastError();
        PCV_HIP(hipStreamSynchronize(s->ctx->stream));
        drop_mid_copies(s);
        s->mid_gave_way = true;
        e = hipMalloc((void**)&g.blk, bytes);
    }
    if (e != hipSuccess && s->screen_copy == PCV_SCREEN_COPY_AUTO && !s->screen_copy_gave_way) {
        // the rows themselves come first: give the screening copies back and scan the f32 rows from now on
        (void)hipGetLastError();
        PCV_HIP(hipStreamSynchronize(s->ctx->stream));
        drop_screening_copies(s);
        s->screen_copy_gave_way = true;
        e = hipMalloc((void**)&g.blk, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        PCV_FAIL(PCV_ERR_DEVICE, "hipMalloc of %.2f GB for %lld corpus rows failed: %s", bytes / 1e9,
                 (long long)cap_rows, hipGetErrorString(e));
    }
    try {
        PCV_HIP(hipMalloc((void**)&g.scale, (size_t)g.cap_rows * sizeof(float)));
        if (with_ids) PCV_HIP(hipMalloc((void**)&g.ids, (size_t)g.cap_rows * sizeof(int64_t)));
        PCV_HIP(hipMemsetAsync(g.blk, 0, bytes, s->ctx->stream));
        PCV_HIP(hipMemsetAsync(g.scale, 0, (size_t)g.cap_rows * sizeof(float), s->ctx->stream));
        if (with_ids) PCV_HIP(hipMemsetAsync(g.ids, 0xff, (size_t)g.cap_rows * sizeof(int64_t), s->ctx->stream));
    } catch (...) {
        free_segment(g);
        throw;
    }
    return g;
}

void assign_positions(pcv_searcher* s) {
    int64_t pos = s->shard_offset;
    for (auto& src : s->sources)
        for (auto& g : src.segs) {
            g.pos0 = pos;
            pos += g.nrows;
        }
}

// Where the next `n` rows of a source go: the spare room of its last segment first, then new segments.
// A new segment is sized for what is announced (pcv_searcher_reserve) or, failing that, for as many rows
// again as the source already holds, so that a source fed by many small adds ends up in O(log n)
// segments (search.rs:138-140 sizes its per-source index from the row count the same way).
struct Piece {
    size_t seg;  // index into src.segs
    int64_t n;
};
std::vector<Piece> place_rows(pcv_searcher* s, Source& src, int64_t n, bool with_ids) {
    settle_mid_build(s, true);
    std::vector<Piece> out;
    int64_t remaining = n;
    if (!src.segs.empty()) {
        Segment& tail = src.segs.back();
        if ((tail.ids != nullptr) == with_ids && tail.nrows < tail.cap_rows) {
            const int64_t m = std::min<int64_t>(remaining, (int64_t)tail.cap_rows - tail.nrows);
            out.push_back({src.segs.size() - 1, m});
            remaining -= m;
        }
    }
    int64_t have = src.rows() + (n - remaining);
    while (remaining > 0) {
        // room for the rows at hand plus, for later adds, as many again as the source holds by then (at least
        // kMinSpareRows, at most kGrowCapRows) — or exactly what the host announced
        const int64_t announced = std::max<int64_t>(0, src.reserve - have);
        const int64_t spare = announced > 0 ? std::max<int64_t>(0, announced - remaining)
                                            : std::min<int64_t>(std::max<int64_t>(have + remaining, kMinSpareRows), kGrowCapRows);
        int64_t cap = std::min<int64_t>(remaining + spare, kMaxSegRows);
        const int64_t need = std::min<int64_t>(remaining, kMaxSegRows);
        Segment g;
        try {
            g = alloc_segment(s, cap, with_ids);
        } catch (const Error& e) {
            if (e.status != PCV_ERR_DEVICE || cap == need) throw;
            g = alloc_segment(s, need, with_ids);  // no room for the spare rows: take exactly what is needed
        }
        src.segs.push_back(g);
        const int64_t m = std::min<int64_t>(remaining, (int64_t)g.cap_rows);
        out.push_back({src.segs.size() - 1, m});
        remaining -= m;
        have += m;
    }
    return out;
}

// build_sources row insert (search.rs:87-113,146-148): rows go to the device as they arrive — staged in
// steps of kStageRows and packed into the blocked layout — so the host never holds more than the caller's
// own buffer.  `rows` is row-major f32 or, equivalently on this little-endian host, the blob bytes.
void append_rows(pcv_searcher* s, Source& src, const int64_t* ids, const void* rows, int64_t n) {
    hipStream_t st = s->ctx->stream;
    const size_t row_bytes = (size_t)s->D * sizeof(float);
    s->dirty = true;
    int64_t done = 0;
    for (const Piece& pc : place_rows(s, src, n, true)) {
        for (int64_t off = 0; off < pc.n;) {
            Segment& g = src.segs[pc.seg];
            const int64_t m = std::min<int64_t>(kStageRows, pc.n - off);
            s->d_stage.ensure((size_t)std::min<int64_t>(kStageRows, std::max<int64_t>(m, std::min<int64_t>(n, kStageRows))) * s->D);
            PCV_HIP(hipMemcpyAsync(s->d_stage.p, (const uint8_t*)rows + (size_t)done * row_bytes, (size_t)m * row_bytes,
                                   hipMemcpyHostToDevice, st));
            launch_pack_rows(st, s->d_stage.p, m, s->D, s->D4, g.blk, g.nrows);
            if (ids)
                PCV_HIP(hipMemcpyAsync(g.ids + g.nrows, ids + done, (size_t)m * sizeof(int64_t), hipMemcpyHostToDevice, st));
            else
                launch_iota_ids(st, g.ids + g.nrows, src.next_implicit_id + done, m);
            PCV_HIP(hipStreamSynchronize(st));  // the staging buffer (and the caller's memory) are free again
            g.nrows += (uint32_t)m;
            off += m;
            done += m;
        }
    }
    src.next_implicit_id += n;
}

// Screening copies (scan.h) of the rows that got their scale since the last finalize.  Asked for explicitly: a failed
// allocation is an error; AUTO: it switches the copies off for this searcher (the f32 rows are scanned instead).
void build_screening_copies(pcv_searcher* s, Source& src) {
    const int kind = copy_kind_wanted(s);
    if (kind == 0) return;
    if (kind == 2 && s->Dp > 1024)
        PCV_FAIL(PCV_ERR_UNSUPPORTED, "the int8 screening copy goes up to 1024 features (this index has %d): use PCV_SCREEN_COPY_BF16 or AUTO", s->D);
    hipStream_t st = s->ctx->stream;
    for (auto& g : src.segs) {
        if (g.nrows == 0) continue;
        const bool present = kind == 1 ? g.blk16 != nullptr : g.blk8 != nullptr;
        if (present && g.copied_rows >= g.scaled_rows) continue;
        if (!present) {
            // (a copy of the other kind, left from a mode change, goes first)
            if (g.blk16) (void)hipFree(g.blk16);
            if (g.blk8) (void)hipFree(g.blk8);
            if (g.scale8) (void)hipFree(g.scale8);
            g.blk16 = g.blk8 = nullptr;
            g.scale8 = nullptr;
            g.copied_rows = 0;
            const size_t nblk = g.cap_rows / kBlockRows;
            const size_t bytes = kind == 1 ? nblk * (s->D4 / 2) * 32 * sizeof(uint4) : nblk * (size_t)(((s->Dp + 127) & ~127) / 16) * 32 * sizeof(uint4);
            hipError_t e = s->fail_copy_alloc ? hipErrorOutOfMemory  // (PCV_TUNE_FAIL_COPY_ALLOC)
                                              : hipMalloc(kind == 1 ? (void**)&g.blk16 : (void**)&g.blk8, bytes);
            if (e != hipSuccess && !s->fail_copy_alloc && s->mids_present && s->mid_copy == PCV_MID_COPY_AUTO) {
                // a mid copy AUTO built by itself gives way to the screening copy the host asked for
                (void)hipGetLastError();
                PCV_HIP(hipStreamSynchronize(st));
                drop_mid_copies(s);
                s->mid_gave_way = true;
                e = hipMalloc(kind == 1 ? (void**)&g.blk16 : (void**)&g.blk8, bytes);
            }
            if (e == hipSuccess && kind == 2) {
                e = hipMalloc((void**)&g.scale8, (size_t)(g.cap_rows / kBlockRows) * kScale8Stride * sizeof(float));
                if (e != hipSuccess) {
                    (void)hipFree(g.blk8);
                    g.blk8 = nullptr;
                }
            }
            if (e != hipSuccess) {
                (void)hipGetLastError();
                g.blk16 = g.blk8 = nullptr;
                g.scale8 = nullptr;
                if (s->screen_copy != PCV_SCREEN_COPY_AUTO)
                    PCV_FAIL(PCV_ERR_DEVICE, "hipMalloc of %.2f GB for the screening copy of %u rows failed: %s", bytes / 1e9, g.cap_rows,
                             hipGetErrorString(e));
                PCV_HIP(hipStreamSynchronize(st));
                drop_screening_copies(s);
                s->screen_copy_gave_way = true;
                return;
            }
        }
        if (kind == 1)
            launch_coarse_pack(st, g.blk, g.scale, g.blk16, g.copied_rows / kBlockRows, g.nblocks(), s->D4);
        else
            launch_coarse_pack8(st, g.blk, g.scale, g.blk8, g.scale8, g.copied_rows / kBlockRows, g.nblocks(), s->D4);
        g.copied_rows = g.scaled_rows;
    }
}

void do_finalize(pcv_searcher* s) {
    settle_mid_build(s, true);
    hipStream_t st = s->ctx->stream;
    for (auto& src : s->sources) {
        for (auto& g : src.segs) {
            if (g.scaled_rows >= g.nrows) continue;
            launch_row_scales(st, g.blk, g.scaled_rows / kBlockRows, g.nblocks(), g.nrows, s->D4, s->metric, g.scale,
                              s->d_max_norm_bits);
            g.scaled_rows = g.nrows;
        }
        build_screening_copies(s, src);
        // a reserved but never filled tail is given back
        while (!src.segs.empty() && src.segs.back().nrows == 0) {
            PCV_HIP(hipStreamSynchronize(st));
            free_segment(src.segs.back());
            src.segs.pop_back();
        }
        src.reserve = 0;
    }
    // sources emptied by clear_source and never refilled disappear (rebuild_source, search.rs:67-69)
    s->sources.erase(std::remove_if(s->sources.begin(), s->sources.end(),
                                    [](const Source& x) { return x.segs.empty(); }),
                     s->sources.end());
    if (s->mid_copy == PCV_MID_COPY_ON || (s->mid_copy == PCV_MID_COPY_AUTO && s->mids_present))
        build_mid_copies(s, s->mid_copy == PCV_MID_COPY_ON);  // new rows join the copy that is there
    assign_positions(s);
    s->copies_kind = s->sources.empty() ? 0 : copy_kind_wanted(s);
    for (const auto& src : s->sources)
        for (const auto& g : src.segs)
            if (g.nrows > 0 && ((s->copies_kind == 1 ? g.blk16 == nullptr : g.blk8 == nullptr) || g.copied_rows < g.nrows)) s->copies_kind = 0;
    if (s->mids_present)
        for (const auto& src : s->sources)
            for (const auto& g : src.segs)
                if (g.nrows > 0 && (!g.mid16 || g.mid_rows < g.nrows)) s->mids_present = false;
    uint32_t bits = 0;
    PCV_HIP(hipMemcpyAsync(&bits, s->d_max_norm_bits, 4, hipMemcpyDeviceToHost, st));
    PCV_HIP(hipStreamSynchronize(st));
    PCV_HIP(hipGetLastError());
    std::memcpy(&s->max_norm, &bits, 4);
    s->d_stage.release();
    s->dirty = false;
}

struct SelSeg {
    const Segment* g;
};

// source_ids == NULL: every source; otherwise exactly the n_sources listed ones (search.rs:166 — an
// empty list matches nothing)
std::vector<SelSeg> select_segments(pcv_searcher* s, const int64_t* source_ids, int n_sources) {
    std::vector<SelSeg> out;
    const bool all = source_ids == nullptr;
    for (const auto& src : s->sources) {
        // rows staged for a rebuild (PCV_STAGING_SOURCE) are nobody's rows yet: "every source" does not mean them — between the
        // staging finalize and the swap an all-sources search would otherwise return the new rows beside the old ones
        bool sel = all && src.id != PCV_STAGING_SOURCE;
        for (int i = 0; i < n_sources && !sel; ++i) sel = (source_ids[i] == src.id);  // search.rs:166
        if (!sel) continue;
        for (const auto& g : src.segs)
            if (g.nrows > 0) out.push_back({&g});
    }
    return out;
}

void ensure_workspace(pcv_searcher* s) {
    const size_t Q = kMfmaQueries;
    s->d_qf32.ensure(Q * s->Dp);
    s->d_qraw.ensure(Q * s->Dp);
    s->d_qbf16.ensure(Q * s->Dp);
    s->d_q8.ensure(Q * (size_t)((s->Dp + 127) & ~127));
    s->d_q8c.ensure(Q * 4);
    s->d_spec.ensure(Q);
    s->d_margin.ensure(Q);
    s->d_margin32.ensure(Q);
    s->d_tau.ensure((Q + 1) * kHot);  // (+ the side-by-side copy of the thresholds: ScanParams::tau_c)
    s->d_slots.ensure(Q * kMaxK);
    s->d_cnt.ensure(Q * kHot);
    s->d_cand.ensure(Q * s->cand_cap);
    s->d_cand_s.ensure(Q * s->cand_cap);
    s->d_hits.ensure(Q * kMaxK);
    if (!s->pin) PCV_HIP(hipHostMalloc((void**)&s->pin, sizeof(pcv_searcher::Pinned), hipHostMallocDefault));
    for (auto& e : s->ev)
        if (!e) PCV_HIP(hipEventCreate(&e));
}

// offsets inside the pass block
struct PassLayout {
    size_t off_seg, off_ceil, off_q, total;
};
// `with_ceil`: the pass has ceilings (scan.h, CeilRec: a search for more than kMaxK results) — their records sit between the
// segment table and the queries, so that they travel with every attempt of the pass
PassLayout pass_layout(const pcv_searcher* s, size_t nseg, bool with_ceil) {
    PassLayout L;
    L.off_seg = align_up(sizeof(ScanParams));
    L.off_ceil = align_up(L.off_seg + nseg * sizeof(SegDesc));
    L.off_q = with_ceil ? align_up(L.off_ceil + (size_t)kMfmaQueries * sizeof(CeilRec)) : L.off_ceil;
    L.total = L.off_q + (size_t)kMfmaQueries * s->D * sizeof(float);
    return L;
}
void ensure_pass_block(pcv_searcher* s, size_t nseg) {
    const size_t want = pass_layout(s, nseg, true).total;
    if (want <= s->pass_cap) return;
    PCV_HIP(hipStreamSynchronize(s->ctx->stream));
    if (s->pin_pass) (void)hipHostFree(s->pin_pass);
    if (s->d_pass) (void)hipFree(s->d_pass);
    s->pin_pass = s->d_pass = nullptr;
    s->pass_cap = 0;
    const size_t cap = want + want / 2;
    PCV_HIP(hipHostMalloc((void**)&s->pin_pass, cap, hipHostMallocDefault));
    PCV_HIP(hipMalloc((void**)&s->d_pass, cap));
    s->pass_cap = cap;
}

// Queue one pass (<= pass_queries() queries over any number of segments) on the context stream without
// waiting for it: one H2D of (parameters, segment table, queries), then prep_seed, scan and
// rescore_select, which writes the [B][k] hits into `d_out` (nullptr = s->d_hits) and — if `download` —
// into pinned host memory as well; the survivor counts always come back that way.  `d_flag` != nullptr
// receives the overflow record (scan.h).
void enqueue_pass(pcv_searcher* s, const float* queries_host, int B, const SelSeg* segs, int nseg, int k, int kernel,
                  pcv_hit_dev* d_out, bool download, pcv_hit_dev* d_flag, const CeilRec* ceil_host = nullptr,
                  bool queries_on_device = false) {
    // `queries_on_device`: queries_host is a DEVICE pointer (embeddings that never left the GPU: encode -> gather -> search of
    // BASELINE configs[4]); the pass copies them device to device behind its parameter upload.
    const auto t_begin = std::chrono::steady_clock::now();
    hipStream_t st = s->ctx->stream;
    ensure_workspace(s);
    ensure_pass_block(s, (size_t)nseg);
    if (!s->state_clean) launch_reset_scan_state(st, s->d_tau.p, s->d_slots.p, s->d_cnt.p);
    s->state_clean = false;  // until finish_pass has seen the pass through
    const PassLayout L = pass_layout(s, (size_t)nseg, ceil_host != nullptr);
    ScanParams& p = *reinterpret_cast<ScanParams*>(s->pin_pass);
    SegDesc* tab = reinterpret_cast<SegDesc*>(s->pin_pass + L.off_seg);
    p = ScanParams{};
    uint32_t blk0 = 0;
    int64_t rows = 0;
    bool have_mid = true;
    // stream the screening copies iff every selected segment has one of the kind the searcher keeps
    int src_kind = (kernel == PCV_KERNEL_MFMA) ? copy_kind_wanted(s) : 0;
    if (src_kind == 2 && (mfma8_pass_queries(s->Dp) < B || s->Dp > 1024)) src_kind = 0;
    const int src_wanted = src_kind;
    for (int i = 0; i < nseg; ++i) {
        const Segment& g = *segs[i].g;
        const bool mid = g.mid16 != nullptr && g.mid_rows >= g.nrows;  // (a copy that does not cover every row yet is not used)
        tab[i] = SegDesc{g.blk, g.scale, g.ids, g.id0, g.pos0, g.nrows, g.nblocks(), blk0, 0, g.blk16, g.blk8, mid ? g.mid16 : nullptr, mid ? g.scale16 : nullptr, g.scale8};
        have_mid = have_mid && mid;
        if ((src_kind == 1 ? g.blk16 == nullptr : (src_kind == 2 ? g.blk8 == nullptr : false)) || g.copied_rows < g.nrows) src_kind = 0;
        PCV_REQUIRE((uint64_t)blk0 + g.nblocks() < 0xffffff00ull, "search: more than 2^32 row blocks in one launch");
        blk0 += g.nblocks();
        rows += g.nrows;
    }
    if (src_kind != 2) {
        // the mid screen's bound uses |q'|_1, which only the int8 path's quantize_queries_kernel computes: passes that stream
        // the bf16 copy or the f32 rows (few coarse survivors anyway) go straight to the f32 row
        for (int i = 0; i < nseg; ++i) {
            tab[i].mid16 = nullptr;
            tab[i].scale16 = nullptr;
        }
        have_mid = false;
    }
    PCV_REQUIRE(B <= 128 || (kernel == PCV_KERNEL_MFMA && src_kind == 2 && src_wanted == 2), "search: %d queries in one pass without int8 copies of every selected row", B);
    p.seg = reinterpret_cast<const SegDesc*>(s->d_pass + L.off_seg);
    p.nseg = nseg;
    p.total_blocks = blk0;
    p.D = s->D;
    p.D4 = s->D4;
    p.B = B;
    p.k = k;
    p.metric = s->metric;
    p.tile_rows = kernel == PCV_KERNEL_MFMA ? mfma_tile_rows(B) : 0u;
    p.queries = reinterpret_cast<const float*>(s->d_pass + L.off_q);
    if (ceil_host) {
        std::memcpy(s->pin_pass + L.off_ceil, ceil_host, (size_t)B * sizeof(CeilRec));
        p.ceil = reinterpret_cast<const CeilRec*>(s->d_pass + L.off_ceil);
    }
    p.qf32 = s->d_qf32.p;
    p.qbf16 = s->d_qbf16.p;
    p.q8 = s->d_q8.p;
    p.q8c = s->d_q8c.p;
    p.qraw = s->d_qraw.p;
    p.margin = s->d_margin.p;
    p.margin32 = s->d_margin32.p;
    p.tau = s->d_tau.p;
    p.tau_c = s->d_tau.p + (size_t)kMfmaQueries * kHot;
    p.slots = s->d_slots.p;
    p.cand_cnt = s->d_cnt.p;
    p.cand = s->d_cand.p;
    p.cand_s = s->d_cand_s.p;
    p.out = d_out ? d_out : s->d_hits.p;
    p.out_host = download ? s->pin->hits : nullptr;
    p.cnt_host = s->pin->cnt;
    p.coarse_host = s->pin->coarse;
    p.flag_rec = d_flag;
    p.cand_cap = s->cand_cap;
    p.flags = (s->scan_flags & ~(16u | 64u)) | (src_kind == 1 ? 16u : 0u) | (src_kind == 2 ? 64u : 0u);
    const uint32_t seed_parts = ((s->scan_flags >> 16) & 0xff) ? ((s->scan_flags >> 16) & 0xff) : kSeedParts;  // tuning
    p.seed_blocks = std::min<uint32_t>(std::min<uint32_t>(seed_parts, kSeedParts) * kSeedPartRows / kBlockRows, segs[0].g->nblocks());
    p.seed_shift = 0;  // the seed blocks are every 2^shift-th block of segment 0, the largest stride that fits
    while (((uint64_t)p.seed_blocks << (p.seed_shift + 1)) <= segs[0].g->nblocks() && p.seed_shift < 20) ++p.seed_shift;
    // |s - c| bounds of the screening scores, relative to |q||x| (DESIGN.md §screening error): an f32 FMA
    // chain in any order, and one bf16 rounding per operand on top of it
    // speculative start threshold (scan.h): the int8 scan only (its queue has the kernel that sets it); j-th largest
    // seed slot with the smallest j whose guess fails with probability < 1e-6 on rows in an order unrelated to the query
    p.spec = s->d_spec.p;
    p.spec_rank = 0;
    p.spec_gap = NAN;
    p.spec_spread = 0.0f;
    p.spec_base_host = s->pin->spec_base;
    p.spec_top_host = s->pin->spec_top;
    p.kth_host = s->pin->kth;
    if (s->gap_rows != rows || s->gap_k != k || s->gap_nseg != nseg) {  // another shape: learn afresh
        s->gap_rows = rows;
        s->gap_k = k;
        s->gap_nseg = nseg;
        s->gaps.reset();
    }
    // (no guess under a ceiling: the check at the end of the pass counts survivors, not survivors that count)
    if (kernel == PCV_KERNEL_MFMA && !s->spec_hold && s->spec_rest == 0 && !(s->scan_flags & 32u) && k >= 2 && !ceil_host) {
        if (!(s->scan_flags & 128u)) p.spec_gap = s->gaps.gap();
        p.spec_spread = (float)s->gaps.spread;
        const double r = (double)std::min<int64_t>(tab[0].nrows, (int64_t)p.seed_blocks * kBlockRows) / (double)std::max<int64_t>(rows, 1);  // (seed rows) / rows
        double binom = 1.0, rj = 1.0;
        for (int j = 1; j < k && r < 0.25; ++j) {
            binom *= (double)(k - j) / (double)j;  // C(k-1, j)
            rj *= r;
            if (binom * rj < 1e-6) {
                p.spec_rank = j;
                break;
            }
        }
    }
#ifdef PCV_STAMPS
    s->d_stamps.ensure(8 * 65536);
    PCV_HIP(hipMemsetAsync(s->d_stamps.p, 0, 8 * 65536 * sizeof(unsigned long long), st));
    p.stamps = s->d_stamps.p;
#endif
    p.eps32 = (float)(s->Dp + 16) * 1.2e-7f;
    p.eps16 = 0.0039101f + 2.0f * p.eps32;
    p.max_norm = s->max_norm;

    size_t bytes = L.off_q;
    const float* queries_dev = nullptr;
    if (queries_host && queries_on_device) {
        queries_dev = queries_host;
    } else if (queries_host) {  // nullptr: the queries of the previous attempt are still on the device
        std::memcpy(s->pin_pass + L.off_q, queries_host, (size_t)B * s->D * sizeof(float));
        bytes += (size_t)B * s->D * sizeof(float);
    }
    const ScanParams* dp = reinterpret_cast<const ScanParams*>(s->d_pass);
    // `timed`: with the event records around the scan kernel.  Records captured into a graph do not give times on
    // replay, so a replayed pass is bracketed from outside (ev[0], ev[3]) and its scan time is taken as the share of that
    // total which the scan kernel had in the plain launches of the same shape.
    auto launch_pass = [&](bool timed) {
        if (timed) PCV_HIP(hipEventRecord(s->ev[0], st));
        launch_upload(st, s->pin_pass, s->d_pass, bytes);
        if (queries_dev)
            PCV_HIP(hipMemcpyAsync(s->d_pass + L.off_q, queries_dev, (size_t)B * s->D * sizeof(float), hipMemcpyDeviceToDevice, st));
        launch_prep_seed(st, p, dp, tab[0]);
        if (timed) PCV_HIP(hipEventRecord(s->ev[1], st));
        if (kernel == PCV_KERNEL_MFMA && src_kind == 2)
            launch_scan_mfma8(st, p, dp, s->ctx->num_cus);
        else if (kernel == PCV_KERNEL_MFMA)
            launch_scan_mfma(st, p, dp, s->ctx->num_cus);
        else
            launch_scan_wave(st, p, dp, s->ctx->num_cus);
        if (timed) PCV_HIP(hipEventRecord(s->ev[2], st));
        launch_rescore_select(st, p, dp);
        if (timed) PCV_HIP(hipEventRecord(s->ev[3], st));
    };
    pcv_searcher::PassShape shape;
    shape.B = B;
    shape.k = k;
    shape.kernel = kernel;
    shape.src_kind = src_kind;
    shape.guess = (p.spec_rank > 0 || p.spec_gap == p.spec_gap) ? 1 : 0;
    shape.nseg = nseg;
    shape.total_blocks = blk0;
    shape.seed_blocks = p.seed_blocks;
    shape.flags = p.flags;
    shape.seg0_rows = tab[0].nrows;
    shape.bytes = bytes;
    shape.pin = s->pin_pass;
    shape.dev = s->d_pass;
    shape.seg0_blk = tab[0].blk;
    shape.seg0_scale = tab[0].scale;
    // only where queueing is a visible share of the pass: up to kGraphRows rows (a longer pass is launched plainly and
    // timed kernel by kernel, which is what the roofline figures are taken from)
    const bool small = rows <= kGraphRows && s->use_graph && !ceil_host && !queries_dev;
    bool replayed = false;
    if (small && s->graph_exec && shape == s->graph_shape) {
        PCV_HIP(hipEventRecord(s->ev[0], st));
        PCV_HIP(hipGraphLaunch(s->graph_exec, st));
        PCV_HIP(hipEventRecord(s->ev[3], st));
        replayed = true;
    } else if (small && shape == s->last_shape && ++s->shape_seen >= 3 && s->shape_fixed_ms >= 0.0f) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        PCV_HIP(hipEventRecord(s->ev[0], st));
        bool ok = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            try {
                launch_pass(false);
            } catch (...) {
                ok = false;
            }
            ok = (hipStreamEndCapture(st, &graph) == hipSuccess) && ok && graph != nullptr;
        }
        if (ok && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
            s->graph_exec = exec;
            s->graph_shape = shape;
            s->graph_fixed_ms = s->shape_fixed_ms;
            PCV_HIP(hipGraphLaunch(exec, st));
            PCV_HIP(hipEventRecord(s->ev[3], st));
            replayed = true;
        } else {  // capture is not possible here: stay with plain launches for good
            (void)hipGetLastError();
            s->use_graph = false;
            launch_pass(true);
        }
        if (graph) (void)hipGraphDestroy(graph);
    } else {
        if (!(shape == s->last_shape)) {
            s->last_shape = shape;
            s->shape_seen = 1;
            s->shape_fixed_ms = -1.0f;
        }
        launch_pass(true);
    }
    s->pending.replayed = replayed;
    s->pending.active = true;
    s->pending.done = false;
    s->pending.B = B;
    s->pending.rows = rows;
    s->pending.src = src_kind;
    s->pending.mid = have_mid && nseg > 0;
    // what the scan kernel of this pass must pull from HBM, per 32-row block: the int8 pieces + the block's scale; the bf16 pieces; or the f32 pieces + the 32 row scales
    const int64_t Dp8 = (s->Dp + 127) & ~127;
    s->pending.stream_bytes = (int64_t)blk0 * (src_kind == 2 ? Dp8 * kBlockRows + (int64_t)kScale8Stride * 4
                                               : src_kind == 1 ? (int64_t)s->Dp * 2 * kBlockRows
                                                               : (int64_t)s->Dp * 4 * kBlockRows + kBlockRows * 4);
    s->pending.learned = p.spec_gap == p.spec_gap;
    s->pending.guessing = p.spec_rank > 0 || s->pending.learned;
    s->stats.host_enqueue_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
}

// Collect a queued pass: wait for the stream, book the statistics, and report whether a candidate list
// overflowed — nothing was lost then but the stored prefix is incomplete, so the lists are grown to what
// the pass needed and the caller repeats it (tau restarts, so the need can only be met or shrink on
// data that is not adversarially ordered; bounded by the row count).
bool finish_pass(pcv_searcher* s) {
    PCV_REQUIRE(s->pending.active, "no pass is pending");
    s->pending.active = false;
    const auto t_begin = std::chrono::steady_clock::now();
    PCV_HIP(hipStreamSynchronize(s->ctx->stream));  // also when nothing was launched: the caller's exchange may be queued
    s->stats.host_wait_ms += std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    PCV_HIP(hipGetLastError());
    if (s->pending.done) return false;
#ifdef PCV_STAMPS
    if (const char* f = getenv("PCV_STAMPS_FILE")) {  // one record per pass: 65536 waves x 8 words
        std::vector<unsigned long long> h(8 * 65536);
        PCV_HIP(hipMemcpy(h.data(), s->d_stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE* fp = fopen(f, "ab")) {
            fwrite(h.data(), sizeof(unsigned long long), h.size(), fp);
            fclose(fp);
        }
    }
#endif
    s->state_clean = true;  // rescore_select_kernel left the scan state as a pass expects it
    const int B = s->pending.B;
    const int64_t rows = s->pending.rows;
    float ms_scan = 0, ms_total = 0;
    (void)hipEventElapsedTime(&ms_total, s->ev[0], s->ev[3]);
    if (s->pending.replayed) {
        ms_scan = ms_total * s->graph_fixed_ms;
    } else {
        (void)hipEventElapsedTime(&ms_scan, s->ev[1], s->ev[2]);
        s->shape_fixed_ms = ms_total > 0.0f ? std::min(1.0f, std::max(0.0f, ms_scan / ms_total)) : 0.0f;
    }
    s->stats.scan_ms += ms_scan;
    s->stats.total_ms += ms_total;
    s->stats.scan_launches += 1;
    s->stats.rows_scanned += rows;
    s->stats.bytes_algorithmic += rows * (int64_t)s->D * 4;
    s->stats.bytes_streamed += s->pending.stream_bytes;
    s->stats.screening_copy = s->pending.src;

    const uint32_t* cnt = s->pin->cnt;
    uint32_t mx = 0;
    int64_t sum = 0;
    bool guess_failed = false;
    for (int b = 0; b < B; ++b) {
        if (cnt[b] == kSpecFailed) {  // fewer than k rows at the speculative threshold (scan.h): repeat without it
            guess_failed = true;
            continue;
        }
        mx = std::max(mx, cnt[b]);
        sum += cnt[b];
    }
    int64_t coarse = 0;
    for (int b = 0; b < B; ++b) {
        coarse += s->pin->coarse[b];
        s->stats.mid_survivors += s->pin->coarse[kMfmaQueries + b];
    }
    s->stats.coarse_survivors += coarse;
    s->stats.mid_copy = s->pending.mid ? 1 : 0;
    // AUTO: a corpus whose coarse screen keeps letting thousands of rows per query through gets its mid copy (built by the
    // next search call, before its passes: search_hits), and so does one where the survivors' f32 rows are a visible share
    // of the pass's traffic (a survivor pulls every 128-byte line its 16-byte pieces lie in: 12 KB at 384-d, 24 KB at 768-d;
    // measured gain of the copy: 12.5M x 384 3.5 %, 50M x 768 3.8 %, nothing at 100M x 384 where the share is 1.2 %)
    const int64_t fine_bytes = coarse * (int64_t)s->Dp * 32;
    if (s->pending.src == 2 && !s->pending.mid &&
        (coarse > kMidTrigger * (int64_t)B || fine_bytes * kMidShare > (int64_t)s->pending.stream_bytes))
        s->mid_hot_passes += 1;
    else
        s->mid_hot_passes = 0;
    if (mx <= s->cand_cap && !guess_failed) {
        s->stats.candidates += sum;
        s->spec_hold = false;
        if (s->spec_rest > 0) s->spec_rest -= 1;
        if (s->spec_penalty > 0 && ++s->spec_clean >= 4096) s->spec_penalty = s->spec_clean = 0;
        if (s->pending.guessing) {  // what this pass teaches about the gap (scan.h: spec_gap)
            for (int b = 0; b < B; ++b) {
                const float d = s->pin->kth[b] - s->pin->spec_base[b];
                const float sp = s->pin->spec_top[b] - s->pin->spec_base[b];
                if (d == d && std::isfinite(d) && sp == sp && std::isfinite(sp)) s->gaps.add(d, sp);
            }
        }
        if (s->gaps.holdoff > 0) s->gaps.holdoff -= 1;
        return false;
    }
    if (guess_failed) {
        s->stats.speculation_reruns += 1;
        s->spec_hold = true;
        if (s->pending.learned) {  // what was learned did not hold: start over, and not before 256 passes have gone by
            s->gaps.reset();
            s->gaps.holdoff = 256;
        } else {  // the seed rows were not a fair sample for this query: no guesses for a while
            s->spec_penalty = std::min(1024, s->spec_penalty ? 2 * s->spec_penalty : 16);
            s->spec_rest = s->spec_penalty;
            s->spec_clean = 0;
        }
        if (mx <= s->cand_cap) return true;
    }
    s->stats.overflow_reruns += 1;
    uint64_t want = (uint64_t)mx + mx / 4 + 1024;
    want = std::min<uint64_t>(want, (uint64_t)rows + 1024);
    s->cand_cap = (uint32_t)std::max<uint64_t>(want, s->cand_cap * 2ull);
    s->d_cand.ensure((size_t)kMfmaQueries * s->cand_cap);
    s->d_cand_s.ensure((size_t)kMfmaQueries * s->cand_cap);
    return true;
}

// One pass, synchronously.  Leaves [B][k] hits in `d_out` (nullptr = s->d_hits) and, if `download`, in s->pin->hits.
void run_pass(pcv_searcher* s, const float* queries_host, int B, const SelSeg* segs, int nseg, int k, int kernel,
              pcv_hit_dev* d_out, bool download, const CeilRec* ceil_host = nullptr) {
    for (int attempt = 0;; ++attempt) {
        enqueue_pass(s, attempt == 0 ? queries_host : nullptr, B, segs, nseg, k, kernel, d_out, download, nullptr, ceil_host);
        if (!finish_pass(s)) return;
        PCV_REQUIRE(attempt < 7, "candidate lists still overflow after %d reruns", attempt + 1);
    }
}

bool hit_better(const pcv_hit_dev& a, const pcv_hit_dev& b) {
    if (a.score != b.score) return a.score > b.score;
    return a.pos < b.pos;
}

int pick_kernel(const pcv_searcher* s, int B) {
    const bool mfma_ok = mfma_pass_queries(s->Dp) > 0;  // the query tile must fit the LDS
    if (s->kernel == PCV_KERNEL_MFMA && !mfma_ok)
        PCV_FAIL(PCV_ERR_UNSUPPORTED, "the MFMA kernel cannot hold a %d-d query tile in LDS", s->D);
    if (s->kernel == PCV_KERNEL_WAVE || s->kernel == PCV_KERNEL_MFMA) return s->kernel;
    // with screening copies the MFMA kernel streams half the bytes of the wave kernel, whatever the batch
    if (mfma_ok && s->copies_kind != 0) return PCV_KERNEL_MFMA;
    return (B <= kMaxWaveQueries || !mfma_ok) ? PCV_KERNEL_WAVE : PCV_KERNEL_MFMA;
}

// `among_ranks`: the split must be the same on every rank of a sharded search, whatever copies each of them holds
int pass_queries(const pcv_searcher* s, int kernel, bool among_ranks = false) {
    if (kernel == PCV_KERNEL_WAVE) return kMaxWaveQueries;
    // every row has its int8 copy: the int8 scan's pass (256 queries up to 384-d); else what the bf16 / f32-row scans take.
    // Among ranks only if the host has said that this holds on EVERY rank (pcv_searcher_allow_wide_sharded_pass): the split of a
    // batch into passes is part of the exchange's protocol.
    if ((!among_ranks || s->wide_sharded) && s->copies_kind == 2 && s->Dp <= 1024)
        return std::min(kMfmaQueries, std::max(mfma8_pass_queries(s->Dp), mfma_pass_queries(s->Dp)));
    return mfma_pass_queries(s->Dp);
}

void check_search_args(const pcv_searcher* s, const float* queries, int n_queries, int k, const char* who, int k_max = kMaxK) {
    PCV_REQUIRE(!s->dirty, "%s: rows were added or cleared without pcv_searcher_finalize", who);
    PCV_REQUIRE(queries != nullptr && n_queries > 0, "%s: no queries", who);
    PCV_REQUIRE(k > 0 && k <= k_max, "%s: num_results %d outside [1,%d]", who, k, k_max);
}

// The ceilings of the pass that follows one whose last hits are `last` (one per query; pos < 0: that pass came back short —
// the query has all the rows there are): scan.h, CeilRec.  The band around the boundary is the fine screen's margin,
// 2 eps32 relative to |q||x| (x <= the largest row norm for the dot metric; 1 for cosine), plus the rounding of the f64 score to f32.
void next_ceilings(const pcv_searcher* s, const float* queries, int B, const pcv_hit_dev* last, CeilRec* out) {
    const float eps32 = (float)(s->Dp + 16) * 1.2e-7f;
    for (int q = 0; q < B; ++q) {
        CeilRec& c = out[q];
        if (last[q].pos < 0) {
            c = CeilRec{-INFINITY, INT64_MAX, -INFINITY, -INFINITY};
            continue;
        }
        double unit = 1.0;
        if (s->metric == PCV_METRIC_DOT) {
            double nq = 0.0;
            for (int i = 0; i < s->D; ++i) nq += (double)queries[(size_t)q * s->D + i] * (double)queries[(size_t)q * s->D + i];
            unit = std::sqrt(nq) * (double)s->max_norm;
        }
        const double band = 2.5 * (double)eps32 * unit + std::fabs(last[q].score) * 2.4e-7 + 1e-37;
        c.score = last[q].score;
        c.pos = last[q].pos;
        c.lo = std::nextafterf((float)(last[q].score - band), -INFINITY);
        c.hi = std::nextafterf((float)(last[q].score + band), INFINITY);
    }
}

// Full search: any number of queries / segments; result [n_queries][k] hits on the host.
void search_hits(pcv_searcher* s, const float* queries, int n_queries, const int64_t* source_ids, int n_sources,
                 int k, std::vector<pcv_hit_dev>& out) {
    check_search_args(s, queries, n_queries, k, "search", 1 << 24);
    PCV_HIP(hipSetDevice(s->ctx->device));
    std::vector<SelSeg> segs = select_segments(s, source_ids, n_sources);
    const pcv_hit_dev none{NAN, -1, -1};
    out.assign((size_t)n_queries * k, none);
    s->stats = pcv_scan_stats{};
    const int kernel = pick_kernel(s, n_queries);
    s->stats.kernel_used = kernel;
    if (segs.empty()) return;
    const int qstep = pass_queries(s, kernel);
    maybe_build_mid_copies(s);
    std::vector<CeilRec> ceil;
    std::vector<pcv_hit_dev> last;
    for (int q0 = 0; q0 < n_queries; q0 += qstep) {
        const int B = std::min(qstep, n_queries - q0);
        const float* qs = queries + (size_t)q0 * s->D;
        // One pass ranks up to kMaxK results.  More (search.rs:157-182 has no limit): the next kMaxK below the last hit of the
        // pass before, and so on — each pass exact among the rows that rank after its ceiling (the statistics add up over the passes).
        for (int got = 0; got < k; got += kMaxK) {
            const int kk = std::min(kMaxK, k - got);
            if (got > 0) {
                ceil.resize((size_t)B);
                next_ceilings(s, qs, B, last.data(), ceil.data());
            }
            run_pass(s, qs, B, segs.data(), (int)segs.size(), kk, kernel, nullptr, true, got > 0 ? ceil.data() : nullptr);
            last.resize((size_t)B);
            bool more = false;
            for (int q = 0; q < B; ++q) {  // (the hits came down with the pass)
                std::memcpy(out.data() + (size_t)(q0 + q) * k + got, s->pin->hits + (size_t)q * kk, (size_t)kk * sizeof(pcv_hit_dev));
                last[(size_t)q] = s->pin->hits[(size_t)q * kk + kk - 1];  // pos < 0: fewer than kk rows counted — nothing is left
                more = more || last[(size_t)q].pos >= 0;
            }
            if (!more) break;
        }
    }
}

// hits -> the reference's result convention
void hits_to_outputs(int metric, int D, const pcv_hit_dev* hits, int n_queries, int k, int64_t* out_ids,
                     float* out_scores, int* out_counts) {
    for (int q = 0; q < n_queries; ++q) {
        int cnt = 0;
        for (int j = 0; j < k; ++j) {
            const pcv_hit_dev& h = hits[(size_t)q * k + j];
            const bool ok = h.pos >= 0;
            if (ok) ++cnt;
            if (out_ids) out_ids[(size_t)q * k + j] = ok ? h.id : -1;
            if (out_scores) {
                float v = NAN;
                if (ok) {
                    if (metric == PCV_METRIC_DOT) {
                        const double d = 1.0 - h.score / (double)D;  // search.rs:275
                        v = (float)(d > 0.0 ? d : 0.0);              // search.rs:277
                    } else {
                        v = (float)h.score;
                    }
                }
                out_scores[(size_t)q * k + j] = v;
            }
        }
        if (out_counts) out_counts[q] = cnt;
    }
}

// The per-shard pass of the begin/end protocol; the caller holds s->mu.
void device_begin(pcv_searcher* s, const float* queries, int n_queries, const int64_t* source_ids, int n_sources, int k,
                  pcv_hit_dev* out, bool queries_on_device = false) {
    check_search_args(s, queries, n_queries, k, "search_device_begin");
    PCV_REQUIRE(!s->pending.active, "search_device_begin: the previous pass was not collected (search_device_end)");
    PCV_HIP(hipSetDevice(s->ctx->device));
    std::vector<SelSeg> segs = select_segments(s, source_ids, n_sources);
    const int kernel = pick_kernel(s, n_queries);
    // Only a condition every rank evaluates alike may refuse: the ranks of a sharded search must all
    // take the same protocol (the exchanged payload differs by the overflow record).
    if (n_queries > pass_queries(s, kernel, true))
        PCV_FAIL(PCV_ERR_UNSUPPORTED, "search_device_begin: %d queries need more than one pass%s", n_queries,
                 s->wide_sharded ? " (wide sharded passes were allowed, but not every row of THIS rank has its int8 screening copy)" : "");
    const size_t n = (size_t)n_queries * k;
    s->stats = pcv_scan_stats{};
    s->stats.kernel_used = kernel;
    if (segs.empty()) {
        // this shard holds none of the selected rows: hand over the same layout — empty lists and a clear
        // overflow record — from pinned memory; search_device_end waits for the copy like for a pass
        ensure_workspace(s);
        const pcv_hit_dev none{NAN, -1, -1};
        for (size_t i = 0; i < n; ++i) s->pin->hits[i] = none;
        s->pin->hits[n] = pcv_hit_dev{0.0, 0, 0};
        PCV_HIP(hipMemcpyAsync(out, s->pin->hits, (n + 1) * sizeof(pcv_hit_dev), hipMemcpyHostToDevice, s->ctx->stream));
        s->pending.active = true;
        s->pending.done = true;
        return;
    }
    maybe_build_mid_copies(s);  // (each rank by its own statistics: the copy changes no result and no protocol)
    enqueue_pass(s, queries, n_queries, segs.data(), (int)segs.size(), k, kernel, out, false, out + n, nullptr, queries_on_device);
}

}  // namespace

extern "C" {

pcv_status pcv_searcher_create(pcv_ctx* ctx, int dim, int metric, pcv_searcher** out) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && out != nullptr, "searcher_create: NULL argument");
        *out = nullptr;
        PCV_REQUIRE(dim > 0 && dim <= 8192, "searcher_create: dim %d outside [1,8192]", dim);
        PCV_REQUIRE(metric == PCV_METRIC_COSINE || metric == PCV_METRIC_DOT, "searcher_create: unknown metric %d",
                    metric);
        PCV_HIP(hipSetDevice(ctx->device));
        auto s = std::make_unique<pcv_searcher>();
        s->ctx = ctx;
        s->D = dim;
        s->Dp = (dim + 63) / 64 * 64;
        s->D4 = s->Dp / 4;
        s->metric = metric;
        if (const char* f = getenv("PCV_SCAN_FLAGS")) s->scan_flags = (uint32_t)strtoul(f, nullptr, 0);
        if (getenv("PCV_NO_SCAN_GRAPH")) s->use_graph = false;
        if (const char* f = getenv("PCV_SCREEN_COPY")) {  // 0 off, 1 on, 2 auto
            const int mode = atoi(f);
            if (mode >= PCV_SCREEN_COPY_OFF && mode <= PCV_SCREEN_COPY_INT8) s->screen_copy = mode;
        }
        PCV_HIP(hipMalloc((void**)&s->d_max_norm_bits, 4));
        // on the stream the row_scales atomics will run on (the context stream does not wait for the null stream)
        hipError_t e = hipMemsetAsync(s->d_max_norm_bits, 0, 4, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            (void)hipFree(s->d_max_norm_bits);
            PCV_FAIL(PCV_ERR_DEVICE, "searcher_create: %s", hipGetErrorString(e));
        }
        *out = s.release();
    });
}

pcv_status pcv_searcher_destroy(pcv_searcher* s) {
    return guarded([&] {
        if (!s) return;
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        try {
            settle_mid_build(s, true);
        } catch (...) {
        }
        if (s->side) {
            (void)hipStreamSynchronize(s->side);
            (void)hipStreamDestroy(s->side);
            (void)hipEventDestroy(s->side_go);
            (void)hipEventDestroy(s->mid_done);
        }
        for (auto& src : s->sources)
            for (auto& g : src.segs) free_segment(g);
        s->d_stage.release();
        s->d_qf32.release();
        s->d_qraw.release();
        s->d_margin.release();
        s->d_margin32.release();
        s->d_qbf16.release();
        s->d_q8.release();
        s->d_q8c.release();
        s->d_spec.release();
        s->d_cand_s.release();
        s->d_tau.release();
        s->d_slots.release();
        s->d_cnt.release();
        s->d_cand.release();
        s->d_hits.release();
        if (s->graph_exec) (void)hipGraphExecDestroy(s->graph_exec);
        if (s->pin) (void)hipHostFree(s->pin);
        if (s->pin_pass) (void)hipHostFree(s->pin_pass);
        if (s->d_pass) (void)hipFree(s->d_pass);
        if (s->d_max_norm_bits) (void)hipFree(s->d_max_norm_bits);
        for (auto& e : s->ev)
            if (e) (void)hipEventDestroy(e);
        delete s;
    });
}

pcv_status pcv_searcher_reserve(pcv_searcher* s, int64_t source_id, int64_t n_rows) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "reserve: searcher is NULL");
        PCV_REQUIRE(n_rows >= 0, "reserve: negative row count");
        std::lock_guard<std::mutex> lk(s->mu);
        Source& src = s->get_or_add_source(source_id);
        src.reserve = src.rows() + n_rows;
    });
}

pcv_status pcv_searcher_add_rows(pcv_searcher* s, int64_t source_id, const int64_t* ids, const float* rows,
                                 int64_t n) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "add_rows: searcher is NULL");
        PCV_REQUIRE(n >= 0 && (rows != nullptr || n == 0), "add_rows: bad rows/n");
        std::lock_guard<std::mutex> lk(s->mu);
        Source& src = s->get_or_add_source(source_id);
        if (n == 0) return;
        PCV_HIP(hipSetDevice(s->ctx->device));
        append_rows(s, src, ids, rows, n);
    });
}

pcv_status pcv_searcher_add_blobs(pcv_searcher* s, int64_t source_id, const int64_t* ids, const uint8_t* blobs,
                                  int64_t n) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "add_blobs: searcher is NULL");
        PCV_REQUIRE(n >= 0 && (blobs != nullptr || n == 0), "add_blobs: bad blobs/n");
        // deserialize_embedding (search.rs:281-286) reads little-endian f32; this library only runs on
        // little-endian hosts (x86-64 + gfx950), where that is the identity on the bytes
        const uint32_t probe = 1;
        PCV_REQUIRE(*reinterpret_cast<const uint8_t*>(&probe) == 1, "add_blobs: big-endian host");
        std::lock_guard<std::mutex> lk(s->mu);
        Source& src = s->get_or_add_source(source_id);
        if (n == 0) return;
        PCV_HIP(hipSetDevice(s->ctx->device));
        append_rows(s, src, ids, blobs, n);
    });
}

static pcv_status add_synthetic(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed, int64_t first_row,
                                int normalize, int n_clusters, float noise, float amp_lo = 0.0f, float amp_hi = 0.0f) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "add_synthetic: searcher is NULL");
        PCV_REQUIRE(n >= 0, "add_synthetic: negative row count");
        PCV_REQUIRE(n_clusters >= 0 && noise >= 0.0f && std::isfinite(noise), "add_synthetic: bad cluster shape");
        PCV_REQUIRE(std::isfinite(amp_lo) && std::isfinite(amp_hi) && amp_lo >= 0.0f && amp_hi >= amp_lo, "add_synthetic: bad amplitude range");
        PCV_REQUIRE(s->D % 4 == 0, "add_synthetic: dim %d is not a multiple of 4", s->D);
        std::lock_guard<std::mutex> lk(s->mu);
        Source& src = s->get_or_add_source(source_id);
        if (n == 0) return;
        PCV_HIP(hipSetDevice(s->ctx->device));
        hipStream_t st = s->ctx->stream;
        // one segment per <= 2^31 rows, generated in place
        const int64_t kMaxRows = (int64_t)1 << 31;
        for (int64_t r0 = 0; r0 < n; r0 += kMaxRows) {
            const int64_t m = std::min(kMaxRows, n - r0);
            Segment g = alloc_segment(s, m, false);
            g.id0 = first_row + r0;
            try {
                launch_synth_fill(st, g.blk, (uint32_t)m, 0, s->D, s->D4, seed, first_row + r0, normalize,
                                  (uint32_t)n_clusters, noise, amp_lo, amp_hi);
                PCV_HIP(hipStreamSynchronize(st));
                PCV_HIP(hipGetLastError());
            } catch (...) {
                free_segment(g);
                throw;
            }
            g.nrows = (uint32_t)m;
            src.segs.push_back(g);
            s->dirty = true;
        }
    });
}

pcv_status pcv_searcher_add_synthetic(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                      int64_t first_row, int normalize) {
    return add_synthetic(s, source_id, n, seed, first_row, normalize, 0, 0.0f);
}

pcv_status pcv_searcher_add_synthetic_clustered(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                                int64_t first_row, int normalize, int n_clusters, float noise) {
    return add_synthetic(s, source_id, n, seed, first_row, normalize, n_clusters, noise);
}

pcv_status pcv_searcher_add_synthetic_scaled(pcv_searcher* s, int64_t source_id, int64_t n, uint64_t seed,
                                             int64_t first_row, float amp_lo, float amp_hi) {
    if (!(amp_hi > amp_lo) || !(amp_lo > 0.0f)) {
        set_error("add_synthetic_scaled: amplitudes must satisfy 0 < amp_lo < amp_hi");
        return PCV_ERR_INVALID;
    }
    return add_synthetic(s, source_id, n, seed, first_row, 0, 0, 0.0f, amp_lo, amp_hi);
}

pcv_status pcv_searcher_clear_source(pcv_searcher* s, int64_t source_id) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "clear_source: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        Source* src = s->find_source(source_id);
        if (!src) return;
        PCV_HIP(hipSetDevice(s->ctx->device));
        PCV_HIP(hipStreamSynchronize(s->ctx->stream));
        settle_mid_build(s, true);
        if (!src->segs.empty()) s->screen_copy_gave_way = s->mid_gave_way = false;  // rows are given back: AUTO may try its copies again
        for (auto& g : src->segs) free_segment(g);
        src->segs.clear();
        src->next_implicit_id = 0;
        src->reserve = 0;
        s->dirty = true;
    });
}

pcv_status pcv_searcher_replace_source(pcv_searcher* s, int64_t from_source_id, int64_t to_source_id) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "replace_source: searcher is NULL");
        PCV_REQUIRE(from_source_id != to_source_id, "replace_source: a source cannot replace itself");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "replace_source: a queued pass has not been collected");
        PCV_HIP(hipSetDevice(s->ctx->device));
        PCV_HIP(hipStreamSynchronize(s->ctx->stream));
        settle_mid_build(s, true);
        std::vector<Segment> fresh;
        int64_t next_id = 0;
        for (size_t i = 0; i < s->sources.size(); ++i)
            if (s->sources[i].id == from_source_id) {
                fresh = std::move(s->sources[i].segs);
                next_id = s->sources[i].next_implicit_id;
                s->sources.erase(s->sources.begin() + (std::ptrdiff_t)i);
                break;
            }
        Source* to = s->find_source(to_source_id);
        if (!to && !fresh.empty()) to = &s->get_or_add_source(to_source_id);
        if (to) {  // (the old rows keep their place among the sources: positions of the others do not move)
            // The staged rebuild holds the old and the new rows of a source at once.  If AUTO gave its copies up to fit them
            // (alloc_segment: the rows come first), the old rows going now is the room to have them again: the next finalize
            // tries — once; another failure gives them up again.  (Without this every later search of a searcher that once
            // rebuilt a large source on a nearly full device streamed the f32 rows, 4x the bytes, silently.)
            if (!to->segs.empty()) s->screen_copy_gave_way = s->mid_gave_way = false;
            for (auto& g : to->segs) free_segment(g);
            to->segs = std::move(fresh);
            to->next_implicit_id = next_id;
            to->reserve = 0;
        }
        s->dirty = true;  // positions are handed out again by finalize (an emptied source disappears there)
    });
}

pcv_status pcv_searcher_finalize(pcv_searcher* s) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "finalize: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_HIP(hipSetDevice(s->ctx->device));
        do_finalize(s);
    });
}

pcv_status pcv_searcher_dim(pcv_searcher* s, int* out_dim) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_dim != nullptr, "searcher_dim: NULL argument");
        *out_dim = s->D;
    });
}

pcv_status pcv_searcher_num_rows(pcv_searcher* s, int64_t* out_rows) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_rows != nullptr, "num_rows: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        int64_t n = 0;
        for (auto& src : s->sources)
            if (src.id != PCV_STAGING_SOURCE) n += src.rows();  // (staged rows are not the searcher's rows yet)
        *out_rows = n;
    });
}

pcv_status pcv_searcher_num_segments(pcv_searcher* s, int* out_n) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_n != nullptr, "num_segments: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        int n = 0;
        for (auto& src : s->sources) n += (int)src.segs.size();
        *out_n = n;
    });
}

pcv_status pcv_searcher_num_sources(pcv_searcher* s, int* out_n) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_n != nullptr, "num_sources: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        int n = 0;
        for (const auto& src : s->sources) n += src.id != PCV_STAGING_SOURCE ? 1 : 0;
        *out_n = n;
    });
}

pcv_status pcv_searcher_source_ids(pcv_searcher* s, int64_t* out_ids, int cap) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_ids != nullptr, "source_ids: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        size_t n = 0;
        for (const auto& src : s->sources) n += src.id != PCV_STAGING_SOURCE ? 1 : 0;
        PCV_REQUIRE(cap >= (int)n, "source_ids: capacity %d < %zu sources", cap, n);
        n = 0;
        for (const auto& src : s->sources)
            if (src.id != PCV_STAGING_SOURCE) out_ids[n++] = src.id;
    });
}

pcv_status pcv_searcher_source_num_rows(pcv_searcher* s, int64_t source_id, int64_t* out_rows) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out_rows != nullptr, "source_num_rows: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->dirty, "source_num_rows: pending rows; call pcv_searcher_finalize first");
        *out_rows = 0;
        if (Source* src = s->find_source(source_id)) *out_rows = src->rows();
    });
}

pcv_status pcv_searcher_get_rows(pcv_searcher* s, const int64_t* positions, int64_t n, float* out_rows,
                                 int64_t* out_ids) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && positions != nullptr && out_rows != nullptr && n >= 0, "get_rows: bad argument");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->dirty, "get_rows: pending rows; call pcv_searcher_finalize first");
        if (n == 0) return;
        PCV_HIP(hipSetDevice(s->ctx->device));
        hipStream_t st = s->ctx->stream;
        std::vector<SegDesc> segs;
        for (auto& src : s->sources)
            for (auto& g : src.segs)
                segs.push_back(SegDesc{g.blk, g.scale, g.ids, g.id0, g.pos0, g.nrows, g.nblocks(), 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr});
        DevBuf<SegDesc> d_segs;
        DevBuf<int64_t> d_pos, d_ids;
        DevBuf<float> d_rows;
        try {
            d_segs.ensure(segs.size() + 1);
            d_pos.ensure(n);
            d_ids.ensure(n);
            d_rows.ensure((size_t)n * s->D);
            PCV_HIP(hipMemcpyAsync(d_segs.p, segs.data(), segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, st));
            PCV_HIP(hipMemcpyAsync(d_pos.p, positions, (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice, st));
            launch_gather_rows(st, d_segs.p, (int)segs.size(), d_pos.p, n, s->D, s->D4, d_rows.p, d_ids.p);
            PCV_HIP(hipMemcpyAsync(out_rows, d_rows.p, (size_t)n * s->D * sizeof(float), hipMemcpyDeviceToHost, st));
            if (out_ids)
                PCV_HIP(hipMemcpyAsync(out_ids, d_ids.p, (size_t)n * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            PCV_HIP(hipStreamSynchronize(st));
        } catch (...) {
            d_segs.release();
            d_pos.release();
            d_ids.release();
            d_rows.release();
            throw;
        }
        d_segs.release();
        d_pos.release();
        d_ids.release();
        d_rows.release();
    });
}

pcv_status pcv_searcher_set_kernel(pcv_searcher* s, int kernel) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "set_kernel: searcher is NULL");
        PCV_REQUIRE(kernel >= PCV_KERNEL_AUTO && kernel <= PCV_KERNEL_MFMA, "set_kernel: unknown kernel %d", kernel);
        std::lock_guard<std::mutex> lk(s->mu);
        s->kernel = kernel;
    });
}

pcv_status pcv_searcher_set_screening_copy(pcv_searcher* s, int mode) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "searcher_set_screening_copy: searcher is NULL");
        PCV_REQUIRE(mode >= PCV_SCREEN_COPY_OFF && mode <= PCV_SCREEN_COPY_INT8, "searcher_set_screening_copy: unknown mode %d", mode);
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_HIP(hipSetDevice(s->ctx->device));
        if (mode == PCV_SCREEN_COPY_OFF) {
            PCV_HIP(hipStreamSynchronize(s->ctx->stream));
            drop_screening_copies(s);
        }
        s->screen_copy = mode;
        s->screen_copy_gave_way = false;
    });
}

pcv_status pcv_searcher_set_mid_copy(pcv_searcher* s, int mode) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "searcher_set_mid_copy: searcher is NULL");
        PCV_REQUIRE(mode >= PCV_MID_COPY_OFF && mode <= PCV_MID_COPY_ON, "searcher_set_mid_copy: unknown mode %d", mode);
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "searcher_set_mid_copy: a queued pass has not been collected");
        PCV_HIP(hipSetDevice(s->ctx->device));
        if (mode == PCV_MID_COPY_OFF) {
            PCV_HIP(hipStreamSynchronize(s->ctx->stream));
            drop_mid_copies(s);
        }
        if (mode == PCV_MID_COPY_ON && !s->mids_present) s->dirty = true;  // built by the next finalize
        s->mid_copy = mode;
        s->mid_gave_way = false;
        s->mid_hot_passes = 0;
    });
}

pcv_status pcv_searcher_allow_wide_sharded_pass(pcv_searcher* s, int on) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "allow_wide_sharded_pass: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "allow_wide_sharded_pass: a queued pass has not been collected");
        s->wide_sharded = on != 0;
    });
}

pcv_status pcv_searcher_wait_background(pcv_searcher* s) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "wait_background: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_HIP(hipSetDevice(s->ctx->device));
        settle_mid_build(s, true);
    });
}

pcv_status pcv_searcher_set_candidate_capacity(pcv_searcher* s, uint32_t n_candidates) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "set_candidate_capacity: searcher is NULL");
        PCV_REQUIRE(n_candidates >= 16 && n_candidates <= (1u << 24), "set_candidate_capacity: %u outside [16, 2^24]", n_candidates);
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "set_candidate_capacity: a queued pass has not been collected");
        PCV_HIP(hipSetDevice(s->ctx->device));
        PCV_HIP(hipStreamSynchronize(s->ctx->stream));
        s->cand_cap = n_candidates;  // the lists are (re)sized to it by the next pass; a pass that needs more grows them
        s->d_cand.release();
        s->d_cand_s.release();
    });
}

pcv_status pcv_searcher_set_tuning(pcv_searcher* s, uint32_t flags) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "set_tuning: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "set_tuning: a queued pass has not been collected");
        s->fail_copy_alloc = (flags & (uint32_t)PCV_TUNE_FAIL_COPY_ALLOC) != 0;
        s->scan_flags = flags & 0x3fffffffu;
    });
}

pcv_status pcv_searcher_set_shard_offset(pcv_searcher* s, int64_t first_global_pos) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "set_shard_offset: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        s->shard_offset = first_global_pos;
        assign_positions(s);
    });
}

pcv_status pcv_searcher_search(pcv_searcher* s, const float* queries, int n_queries, const int64_t* source_ids,
                               int n_sources, int k, int64_t* out_ids, float* out_scores, int* out_counts) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "search: searcher is NULL");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_REQUIRE(!s->pending.active, "search: a pass queued by search_device_begin has not been collected");
        std::vector<pcv_hit_dev> hits;
        search_hits(s, queries, n_queries, source_ids, n_sources, k, hits);
        hits_to_outputs(s->metric, s->D, hits.data(), n_queries, k, out_ids, out_scores, out_counts);
    });
}

pcv_status pcv_searcher_search_device(pcv_searcher* s, const float* queries, int n_queries,
                                      const int64_t* source_ids, int n_sources, int k, void* d_out, int async) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && d_out != nullptr, "search_device: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        check_search_args(s, queries, n_queries, k, "search_device");
        PCV_REQUIRE(!s->pending.active, "search_device: a pass queued by search_device_begin has not been collected");
        PCV_HIP(hipSetDevice(s->ctx->device));
        std::vector<SelSeg> segs = select_segments(s, source_ids, n_sources);
        const int kernel = pick_kernel(s, n_queries);
        const int qstep = pass_queries(s, kernel);
        s->stats = pcv_scan_stats{};
        s->stats.kernel_used = kernel;
        pcv_hit_dev* out = (pcv_hit_dev*)d_out;
        if (segs.empty()) {
            const pcv_hit_dev none{NAN, -1, -1};
            std::vector<pcv_hit_dev> hits((size_t)n_queries * k, none);
            PCV_HIP(hipMemcpyAsync(out, hits.data(), hits.size() * sizeof(pcv_hit_dev), hipMemcpyHostToDevice, s->ctx->stream));
            PCV_HIP(hipStreamSynchronize(s->ctx->stream));  // `hits` dies with this scope
            return;
        }
        maybe_build_mid_copies(s);
        // results stay on the device: every pass writes its slice of the caller's list
        for (int q0 = 0; q0 < n_queries; q0 += qstep) {
            const int B = std::min(qstep, n_queries - q0);
            run_pass(s, queries + (size_t)q0 * s->D, B, segs.data(), (int)segs.size(), k, kernel, out + (size_t)q0 * k, false);
        }
        (void)async;  // every pass has been collected: nothing is left in flight
    });
}

pcv_status pcv_searcher_search_device_begin(pcv_searcher* s, const float* queries, int n_queries,
                                            const int64_t* source_ids, int n_sources, int k, void* d_out) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && d_out != nullptr, "search_device_begin: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        device_begin(s, queries, n_queries, source_ids, n_sources, k, (pcv_hit_dev*)d_out);
    });
}

pcv_status pcv_searcher_search_device_begin_dq(pcv_searcher* s, const void* d_queries, int n_queries,
                                               const int64_t* source_ids, int n_sources, int k, void* d_out) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && d_out != nullptr, "search_device_begin_dq: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        device_begin(s, (const float*)d_queries, n_queries, source_ids, n_sources, k, (pcv_hit_dev*)d_out, true);
    });
}

pcv_status pcv_searcher_search_device_end(pcv_searcher* s, int* out_overflowed) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "search_device_end: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        PCV_HIP(hipSetDevice(s->ctx->device));
        const bool over = finish_pass(s);
        if (out_overflowed) *out_overflowed = over ? 1 : 0;
    });
}

pcv_status pcv_searcher_repeat_without_guess(pcv_searcher* s) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr, "repeat_without_guess: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        s->spec_hold = true;  // cleared by the next pass that completes (finish_pass)
    });
}

namespace {
// merge scratch of a context: device output and its pinned host copy for `n` records
void ensure_merge_scratch(pcv_ctx* ctx, size_t n) {
    if (ctx->merge_cap >= n) return;
    PCV_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->merge_dev) (void)hipFree(ctx->merge_dev);
    if (ctx->merge_pin) (void)hipHostFree(ctx->merge_pin);
    ctx->merge_dev = ctx->merge_pin = nullptr;
    ctx->merge_cap = 0;
    PCV_HIP(hipMalloc(&ctx->merge_dev, n * sizeof(pcv_hit_dev)));
    PCV_HIP(hipHostMalloc(&ctx->merge_pin, n * sizeof(pcv_hit_dev), hipHostMallocDefault));
    ctx->merge_cap = n;
}

void merge_lists(pcv_ctx* ctx, int metric, int dim, const void* d_lists, int n_shards, int n_queries, int k,
                 int64_t* out_ids, float* out_scores, int* out_counts, int* out_any_overflow) {
    PCV_REQUIRE(ctx != nullptr && d_lists != nullptr, "merge_topk: NULL argument");
    PCV_REQUIRE(n_shards > 0 && n_queries > 0 && k > 0 && k <= kMaxK, "merge_topk: bad shape");
    PCV_HIP(hipSetDevice(ctx->device));
    const int flagged = out_any_overflow != nullptr;
    const size_t n = (size_t)n_queries * k;
    ensure_merge_scratch(ctx, n + 1);
    pcv_hit_dev* d_out = (pcv_hit_dev*)ctx->merge_dev;
    pcv_hit_dev* h_out = (pcv_hit_dev*)ctx->merge_pin;
    launch_merge(ctx->stream, (const pcv_hit_dev*)d_lists, n_shards, n_queries, k, d_out, flagged);
    PCV_HIP(hipMemcpyAsync(h_out, d_out, (n + flagged) * sizeof(pcv_hit_dev), hipMemcpyDeviceToHost, ctx->stream));
    PCV_HIP(hipStreamSynchronize(ctx->stream));
    PCV_HIP(hipGetLastError());
    if (flagged) *out_any_overflow = h_out[n].pos != 0;
    hits_to_outputs(metric, dim, h_out, n_queries, k, out_ids, out_scores, out_counts);
}
}  // namespace

pcv_status pcv_merge_topk(pcv_ctx* ctx, int metric, int dim, const void* d_lists, int n_shards, int n_queries,
                          int k, int64_t* out_ids, float* out_scores, int* out_counts) {
    return guarded([&] {
        merge_lists(ctx, metric, dim, d_lists, n_shards, n_queries, k, out_ids, out_scores, out_counts, nullptr);
    });
}

pcv_status pcv_merge_topk_flagged(pcv_ctx* ctx, int metric, int dim, const void* d_lists, int n_shards,
                                  int n_queries, int k, int64_t* out_ids, float* out_scores, int* out_counts,
                                  int* out_any_overflow) {
    return guarded([&] {
        PCV_REQUIRE(out_any_overflow != nullptr, "merge_topk_flagged: NULL argument");
        merge_lists(ctx, metric, dim, d_lists, n_shards, n_queries, k, out_ids, out_scores, out_counts,
                    out_any_overflow);
    });
}

pcv_status pcv_merge_topk_host(int metric, int dim, const pcv_hit* lists, int n_shards, int n_queries, int k,
                               int64_t* out_ids, float* out_scores, int* out_counts) {
    return guarded([&] {
        PCV_REQUIRE(lists != nullptr, "merge_topk_host: lists is NULL");
        PCV_REQUIRE(n_shards > 0 && n_queries > 0 && k > 0 && k <= kMaxK, "merge_topk_host: bad shape");
        static_assert(sizeof(pcv_hit) == sizeof(pcv_hit_dev), "hit layout");
        const pcv_hit_dev* L = reinterpret_cast<const pcv_hit_dev*>(lists);
        const pcv_hit_dev none{NAN, -1, -1};
        std::vector<pcv_hit_dev> out((size_t)n_queries * k, none), m;
        for (int q = 0; q < n_queries; ++q) {
            m.clear();
            for (int r = 0; r < n_shards; ++r)
                for (int j = 0; j < k; ++j) {
                    const pcv_hit_dev& e = L[((size_t)r * n_queries + q) * k + j];
                    if (e.pos >= 0 && e.score == e.score) m.push_back(e);
                }
            std::sort(m.begin(), m.end(), hit_better);  // search.rs:179, canonical order
            for (int j = 0; j < k && j < (int)m.size(); ++j) out[(size_t)q * k + j] = m[j];  // search.rs:180
        }
        hits_to_outputs(metric, dim, out.data(), n_queries, k, out_ids, out_scores, out_counts);
    });
}

// ---- native RCCL exchange ------------------------------------------------------------------------
extern "C++" {
struct NcclId {  // ncclUniqueId of rccl.h: 128 opaque bytes, passed by value
    char internal[128];
};
namespace {
// the handful of RCCL entry points used, resolved from librccl.so.1 at first use (rccl.h signatures)
struct Rccl {
    int (*GetUniqueId)(void* id) = nullptr;
    int (*CommInitRank)(void** comm, int nranks, NcclId id, int rank) = nullptr;
    int (*AllGather)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t st) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;  // first use may come from two contexts' threads at once
    std::call_once(once, [] {
        // RCCL must sit on the HIP runtime this library is bound to.  A process may hold a second
        // runtime + RCCL pair (PyTorch bundles its own, with the same sonames), so RCCL is looked up by
        // path next to our libamdhip64 first and by soname only after that.
        void* h = nullptr;
        Dl_info info;
        if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                for (const char* name : {"librccl.so.1", "librccl.so"})
                    if (!h) h = dlopen((dir + name).c_str(), RTLD_NOW | RTLD_LOCAL);
            }
        }
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) {
            r.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
            r.CommInitRank = (int (*)(void**, int, NcclId, int))dlsym(h, "ncclCommInitRank");
            r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
            r.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
            r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
            r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
        }
    });
    return r;
}
void rccl_check(int rc, const char* what) {
    if (rc != 0) {
        Rccl& r = rccl();
        PCV_FAIL(PCV_ERR_DEVICE, "%s failed: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
    }
}
}  // namespace

struct pcv_comm {
    pcv_ctx* ctx = nullptr;
    void* comm = nullptr;
    int world = 1, rank = 0;
    DevBuf<pcv_hit_dev> d_local, d_gathered, d_merged;
    pcv_hit_dev* pin_hits = nullptr;
    size_t pin_cap = 0;
};
}  // extern "C++"

pcv_status pcv_comm_unique_id(uint8_t out_id[128]) {
    return guarded([&] {
        PCV_REQUIRE(out_id != nullptr, "comm_unique_id: NULL argument");
        if (!rccl().ok) PCV_FAIL(PCV_ERR_UNSUPPORTED, "RCCL (librccl.so.1) is not available: %s", dlerror() ? dlerror() : "");
        NcclId id;
        rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
        std::memcpy(out_id, id.internal, 128);
    });
}

pcv_status pcv_comm_create(pcv_ctx* ctx, int world_size, int rank, const uint8_t id_bytes[128], pcv_comm** out) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && id_bytes != nullptr && out != nullptr, "comm_create: NULL argument");
        *out = nullptr;
        PCV_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, "comm_create: rank %d / world %d", rank, world_size);
        if (!rccl().ok) PCV_FAIL(PCV_ERR_UNSUPPORTED, "RCCL (librccl.so.1) is not available");
        PCV_HIP(hipSetDevice(ctx->device));
        auto c = std::make_unique<pcv_comm>();
        c->ctx = ctx;
        c->world = world_size;
        c->rank = rank;
        NcclId id;
        std::memcpy(id.internal, id_bytes, 128);
        rccl_check(rccl().CommInitRank(&c->comm, world_size, id, rank), "ncclCommInitRank");
        *out = c.release();
    });
}

pcv_status pcv_comm_destroy(pcv_comm* c) {
    return guarded([&] {
        if (!c) return;
        (void)hipSetDevice(c->ctx->device);
        (void)hipStreamSynchronize(c->ctx->stream);
        if (c->comm) rccl().CommDestroy(c->comm);
        c->d_local.release();
        c->d_gathered.release();
        c->d_merged.release();
        if (c->pin_hits) (void)hipHostFree(c->pin_hits);
        delete c;
    });
}

pcv_status pcv_comm_all_gather(pcv_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank) {
    return guarded([&] {
        PCV_REQUIRE(c != nullptr && d_send != nullptr && d_recv != nullptr && bytes_per_rank > 0, "comm_all_gather: bad argument");
        PCV_HIP(hipSetDevice(c->ctx->device));
        rccl_check(rccl().AllGather(d_send, d_recv, bytes_per_rank, /*ncclInt8*/ 0, c->comm, c->ctx->stream), "ncclAllGather");
    });
}

static pcv_status search_sharded_impl(pcv_searcher* s, pcv_comm* c, const float* queries, bool queries_on_device, int n_queries,
                                      const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                                      float* out_scores, int* out_counts) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && c != nullptr, "search_sharded: NULL argument");
        PCV_REQUIRE(s->ctx == c->ctx, "search_sharded: searcher and communicator live on different contexts");
        // the whole step — local pass, exchange, merge, collection — is one critical section of the searcher:
        // it shares the pass workspace, the pinned blocks and the stream with plain searches
        std::lock_guard<std::mutex> lk(s->mu);
        check_search_args(s, queries, n_queries, k, "search_sharded");
        PCV_REQUIRE(!s->pending.active, "search_sharded: a pass queued by search_device_begin has not been collected");
        const size_t n = (size_t)n_queries * k;
        hipStream_t st = s->ctx->stream;
        PCV_HIP(hipSetDevice(s->ctx->device));
        c->d_local.ensure(n + 1);
        c->d_gathered.ensure((n + 1) * c->world);
        c->d_merged.ensure(n + 1);
        if (c->pin_cap < n + 1) {
            if (c->pin_hits) (void)hipHostFree(c->pin_hits);
            c->pin_hits = nullptr;
            c->pin_cap = 0;
            PCV_HIP(hipHostMalloc((void**)&c->pin_hits, (n + 1) * sizeof(pcv_hit_dev), hipHostMallocDefault));
            c->pin_cap = n + 1;
        }
        auto exchange = [&](const pcv_hit_dev* local, size_t nq, int flagged) {  // all-gather + merge + download, queued behind the local pass
            const size_t rec = nq * k + flagged;
            rccl_check(rccl().AllGather(local, c->d_gathered.p, rec * sizeof(pcv_hit_dev), /*ncclInt8*/ 0, c->comm, st),
                       "ncclAllGather");
            launch_merge(st, c->d_gathered.p, c->world, (int)nq, k, c->d_merged.p, flagged);
            PCV_HIP(hipMemcpyAsync(c->pin_hits, c->d_merged.p, rec * sizeof(pcv_hit_dev), hipMemcpyDeviceToHost, st));
        };
        const int qstep = pass_queries(s, pick_kernel(s, n_queries), true);  // every rank computes the same split
        std::vector<pcv_hit_dev> all(n);
        pcv_scan_stats total{};
        auto accumulate = [&] {  // every attempt starts its own statistics (device_begin)
            total.rows_scanned += s->stats.rows_scanned;
            total.bytes_algorithmic += s->stats.bytes_algorithmic;
            total.scan_ms += s->stats.scan_ms;
            total.total_ms += s->stats.total_ms;
            total.candidates += s->stats.candidates;
            total.scan_launches += s->stats.scan_launches;
            total.overflow_reruns += s->stats.overflow_reruns;
            total.speculation_reruns += s->stats.speculation_reruns;
            total.host_enqueue_ms += s->stats.host_enqueue_ms;
            total.host_wait_ms += s->stats.host_wait_ms;
            total.kernel_used = s->stats.kernel_used;
            total.bytes_streamed += s->stats.bytes_streamed;
            total.coarse_survivors += s->stats.coarse_survivors;
            total.mid_survivors += s->stats.mid_survivors;
            total.mid_copy = s->stats.mid_copy;
            total.screening_copy = s->stats.screening_copy;
        };
        for (int q0 = 0; q0 < n_queries; q0 += qstep) {
            const int B = std::min(qstep, n_queries - q0);
            const size_t nb = (size_t)B * k;
            // One pass: everything is queued back to back and the host waits once.  Whether a list
            // overflowed somewhere is part of the exchanged payload, so all ranks repeat (or not) together.
            for (int attempt = 0;; ++attempt) {
                device_begin(s, queries + (size_t)q0 * s->D, B, source_ids, n_sources, k, c->d_local.p, queries_on_device);
                try {
                    exchange(c->d_local.p, (size_t)B, 1);
                } catch (...) {
                    try {
                        finish_pass(s);  // collect the queued pass before reporting
                    } catch (...) {
                    }
                    throw;
                }
                finish_pass(s);  // waits for the stream (pass, all-gather, merge, download); grows this rank's lists if needed
                const bool again = c->pin_hits[nb].pos != 0;
                if (!again) std::memcpy(all.data() + (size_t)q0 * k, c->pin_hits, nb * sizeof(pcv_hit_dev));
                accumulate();
                if (!again) break;
                // Some rank's pass was incomplete (a list overflowed, or a speculative threshold did not hold — one flag
                // covers both): the repeat runs without a guess on EVERY rank, so that guesses failing on different ranks
                // in different attempts cannot use up the limit; what is left to repeat for are overflows, and each of
                // those grows the lists of the rank it happened on.
                s->spec_hold = true;
                PCV_REQUIRE(attempt < 7,
                            "search_sharded: the step is still incomplete on some rank after %d repeats (on this rank: %d candidate-list "
                            "overflows, %d speculative thresholds that did not hold)",
                            attempt + 1, total.overflow_reruns, total.speculation_reruns);
            }
        }
        s->stats = total;
        hits_to_outputs(s->metric, s->D, all.data(), n_queries, k, out_ids, out_scores, out_counts);
    });
}

pcv_status pcv_searcher_search_sharded(pcv_searcher* s, pcv_comm* c, const float* queries, int n_queries,
                                       const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                                       float* out_scores, int* out_counts) {
    return search_sharded_impl(s, c, queries, false, n_queries, source_ids, n_sources, k, out_ids, out_scores, out_counts);
}

pcv_status pcv_searcher_search_sharded_dq(pcv_searcher* s, pcv_comm* c, const void* d_queries, int n_queries,
                                          const int64_t* source_ids, int n_sources, int k, int64_t* out_ids,
                                          float* out_scores, int* out_counts) {
    return search_sharded_impl(s, c, (const float*)d_queries, true, n_queries, source_ids, n_sources, k, out_ids, out_scores, out_counts);
}

static pcv_status similarity(pcv_ctx* ctx, const float* a, int B, const float* m, int64_t N, int dim, float* out,
                             int cosine) {
    return guarded([&] {
        PCV_REQUIRE(ctx != nullptr && a != nullptr && m != nullptr && out != nullptr, "similarity: NULL argument");
        PCV_REQUIRE(B > 0 && N > 0 && dim > 0, "similarity: empty input");
        PCV_HIP(hipSetDevice(ctx->device));
        DevBuf<float> d_a, d_m, d_o;
        try {
            d_a.ensure((size_t)B * dim);
            d_m.ensure((size_t)N * dim);
            d_o.ensure((size_t)B * N);
            PCV_HIP(hipMemcpyAsync(d_a.p, a, (size_t)B * dim * 4, hipMemcpyHostToDevice, ctx->stream));
            PCV_HIP(hipMemcpyAsync(d_m.p, m, (size_t)N * dim * 4, hipMemcpyHostToDevice, ctx->stream));
            launch_similarity_matrix(ctx->stream, d_a.p, B, d_m.p, N, dim, cosine, d_o.p);
            PCV_HIP(hipMemcpyAsync(out, d_o.p, (size_t)B * N * 4, hipMemcpyDeviceToHost, ctx->stream));
            PCV_HIP(hipStreamSynchronize(ctx->stream));
            PCV_HIP(hipGetLastError());
        } catch (...) {
            d_a.release();
            d_m.release();
            d_o.release();
            throw;
        }
        d_a.release();
        d_m.release();
        d_o.release();
    });
}

pcv_status pcv_dot_product(pcv_ctx* ctx, const float* a, int B, const float* m, int64_t N, int dim, float* out) {
    return similarity(ctx, a, B, m, N, dim, out, 0);
}
pcv_status pcv_cosine_similarity(pcv_ctx* ctx, const float* a, int B, const float* m, int64_t N, int dim,
                                 float* out) {
    return similarity(ctx, a, B, m, N, dim, out, 1);
}

pcv_status pcv_searcher_last_stats(pcv_searcher* s, pcv_scan_stats* out) {
    return guarded([&] {
        PCV_REQUIRE(s != nullptr && out != nullptr, "last_stats: NULL argument");
        std::lock_guard<std::mutex> lk(s->mu);
        *out = s->stats;
    });
}

}  // extern "C"
