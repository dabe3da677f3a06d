// Host side of the C ABI declared in include/sebm25.h: argument checking, device memory,
// the build / search pipelines as sequences of kernel launches, and result copies.
#include <algorithm>
#include <thread>
#include <unordered_set>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.hpp"

using namespace sebm25;

namespace {
thread_local std::string g_thread_err;

constexpr uint32_t MAX_K = 2048;
constexpr uint32_t MAX_QUERY_TERMS = 0xFFFFFFFFu;  // sebm25_search: no limit of its own (a query holds at most n_terms distinct terms)
constexpr uint32_t RED_MAX_CHUNKS = 16384;  // wavefronts
}  // namespace

struct H2DJob {
    void* dst;
    const void* src;
    size_t bytes;
};

struct SearchPlanVecs {
    std::vector<uint32_t> flat, q_toff, q_nt, q_order, q_ns, first, nlists, bucket;
    std::vector<uint64_t> q_len, item_len;
    std::vector<QItem> items, unsorted;
};
struct NaHost;

struct sebm25_index {
    int device = 0;
    hipStream_t stream = nullptr;
    std::vector<H2DJob> h2d;  // host arrays waiting to be copied in (stage_in / flush_h2d)
    std::string err;
    std::mutex mu;
    bool built = false, finalized = false;

    uint64_t R = 0;
    uint32_t V = 0, n_docs = 0, n_articles = 0;
    RowFmt fmt{1, 1, 59};
    uint32_t n_pass = 0;
    int sort_bits = SORT_MAX_BITS;  // widest radix digit (tuning knob: SEBM25_SORT_BITS = 4..10)
    bool big_weights = false;       // some posting weight needs more than 16 bits: queries take the general kernel
    uint64_t P = 0, P_web = 0, WH = 0;
    uint32_t web_base = 0, news_base = 0;
    sebm25_doc_totals local_totals{}, used_totals{};
    uint64_t N = 1;
    double avgdl = 0.0;

    DevBuf sortA, sortB, chunk_hist, group_off, bin_start, scan_tmp;  // bin_start: one array per pass parity
    DevBuf row_ptr, news_ptr, df_local, df_global, term_rows, idf, idf_used, doc_len;
    DevBuf scalars;  // u64[8]: 0 kept postings, 1 web heads, 2 n_done, 3 n_wc, 4 sum_wc, 5 kept web postings; u32 err at [7]
    DevBuf st_term, st_unit, st_tf, st_meta, st_wc, st_flags, st_misc;
    DevBuf url_rank_web, url_rank_news;
    bool has_rank_web = false, has_rank_news = false, has_df_global = false;
    uint32_t rank_base_web = 0, rank_shift_web = 0, rank_base_news = 0, rank_shift_news = 0;  // units whose key needs no array
    Slot* post = nullptr;  // one slot per sorted row: unit, weight in tenths (or POST_HOLE)
    const uint64_t* sorted_rows = nullptr;  // the sorted packed rows this segment was built from (read again by updates)
    // A build whose last sort pass wrote the posting slots has no sorted rows: the input of that pass (the
    // rows ordered by the earlier digits) is kept with the pass's offsets, and the rows are sorted out of it
    // when an update, a compaction or the exact second look at unusual runs first asks for them.
    const uint64_t* unsorted_rows = nullptr;
    int last_shift = 0, last_bits = 0;
    bool keep_rows = false;   // this segment's sorted rows are wanted right away (update segments)
    bool fused = true;        // SEBM25_FUSED_POSTINGS=0: always the separate posting pass
    bool built_fused = false; // the slots this segment holds were written by the last sort pass
    uint64_t max_rows = MAX_ROWS;  // rows one shard holds (SEBM25_TEST_MAX_ROWS lowers it: tests of what a rank past it does)
    const uint32_t* wc_dev = nullptr;       // device views of the last build's word_count / doc_flags arrays
    const uint8_t* flags_dev = nullptr;

    // ---- incremental updates.  The handle itself is the base segment; every sebm25_update adds one more
    // (a handle of its own that only the base refers to).  A unit's live rows are in exactly one segment.
    std::vector<sebm25_index*> deltas;
    std::vector<sebm25_index*> spares;   // segment objects of dropped update segments, kept with their device buffers for the next update
    DevBuf dead_web, dead_news;  // bitmaps over this segment's unit spaces: rows of these units are dead here
    bool has_dead = false;
    uint64_t dead_rows = 0;
    // sizes and sums over all segments (equal to V / n_docs / n_articles / P / P_web while there are no deltas)
    uint32_t V_cur = 0, n_docs_cur = 0, n_articles_cur = 0;
    DevBuf df_sum, term_rows_sum, chg_web, chg_news, chg_coarse, upd_ids, upd_counts;

    std::vector<uint32_t> h_row_ptr, h_news_ptr;
    std::vector<uint32_t> h_list_len;   // [2 V]: web and news list length of every term side by side (what the planner reads)
    bool host_ptrs_valid = false;
    SearchPlanVecs plan_vecs;    // host plan storage of sebm25_search, kept between calls
    NaHost* na_host = nullptr;   // host plan storage of sebm25_search_noalloc (pinned, fixed size)

    // search scratch
    DevBuf q_items, q_terms, q_first, q_n, q_wide;
    DevBuf o_unit, o_score, o_mat, o_tie, o_count, t_unit, t_score, t_mat, t_tie, t_count;
    DevBuf o_pack;                 // small host-bound results: all five arrays in one buffer, one copy back
    void* o_host = nullptr;        // its pinned host twin (grow-only)
    size_t o_host_cap = 0;
    DevBuf m_in[5];
    DevBuf gen_rows, gen_cdf;

    PhaseTimer timer;
};

extern "C" void sebm25_destroy(sebm25_index* ix);

namespace {

int fail(sebm25_index* ix, int code, const std::string& msg) {
    if (ix) ix->err = msg;
    g_thread_err = msg;
    return code;
}

#define HIP_TRY(ix, expr)                                                                             \
    do {                                                                                              \
        hipError_t e__ = (expr);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            (void)hipGetLastError();                                                                  \
            return fail(ix, e__ == hipErrorOutOfMemory ? SEBM25_E_NOMEM : SEBM25_E_HIP,               \
                        std::string(#expr) + ": " + hipGetErrorString(e__));                          \
        }                                                                                             \
    } while (0)

}  // namespace

// ---- the entry point that allocates nothing ------------------------------------------------
// Everything sebm25_search keeps in grow-on-demand device buffers (work items, deduped terms, the
// temporary result lists of split queries) lives here in one file-scope device array of fixed size.
// Each GPU holds its own copy at its own address: sebm25_create looks the address up once per
// device (hipGetSymbolAddress with that device current) and host code only ever uses that address.
namespace {
constexpr uint32_t PLAN_SUB = 16, PLAN_PER_GROUP = 65 * PLAN_SUB;  // counting-sort buckets of the query planner
constexpr uint32_t NA_MAX_QUERIES = 1024, NA_MAX_TERMS = 4096, NA_MAX_ITEMS = 4096, NA_MAX_K = 256;
constexpr uint32_t NA_TMP_ENTRIES = 131072;  // temporary lists x k
constexpr size_t na_align(size_t x) { return (x + 255) / 256 * 256; }
constexpr size_t NA_OFF_ITEMS = 0;
constexpr size_t NA_OFF_TERMS = NA_OFF_ITEMS + na_align((size_t)NA_MAX_ITEMS * sizeof(QItem));
constexpr size_t NA_OFF_FIRST = NA_OFF_TERMS + na_align((size_t)NA_MAX_TERMS * 4);
constexpr size_t NA_OFF_NLISTS = NA_OFF_FIRST + na_align((size_t)NA_MAX_QUERIES * 4);
constexpr size_t NA_OFF_T_UNIT = NA_OFF_NLISTS + na_align((size_t)NA_MAX_QUERIES * 4);
constexpr size_t NA_OFF_T_SCORE = NA_OFF_T_UNIT + na_align((size_t)NA_TMP_ENTRIES * 4);
constexpr size_t NA_OFF_T_MAT = NA_OFF_T_SCORE + na_align((size_t)NA_TMP_ENTRIES * 8);
constexpr size_t NA_OFF_T_TIE = NA_OFF_T_MAT + na_align((size_t)NA_TMP_ENTRIES * 4);
constexpr size_t NA_OFF_T_COUNT = NA_OFF_T_TIE + na_align((size_t)NA_TMP_ENTRIES * 4);
constexpr size_t NA_SCRATCH_BYTES = NA_OFF_T_COUNT + na_align((size_t)NA_MAX_ITEMS * 4);  // about 2.8 MB
constexpr int NA_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[NA_SCRATCH_BYTES];

namespace {
std::mutex g_na_mu;
unsigned char* g_na_ptr[NA_MAX_DEVICES];   // device address of `scratch` on each GPU (0: not looked up yet)
std::mutex g_na_device_mu[NA_MAX_DEVICES];  // calls on one GPU share its array: one at a time within this process

}  // namespace
// host-side plan storage of the no-alloc entry point, allocated once per handle by sebm25_create
struct NaHost {
    uint32_t flat[NA_MAX_TERMS];
    uint32_t q_toff[NA_MAX_QUERIES], q_nt[NA_MAX_QUERIES], q_order[NA_MAX_QUERIES], q_ns[NA_MAX_QUERIES], first[NA_MAX_QUERIES], nlists[NA_MAX_QUERIES];
    uint64_t q_len[NA_MAX_QUERIES];
    QItem items[NA_MAX_ITEMS], unsorted[NA_MAX_ITEMS];
    uint64_t item_len[NA_MAX_ITEMS];
    uint32_t bucket[(SCORE_FAST_T + 1) * PLAN_PER_GROUP + 1];
};
namespace {

// once per device, with that device current
int na_setup_device(sebm25_index* ix, int device) {
    if (device < 0 || device >= NA_MAX_DEVICES) return SEBM25_OK;  // no-alloc entry unavailable there; reported at call time
    std::lock_guard<std::mutex> lk(g_na_mu);
    if (g_na_ptr[device]) return SEBM25_OK;
    void* p = nullptr;
    HIP_TRY(ix, hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    g_na_ptr[device] = static_cast<unsigned char*>(p);
    return SEBM25_OK;
}
}  // namespace


namespace {

// No exception crosses the C boundary: allocation failures of the host containers become SEBM25_E_NOMEM.
int fail_quiet(sebm25_index* ix, int code, const char* msg) noexcept {
    try {
        return fail(ix, code, msg);
    } catch (...) {
        return code;
    }
}
#define ABI_TRY try {
#define ABI_CATCH(ix)                                                                 \
    }                                                                                 \
    catch (const std::bad_alloc&) { return fail_quiet(ix, SEBM25_E_NOMEM, "out of host memory"); } \
    catch (const std::exception& e) { return fail_quiet(ix, SEBM25_E_INTERNAL, e.what()); }        \
    catch (...) { return fail_quiet(ix, SEBM25_E_INTERNAL, "unknown exception"); }

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

int bits_for(uint64_t n_values) {  // bits needed to hold 0 .. n_values - 1
    int b = 1;
    while (b < 64 && (1ull << b) < n_values) ++b;
    return b;
}

// device view of a caller array: in place when it already lives on the GPU, else staged
template <class T>
int stage_in(sebm25_index* ix, const T* src, size_t n, DevBuf& buf, const T** out) {
    if (n == 0 || !src) {
        *out = nullptr;
        return SEBM25_OK;
    }
    // kernels read rows with 16-byte vector loads: a device array that is not 16-byte aligned
    // (a view into a larger allocation) is copied once into aligned staging memory
    if (is_device_ptr(src) && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        *out = src;
        return SEBM25_OK;
    }
    HIP_TRY(ix, buf.ensure(n * sizeof(T)));
    if (is_device_ptr(src)) {
        HIP_TRY(ix, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyDeviceToDevice, ix->stream));
    } else {
        ix->h2d.push_back({buf.p, src, n * sizeof(T)});  // host memory: copied by flush_h2d
    }
    *out = buf.as<T>();
    return SEBM25_OK;
}

// Copies the queued host arrays to the device.  One copy from pageable memory runs at about
// 30 GB/s (the runtime stages it through pinned memory on the calling thread); large inputs are
// therefore cut into slices that several threads copy at once, each on a stream of its own.
int flush_h2d(sebm25_index* ix) {
    std::vector<H2DJob> jobs;
    jobs.swap(ix->h2d);
    size_t total = 0;
    for (const H2DJob& j : jobs) total += j.bytes;
    if (total == 0) return SEBM25_OK;
    constexpr size_t PAR_MIN = 64u << 20;
    constexpr int THREADS = 4;
    if (total < PAR_MIN) {
        for (const H2DJob& j : jobs) HIP_TRY(ix, hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyHostToDevice, ix->stream));
        return SEBM25_OK;
    }
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));  // the staging buffers may still be read by earlier work
    hipError_t errs[THREADS];
    std::thread th[THREADS];
    const int dev = ix->device;
    auto work = [&](int t) {
        hipError_t e = hipSetDevice(dev);
        hipStream_t s = nullptr;
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (const H2DJob& j : jobs) {
            if (e != hipSuccess) break;
            // slice t of every array, cut at 4 KB multiples
            const size_t per = ((j.bytes + THREADS - 1) / THREADS + 4095) / 4096 * 4096;
            const size_t off = std::min(j.bytes, per * (size_t)t), len = std::min(j.bytes - off, per);
            if (len)
                e = hipMemcpyAsync(static_cast<char*>(j.dst) + off, static_cast<const char*>(j.src) + off, len,
                                   hipMemcpyHostToDevice, s);
        }
        if (s) {
            const hipError_t e2 = hipStreamSynchronize(s);
            if (e == hipSuccess) e = e2;
            (void)hipStreamDestroy(s);
        }
        errs[t] = e;
    };
    for (int t = 0; t < THREADS; ++t) {
        errs[t] = hipSuccess;
        try {
            th[t] = std::thread(work, t);
        } catch (...) {  // no thread to be had: this slice is copied here, nothing crosses the C boundary
            work(t);
        }
    }
    for (int t = 0; t < THREADS; ++t)
        if (th[t].joinable()) th[t].join();
    for (int t = 0; t < THREADS; ++t) HIP_TRY(ix, errs[t]);
    return SEBM25_OK;
}

int copy_out(sebm25_index* ix, void* dst, const void* src_dev, size_t bytes) {
    if (!dst || bytes == 0) return SEBM25_OK;
    HIP_TRY(ix, hipMemcpyAsync(dst, src_dev, bytes, hipMemcpyDefault, ix->stream));
    return SEBM25_OK;
}

void phase_begin(sebm25_index* ix, const char* name) {
    PhaseTimer& t = ix->timer;
    if (!t.enabled) return;
    if (t.ev.size() < 2 * (t.used + 1)) {
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        t.ev.push_back(a);
        t.ev.push_back(b);
    }
    if (t.names.size() <= t.used) t.names.push_back(name); else t.names[t.used] = name;
    (void)hipEventRecord(t.ev[2 * t.used], ix->stream);
}
void phase_end(sebm25_index* ix) {
    PhaseTimer& t = ix->timer;
    if (!t.enabled) return;
    (void)hipEventRecord(t.ev[2 * t.used + 1], ix->stream);
    ++t.used;
}
void phases_reset(sebm25_index* ix) { ix->timer.used = 0; }
void phases_collect(sebm25_index* ix) {
    PhaseTimer& t = ix->timer;
    if (!t.enabled) return;
    t.ms.assign(t.used, 0.f);
    for (size_t i = 0; i < t.used; ++i) {
        (void)hipEventSynchronize(t.ev[2 * i + 1]);
        (void)hipEventElapsedTime(&t.ms[i], t.ev[2 * i], t.ev[2 * i + 1]);
    }
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int refresh_host_ptrs(sebm25_index* ix) {
    if (ix->host_ptrs_valid) return SEBM25_OK;
    ix->h_row_ptr.resize((size_t)ix->V + 1);
    ix->h_news_ptr.resize(ix->V);
    ix->h_list_len.resize((size_t)ix->V * 2);
    HIP_TRY(ix, hipMemcpyAsync(ix->h_row_ptr.data(), ix->row_ptr.p, ((size_t)ix->V + 1) * 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(ix, hipMemcpyAsync(ix->h_news_ptr.data(), ix->news_ptr.p, (size_t)ix->V * 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    for (size_t t = 0; t < ix->V; ++t) {
        ix->h_list_len[2 * t] = ix->h_news_ptr[t] - ix->h_row_ptr[t];
        ix->h_list_len[2 * t + 1] = ix->h_row_ptr[t + 1] - ix->h_news_ptr[t];
    }
    ix->host_ptrs_valid = true;
    return SEBM25_OK;
}

// Update segments that are dropped keep their (grow-only) device buffers for the next update: freeing and
// allocating the fifteen buffers of a segment costs more than building a small segment does.
constexpr size_t SPARE_SEGMENTS = 8, SPARE_MAX_BYTES = (size_t)256 << 20;
void drop_deltas(sebm25_index* ix) {
    for (sebm25_index* d : ix->deltas) {
        if (ix->spares.size() < SPARE_SEGMENTS && d->sortA.cap <= SPARE_MAX_BYTES && d->deltas.empty()) {
            d->built = d->finalized = false;
            d->has_dead = false;
            d->dead_rows = 0;
            d->sorted_rows = nullptr;
            d->host_ptrs_valid = false;
            ix->spares.push_back(d);
        } else {
            sebm25_destroy(d);
        }
    }
    ix->deltas.clear();
}
sebm25_index* take_segment(sebm25_index* ix) {
    sebm25_index* seg = nullptr;
    if (!ix->spares.empty()) {
        seg = ix->spares.back();
        ix->spares.pop_back();
    } else {
        seg = new (std::nothrow) sebm25_index();
    }
    if (seg) {
        seg->device = ix->device;
        seg->stream = ix->stream;
        seg->sort_bits = ix->sort_bits;
        seg->fused = ix->fused;
        seg->keep_rows = true;  // read again at once: rows of unlisted units, the fold with the next update's rows
    }
    return seg;
}

int do_finalize(sebm25_index* ix, const uint32_t* df_global, const sebm25_doc_totals* tg) {
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "finalize: no index built on this handle");
    const uint32_t* df_dev = ix->deltas.empty() ? ix->df_local.as<uint32_t>() : ix->df_sum.as<uint32_t>();
    ix->has_df_global = false;
    if (df_global) {
        HIP_TRY(ix, ix->df_global.ensure((size_t)ix->V_cur * 4));
        HIP_TRY(ix, hipMemcpyAsync(ix->df_global.p, df_global, (size_t)ix->V_cur * 4, hipMemcpyDefault, ix->stream));
        df_dev = ix->df_global.as<uint32_t>();
        ix->has_df_global = true;
    }
    ix->used_totals = tg ? *tg : ix->local_totals;
    ix->N = ix->used_totals.n_done ? ix->used_totals.n_done : 1;  // `or 1`
    ix->avgdl = ix->used_totals.n_wc ? (double)ix->used_totals.sum_wc / (double)ix->used_totals.n_wc : 0.0;  // `or 0.0`
    phase_begin(ix, "idf");
    launch_idf(df_dev, ix->V_cur, ix->N, nullptr, ix->idf.as<double>(), ix->idf_used.as<double>(), ix->stream);
    phase_end(ix);
    HIP_TRY(ix, hipGetLastError());
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    ix->finalized = true;
    return SEBM25_OK;
}

// The separate posting pass over sorted rows: slots into `slots` (the sort buffer that does not hold the rows), list
// bounds, df and row counts.  Counters in ix->scalars [0], [1], [5] and the per-term arrays are (re)initialised here.
int post_sorted(sebm25_index* ix, const uint64_t* sorted, uint64_t* slots) {
    const uint64_t R = ix->R;
    const uint32_t V = ix->V;
    uint64_t* scal = ix->scalars.as<uint64_t>();
    uint32_t* err_dev = reinterpret_cast<uint32_t*>(scal + 7);
    uint32_t* row_ptr = ix->row_ptr.as<uint32_t>();
    uint32_t* news_ptr = ix->news_ptr.as<uint32_t>();
    ix->post = reinterpret_cast<Slot*>(slots);
    ix->built_fused = false;
    // one contiguous range of rows per wavefront
    uint64_t chunk2 = (R + RED_MAX_CHUNKS - 1) / RED_MAX_CHUNKS;
    chunk2 = (chunk2 + RED_BATCH - 1) / RED_BATCH * RED_BATCH;
    const uint32_t n_chunks = (uint32_t)((R + chunk2 - 1) / chunk2);
    HIP_TRY(ix, hipMemsetAsync(news_ptr, 0xFF, (size_t)V * 4, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(row_ptr, 0xFF, ((size_t)V + 1) * 4, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(ix->df_local.p, 0, (size_t)V * 4, ix->stream));
    HIP_TRY(ix, ix->scan_tmp.ensure(std::max(row_ptr_fill_tmp_words(V), postings_tmp_words(n_chunks)) * 4));
    phase_begin(ix, "postings");
    launch_postings(sorted, R, chunk2, n_chunks, ix->fmt, V, ix->post, row_ptr, news_ptr,
                    ix->df_local.as<uint32_t>(), scal, ix->scan_tmp.as<uint32_t>(), err_dev, ix->stream);
    phase_end(ix);
    phase_begin(ix, "term_stats");
    launch_row_ptr_fill(row_ptr, V, R, ix->scan_tmp.as<uint32_t>(), ix->stream);
    launch_term_stats(V, row_ptr, news_ptr, ix->term_rows.as<uint32_t>(), ix->stream);
    phase_end(ix);
    HIP_TRY(ix, hipGetLastError());
    return SEBM25_OK;
}

// Sorted rows of a segment whose last sort pass wrote posting slots instead: that pass again, this time writing the
// rows, from its kept input and offsets into a buffer that then replaces the input's.
int ensure_sorted_rows(sebm25_index* ix, sebm25_index* seg) {
    if (seg->sorted_rows || seg->R == 0 || !seg->unsorted_rows) return SEBM25_OK;
    const size_t Rcap = std::max<uint64_t>(seg->R, 1);
    DevBuf out;
    HIP_TRY(ix, out.ensure((Rcap + POST_PAD) * 8));
    BuildCtx ctx{};
    ctx.fmt = seg->fmt;
    ctx.n_terms = seg->V; ctx.n_docs = seg->n_docs; ctx.n_articles = seg->n_articles;
    ctx.err = reinterpret_cast<uint32_t*>(seg->scalars.as<uint64_t>() + 7);
    const uint32_t G = (uint32_t)((seg->R + SORT_TILE - 1) / SORT_TILE);
    launch_sort_scatter(2, RawRows{}, seg->unsorted_rows, out.as<uint64_t>(), seg->R, G, seg->last_shift, seg->last_bits, ctx,
                        seg->chunk_hist.as<unsigned short>(), seg->group_off.as<uint32_t>(), ix->stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        out.release();
        HIP_TRY(ix, e);
    }
    DevBuf& in_buf = seg->unsorted_rows == seg->sortA.as<uint64_t>() ? seg->sortA : seg->sortB;
    in_buf.release();
    in_buf = out;
    seg->sorted_rows = in_buf.as<uint64_t>();
    seg->unsorted_rows = nullptr;
    return SEBM25_OK;
}

// Sorts the rows of a segment and writes its posting slots, list bounds, df and row counts.  The rows are the
// caller's raw arrays (start_packed == nullptr) or packed rows that already sit in sort buffer A (compaction).
// ix->R / V / fmt / n_docs / n_articles describe the segment; buffers are sized here.
// presorted: start_packed (sort buffer A) already holds the rows in their final order; only the postings are made
int sort_and_post(sebm25_index* ix, RawRows raw, const uint64_t* start_packed, bool unit_major, const uint8_t* flags_dev,
                  bool presorted = false) {
    const uint64_t R = ix->R;
    const uint32_t V = ix->V;
    const RowFmt fmt = ix->fmt;
    const size_t Rcap = std::max<uint64_t>(R, 1);
    // each sort buffer later holds the two posting slot arrays, padded by POST_PAD entries each
    HIP_TRY(ix, ix->sortA.ensure((Rcap + POST_PAD) * 8));
    HIP_TRY(ix, ix->sortB.ensure((Rcap + POST_PAD) * 8));
    HIP_TRY(ix, ix->row_ptr.ensure(((size_t)V + 1) * 4));
    HIP_TRY(ix, ix->news_ptr.ensure((size_t)V * 4));
    HIP_TRY(ix, ix->df_local.ensure((size_t)V * 4));
    HIP_TRY(ix, ix->term_rows.ensure((size_t)V * 4));
    uint64_t* scal = ix->scalars.as<uint64_t>();
    uint32_t* err_dev = reinterpret_cast<uint32_t*>(scal + 7);
    uint32_t* row_ptr = ix->row_ptr.as<uint32_t>();
    uint32_t* news_ptr = ix->news_ptr.as<uint32_t>();
    ix->sorted_rows = nullptr;
    ix->unsorted_rows = nullptr;

    if (R == 0) {
        HIP_TRY(ix, hipMemsetAsync(row_ptr, 0, ((size_t)V + 1) * 4, ix->stream));
        HIP_TRY(ix, hipMemsetAsync(news_ptr, 0, (size_t)V * 4, ix->stream));
        HIP_TRY(ix, hipMemsetAsync(ix->df_local.p, 0, (size_t)V * 4, ix->stream));
        HIP_TRY(ix, hipMemsetAsync(ix->term_rows.p, 0, (size_t)V * 4, ix->stream));
        ix->post = ix->sortA.as<Slot>();
        ix->n_pass = 0;
        ix->built_fused = false;
        return SEBM25_OK;
    }
    // ---- stable LSD radix sort on the key bits of the packed row
    const int key_lo = unit_major ? fmt.term_shift() : fmt.key_shift();
    const int key_bits = 64 - key_lo;
    const int n_pass = presorted ? 0 : (key_bits + ix->sort_bits - 1) / ix->sort_bits;
    ix->n_pass = (uint32_t)n_pass;
    // one chunk = one tile = the work of one scatter workgroup
    const uint64_t chunk_rows = SORT_TILE;
    const uint32_t G = (uint32_t)((R + chunk_rows - 1) / chunk_rows);
    const uint32_t n_groups = (G + SORT_GROUP - 1) / SORT_GROUP;
    HIP_TRY(ix, ix->chunk_hist.ensure((size_t)SORT_MAX_BINS * G * 2));
    HIP_TRY(ix, ix->group_off.ensure((size_t)SORT_MAX_BINS * n_groups * 4));
    HIP_TRY(ix, ix->bin_start.ensure((size_t)SORT_MAX_BINS * 4 * 2));
    HIP_TRY(ix, ix->scan_tmp.ensure(std::max(sort_offsets_tmp_words(), scatter_post_tmp_words(G)) * 4));
    // Two passes over the term bits of unit-major raw rows (every vocabulary of 1025 .. 2^20 terms): the second
    // pass writes the posting slots itself and the sorted rows are never stored.
    const bool fuse = ix->fused && !ix->keep_rows && !start_packed && !presorted && unit_major && n_pass == 2;

    BuildCtx ctx{};
    ctx.fmt = fmt;
    ctx.n_terms = V; ctx.n_docs = ix->n_docs; ctx.n_articles = ix->n_articles;
    ctx.check_order = unit_major ? 1u : 0u;
    ctx.doc_len = ix->doc_len.as<uint32_t>();
    ctx.doc_flags = ix->n_docs ? flags_dev : nullptr;
    ctx.err = err_dev;

    // at most ceil(54 key bits / 4-bit digits) = 14 passes
    static const char* hist_names[] = {"sort_hist_0", "sort_hist_1", "sort_hist_2", "sort_hist_3", "sort_hist_4", "sort_hist_5", "sort_hist_6", "sort_hist_7",
                                       "sort_hist_8", "sort_hist_9", "sort_hist_10", "sort_hist_11", "sort_hist_12", "sort_hist_13", "sort_hist_14", "sort_hist_15"};
    static const char* scan_names[] = {"sort_scan_0", "sort_scan_1", "sort_scan_2", "sort_scan_3", "sort_scan_4", "sort_scan_5", "sort_scan_6", "sort_scan_7",
                                       "sort_scan_8", "sort_scan_9", "sort_scan_10", "sort_scan_11", "sort_scan_12", "sort_scan_13", "sort_scan_14", "sort_scan_15"};
    static const char* scat_names[] = {"sort_scatter_0", "sort_scatter_1", "sort_scatter_2", "sort_scatter_3", "sort_scatter_4", "sort_scatter_5", "sort_scatter_6", "sort_scatter_7",
                                       "sort_scatter_8", "sort_scatter_9", "sort_scatter_10", "sort_scatter_11", "sort_scatter_12", "sort_scatter_13", "sort_scatter_14", "sort_scatter_15"};
    if (n_pass > 16) return fail(ix, SEBM25_E_INTERNAL, "build: more than 16 sort passes");
    const uint64_t* src_packed = start_packed;
    uint64_t* bufs[2] = {ix->sortA.as<uint64_t>(), ix->sortB.as<uint64_t>()};
    const int first_dst = start_packed ? 1 : 0;  // packed rows start in buffer A: the first pass writes B
    int shift = key_lo, bits_before = 0;
    for (int p = 0; p < n_pass; ++p) {
        const int bits = key_bits / n_pass + (p < key_bits % n_pass ? 1 : 0);
        const int mode = (p == 0 && !start_packed) ? (shift >= fmt.term_shift() ? 0 : 1) : 2;
        uint64_t* dst = bufs[(p + first_dst) & 1];
        unsigned short* prefix = ix->chunk_hist.as<unsigned short>();
        uint32_t* group_off = ix->group_off.as<uint32_t>();
        uint32_t* bin_start = ix->bin_start.as<uint32_t>() + (size_t)(p & 1) * SORT_MAX_BINS;
        phase_begin(ix, hist_names[p]);
        launch_sort_hist(mode, raw, src_packed, R, chunk_rows, G, shift, bits, ctx, prefix, group_off, ix->stream);
        phase_end(ix);
        phase_begin(ix, scan_names[p]);
        launch_sort_offsets(group_off, n_groups, bits, bin_start, ix->scan_tmp.as<uint32_t>(), ix->stream);
        phase_end(ix);
        if (fuse && p == 1) {
            // the posting slots take the place the sorted rows would have had
            ix->post = reinterpret_cast<Slot*>(dst);
            HIP_TRY(ix, hipMemsetAsync(news_ptr, 0xFF, (size_t)V * 4, ix->stream));
            HIP_TRY(ix, hipMemsetAsync(ix->df_local.p, 0, (size_t)V * 4, ix->stream));
            phase_begin(ix, "first_slots");
            launch_first_slots(src_packed, R, G, shift, bits_before, bits, ix->bin_start.as<uint32_t>(), prefix, group_off, bin_start,
                               V, row_ptr, ix->post + R, ix->stream);
            phase_end(ix);
            phase_begin(ix, "scatter_postings");
            launch_sort_scatter_post(src_packed, R, G, shift, bits, ctx, prefix, group_off, ix->post, news_ptr,
                                     ix->df_local.as<uint32_t>(), ix->scan_tmp.as<uint32_t>(), scal, ix->stream);
            phase_end(ix);
            phase_begin(ix, "term_stats");
            launch_term_stats_dup(V, row_ptr, news_ptr, ix->term_rows.as<uint32_t>(), ix->df_local.as<uint32_t>(), ix->stream);
            phase_end(ix);
            HIP_TRY(ix, hipGetLastError());
            ix->unsorted_rows = src_packed;
            ix->built_fused = true;
            ix->last_shift = shift;
            ix->last_bits = bits;
            return SEBM25_OK;
        }
        phase_begin(ix, scat_names[p]);
        launch_sort_scatter(mode, raw, src_packed, dst, R, G, shift, bits, ctx, prefix, group_off, ix->stream);
        phase_end(ix);
        src_packed = dst;
        shift += bits;
        bits_before = bits;
    }
    HIP_TRY(ix, hipGetLastError());
    const uint64_t* sorted = src_packed;
    uint64_t* other = sorted == bufs[0] ? bufs[1] : bufs[0];  // the buffer the last pass did not write
    // postings live in the free sort buffer: unit ids in its first half, weights in the second
    int rc = post_sorted(ix, sorted, other);
    if (rc) return rc;
    ix->sorted_rows = sorted;
    return SEBM25_OK;
}

// reads the counters a build leaves in ix->scalars, runs the rare exact second look, reports device-side errors
int collect_build(sebm25_index* ix, uint64_t (&h)[8], bool* redone = nullptr) {
    if (redone) *redone = false;
    uint64_t* scal = ix->scalars.as<uint64_t>();
    uint32_t* err_dev = reinterpret_cast<uint32_t*>(scal + 7);
    HIP_TRY(ix, hipMemcpyAsync(h, scal, 64, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    uint32_t dev_err = (uint32_t)(h[7] & 0xFFFFFFFFull);
    if ((dev_err & DEV_NEED_FIXUP) && !(dev_err & (DEV_ERR_RANGE | DEV_ERR_ORDER)) && ix->unsorted_rows) {
        // the last sort pass met runs it does not turn into slots itself (see post_tile): the rows are sorted out
        // and the separate posting pass redoes slots, list bounds and counts
        int rc = ensure_sorted_rows(ix, ix);
        if (rc) return rc;
        HIP_TRY(ix, hipMemsetAsync(scal, 0, 16, ix->stream));       // kept postings, web heads
        HIP_TRY(ix, hipMemsetAsync(scal + 5, 0, 8, ix->stream));    // kept web postings
        HIP_TRY(ix, hipMemsetAsync(scal + 7, 0, 8, ix->stream));    // device error word
        uint64_t* slots = ix->sorted_rows == ix->sortA.as<uint64_t>() ? ix->sortB.as<uint64_t>() : ix->sortA.as<uint64_t>();
        if ((rc = post_sorted(ix, ix->sorted_rows, slots))) return rc;
        if (redone) *redone = true;
        HIP_TRY(ix, hipMemcpyAsync(h, scal, 64, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(ix, hipStreamSynchronize(ix->stream));
        dev_err = (uint32_t)(h[7] & 0xFFFFFFFFull);
    }
    if ((dev_err & DEV_NEED_FIXUP) && !(dev_err & (DEV_ERR_RANGE | DEV_ERR_ORDER))) {
        // rare runs the streaming kernel leaves to an exact second look
        launch_postings_fixup(ix->sorted_rows, ix->R, ix->fmt, ix->post, err_dev, ix->stream);
        HIP_TRY(ix, hipGetLastError());
        HIP_TRY(ix, hipMemcpyAsync(h + 7, scal + 7, 8, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(ix, hipStreamSynchronize(ix->stream));
        dev_err = (uint32_t)(h[7] & 0xFFFFFFFFull);
    }
    if (dev_err & DEV_ERR_RANGE)
        return fail(ix, SEBM25_E_RANGE, "build: a row has term_id >= n_terms, a unit id outside its id space, an invalid "
                                        "source_type, a tf too wide for the packed row, or a posting weight above 2^32 - 2 tenths");
    if (dev_err & DEV_ERR_ORDER)
        return fail(ix, SEBM25_E_ORDER, "build: rows were declared unit-major (SEBM25_ROWS_UNIT_MAJOR) but are not");
    ix->big_weights = (dev_err & DEV_BIG_WEIGHT) != 0;
    ix->P = h[0];
    ix->WH = h[1];
    ix->P_web = h[5];
    return SEBM25_OK;
}

// and_finalize: the single-GPU job.  idf is queued behind the document totals with N read on the device, so the whole
// job waits for the GPU once.
int do_build_local(sebm25_index* ix, const sebm25_rows* in, bool and_finalize = false) {
    if (!in) return fail(ix, SEBM25_E_INVALID, "build: rows is NULL");
    if (in->n_terms == 0) return fail(ix, SEBM25_E_INVALID, "build: n_terms must be >= 1");
    if (in->n_rows > ix->max_rows) return fail(ix, SEBM25_E_RANGE, "build: more than 2^32 - 4096 rows in one shard");
    if (in->n_rows && (!in->term_id || !in->unit_id || !in->tf || !in->meta))
        return fail(ix, SEBM25_E_INVALID, "build: a row array is NULL");
    if (in->flags & ~SEBM25_ROWS_UNIT_MAJOR) return fail(ix, SEBM25_E_INVALID, "build: unknown flag");

    ix->built = ix->finalized = false;
    ix->host_ptrs_valid = false;
    drop_deltas(ix);
    ix->has_dead = false;
    ix->dead_rows = 0;
    ix->sorted_rows = nullptr;
    ix->unsorted_rows = nullptr;
    ix->h_row_ptr.resize((size_t)in->n_terms + 1);  // filled by the first search; sized here so that a search never allocates them
    ix->h_news_ptr.resize(in->n_terms);
    ix->h_list_len.resize((size_t)in->n_terms * 2);
    const uint64_t R = in->n_rows;
    const uint32_t V = in->n_terms;
    RowFmt fmt;
    fmt.term_bits = bits_for(V);
    fmt.unit_bits = bits_for(std::max<uint64_t>(std::max(in->n_docs, in->n_articles), 1));
    fmt.tf_bits = 61 - fmt.term_bits - fmt.unit_bits;
    if (fmt.tf_bits < 8)
        return fail(ix, SEBM25_E_RANGE, "build: term and unit id spaces leave fewer than 8 bits for tf in the packed row");
    ix->R = R; ix->V = V; ix->n_docs = in->n_docs; ix->n_articles = in->n_articles; ix->fmt = fmt;
    ix->V_cur = V; ix->n_docs_cur = in->n_docs; ix->n_articles_cur = in->n_articles;
    phases_reset(ix);

    // ---- device memory
    HIP_TRY(ix, ix->idf.ensure((size_t)V * 8));
    HIP_TRY(ix, ix->idf_used.ensure((size_t)V * 8));
    HIP_TRY(ix, ix->doc_len.ensure(std::max<size_t>(in->n_docs, 1) * 4));
    HIP_TRY(ix, ix->scalars.ensure(64));
    HIP_TRY(ix, hipMemsetAsync(ix->scalars.p, 0, 64, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(ix->doc_len.p, 0, std::max<size_t>(in->n_docs, 1) * 4, ix->stream));
    uint64_t* scal = ix->scalars.as<uint64_t>();
    const size_t Rcap = std::max<uint64_t>(R, 1);
    HIP_TRY(ix, ix->sortA.ensure((Rcap + POST_PAD) * 8));  // before the inputs are staged: a failure here costs no copy
    HIP_TRY(ix, ix->sortB.ensure((Rcap + POST_PAD) * 8));

    // ---- inputs
    RawRows raw{};
    const uint32_t* wc_dev = nullptr;
    const uint8_t* flags_dev = nullptr;
    int rc;
    ix->h2d.clear();
    if ((rc = stage_in(ix, in->term_id, R, ix->st_term, &raw.term))) return rc;
    if ((rc = stage_in(ix, in->unit_id, R, ix->st_unit, &raw.unit))) return rc;
    if ((rc = stage_in(ix, in->tf, R, ix->st_tf, &raw.tf))) return rc;
    if ((rc = stage_in(ix, in->meta, R, ix->st_meta, &raw.meta))) return rc;
    if ((rc = stage_in(ix, in->word_count, in->n_docs, ix->st_wc, &wc_dev))) return rc;
    if ((rc = stage_in(ix, in->doc_flags, in->n_docs, ix->st_flags, &flags_dev))) return rc;
    if ((rc = flush_h2d(ix))) return rc;
    ix->wc_dev = wc_dev;
    ix->flags_dev = flags_dev;

    if ((rc = sort_and_post(ix, raw, nullptr, (in->flags & SEBM25_ROWS_UNIT_MAJOR) != 0, flags_dev))) return rc;
    phase_begin(ix, "doc_stats");
    launch_doc_stats(wc_dev, ix->doc_len.as<uint32_t>(), flags_dev, in->n_docs, scal + 2, ix->stream);
    phase_end(ix);
    HIP_TRY(ix, hipGetLastError());

    if (and_finalize) {
        phase_begin(ix, "idf");
        launch_idf(ix->df_local.as<uint32_t>(), V, 1, scal + 2, ix->idf.as<double>(), ix->idf_used.as<double>(), ix->stream);
        phase_end(ix);
        HIP_TRY(ix, hipGetLastError());
    }
    uint64_t h[8];
    bool redone = false;
    if ((rc = collect_build(ix, h, &redone))) return rc;
    ix->local_totals.n_done = h[2];
    ix->local_totals.n_wc = h[3];
    ix->local_totals.sum_wc = h[4];
    ix->local_totals.n_postings = h[0];
    ix->built = true;
    if (and_finalize) {
        if (redone) return do_finalize(ix, nullptr, nullptr);   // df was made again by the separate posting pass
        ix->has_df_global = false;
        ix->used_totals = ix->local_totals;
        ix->N = h[2] ? h[2] : 1;                                             // `or 1`
        ix->avgdl = h[3] ? (double)h[4] / (double)h[3] : 0.0;                // `or 0.0`
        ix->finalized = true;
    }
    return SEBM25_OK;
}

// ---- compaction: all segments folded into one, on the device --------------------------------
int do_compact(sebm25_index* ix);
int fold_deltas(sebm25_index* ix);

// ---- incremental update --------------------------------------------------------------------
std::vector<sebm25_index*> all_segments(sebm25_index* ix) {
    std::vector<sebm25_index*> v{ix};
    v.insert(v.end(), ix->deltas.begin(), ix->deltas.end());
    return v;
}

int ensure_dead_bitmaps(sebm25_index* ix, sebm25_index* seg) {
    if (seg->has_dead) return SEBM25_OK;
    const size_t ww = ((size_t)seg->n_docs + 31) / 32 + 1, wn = ((size_t)seg->n_articles + 31) / 32 + 1;
    HIP_TRY(ix, seg->dead_web.ensure(ww * 4));
    HIP_TRY(ix, seg->dead_news.ensure(wn * 4));
    HIP_TRY(ix, hipMemsetAsync(seg->dead_web.p, 0, ww * 4, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(seg->dead_news.p, 0, wn * 4, ix->stream));
    seg->has_dead = true;
    return SEBM25_OK;
}

// df and row counts per term summed over the segments
int sum_term_counts(sebm25_index* ix) {
    const size_t V = ix->V_cur;
    HIP_TRY(ix, ix->df_sum.ensure(V * 4));
    HIP_TRY(ix, ix->term_rows_sum.ensure(V * 4));
    HIP_TRY(ix, hipMemsetAsync(ix->df_sum.p, 0, V * 4, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(ix->term_rows_sum.p, 0, V * 4, ix->stream));
    for (sebm25_index* seg : all_segments(ix)) {
        launch_add_u32(ix->df_sum.as<uint32_t>(), seg->df_local.as<uint32_t>(), seg->V, ix->stream);
        launch_add_u32(ix->term_rows_sum.as<uint32_t>(), seg->term_rows.as<uint32_t>(), seg->V, ix->stream);
    }
    HIP_TRY(ix, hipGetLastError());
    return SEBM25_OK;
}

// grows a per-term or per-doc array, keeping what it holds and zero-filling the rest
int grow_keep(sebm25_index* ix, DevBuf& buf, size_t old_bytes, size_t new_bytes) {
    if (new_bytes <= buf.cap) {
        if (new_bytes > old_bytes) HIP_TRY(ix, hipMemsetAsync(static_cast<char*>(buf.p) + old_bytes, 0, new_bytes - old_bytes, ix->stream));
        return SEBM25_OK;
    }
    DevBuf nb;
    HIP_TRY(ix, nb.ensure(new_bytes));
    hipError_t e = hipSuccess;
    if (old_bytes) e = hipMemcpyAsync(nb.p, buf.p, old_bytes, hipMemcpyDeviceToDevice, ix->stream);
    if (e == hipSuccess) e = hipMemsetAsync(static_cast<char*>(nb.p) + old_bytes, 0, new_bytes - old_bytes, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        nb.release();
        HIP_TRY(ix, e);
    }
    buf.release();
    buf = nb;
    return SEBM25_OK;
}

int do_update(sebm25_index* ix, const sebm25_rows* in, const uint32_t* chg_docs, uint32_t n_cd, const uint32_t* chg_arts,
              uint32_t n_ca) {
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "update: no index built on this handle");
    if (!in) return fail(ix, SEBM25_E_INVALID, "update: rows is NULL");
    if ((n_cd && !chg_docs) || (n_ca && !chg_arts)) return fail(ix, SEBM25_E_INVALID, "update: a list of changed units is NULL");
    if (in->n_terms < ix->V_cur || in->n_docs < ix->n_docs_cur || in->n_articles < ix->n_articles_cur)
        return fail(ix, SEBM25_E_INVALID, "update: the term, doc and article id spaces can only grow");
    const uint32_t n_docs = in->n_docs, n_arts = in->n_articles, V = in->n_terms;
    int rc;
    const bool dbg = getenv("SEBM25_DEBUG_TIMES") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lap = [&](const char* what) {
        if (!dbg) return;
        (void)hipStreamSynchronize(ix->stream);
        const double t1 = now();
        fprintf(stderr, "[sebm25 update] %-22s %.3f ms\n", what, t1 - t0);
        t0 = t1;
    };

    // ---- the units this update replaces or removes, as bitmaps over the new id spaces
    const size_t ww = ((size_t)n_docs + 31) / 32 + 1, wn = ((size_t)n_arts + 31) / 32 + 1;
    HIP_TRY(ix, ix->chg_web.ensure(ww * 4));
    HIP_TRY(ix, ix->chg_news.ensure(wn * 4));
    HIP_TRY(ix, hipMemsetAsync(ix->chg_web.p, 0, ww * 4, ix->stream));
    HIP_TRY(ix, hipMemsetAsync(ix->chg_news.p, 0, wn * 4, ix->stream));
    HIP_TRY(ix, ix->upd_ids.ensure(((size_t)n_cd + n_ca + 1) * 4));
    uint32_t* ids_dev = ix->upd_ids.as<uint32_t>();
    if (n_cd) HIP_TRY(ix, hipMemcpyAsync(ids_dev, chg_docs, (size_t)n_cd * 4, hipMemcpyDefault, ix->stream));
    if (n_ca) HIP_TRY(ix, hipMemcpyAsync(ids_dev + n_cd, chg_arts, (size_t)n_ca * 4, hipMemcpyDefault, ix->stream));
    launch_bitmap_set(ids_dev, n_cd, n_docs, 0, ix->chg_web.as<uint32_t>(), ix->stream);
    launch_bitmap_set(ids_dev + n_cd, n_ca, n_arts, 0, ix->chg_news.as<uint32_t>(), ix->stream);
    // coarse copies (one bit per 2^shift units) small enough for every block of the dead-row scan to hold in LDS
    uint32_t shift = 6;
    while ((((size_t)n_docs >> shift) + ((size_t)n_arts >> shift)) / 8 > 16384) ++shift;
    const uint32_t cw = (uint32_t)(((size_t)n_docs >> shift) / 32 + 1), c_words = cw + (uint32_t)(((size_t)n_arts >> shift) / 32 + 1);
    HIP_TRY(ix, ix->chg_coarse.ensure((size_t)c_words * 4));
    HIP_TRY(ix, hipMemsetAsync(ix->chg_coarse.p, 0, (size_t)c_words * 4, ix->stream));
    launch_bitmap_set(ids_dev, n_cd, n_docs, shift, ix->chg_coarse.as<uint32_t>(), ix->stream);
    launch_bitmap_set(ids_dev + n_cd, n_ca, n_arts, shift, ix->chg_coarse.as<uint32_t>() + cw, ix->stream);
    HIP_TRY(ix, hipGetLastError());
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));  // the id lists may be pageable host memory

    lap("bitmaps");
    // ---- the new rows become a segment of their own (nothing that exists is touched until this has succeeded)
    sebm25_index* seg = take_segment(ix);
    if (!seg) return fail(ix, SEBM25_E_NOMEM, "update: out of host memory");
    rc = do_build_local(seg, in);
    uint32_t listed_err = 0;
    if (!rc && seg->R) {
        uint32_t* err_dev = reinterpret_cast<uint32_t*>(seg->scalars.as<uint64_t>() + 7);
        hipError_t e = hipMemsetAsync(err_dev, 0, 4, ix->stream);
        launch_check_listed(seg->sorted_rows, seg->R, seg->fmt, ix->chg_web.as<uint32_t>(), ix->chg_news.as<uint32_t>(), err_dev,
                            ix->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&listed_err, err_dev, 4, hipMemcpyDeviceToHost, ix->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            rc = fail(seg, SEBM25_E_HIP, std::string("update: ") + hipGetErrorString(e));
        } else if (listed_err) {
            rc = fail(seg, SEBM25_E_INVALID, "update: a row belongs to a unit that is not listed as changed");
        }
    }
    if (rc) {
        const std::string msg = seg->err;
        sebm25_destroy(seg);
        return fail(ix, rc, msg);
    }

    lap("build of the segment");
    // ---- everything that can fail for want of memory or for the sizes involved happens before a row of the index is
    // touched: a failure up to here leaves the index as it was
    const std::vector<sebm25_index*> segs = all_segments(ix);
    auto give_up = [&](int code) {   // error text already set on ix
        sebm25_destroy(seg);
        return code;
    };
    {
        // the fold of this update's rows with those of the update segment that is there repacks both to the row format of
        // the id spaces as they will be
        RowFmt nf;
        nf.term_bits = bits_for(V);
        nf.unit_bits = bits_for(std::max<uint64_t>(std::max(n_docs, n_arts), 1));
        nf.tf_bits = 61 - nf.term_bits - nf.unit_bits;
        uint64_t n_upd = seg->R;
        bool narrow = false;
        for (const sebm25_index* d : ix->deltas) {
            n_upd += d->R - d->dead_rows;
            narrow |= nf.tf_bits < d->fmt.tf_bits && nf.tf_bits < 32;
        }
        if (!ix->deltas.empty() && narrow)
            return give_up(fail(ix, SEBM25_E_RANGE, "update: the grown id spaces leave too few bits for tf in the packed row (rebuild)"));
        if (n_upd > MAX_ROWS) return give_up(fail(ix, SEBM25_E_RANGE, "update: more than 2^32 - 4096 rows in the update segments (compact)"));
    }
    auto prepare = [&]() -> int {
        int r;
        HIP_TRY(ix, ix->upd_counts.ensure(segs.size() * 4 * 8));
        for (sebm25_index* g : segs) {
            if ((r = ensure_dead_bitmaps(ix, g))) return r;
            if ((r = ensure_sorted_rows(ix, g))) return r;
        }
        if ((r = grow_keep(ix, ix->doc_len, (size_t)ix->n_docs_cur * 4, std::max<size_t>(n_docs, 1) * 4))) return r;
        if ((r = grow_keep(ix, ix->idf, (size_t)ix->V_cur * 8, (size_t)V * 8))) return r;
        if ((r = grow_keep(ix, ix->idf_used, (size_t)ix->V_cur * 8, (size_t)V * 8))) return r;
        const size_t sums_held = ix->deltas.empty() ? 0 : (size_t)ix->V_cur * 4;
        if ((r = grow_keep(ix, ix->df_sum, sums_held, (size_t)V * 4))) return r;
        if ((r = grow_keep(ix, ix->term_rows_sum, sums_held, (size_t)V * 4))) return r;
        HIP_TRY(ix, hipMemsetAsync(ix->upd_counts.p, 0, segs.size() * 4 * 8, ix->stream));
        return SEBM25_OK;
    };
    if ((rc = prepare())) return give_up(rc);

    // ---- from here on rows change.  Only a failing HIP call can interrupt it; the handle then counts as unbuilt, so that a
    // half-updated index is never searched.
    ix->finalized = false;
    std::vector<unsigned long long> counts(segs.size() * 4);
    uint64_t h[3];
    auto apply = [&]() -> int {
        // in every segment that exists, the rows of the changed units die
        for (size_t s = 0; s < segs.size(); ++s) {
            sebm25_index* g = segs[s];
            launch_tombstone(g->sorted_rows, g->R, g->fmt, ix->chg_web.as<uint32_t>(), ix->chg_news.as<uint32_t>(),
                             ix->chg_coarse.as<uint32_t>(), cw, c_words, shift, g->dead_web.as<uint32_t>(), g->dead_news.as<uint32_t>(), g->post, g->df_local.as<uint32_t>(),
                             g->term_rows.as<uint32_t>(), ix->upd_counts.as<unsigned long long>() + 4 * s, ix->stream);
            launch_bitmap_or(g->dead_web.as<uint32_t>(), ix->chg_web.as<uint32_t>(), (uint32_t)(((size_t)g->n_docs + 31) / 32), ix->stream);
            launch_bitmap_or(g->dead_news.as<uint32_t>(), ix->chg_news.as<uint32_t>(), (uint32_t)(((size_t)g->n_articles + 31) / 32), ix->stream);
        }
        HIP_TRY(ix, hipGetLastError());
        HIP_TRY(ix, hipMemcpyAsync(counts.data(), ix->upd_counts.p, counts.size() * 8, hipMemcpyDeviceToHost, ix->stream));
        lap("dead rows");
        // per-doc body lengths: the new segment's for changed docs
        launch_doc_len_merge(ix->doc_len.as<uint32_t>(), seg->doc_len.as<uint32_t>(), ix->chg_web.as<uint32_t>(), n_docs, ix->stream);
        // document totals from the documents columns as they are now (bm25_stats.py:9,12)
        uint64_t* scal = ix->scalars.as<uint64_t>();
        HIP_TRY(ix, hipMemsetAsync(scal + 2, 0, 24, ix->stream));
        launch_doc_stats(seg->wc_dev, ix->doc_len.as<uint32_t>(), seg->flags_dev, n_docs, scal + 2, ix->stream);
        HIP_TRY(ix, hipGetLastError());
        HIP_TRY(ix, hipMemcpyAsync(h, scal + 2, 24, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(ix, hipStreamSynchronize(ix->stream));
        return SEBM25_OK;
    };
    if ((rc = apply())) {
        ix->built = false;
        return give_up(rc);
    }
    // ---- the handle now spans the new id spaces
    if (n_docs > ix->n_docs_cur) ix->has_rank_web = false;      // the rank arrays no longer cover every unit: the caller sets them again
    if (n_arts > ix->n_articles_cur) ix->has_rank_news = false;
    ix->deltas.push_back(seg);
    ix->V_cur = V; ix->n_docs_cur = n_docs; ix->n_articles_cur = n_arts;
    uint64_t P_all = seg->P;
    for (size_t s = 0; s < segs.size(); ++s) {
        sebm25_index* g = segs[s];
        g->P -= counts[4 * s];
        g->P_web -= counts[4 * s + 1];
        g->WH -= counts[4 * s + 2];
        g->dead_rows += counts[4 * s + 3];
        P_all += g->P;
    }
    ix->local_totals.n_done = h[0];
    ix->local_totals.n_wc = h[1];
    ix->local_totals.sum_wc = h[2];
    ix->local_totals.n_postings = P_all;
    lap("doc lengths, document totals");
    // one update segment again.  The fold only saves later searches a segment: when it cannot be done (memory), the
    // segments stay as they are, which is a valid index
    bool summed = false;
    if (ix->deltas.size() > 1) {
        if (fold_deltas(ix) == SEBM25_OK)
            summed = true;   // the fold ends with the sums over the segments
        else
            (void)hipGetLastError();
    }
    lap("update segments folded");
    if (!summed && (rc = sum_term_counts(ix))) {   // buffers are in place: only a failing HIP call ends up here
        ix->built = false;
        return rc;
    }
    return SEBM25_OK;
}


// row format for the id spaces as they are now
RowFmt current_fmt(const sebm25_index* ix) {
    RowFmt fmt;
    fmt.term_bits = bits_for(ix->V_cur);
    fmt.unit_bits = bits_for(std::max<uint64_t>(std::max(ix->n_docs_cur, ix->n_articles_cur), 1));
    fmt.tf_bits = 61 - fmt.term_bits - fmt.unit_bits;
    return fmt;
}

// The update segments become one: their live rows are gathered (repacked to the current row format), sorted and
// turned into postings like any segment.  An update calls this, so that a search never scores more than the base
// segment and one update segment (every further segment costs a batch its planning and launches again), and a
// compaction, which then merges that one sorted segment into the base.
int fold_deltas(sebm25_index* ix) {
    if (ix->deltas.empty()) return SEBM25_OK;
    const RowFmt fmt = current_fmt(ix);
    if (ix->deltas.size() == 1) {
        const sebm25_index* d = ix->deltas[0];
        if (d->dead_rows == 0 && d->fmt.term_bits == fmt.term_bits && d->fmt.unit_bits == fmt.unit_bits) return SEBM25_OK;
    }
    uint64_t n_upd = 0;
    for (const sebm25_index* d : ix->deltas) {
        n_upd += d->R - d->dead_rows;
        if (fmt.tf_bits < d->fmt.tf_bits && fmt.tf_bits < 32)
            return fail(ix, SEBM25_E_RANGE, "update: the grown id spaces leave too few bits for tf in the packed row (rebuild)");
    }
    if (n_upd > MAX_ROWS) return fail(ix, SEBM25_E_RANGE, "update: more than 2^32 - 4096 rows in the update segments (compact)");
    int rc;
    for (sebm25_index* d : ix->deltas)
        if ((rc = ensure_dead_bitmaps(ix, d))) return rc;
    sebm25_index* one = take_segment(ix);
    if (!one) return fail(ix, SEBM25_E_NOMEM, "update: out of host memory");
    auto give_back = [&](int code, const std::string& msg) {
        sebm25_destroy(one);
        return fail(ix, code, msg);
    };
    one->R = n_upd; one->V = one->V_cur = ix->V_cur; one->n_docs = one->n_docs_cur = ix->n_docs_cur;
    one->n_articles = one->n_articles_cur = ix->n_articles_cur; one->fmt = fmt;
    one->has_dead = false; one->dead_rows = 0; one->host_ptrs_valid = false;
    one->h_row_ptr.resize((size_t)one->V + 1);
    one->h_news_ptr.resize(one->V);
    one->h_list_len.resize((size_t)one->V * 2);
    hipError_t e = one->scalars.ensure(64);
    if (e == hipSuccess) e = hipMemsetAsync(one->scalars.p, 0, 64, ix->stream);
    if (e == hipSuccess) e = one->sortA.ensure((std::max<uint64_t>(n_upd, 1) + POST_PAD) * 8);
    if (e == hipSuccess) e = ix->upd_counts.ensure(64);
    if (e == hipSuccess) e = hipMemsetAsync(ix->upd_counts.p, 0, 8, ix->stream);
    unsigned long long got = 0;
    if (e == hipSuccess) {
        for (sebm25_index* d : ix->deltas)
            launch_gather_live(d->sorted_rows, d->R, d->fmt, fmt, d->dead_web.as<uint32_t>(), d->dead_news.as<uint32_t>(),
                               one->sortA.as<uint64_t>(), ix->upd_counts.as<unsigned long long>(), ix->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&got, ix->upd_counts.p, 8, hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return give_back(e == hipErrorOutOfMemory ? SEBM25_E_NOMEM : SEBM25_E_HIP, std::string("update: ") + hipGetErrorString(e));
    }
    if (got != n_upd) return give_back(SEBM25_E_INTERNAL, "update: live row count of the update segments disagrees with their bookkeeping");
    if ((rc = sort_and_post(one, RawRows{}, one->sortA.as<uint64_t>(), false, nullptr))) return give_back(rc, one->err);
    uint64_t h[8];
    if ((rc = collect_build(one, h))) return give_back(rc, one->err);
    one->built = true;
    drop_deltas(ix);
    ix->deltas.push_back(one);
    return sum_term_counts(ix);
}

int do_compact(sebm25_index* ix) {
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "compact: no index built on this handle");
    if (ix->deltas.empty() && ix->dead_rows == 0) return SEBM25_OK;
    int rc;
    if ((rc = fold_deltas(ix))) return rc;   // at most one update segment from here on: sorted, no dead rows, current row format
    const RowFmt fmt = current_fmt(ix);
    if (fmt.tf_bits < ix->fmt.tf_bits && fmt.tf_bits < 32)
        return fail(ix, SEBM25_E_RANGE, "compact: the grown id spaces leave too few bits for tf in the packed row (rebuild)");
    const uint64_t n_upd = ix->deltas.empty() ? 0 : ix->deltas[0]->R;
    const uint64_t* upd_rows = ix->deltas.empty() ? nullptr : ix->deltas[0]->sorted_rows;
    const uint64_t total = ix->R - ix->dead_rows + n_upd;
    if (total > MAX_ROWS) return fail(ix, SEBM25_E_RANGE, "compact: more than 2^32 - 4096 live rows");
    if ((rc = ensure_dead_bitmaps(ix, ix))) return rc;
    if ((rc = ensure_sorted_rows(ix, ix))) return rc;
    // ---- merge: the base segment's rows are sorted already.  Live rows per tile, their running sum, the insertion
    // point of every update row, then one pass that writes both to their merged places.
    // The base segment's sorted rows sit in one of its two sort buffers and its posting slots in the other; the
    // slots are rebuilt anyway, so their buffer takes the merged rows when it is large enough (no allocation).
    const size_t cap = std::max<uint64_t>(total, 1);
    const size_t need = (cap + POST_PAD) * 8;
    DevBuf* slot_buf = ix->sorted_rows == ix->sortA.as<uint64_t>() ? &ix->sortB : &ix->sortA;
    DevBuf nA;
    const bool in_place = ix->sorted_rows && slot_buf->cap >= need;
    if (!in_place) HIP_TRY(ix, nA.ensure(need));
    uint64_t* merged = in_place ? slot_buf->as<uint64_t>() : nA.as<uint64_t>();
    const uint32_t n_tiles = merge_tiles_for(ix->R);
    hipError_t e = ix->chunk_hist.ensure(((size_t)n_tiles + 1) * 4);
    if (e == hipSuccess) e = ix->scan_tmp.ensure(scan_tmp_words((size_t)n_tiles + 1) * 4);
    if (e == hipSuccess) e = ix->upd_ids.ensure(std::max<uint64_t>(n_upd, 1) * 4);
    uint32_t* tile_off = ix->chunk_hist.as<uint32_t>();
    uint32_t* ins = ix->upd_ids.as<uint32_t>();
    uint32_t live_base = 0;
    if (e == hipSuccess) e = hipMemsetAsync(tile_off + n_tiles, 0, 4, ix->stream);
    if (e == hipSuccess) {
        launch_live_tile_counts(ix->sorted_rows, ix->R, ix->fmt, ix->dead_web.as<uint32_t>(), ix->dead_news.as<uint32_t>(), tile_off, ix->stream);
        launch_exclusive_scan_u32(tile_off, (size_t)n_tiles + 1, ix->scan_tmp.as<uint32_t>(), ix->stream);
        launch_insert_points(upd_rows, n_upd, fmt, ix->sorted_rows, ix->R, ix->fmt, ins, ix->stream);
        launch_merge_tiles(ix->sorted_rows, ix->R, ix->fmt, ix->dead_web.as<uint32_t>(), ix->dead_news.as<uint32_t>(), tile_off, upd_rows,
                           n_upd, ins, fmt, merged, ix->stream);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&live_base, tile_off + n_tiles, 4, hipMemcpyDeviceToHost, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    const unsigned long long n_live = (unsigned long long)live_base + n_upd;
    if (e != hipSuccess || n_live != total) {
        nA.release();
        ix->built = ix->built && !in_place;  // the slots of the base segment were overwritten
        if (e != hipSuccess) HIP_TRY(ix, e);
        return fail(ix, SEBM25_E_INTERNAL, "compact: live row count disagrees with the segments' bookkeeping");
    }
    // ---- from here on the handle is the one new segment, its rows in sort buffer A
    drop_deltas(ix);
    if (in_place) {
        if (slot_buf == &ix->sortB) std::swap(ix->sortA, ix->sortB);
    } else {
        ix->sortA.release();
        ix->sortA = nA;
    }
    ix->R = n_live; ix->V = ix->V_cur; ix->n_docs = ix->n_docs_cur; ix->n_articles = ix->n_articles_cur; ix->fmt = fmt;
    ix->has_dead = false;
    ix->dead_rows = 0;
    ix->host_ptrs_valid = false;
    ix->h_row_ptr.resize((size_t)ix->V + 1);
    ix->h_news_ptr.resize(ix->V);
    ix->h_list_len.resize((size_t)ix->V * 2);
    phases_reset(ix);
    uint64_t* scal = ix->scalars.as<uint64_t>();
    HIP_TRY(ix, hipMemsetAsync(scal, 0, 64, ix->stream));
    // sort_and_post sizes buffer B; buffer A already holds the rows in order (ensure keeps a buffer that is large enough)
    if ((rc = sort_and_post(ix, RawRows{}, ix->sortA.as<uint64_t>(), false, nullptr, true))) {
        ix->built = false;
        return rc;
    }
    uint64_t h[8];
    if ((rc = collect_build(ix, h))) {
        ix->built = false;
        return rc;
    }
    ix->local_totals.n_postings = ix->P;
    return SEBM25_OK;
}

}  // namespace

// ============================================================================ C ABI =======

extern "C" {

const char* sebm25_last_error(const sebm25_index* ix) { return ix ? ix->err.c_str() : g_thread_err.c_str(); }

int sebm25_create(int device, sebm25_index** out) {
    ABI_TRY
    if (!out) return fail(nullptr, SEBM25_E_INVALID, "create: out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, SEBM25_E_HIP, "create: no HIP device is visible");
    }
    if (device < 0 || device >= n) return fail(nullptr, SEBM25_E_INVALID, "create: device index out of range");
    sebm25_index* ix = new (std::nothrow) sebm25_index();
    if (!ix) return fail(nullptr, SEBM25_E_NOMEM, "create: out of host memory");
    ix->device = device;
    if (const char* e = getenv("SEBM25_SORT_BITS")) ix->sort_bits = std::min(std::max(atoi(e), 4), (int)SORT_MAX_BITS);
    if (const char* e = getenv("SEBM25_FUSED_POSTINGS")) ix->fused = atoi(e) != 0;  // 0: always the separate posting pass
    if (const char* e = getenv("SEBM25_TEST_MAX_ROWS")) ix->max_rows = std::min<uint64_t>(MAX_ROWS, strtoull(e, nullptr, 10));
    {
        // set-up of sebm25_search_noalloc: this GPU's address of the fixed device array and the pinned host plan storage
        DeviceGuard g(device);
        int rc = na_setup_device(ix, device);
        void* hp = nullptr;
        if (rc == SEBM25_OK && hipHostMalloc(&hp, sizeof(NaHost), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            hp = nullptr;  // the entry point reports it; everything else works without
        }
        ix->na_host = static_cast<NaHost*>(hp);
        if (rc) {
            g_thread_err = ix->err;
            delete ix;
            return rc;
        }
    }
    *out = ix;
    return SEBM25_OK;
    ABI_CATCH(nullptr)
}

void sebm25_destroy(sebm25_index* ix) {
    if (!ix) return;
    for (sebm25_index* d : ix->deltas) sebm25_destroy(d);
    ix->deltas.clear();
    for (sebm25_index* d : ix->spares) sebm25_destroy(d);
    ix->spares.clear();
    {
        DeviceGuard g(ix->device);
        (void)hipStreamSynchronize(ix->stream);
        for (DevBuf* b : {&ix->dead_web, &ix->dead_news, &ix->df_sum, &ix->term_rows_sum, &ix->chg_web, &ix->chg_news, &ix->chg_coarse, &ix->upd_ids,
                          &ix->upd_counts})
            b->release();
        DevBuf* all[] = {&ix->sortA, &ix->sortB, &ix->chunk_hist, &ix->group_off, &ix->bin_start, &ix->scan_tmp, &ix->row_ptr, &ix->news_ptr,
                         &ix->df_local, &ix->df_global, &ix->term_rows, &ix->idf, &ix->idf_used,
                         &ix->doc_len, &ix->scalars, &ix->st_term, &ix->st_unit, &ix->st_tf, &ix->st_meta, &ix->st_wc,
                         &ix->st_flags, &ix->st_misc, &ix->url_rank_web, &ix->url_rank_news, &ix->q_items, &ix->q_terms,
                         &ix->q_first, &ix->q_n, &ix->q_wide, &ix->o_pack, &ix->o_unit, &ix->o_score, &ix->o_mat, &ix->o_tie, &ix->o_count, &ix->t_unit,
                         &ix->t_score, &ix->t_mat, &ix->t_tie, &ix->t_count, &ix->m_in[0], &ix->m_in[1], &ix->m_in[2],
                         &ix->m_in[3], &ix->m_in[4], &ix->gen_rows, &ix->gen_cdf};
        for (DevBuf* b : all) b->release();
        for (hipEvent_t e : ix->timer.ev) (void)hipEventDestroy(e);
        if (ix->na_host) (void)hipHostFree(ix->na_host);
        if (ix->o_host) (void)hipHostFree(ix->o_host);
    }
    delete ix;
}

int sebm25_set_stream(sebm25_index* ix, void* hip_stream) {
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "set_stream: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->stream = reinterpret_cast<hipStream_t>(hip_stream);
    return SEBM25_OK;
}

int sebm25_enable_timing(sebm25_index* ix, int on) {
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "enable_timing: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->timer.enabled = on != 0;
    ix->timer.used = 0;
    return SEBM25_OK;
}

int sebm25_get_timings(sebm25_index* ix, const char** names_out, float* ms_out, uint32_t cap, uint32_t* n_out) {
    ABI_TRY
    if (!ix || !n_out) return fail(ix, SEBM25_E_INVALID, "get_timings: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    phases_collect(ix);
    const uint32_t n = (uint32_t)ix->timer.ms.size();
    *n_out = n;
    for (uint32_t i = 0; i < n && i < cap; ++i) {
        if (names_out) names_out[i] = ix->timer.names[i];
        if (ms_out) ms_out[i] = ix->timer.ms[i];
    }
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_build_local(sebm25_index* ix, const sebm25_rows* rows) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "build_local: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    return do_build_local(ix, rows);
    ABI_CATCH(ix)
}

int sebm25_finalize(sebm25_index* ix, const uint32_t* df_global, const sebm25_doc_totals* totals_global) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "finalize: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    return do_finalize(ix, df_global, totals_global);
    ABI_CATCH(ix)
}

int sebm25_build(sebm25_index* ix, const sebm25_rows* rows) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "build: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    return do_build_local(ix, rows, true);
    ABI_CATCH(ix)
}

int sebm25_update_local(sebm25_index* ix, const sebm25_rows* rows, const uint32_t* changed_docs, uint32_t n_changed_docs,
                        const uint32_t* changed_articles, uint32_t n_changed_articles) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "update_local: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    return do_update(ix, rows, changed_docs, n_changed_docs, changed_articles, n_changed_articles);
    ABI_CATCH(ix)
}

int sebm25_update(sebm25_index* ix, const sebm25_rows* rows, const uint32_t* changed_docs, uint32_t n_changed_docs,
                  const uint32_t* changed_articles, uint32_t n_changed_articles) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "update: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    int rc = do_update(ix, rows, changed_docs, n_changed_docs, changed_articles, n_changed_articles);
    if (rc) return rc;
    return do_finalize(ix, nullptr, nullptr);
    ABI_CATCH(ix)
}

int sebm25_compact(sebm25_index* ix) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "compact: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    return do_compact(ix);
    ABI_CATCH(ix)
}

int sebm25_local_stats(sebm25_index* ix, uint32_t* df_out, sebm25_doc_totals* totals_out) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "local_stats: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "local_stats: no index built on this handle");
    int rc = copy_out(ix, df_out, ix->deltas.empty() ? ix->df_local.p : ix->df_sum.p, (size_t)ix->V_cur * 4);
    if (rc) return rc;
    if (totals_out) *totals_out = ix->local_totals;
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_get_stats(sebm25_index* ix, uint32_t* df_out, double* idf_out, uint32_t* term_rows_out, uint64_t* n_docs_out,
                     double* avgdl_out) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "get_stats: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->finalized) return fail(ix, SEBM25_E_STATE, "get_stats: statistics are not finalized");
    int rc;
    const void* df_src = ix->has_df_global ? ix->df_global.p : (ix->deltas.empty() ? ix->df_local.p : ix->df_sum.p);
    if ((rc = copy_out(ix, df_out, df_src, (size_t)ix->V_cur * 4))) return rc;
    if ((rc = copy_out(ix, idf_out, ix->idf.p, (size_t)ix->V_cur * 8))) return rc;
    if ((rc = copy_out(ix, term_rows_out, ix->deltas.empty() ? ix->term_rows.p : ix->term_rows_sum.p, (size_t)ix->V_cur * 4))) return rc;
    if (n_docs_out) *n_docs_out = ix->N;
    if (avgdl_out) *avgdl_out = ix->avgdl;
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_get_doc_len(sebm25_index* ix, uint32_t* doc_len_out) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "get_doc_len: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "get_doc_len: no index built on this handle");
    int rc = copy_out(ix, doc_len_out, ix->doc_len.p, (size_t)ix->n_docs_cur * 4);
    if (rc) return rc;
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_set_idf(sebm25_index* ix, const double* idf, const uint8_t* present) {
    ABI_TRY
    if (!ix || !idf) return fail(ix, SEBM25_E_INVALID, "set_idf: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "set_idf: no index built on this handle");
    const double* idf_dev = nullptr;
    const uint8_t* pr_dev = nullptr;
    int rc;
    ix->h2d.clear();
    if ((rc = stage_in(ix, idf, ix->V_cur, ix->st_misc, &idf_dev))) return rc;
    if ((rc = stage_in(ix, present, ix->V_cur, ix->st_tf, &pr_dev))) return rc;
    if ((rc = flush_h2d(ix))) return rc;
    launch_idf_override(idf_dev, pr_dev, ix->V_cur, ix->idf_used.as<double>(), ix->stream);
    HIP_TRY(ix, hipGetLastError());
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    ix->finalized = true;
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_set_url_rank(sebm25_index* ix, int source, const uint32_t* rank) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "set_url_rank: handle is NULL");
    if (source != (int)SEBM25_SRC_WEB && source != (int)SEBM25_SRC_NEWS)
        return fail(ix, SEBM25_E_INVALID, "set_url_rank: source must be 1 (web) or 2 (news)");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "set_url_rank: no index built on this handle");
    const bool web = source == (int)SEBM25_SRC_WEB;
    bool& has = web ? ix->has_rank_web : ix->has_rank_news;
    if (!rank) {
        has = false;
        return SEBM25_OK;
    }
    DevBuf& buf = web ? ix->url_rank_web : ix->url_rank_news;
    const size_t n = web ? ix->n_docs_cur : ix->n_articles_cur;
    HIP_TRY(ix, buf.ensure(std::max<size_t>(n, 1) * 4));
    HIP_TRY(ix, hipMemcpyAsync(buf.p, rank, n * 4, hipMemcpyDefault, ix->stream));
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    (web ? ix->rank_base_web : ix->rank_base_news) = 0;
    (web ? ix->rank_shift_web : ix->rank_shift_news) = 0;
    has = true;
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_set_url_rank_appended(sebm25_index* ix, int source, uint32_t n_base, uint32_t shift, const uint32_t* keys,
                                 uint32_t n_keys) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "set_url_rank_appended: handle is NULL");
    if (source != (int)SEBM25_SRC_WEB && source != (int)SEBM25_SRC_NEWS)
        return fail(ix, SEBM25_E_INVALID, "set_url_rank_appended: source must be 1 (web) or 2 (news)");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "set_url_rank_appended: no index built on this handle");
    const bool web = source == (int)SEBM25_SRC_WEB;
    const size_t n = web ? ix->n_docs_cur : ix->n_articles_cur;
    if ((size_t)n_base + n_keys != n)
        return fail(ix, SEBM25_E_INVALID, "set_url_rank_appended: n_base + n_keys must be the number of units");
    if (n_keys && !keys) return fail(ix, SEBM25_E_INVALID, "set_url_rank_appended: keys is NULL");
    if (shift > 16 || (((uint64_t)n_base + 1) << shift) > 0xFFFFFFFFull)
        return fail(ix, SEBM25_E_RANGE, "set_url_rank_appended: (n_base + 1) << shift must fit 32 bits, shift <= 16");
    DevBuf& buf = web ? ix->url_rank_web : ix->url_rank_news;
    HIP_TRY(ix, buf.ensure(std::max<size_t>(n_keys, 1) * 4));
    if (n_keys) HIP_TRY(ix, hipMemcpyAsync(buf.p, keys, (size_t)n_keys * 4, hipMemcpyDefault, ix->stream));
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    (web ? ix->rank_base_web : ix->rank_base_news) = n_base;
    (web ? ix->rank_shift_web : ix->rank_shift_news) = shift;
    (web ? ix->has_rank_web : ix->has_rank_news) = true;
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_set_unit_base(sebm25_index* ix, uint32_t web_base, uint32_t news_base) {
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "set_unit_base: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->web_base = web_base;
    ix->news_base = news_base;
    return SEBM25_OK;
}

int sebm25_get_info(sebm25_index* ix, sebm25_index_info* out) {
    ABI_TRY
    if (!ix || !out) return fail(ix, SEBM25_E_INVALID, "get_info: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "get_info: no index built on this handle");
    out->n_rows = ix->R - ix->dead_rows;
    out->n_postings = ix->P;
    out->n_web_postings = ix->P_web;
    out->n_dead_rows = ix->dead_rows;
    out->n_update_rows = 0;
    for (const sebm25_index* d : ix->deltas) {
        out->n_update_rows += d->R;
        out->n_rows += d->R - d->dead_rows;
        out->n_postings += d->P;
        out->n_web_postings += d->P_web;
        out->n_dead_rows += d->dead_rows;
    }
    out->n_segments = 1u + (uint32_t)ix->deltas.size();
    out->n_terms = ix->V_cur;
    out->n_docs = ix->n_docs_cur;
    out->n_articles = ix->n_articles_cur;
    out->term_bits = (uint32_t)ix->fmt.term_bits;
    out->unit_bits = (uint32_t)ix->fmt.unit_bits;
    out->tf_bits = (uint32_t)ix->fmt.tf_bits;
    out->sort_passes = ix->n_pass;
    out->fused_postings = ix->built_fused ? 1u : 0u;
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_get_postings(sebm25_index* ix, uint64_t* row_ptr, uint64_t* news_ptr, uint32_t* post_unit,
                        uint32_t* post_weight) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "get_postings: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (!ix->built) return fail(ix, SEBM25_E_STATE, "get_postings: no index built on this handle");
    if ((row_ptr && is_device_ptr(row_ptr)) || (news_ptr && is_device_ptr(news_ptr)))
        return fail(ix, SEBM25_E_INVALID, "get_postings: row_ptr and news_ptr must be host arrays");
    // Every segment keeps one slot per sorted row (holes where rows were merged, filtered or have died); this
    // export compacts them on the host into the CSR the ABI describes, a term's postings from several segments
    // merged by unit.
    const std::vector<sebm25_index*> segs = all_segments(ix);
    std::vector<std::vector<Slot>> sl(segs.size());
    uint64_t P_all = 0;
    for (size_t s = 0; s < segs.size(); ++s) {
        sebm25_index* seg = segs[s];
        seg->stream = ix->stream;
        int rc = refresh_host_ptrs(seg);
        if (rc) {
            if (seg != ix) ix->err = seg->err;
            return rc;
        }
        sl[s].resize(seg->R);
        if (seg->R) HIP_TRY(ix, hipMemcpyAsync(sl[s].data(), seg->post, (size_t)seg->R * sizeof(Slot), hipMemcpyDeviceToHost, ix->stream));
        P_all += seg->P;
    }
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    const bool want_posts = post_unit || post_weight;
    std::vector<uint32_t> cu, cw;
    if (want_posts) {
        cu.reserve(P_all);
        cw.reserve(P_all);
    }
    std::vector<std::pair<uint32_t, uint32_t>> run;
    uint64_t n = 0;
    for (size_t t = 0; t < ix->V_cur; ++t) {
        if (row_ptr) row_ptr[t] = n;
        for (int news = 0; news < 2; ++news) {
            if (news && news_ptr) news_ptr[t] = n;
            run.clear();
            size_t from = 0;
            for (size_t s = 0; s < segs.size(); ++s) {
                const sebm25_index* seg = segs[s];
                if (t >= seg->V) continue;
                const uint32_t b = news ? seg->h_news_ptr[t] : seg->h_row_ptr[t];
                const uint32_t e = news ? seg->h_row_ptr[t + 1] : seg->h_news_ptr[t];
                const size_t before = run.size();
                for (uint32_t i = b; i < e; ++i)
                    if (sl[s][i].y != POST_HOLE) run.emplace_back(sl[s][i].x, sl[s][i].y);
                if (run.size() > before) ++from;
            }
            if (from > 1) std::sort(run.begin(), run.end());
            n += run.size();
            if (want_posts)
                for (const auto& pr : run) {
                    cu.push_back(pr.first);
                    cw.push_back(pr.second);
                }
        }
    }
    if (row_ptr) row_ptr[ix->V_cur] = n;
    if (n != P_all) return fail(ix, SEBM25_E_INTERNAL, "get_postings: slot arrays and posting counts disagree");
    if (post_unit && n) HIP_TRY(ix, hipMemcpy(post_unit, cu.data(), n * 4, hipMemcpyDefault));
    if (post_weight && n) HIP_TRY(ix, hipMemcpy(post_weight, cw.data(), n * 4, hipMemcpyDefault));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

// ---------------------------------------------------------------------------- search -------

}  // extern "C"

namespace {

// Query planning shared by sebm25_search (storage grown on demand) and sebm25_search_noalloc
// (storage fixed at set-up).  Pass 1 dedupes the terms of every query and decides into how many
// unit ranges a long query is cut; pass 2 emits the work items grouped by their number of lists,
// each group longest first.
struct PlanStore {
    uint32_t* flat;      // [flat_cap] deduped, matching terms of all queries
    size_t flat_cap;
    uint32_t *q_toff, *q_nt, *q_order, *q_ns;  // [n_queries] offset into flat, number of terms, list order (stepwise kernels), number of unit ranges
    uint64_t* q_len;                 // [n_queries] postings of the query
    uint32_t *first, *nlists;        // [n_queries] temporary lists of a split query (nlists 0: not split)
    QItem* items;                    // [item_cap] sorted work items (pass 2)
    uint64_t* item_len;              // [item_cap] scratch of pass 2
    QItem* unsorted;                 // [item_cap] scratch of pass 2
    size_t item_cap;
    uint32_t* bucket;                // [plan_buckets() + 1] counting-sort scratch
};
struct PlanInfo {
    uint32_t n_flat = 0, n_items = 0, n_tmp = 0;
    uint64_t n_wide_words = 0;  // cursor storage of the items with more than SCORE_REG_LISTS lists
    uint32_t n_by_t[SCORE_FAST_T + 1] = {};
};
size_t plan_buckets() { return (size_t)(score_fast_max_terms() + 1) * PLAN_PER_GROUP; }

// max_tmp_entries: capacity of the temporary slice lists in entries (lists x k); a plan that would
// need more is made without splitting.  Returns 0 or an error code (message set).
int plan_pass1(sebm25_index* ix, bool news, const uint32_t* q_terms, const uint64_t* q_offsets, uint32_t n_queries,
               uint32_t k, PlanStore& st, PlanInfo& info, uint64_t max_tmp_entries, uint32_t max_query_terms = MAX_QUERY_TERMS) {
    const uint32_t n_units = news ? ix->n_articles : ix->n_docs;
    const uint32_t* lens = ix->h_list_len.data() + (news ? 1 : 0);
    auto list_len = [&](uint32_t t) -> uint64_t { return lens[2 * (size_t)t]; };
    const uint32_t fast_t = ix->big_weights ? 0u : score_fast_max_terms();
    size_t nf = 0;
    uint64_t total_len = 0;
    // the list length of a term is a read at a random place of an array of 2 V entries: ask for those of the terms a
    // few queries ahead while this query is planned
    const uint64_t n_q_terms = q_offsets[n_queries];
    constexpr uint64_t AHEAD = 24;
    std::unordered_set<uint32_t> seen;   // only touched by queries of 64 and more distinct terms
    uint32_t seen_q = 0xFFFFFFFFu;
    for (uint32_t q = 0; q < n_queries; ++q) {
        if (q_offsets[q + 1] < q_offsets[q]) return fail(ix, SEBM25_E_INVALID, "search: q_offsets must be non-decreasing");
        for (uint64_t j = q_offsets[q] + AHEAD, e = std::min(q_offsets[q + 1] + AHEAD, n_q_terms); j < e; ++j) {
            const uint32_t t = q_terms[j];
            if (t < ix->V) __builtin_prefetch(&lens[2 * (size_t)t]);
        }
        const size_t start = nf;
        uint64_t len = 0;
        for (uint64_t j = q_offsets[q]; j < q_offsets[q + 1]; ++j) {
            const uint32_t t = q_terms[j];
            if (t >= ix->V || list_len(t) == 0) continue;  // matches no row
            bool dup = false;
            if (nf - start < 64) {
                for (size_t x = start; x < nf; ++x) dup |= st.flat[x] == t;
            } else {  // a long query: the terms seen so far in a set
                if (seen_q != q) {
                    seen.clear();
                    seen.insert(st.flat + start, st.flat + nf);
                    seen_q = q;
                }
                dup = !seen.insert(t).second;
            }
            if (dup) continue;  // `= ANY($1)`: a repeated term selects its rows once
            if (nf >= st.flat_cap) return fail(ix, SEBM25_E_RANGE, "search: too many query terms for the plan storage");
            st.flat[nf++] = t;
            len += list_len(t);
        }
        if (nf - start > max_query_terms)
            return fail(ix, SEBM25_E_RANGE, "search: more than " + std::to_string(max_query_terms) + " distinct matching terms in one query");
        // Stepwise kernels read the lists shortest first and treat the longest one specially; the sum of a
        // score still runs in the order the caller gave.  order = 3 bits per given position: where that term
        // now sits (identity for the general kernel, which keeps the given order).
        uint32_t order = 0xFAC688u;  // position j -> place j
        const size_t nt = nf - start;
        if (nt >= 2 && nt <= fast_t) {
            uint32_t given[SCORE_FAST_T], phys_of[SCORE_FAST_T];
            for (size_t j = 0; j < nt; ++j) given[j] = st.flat[start + j];
            uint32_t idx[SCORE_FAST_T];
            for (size_t j = 0; j < nt; ++j) idx[j] = (uint32_t)j;
            for (size_t a = 1; a < nt; ++a)  // stable insertion sort on the list length
                for (size_t b = a; b > 0 && list_len(given[idx[b]]) < list_len(given[idx[b - 1]]); --b) std::swap(idx[b], idx[b - 1]);
            order = 0;
            for (size_t p = 0; p < nt; ++p) {
                st.flat[start + p] = given[idx[p]];
                phys_of[idx[p]] = (uint32_t)p;
            }
            for (size_t j = 0; j < nt; ++j) order |= phys_of[j] << (3 * j);
        }
        st.q_toff[q] = (uint32_t)start;
        st.q_nt[q] = (uint32_t)nt;
        st.q_order[q] = order;
        st.q_len[q] = len;
        total_len += len;
    }
    info.n_flat = (uint32_t)nf;
    // postings per work item: enough items to fill the GPU, few enough that per-item set-up
    // (binary searches, final cut) stays small next to the streaming
    static const uint64_t slice_max = [] {
        const char* e = getenv("SEBM25_EXP_SLICE_MAX");  // experiments only
        return e ? (uint64_t)strtoull(e, nullptr, 10) : 524288ull;
    }();
    uint64_t slice = std::min<uint64_t>(std::max<uint64_t>(total_len / 8192, 8192), slice_max);
    slice = std::max<uint64_t>(slice, (uint64_t)k * 32);
    for (int attempt = 0; attempt < 2; ++attempt) {
        uint64_t n_items = 0, n_tmp = 0;
        for (uint32_t q = 0; q < n_queries; ++q) {
            uint32_t ns = 1;
            if (attempt == 0) {
                ns = (uint32_t)std::min<uint64_t>((st.q_len[q] + slice - 1) / slice, 64);
                ns = std::max(1u, std::min(ns, std::max(1u, n_units / 4096u)));
            }
            st.q_ns[q] = ns;
            n_items += ns;
            if (ns > 1) n_tmp += ns;
        }
        info.n_items = (uint32_t)n_items;
        info.n_tmp = (uint32_t)n_tmp;
        if (n_tmp * k <= max_tmp_entries) break;  // else: plan again without splitting
    }
    return SEBM25_OK;
}

// tmp_base: when given, every item writes a temporary list, query q's lists of this plan being tmp_base[q], tmp_base[q] + 1, ...
// (several index segments scored for one batch); first / nlists are then the caller's business
int plan_pass2(sebm25_index* ix, bool news, uint32_t n_queries, PlanStore& st, PlanInfo& info, const uint32_t* tmp_base = nullptr) {
    if (info.n_items > st.item_cap) return fail(ix, SEBM25_E_RANGE, "search: too many work items for the plan storage");
    const uint32_t n_units = news ? ix->n_articles : ix->n_docs;
    const uint32_t fast_t = ix->big_weights ? 0u : score_fast_max_terms();
    const uint32_t n_groups = score_fast_max_terms() + 1;
    uint32_t n = 0, n_tmp = 0;
    info.n_wide_words = 0;
    for (uint32_t q = 0; q < n_queries; ++q) {
        const uint32_t ns = st.q_ns[q];
        if (!tmp_base) {
            st.first[q] = ns > 1 ? n_tmp : 0u;
            st.nlists[q] = ns > 1 ? ns : 0u;
        }
        for (uint32_t sl = 0; sl < ns; ++sl) {
            QItem it{};
            it.q = q;
            it.term_off = st.q_toff[q];
            it.n_terms = st.q_nt[q];
            it.order = st.q_order[q];
            if (it.n_terms > SCORE_REG_LISTS) {  // cursor and end of every list, in device memory
                it.order = (uint32_t)info.n_wide_words;
                info.n_wide_words += 2ull * it.n_terms;
            }
            it.unit_lo = (uint32_t)((uint64_t)n_units * sl / ns);
            it.unit_hi = sl + 1 == ns ? 0xFFFFFFFFu : (uint32_t)((uint64_t)n_units * (sl + 1) / ns);
            it.direct = !tmp_base && ns == 1;
            it.out_slot = tmp_base ? tmp_base[q] + sl : (ns == 1 ? q : n_tmp++);
            st.unsorted[n] = it;
            st.item_len[n] = st.q_len[q] / ns;
            ++n;
        }
    }
    // a query without a matching term still needs its (empty) result row written: the general kernel does that
    auto group_of = [&](uint32_t i) -> uint32_t {
        const uint32_t nt = st.unsorted[i].n_terms;
        return nt >= 1u && nt <= fast_t ? nt - 1u : n_groups - 1u;
    };
    // counting sort on (group, length's bit count and its next four bits): order within 6 %, linear time
    auto bucket_of = [&](uint32_t i) -> uint32_t {
        const uint64_t len = st.item_len[i];
        const uint32_t bits = len ? 64u - (uint32_t)__builtin_clzll(len) : 0u;  // 0..64
        const uint32_t frac = bits > 4 ? (uint32_t)(len >> (bits - 5)) & (PLAN_SUB - 1) : (uint32_t)len & (PLAN_SUB - 1);
        const uint32_t key = bits * PLAN_SUB + frac;  // grows with the length
        return group_of(i) * PLAN_PER_GROUP + (PLAN_PER_GROUP - 1u - key);
    };
    const size_t nb = plan_buckets();
    for (size_t b = 0; b <= nb; ++b) st.bucket[b] = 0;
    for (uint32_t g = 0; g <= SCORE_FAST_T; ++g) info.n_by_t[g] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        ++st.bucket[bucket_of(i) + 1];
        ++info.n_by_t[group_of(i)];
    }
    for (size_t b = 0; b < nb; ++b) st.bucket[b + 1] += st.bucket[b];
    for (uint32_t i = 0; i < n; ++i) st.items[st.bucket[bucket_of(i)]++] = st.unsorted[i];
    return SEBM25_OK;
}

struct SearchBufs {  // device memory of one search call
    QItem* items;
    uint32_t *terms, *first, *nlists;
    uint32_t *t_unit, *t_mat, *t_tie, *t_count;
    double* t_score;
    uint32_t* wide = nullptr;   // cursors of the items with more than SCORE_REG_LISTS lists (info.n_wide_words words)
};

// kernel arguments of one segment's scoring launch: statistics, url ranks and unit bases are the handle's (ix), lists and
// slots the segment's (seg; the handle itself is the base segment)
ScoreArgs score_args_of(const sebm25_index* ix, const sebm25_index* seg, bool news, uint32_t k, const PlanInfo& info,
                        const SearchBufs& b, uint32_t* d_unit, double* d_score, uint32_t* d_mat, uint32_t* d_tie,
                        uint32_t* d_count) {
    ScoreArgs a{};
    a.row_ptr = seg->row_ptr.as<uint32_t>();
    a.news_ptr = seg->news_ptr.as<uint32_t>();
    a.post = seg->post;
    a.idf_used = ix->idf_used.as<double>();
    a.url_rank = news ? (ix->has_rank_news ? ix->url_rank_news.as<uint32_t>() : nullptr)
                      : (ix->has_rank_web ? ix->url_rank_web.as<uint32_t>() : nullptr);
    a.rank_base = news ? ix->rank_base_news : ix->rank_base_web;
    a.rank_shift = news ? ix->rank_shift_news : ix->rank_shift_web;
    a.unit_base = news ? ix->news_base : ix->web_base;
    a.news = news ? 1 : 0;
    a.items = b.items;
    a.q_terms = b.terms;
    a.wide_cur = b.wide;
    a.n_items = info.n_items;
    a.k = k;
    a.out_unit = d_unit; a.out_score = d_score; a.out_matched = d_mat; a.out_tie = d_tie; a.out_count = d_count;
    a.tmp_unit = b.t_unit; a.tmp_score = b.t_score; a.tmp_matched = b.t_mat; a.tmp_tie = b.t_tie; a.tmp_count = b.t_count;
    return a;
}

// uploads one segment's plan and launches its scoring kernels; statistics, url ranks and unit bases are the
// handle's (ix), lists and slots the segment's (seg; the handle itself is the base segment)
int score_segment(sebm25_index* ix, const sebm25_index* seg, bool news, uint32_t k, const PlanStore& st, const PlanInfo& info,
                  const SearchBufs& b, uint32_t* d_unit, double* d_score, uint32_t* d_mat, uint32_t* d_tie, uint32_t* d_count) {
    phase_begin(ix, "search_plan_h2d");
    HIP_TRY(ix, hipMemcpyAsync(b.items, st.items, (size_t)info.n_items * sizeof(QItem), hipMemcpyHostToDevice, ix->stream));
    if (info.n_flat)
        HIP_TRY(ix, hipMemcpyAsync(b.terms, st.flat, (size_t)info.n_flat * 4, hipMemcpyHostToDevice, ix->stream));
    phase_end(ix);
    const ScoreArgs a = score_args_of(ix, seg, news, k, info, b, d_unit, d_score, d_mat, d_tie, d_count);
    phase_begin(ix, "score_topk");
    launch_score(a, info.n_by_t, ix->stream);
    phase_end(ix);
    HIP_TRY(ix, hipGetLastError());
    return SEBM25_OK;
}

// launches scoring (+ the merge of split queries) for a finished plan; results land in d_*
int run_search(sebm25_index* ix, bool news, uint32_t n_queries, uint32_t k, const PlanStore& st, const PlanInfo& info,
               const SearchBufs& b, uint32_t* d_unit, double* d_score, uint32_t* d_mat, uint32_t* d_tie, uint32_t* d_count) {
    phase_begin(ix, "search_plan_h2d");
    HIP_TRY(ix, hipMemcpyAsync(b.items, st.items, (size_t)info.n_items * sizeof(QItem), hipMemcpyHostToDevice, ix->stream));
    if (info.n_flat)
        HIP_TRY(ix, hipMemcpyAsync(b.terms, st.flat, (size_t)info.n_flat * 4, hipMemcpyHostToDevice, ix->stream));
    if (info.n_tmp) {
        HIP_TRY(ix, hipMemcpyAsync(b.first, st.first, (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
        HIP_TRY(ix, hipMemcpyAsync(b.nlists, st.nlists, (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
    }
    phase_end(ix);

    const ScoreArgs a = score_args_of(ix, ix, news, k, info, b, d_unit, d_score, d_mat, d_tie, d_count);
    phase_begin(ix, "score_topk");
    launch_score(a, info.n_by_t, ix->stream);
    phase_end(ix);
    if (info.n_tmp) {
        MergeArgs m{};
        m.n_queries = n_queries; m.k = k;
        m.first = b.first; m.n = b.nlists;
        m.in_unit = a.tmp_unit; m.in_score = a.tmp_score; m.in_matched = a.tmp_matched; m.in_tie = a.tmp_tie; m.in_count = a.tmp_count;
        m.list_stride = 1;
        m.out_unit = d_unit; m.out_score = d_score; m.out_matched = d_mat; m.out_tie = d_tie; m.out_count = d_count;
        phase_begin(ix, "merge_slices");
        launch_merge(m, ix->stream);
        phase_end(ix);
    }
    HIP_TRY(ix, hipGetLastError());
    return SEBM25_OK;
}

// plan storage of sebm25_search in a segment's own vectors (grown on demand); items are sized by the caller
void plan_store_of(sebm25_index* seg, const uint64_t* q_offsets, uint32_t n_queries, PlanStore& st) {
    SearchPlanVecs& v = seg->plan_vecs;
    v.flat.resize(std::max<size_t>(q_offsets[n_queries], 1));
    for (std::vector<uint32_t>* x : {&v.q_toff, &v.q_nt, &v.q_order, &v.q_ns, &v.first, &v.nlists}) x->resize(n_queries);
    v.q_len.resize(n_queries);
    v.bucket.resize(plan_buckets() + 1);
    st.flat = v.flat.data(); st.flat_cap = v.flat.size();
    st.q_toff = v.q_toff.data(); st.q_nt = v.q_nt.data(); st.q_order = v.q_order.data(); st.q_ns = v.q_ns.data(); st.q_len = v.q_len.data();
    st.first = v.first.data(); st.nlists = v.nlists.data(); st.bucket = v.bucket.data();
}
void plan_items_of(sebm25_index* seg, uint32_t n_items, PlanStore& st) {
    SearchPlanVecs& v = seg->plan_vecs;
    v.items.resize(n_items); v.unsorted.resize(n_items); v.item_len.resize(n_items);
    st.items = v.items.data(); st.unsorted = v.unsorted.data(); st.item_len = v.item_len.data(); st.item_cap = n_items;
}

// A batch over an index that has update segments: every segment is scored on its own (each unit's live rows sit
// in exactly one segment, so a segment's scores are complete), every work item writes a temporary list, and one
// merge per query cuts the lists of all segments to the best k.
int search_segments(sebm25_index* ix, bool news, const uint32_t* q_terms, const uint64_t* q_offsets, uint32_t n_queries,
                    uint32_t k, uint32_t* d_unit, double* d_score, uint32_t* d_mat, uint32_t* d_tie, uint32_t* d_count) {
    std::vector<sebm25_index*> segs{ix};
    segs.insert(segs.end(), ix->deltas.begin(), ix->deltas.end());
    const size_t S = segs.size();
    std::vector<PlanStore> st(S);
    std::vector<PlanInfo> info(S);
    std::vector<std::vector<uint32_t>> tmp_base(S, std::vector<uint32_t>(n_queries));
    std::vector<uint32_t> first(n_queries), nlists(n_queries);
    int rc;
    // The base segment is planned first and its kernels launched; the update segments are planned while those run.
    // An update segment is small: its queries are never cut into unit ranges, so every query has exactly one list
    // per update segment and the places of all lists are known once the base segment is planned.
    auto plan_first_pass = [&](size_t s, uint64_t max_tmp_entries) -> int {
        sebm25_index* seg = segs[s];
        seg->stream = ix->stream;
        int r;
        if ((r = refresh_host_ptrs(seg)) || (plan_store_of(seg, q_offsets, n_queries, st[s]), false) ||
            (r = plan_pass1(seg, news, q_terms, q_offsets, n_queries, k, st[s], info[s], max_tmp_entries))) {
            if (seg != ix) ix->err = seg->err;
            return r;
        }
        return SEBM25_OK;
    };
    auto plan_and_launch = [&](size_t s) -> int {
        sebm25_index* seg = segs[s];
        plan_items_of(seg, info[s].n_items, st[s]);
        int r;
        if ((r = plan_pass2(seg, news, n_queries, st[s], info[s], tmp_base[s].data()))) {
            if (seg != ix) ix->err = seg->err;
            return r;
        }
        HIP_TRY(ix, seg->q_items.ensure((size_t)std::max(info[s].n_items, 1u) * sizeof(QItem)));
        HIP_TRY(ix, seg->q_terms.ensure(std::max<size_t>(info[s].n_flat, 1) * 4));
        HIP_TRY(ix, seg->q_wide.ensure(std::max<size_t>(info[s].n_wide_words, 1) * 4));
        SearchBufs b{seg->q_items.as<QItem>(), seg->q_terms.as<uint32_t>(), nullptr, nullptr,
                     ix->t_unit.as<uint32_t>(), ix->t_mat.as<uint32_t>(), ix->t_tie.as<uint32_t>(), ix->t_count.as<uint32_t>(),
                     ix->t_score.as<double>(), seg->q_wide.as<uint32_t>()};
        return score_segment(ix, seg, news, k, st[s], info[s], b, d_unit, d_score, d_mat, d_tie, d_count);
    };
    if ((rc = plan_first_pass(0, ~0ull))) return rc;
    uint32_t n_tmp = 0;
    for (uint32_t q = 0; q < n_queries; ++q) {
        first[q] = n_tmp;
        tmp_base[0][q] = n_tmp;
        n_tmp += st[0].q_ns[q];
        for (size_t s = 1; s < S; ++s) tmp_base[s][q] = n_tmp++;
        nlists[q] = n_tmp - first[q];
    }
    const size_t tk = (size_t)std::max(n_tmp, 1u) * k;
    HIP_TRY(ix, ix->t_unit.ensure(tk * 4));
    HIP_TRY(ix, ix->t_score.ensure(tk * 8));
    HIP_TRY(ix, ix->t_mat.ensure(tk * 4));
    HIP_TRY(ix, ix->t_tie.ensure(tk * 4));
    HIP_TRY(ix, ix->t_count.ensure((size_t)std::max(n_tmp, 1u) * 4));
    HIP_TRY(ix, ix->q_first.ensure((size_t)n_queries * 4));
    HIP_TRY(ix, ix->q_n.ensure((size_t)n_queries * 4));
    if ((rc = plan_and_launch(0))) return rc;
    for (size_t s = 1; s < S; ++s) {
        if ((rc = plan_first_pass(s, 0))) return rc;   // 0 entries for unit-range lists: one work item per query
        if ((rc = plan_and_launch(s))) return rc;
    }
    HIP_TRY(ix, hipMemcpyAsync(ix->q_first.p, first.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(ix, hipMemcpyAsync(ix->q_n.p, nlists.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
    MergeArgs m{};
    m.n_queries = n_queries; m.k = k;
    m.first = ix->q_first.as<uint32_t>(); m.n = ix->q_n.as<uint32_t>();
    m.in_unit = ix->t_unit.as<uint32_t>(); m.in_score = ix->t_score.as<double>(); m.in_matched = ix->t_mat.as<uint32_t>();
    m.in_tie = ix->t_tie.as<uint32_t>(); m.in_count = ix->t_count.as<uint32_t>();
    m.list_stride = 1;
    m.out_unit = d_unit; m.out_score = d_score; m.out_matched = d_mat; m.out_tie = d_tie; m.out_count = d_count;
    phase_begin(ix, "merge_slices");
    launch_merge(m, ix->stream);
    phase_end(ix);
    HIP_TRY(ix, hipGetLastError());
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));  // the host vectors above were the source of asynchronous copies
    return SEBM25_OK;
}

int check_search_args(sebm25_index* ix, int source, const uint32_t* q_terms, const uint64_t* q_offsets, uint32_t n_queries,
                      uint32_t k, uint32_t max_k, const uint32_t* out_unit, const double* out_score,
                      const uint32_t* out_matched, const uint32_t* out_count) {
    if (!ix->built || !ix->finalized) return fail(ix, SEBM25_E_STATE, "search: index is not built and finalized");
    if (source != (int)SEBM25_SRC_WEB && source != (int)SEBM25_SRC_NEWS)
        return fail(ix, SEBM25_E_INVALID, "search: source must be 1 (web) or 2 (news)");
    if (k == 0 || k > max_k) return fail(ix, SEBM25_E_RANGE, "search: k must be in 1.." + std::to_string(max_k));
    if (n_queries == 0) return SEBM25_OK;
    if (!q_offsets || !out_unit || !out_score || !out_matched || !out_count)
        return fail(ix, SEBM25_E_INVALID, "search: NULL argument");
    if (is_device_ptr(q_offsets) || is_device_ptr(q_terms))
        return fail(ix, SEBM25_E_INVALID, "search: query arrays must be host memory");
    if (q_offsets[n_queries] && !q_terms) return fail(ix, SEBM25_E_INVALID, "search: q_terms is NULL");
    return SEBM25_OK;
}

}  // namespace

extern "C" {

int sebm25_search(sebm25_index* ix, int source, const uint32_t* q_terms, const uint64_t* q_offsets, uint32_t n_queries,
                  uint32_t k, uint32_t* out_unit, double* out_score, uint32_t* out_matched, uint32_t* out_tie,
                  uint32_t* out_count) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "search: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    int rc = check_search_args(ix, source, q_terms, q_offsets, n_queries, k, MAX_K, out_unit, out_score, out_matched, out_count);
    if (rc || n_queries == 0) return rc;
    if ((rc = refresh_host_ptrs(ix))) return rc;
    phases_reset(ix);
    const bool news = source == (int)SEBM25_SRC_NEWS;

    // ---- plan (host storage grown on demand)
    const bool segmented = !ix->deltas.empty();
    PlanStore st{};
    PlanInfo info;
    if (!segmented) {
        plan_store_of(ix, q_offsets, n_queries, st);
        if ((rc = plan_pass1(ix, news, q_terms, q_offsets, n_queries, k, st, info, ~0ull))) return rc;
        plan_items_of(ix, info.n_items, st);
        if ((rc = plan_pass2(ix, news, n_queries, st, info))) return rc;
    }

    // ---- device buffers
    const size_t qk = (size_t)n_queries * k, tk = (size_t)std::max(info.n_tmp, 1u) * k;
    const bool dev_out = is_device_ptr(out_unit);
    if (dev_out != is_device_ptr(out_score) || dev_out != is_device_ptr(out_matched) || dev_out != is_device_ptr(out_count) ||
        (out_tie && dev_out != is_device_ptr(out_tie)))
        return fail(ix, SEBM25_E_INVALID, "search: output arrays must be all host or all device memory");
    HIP_TRY(ix, ix->q_items.ensure((size_t)info.n_items * sizeof(QItem)));
    HIP_TRY(ix, ix->q_terms.ensure(std::max<size_t>(info.n_flat, 1) * 4));
    HIP_TRY(ix, ix->q_first.ensure((size_t)n_queries * 4));
    HIP_TRY(ix, ix->q_n.ensure((size_t)n_queries * 4));
    HIP_TRY(ix, ix->t_unit.ensure(tk * 4));
    HIP_TRY(ix, ix->t_score.ensure(tk * 8));
    HIP_TRY(ix, ix->t_mat.ensure(tk * 4));
    HIP_TRY(ix, ix->t_tie.ensure(tk * 4));
    HIP_TRY(ix, ix->t_count.ensure((size_t)std::max(info.n_tmp, 1u) * 4));
    uint32_t *d_unit = out_unit, *d_mat = out_matched, *d_count = out_count, *d_tie = out_tie;
    double* d_score = out_score;
    // results bound for host arrays, when small: the five arrays sit in one device buffer and come back with one
    // copy into pinned memory (five copies into pageable memory cost more than scoring a single query does)
    const size_t pack_bytes = qk * 8 + qk * 4 * (out_tie ? 3 : 2) + (size_t)n_queries * 4;
    const bool packed = !dev_out && pack_bytes <= ((size_t)256 << 10);
    if (packed) {
        HIP_TRY(ix, ix->o_pack.ensure(pack_bytes));
        if (ix->o_host_cap < pack_bytes) {
            if (ix->o_host) (void)hipHostFree(ix->o_host);
            ix->o_host = nullptr;
            ix->o_host_cap = 0;
            const size_t want = std::max<size_t>(pack_bytes, (size_t)64 << 10);
            HIP_TRY(ix, hipHostMalloc(&ix->o_host, want, hipHostMallocDefault));
            ix->o_host_cap = want;
        }
        unsigned char* p = static_cast<unsigned char*>(ix->o_pack.p);
        d_score = reinterpret_cast<double*>(p);
        d_unit = reinterpret_cast<uint32_t*>(p + qk * 8);
        d_mat = d_unit + qk;
        d_tie = out_tie ? d_mat + qk : nullptr;
        d_count = (out_tie ? d_tie : d_mat) + qk;
    } else if (!dev_out) {
        HIP_TRY(ix, ix->o_unit.ensure(qk * 4));
        HIP_TRY(ix, ix->o_score.ensure(qk * 8));
        HIP_TRY(ix, ix->o_mat.ensure(qk * 4));
        HIP_TRY(ix, ix->o_count.ensure((size_t)n_queries * 4));
        d_unit = ix->o_unit.as<uint32_t>();
        d_score = ix->o_score.as<double>();
        d_mat = ix->o_mat.as<uint32_t>();
        d_count = ix->o_count.as<uint32_t>();
        if (out_tie) {
            HIP_TRY(ix, ix->o_tie.ensure(qk * 4));
            d_tie = ix->o_tie.as<uint32_t>();
        }
    }
    HIP_TRY(ix, ix->q_wide.ensure(std::max<size_t>(info.n_wide_words, 1) * 4));
    SearchBufs b{ix->q_items.as<QItem>(), ix->q_terms.as<uint32_t>(), ix->q_first.as<uint32_t>(), ix->q_n.as<uint32_t>(),
                 ix->t_unit.as<uint32_t>(), ix->t_mat.as<uint32_t>(), ix->t_tie.as<uint32_t>(), ix->t_count.as<uint32_t>(),
                 ix->t_score.as<double>(), ix->q_wide.as<uint32_t>()};
    if (segmented)
        rc = search_segments(ix, news, q_terms, q_offsets, n_queries, k, d_unit, d_score, d_mat, d_tie, d_count);
    else
        rc = run_search(ix, news, n_queries, k, st, info, b, d_unit, d_score, d_mat, d_tie, d_count);
    if (rc) return rc;
    if (packed) {
        HIP_TRY(ix, hipMemcpyAsync(ix->o_host, ix->o_pack.p, pack_bytes, hipMemcpyDeviceToHost, ix->stream));
        HIP_TRY(ix, hipStreamSynchronize(ix->stream));
        const unsigned char* h = static_cast<const unsigned char*>(ix->o_host);
        memcpy(out_score, h, qk * 8);
        memcpy(out_unit, h + qk * 8, qk * 4);
        memcpy(out_matched, h + qk * 12, qk * 4);
        if (out_tie) memcpy(out_tie, h + qk * 16, qk * 4);
        memcpy(out_count, h + qk * (out_tie ? 20 : 16), (size_t)n_queries * 4);
        return SEBM25_OK;
    }
    if (!dev_out) {
        if ((rc = copy_out(ix, out_unit, d_unit, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_score, d_score, qk * 8))) return rc;
        if ((rc = copy_out(ix, out_matched, d_mat, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_tie, d_tie, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_count, d_count, (size_t)n_queries * 4))) return rc;
    }
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_search_noalloc(sebm25_index* ix, int source, const uint32_t* q_terms, const uint64_t* q_offsets,
                          uint32_t n_queries, uint32_t k, uint32_t* out_unit, double* out_score, uint32_t* out_matched,
                          uint32_t* out_tie, uint32_t* out_count) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "search_noalloc: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    int rc = check_search_args(ix, source, q_terms, q_offsets, n_queries, k, NA_MAX_K, out_unit, out_score, out_matched, out_count);
    if (rc || n_queries == 0) return rc;
    if (n_queries > NA_MAX_QUERIES)
        return fail(ix, SEBM25_E_RANGE, "search_noalloc: at most 1024 queries per call (use sebm25_search for larger batches)");
    if (!ix->deltas.empty())
        return fail(ix, SEBM25_E_STATE, "search_noalloc: the index has update segments (call sebm25_compact, or use sebm25_search)");
    if (ix->device >= NA_MAX_DEVICES || !g_na_ptr[ix->device] || !ix->na_host)
        return fail(ix, SEBM25_E_STATE, "search_noalloc: the fixed storage was not set up when the handle was created");
    if (!ix->host_ptrs_valid && (ix->h_row_ptr.size() != (size_t)ix->V + 1 || ix->h_news_ptr.size() != ix->V || ix->h_list_len.size() != (size_t)ix->V * 2))
        return fail(ix, SEBM25_E_STATE, "search_noalloc: host copies of the list bounds are not reserved");
    if ((rc = refresh_host_ptrs(ix))) return rc;  // copies only: the host arrays were sized by the build
    if (!is_device_ptr(out_unit) || !is_device_ptr(out_score) || !is_device_ptr(out_matched) || !is_device_ptr(out_count) ||
        (out_tie && !is_device_ptr(out_tie)))
        return fail(ix, SEBM25_E_INVALID, "search_noalloc: output arrays must be device memory");
    phases_reset(ix);
    const bool news = source == (int)SEBM25_SRC_NEWS;
    NaHost& h = *ix->na_host;
    PlanStore st{};
    st.flat = h.flat; st.flat_cap = NA_MAX_TERMS;
    st.q_toff = h.q_toff; st.q_nt = h.q_nt; st.q_order = h.q_order; st.q_ns = h.q_ns; st.q_len = h.q_len;
    st.first = h.first; st.nlists = h.nlists; st.bucket = h.bucket;
    st.items = h.items; st.unsorted = h.unsorted; st.item_len = h.item_len; st.item_cap = NA_MAX_ITEMS;
    PlanInfo info;
    std::lock_guard<std::mutex> dev_lk(g_na_device_mu[ix->device]);
    // no storage for the cursors of wider queries here: at most SCORE_REG_LISTS distinct matching terms per query
    if ((rc = plan_pass1(ix, news, q_terms, q_offsets, n_queries, k, st, info, NA_TMP_ENTRIES, SCORE_REG_LISTS))) return rc;
    if (info.n_items > NA_MAX_ITEMS) {  // too many unit ranges for the fixed item array: plan without splitting
        for (uint32_t q = 0; q < n_queries; ++q) h.q_ns[q] = 1;
        info.n_items = n_queries;
        info.n_tmp = 0;
    }
    if ((rc = plan_pass2(ix, news, n_queries, st, info))) return rc;
    unsigned char* base = g_na_ptr[ix->device];  // the device address, never the array's name
    SearchBufs b{reinterpret_cast<QItem*>(base + NA_OFF_ITEMS), reinterpret_cast<uint32_t*>(base + NA_OFF_TERMS),
                 reinterpret_cast<uint32_t*>(base + NA_OFF_FIRST), reinterpret_cast<uint32_t*>(base + NA_OFF_NLISTS),
                 reinterpret_cast<uint32_t*>(base + NA_OFF_T_UNIT), reinterpret_cast<uint32_t*>(base + NA_OFF_T_MAT),
                 reinterpret_cast<uint32_t*>(base + NA_OFF_T_TIE), reinterpret_cast<uint32_t*>(base + NA_OFF_T_COUNT),
                 reinterpret_cast<double*>(base + NA_OFF_T_SCORE)};
    rc = run_search(ix, news, n_queries, k, st, info, b, out_unit, out_score, out_matched, out_tie, out_count);
    // also after a failure: what was queued may still use this GPU's shared array and the pinned plan, which the next
    // call (of any handle on this GPU) takes over when the device lock is released
    const hipError_t e_sync = hipStreamSynchronize(ix->stream);
    if (rc) {
        (void)hipGetLastError();
        return rc;
    }
    HIP_TRY(ix, e_sync);
    return SEBM25_OK;
    ABI_CATCH(ix)
}

int sebm25_merge_topk(sebm25_index* ix, uint32_t n_lists, uint32_t n_queries, uint32_t k, const uint32_t* in_unit,
                      const double* in_score, const uint32_t* in_matched, const uint32_t* in_tie, const uint32_t* in_count,
                      uint32_t* out_unit, double* out_score, uint32_t* out_matched, uint32_t* out_tie, uint32_t* out_count) {
    ABI_TRY
    if (!ix) return fail(nullptr, SEBM25_E_INVALID, "merge_topk: handle is NULL");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (k == 0 || k > MAX_K) return fail(ix, SEBM25_E_RANGE, "merge_topk: k must be in 1..2048");
    if (n_queries == 0) return SEBM25_OK;
    if (n_lists == 0 || !in_unit || !in_score || !in_matched || !in_tie || !in_count || !out_unit || !out_score ||
        !out_matched || !out_count)
        return fail(ix, SEBM25_E_INVALID, "merge_topk: NULL argument or no lists");
    const size_t lk_n = (size_t)n_lists * n_queries * k, qk = (size_t)n_queries * k;
    MergeArgs m{};
    int rc;
    ix->h2d.clear();
    if ((rc = stage_in(ix, in_unit, lk_n, ix->m_in[0], &m.in_unit))) return rc;
    if ((rc = stage_in(ix, in_score, lk_n, ix->m_in[1], &m.in_score))) return rc;
    if ((rc = stage_in(ix, in_matched, lk_n, ix->m_in[2], &m.in_matched))) return rc;
    if ((rc = stage_in(ix, in_tie, lk_n, ix->m_in[3], &m.in_tie))) return rc;
    if ((rc = stage_in(ix, in_count, (size_t)n_lists * n_queries, ix->m_in[4], &m.in_count))) return rc;
    if ((rc = flush_h2d(ix))) return rc;
    const bool dev_out = is_device_ptr(out_unit);
    if (dev_out != is_device_ptr(out_score) || dev_out != is_device_ptr(out_matched) || dev_out != is_device_ptr(out_count) ||
        (out_tie && dev_out != is_device_ptr(out_tie)))
        return fail(ix, SEBM25_E_INVALID, "merge_topk: output arrays must be all host or all device memory");
    uint32_t *d_unit = out_unit, *d_mat = out_matched, *d_count = out_count, *d_tie = out_tie;
    double* d_score = out_score;
    if (!dev_out) {
        HIP_TRY(ix, ix->o_unit.ensure(qk * 4));
        HIP_TRY(ix, ix->o_score.ensure(qk * 8));
        HIP_TRY(ix, ix->o_mat.ensure(qk * 4));
        HIP_TRY(ix, ix->o_tie.ensure(qk * 4));
        HIP_TRY(ix, ix->o_count.ensure((size_t)n_queries * 4));
        d_unit = ix->o_unit.as<uint32_t>();
        d_score = ix->o_score.as<double>();
        d_mat = ix->o_mat.as<uint32_t>();
        d_tie = ix->o_tie.as<uint32_t>();
        d_count = ix->o_count.as<uint32_t>();
    }
    // lists of query q: q, q + n_queries, q + 2 n_queries, ...
    std::vector<uint32_t> first(n_queries), nl(n_queries, n_lists);
    for (uint32_t q = 0; q < n_queries; ++q) first[q] = q;
    HIP_TRY(ix, ix->q_first.ensure((size_t)n_queries * 4));
    HIP_TRY(ix, ix->q_n.ensure((size_t)n_queries * 4));
    HIP_TRY(ix, hipMemcpyAsync(ix->q_first.p, first.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(ix, hipMemcpyAsync(ix->q_n.p, nl.data(), (size_t)n_queries * 4, hipMemcpyHostToDevice, ix->stream));
    m.n_queries = n_queries; m.k = k;
    m.first = ix->q_first.as<uint32_t>(); m.n = ix->q_n.as<uint32_t>();
    m.list_stride = n_queries;
    m.out_unit = d_unit; m.out_score = d_score; m.out_matched = d_mat; m.out_tie = d_tie; m.out_count = d_count;
    launch_merge(m, ix->stream);
    HIP_TRY(ix, hipGetLastError());
    if (!dev_out) {
        if ((rc = copy_out(ix, out_unit, d_unit, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_score, d_score, qk * 8))) return rc;
        if ((rc = copy_out(ix, out_matched, d_mat, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_tie, d_tie, qk * 4))) return rc;
        if ((rc = copy_out(ix, out_count, d_count, (size_t)n_queries * 4))) return rc;
    }
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

// ---------------------------------------------------------------------------- corpus -------

int sebm25_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out) {
    ABI_TRY
    if (!cdf_out || n_terms == 0) return fail(nullptr, SEBM25_E_INVALID, "zipf_cdf: NULL output or empty vocabulary");
    host_zipf_cdf(n_terms, zipf_s, cdf_out);
    return SEBM25_OK;
    ABI_CATCH(nullptr)
}

int sebm25_gen_queries(uint64_t seed, uint32_t n_terms, double zipf_s, uint32_t min_rank, uint32_t n_queries,
                       uint32_t terms_per_query, uint32_t* q_terms_out) {
    ABI_TRY
    if (!q_terms_out || n_terms == 0) return fail(nullptr, SEBM25_E_INVALID, "gen_queries: NULL output or empty vocabulary");
    std::vector<uint32_t> cdf(n_terms);
    host_zipf_cdf(n_terms, zipf_s, cdf.data());
    if (host_gen_queries(seed, n_terms, cdf.data(), min_rank, n_queries, terms_per_query, q_terms_out))
        return fail(nullptr, SEBM25_E_RANGE, "gen_queries: vocabulary too small for terms_per_query distinct terms");
    return SEBM25_OK;
    ABI_CATCH(nullptr)
}

int sebm25_gen_corpus(sebm25_index* ix, const sebm25_corpus_spec* spec, uint64_t* n_rows_out, uint32_t* term_id,
                      uint32_t* unit_id, uint32_t* tf, uint8_t* meta, uint32_t* word_count) {
    ABI_TRY
    if (!ix || !spec || !n_rows_out) return fail(ix, SEBM25_E_INVALID, "gen_corpus: NULL argument");
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceGuard g(ix->device);
    if (spec->n_terms == 0 || spec->doc_end < spec->doc_begin || spec->article_end < spec->article_begin ||
        spec->body_max < spec->body_min)
        return fail(ix, SEBM25_E_INVALID, "gen_corpus: inconsistent spec");
    if ((uint64_t)spec->title_len + spec->desc_len + spec->body_max > gen_max_draws())
        return fail(ix, SEBM25_E_RANGE, "gen_corpus: title_len + desc_len + body_max exceeds 1024 draws");
    const uint64_t n_units = (uint64_t)(spec->doc_end - spec->doc_begin) + (spec->article_end - spec->article_begin);
    if (n_units == 0) {
        *n_rows_out = 0;
        return SEBM25_OK;
    }
    if (n_units > 0x7FFFFFFFull) return fail(ix, SEBM25_E_RANGE, "gen_corpus: too many units");
    std::vector<uint32_t> cdf(spec->n_terms);
    host_zipf_cdf(spec->n_terms, spec->zipf_s, cdf.data());
    HIP_TRY(ix, ix->gen_cdf.ensure((size_t)spec->n_terms * 4));
    HIP_TRY(ix, hipMemcpyAsync(ix->gen_cdf.p, cdf.data(), (size_t)spec->n_terms * 4, hipMemcpyHostToDevice, ix->stream));
    HIP_TRY(ix, ix->gen_rows.ensure((n_units + 1) * 4));
    HIP_TRY(ix, ix->scan_tmp.ensure(scan_tmp_words(n_units + 1) * 4));
    uint32_t* unit_rows = ix->gen_rows.as<uint32_t>();
    HIP_TRY(ix, hipMemsetAsync(unit_rows + n_units, 0, 4, ix->stream));
    launch_gen_count(*spec, ix->gen_cdf.as<uint32_t>(), unit_rows, ix->stream);
    // exclusive scan over n_units + 1 words: the last word becomes the row total
    launch_exclusive_scan_u32(unit_rows, n_units + 1, ix->scan_tmp.as<uint32_t>(), ix->stream);
    HIP_TRY(ix, hipGetLastError());
    uint32_t total = 0;
    HIP_TRY(ix, hipMemcpyAsync(&total, unit_rows + n_units, 4, hipMemcpyDeviceToHost, ix->stream));
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    *n_rows_out = total;
    if (!term_id && !unit_id && !tf && !meta) return SEBM25_OK;
    if (!term_id || !unit_id || !tf || !meta) return fail(ix, SEBM25_E_INVALID, "gen_corpus: pass all row arrays or none");
    if (!is_device_ptr(term_id) || !is_device_ptr(unit_id) || !is_device_ptr(tf) || !is_device_ptr(meta) ||
        (word_count && !is_device_ptr(word_count)))
        return fail(ix, SEBM25_E_INVALID, "gen_corpus: output arrays must be device memory");
    launch_gen_emit(*spec, ix->gen_cdf.as<uint32_t>(), unit_rows, term_id, unit_id, tf, meta, word_count, ix->stream);
    HIP_TRY(ix, hipGetLastError());
    HIP_TRY(ix, hipStreamSynchronize(ix->stream));
    return SEBM25_OK;
    ABI_CATCH(ix)
}

}  // extern "C"
