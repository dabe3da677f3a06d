// Internal declarations shared by the host side (abi.hip) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sebm25.h"

namespace sebm25 {

// ------------------------------------------------------------------ packed sort row ------
// One token row as a 64-bit word, most significant field first:
//   [ term : Tb ][ src : 1 ][ unit : Ub ][ fc : 2 ][ tf : Fb ]      Tb + Ub + Fb = 61
// src: 0 web, 1 news.  fc: 0 = weight 1.0, 1 = title (3.2), 2 = description (1.7), 3 = row of a web doc
// whose status is not 'done' (counts toward df, yields no posting).
// Sorting on the top Tb + 1 + Ub bits orders rows by (term, source, unit).
struct RowFmt {
    int term_bits, unit_bits, tf_bits;
    __host__ __device__ int term_shift() const { return 64 - term_bits; }
    __host__ __device__ int src_shift() const { return 63 - term_bits; }
    __host__ __device__ int unit_shift() const { return tf_bits + 2; }
    __host__ __device__ int key_shift() const { return tf_bits + 2; }
    __host__ __device__ uint64_t pack(uint32_t term, uint32_t src, uint32_t unit, uint32_t fc, uint64_t tf) const {
        return ((uint64_t)term << term_shift()) | ((uint64_t)src << src_shift()) | ((uint64_t)unit << unit_shift()) |
               ((uint64_t)fc << tf_bits) | tf;
    }
    __host__ __device__ uint32_t term(uint64_t r) const { return (uint32_t)(r >> term_shift()); }
    __host__ __device__ uint32_t src(uint64_t r) const { return (uint32_t)(r >> src_shift()) & 1u; }
    __host__ __device__ uint32_t unit(uint64_t r) const {
        return (uint32_t)(r >> unit_shift()) & (uint32_t)((1ull << unit_bits) - 1ull);
    }
    __host__ __device__ uint32_t fc(uint64_t r) const { return (uint32_t)(r >> tf_bits) & 3u; }
    __host__ __device__ uint64_t tf(uint64_t r) const { return r & ((1ull << tf_bits) - 1ull); }
    __host__ __device__ uint64_t key(uint64_t r) const { return r >> key_shift(); }
};

// weight of a posting slot that holds no posting (a merged or filtered row); real weights are smaller
constexpr uint32_t POST_HOLE = 0xFFFFFFFFu;
// One posting slot per sorted row: x = the row's unit, y = the posting's weight in tenths at the head of a kept
// (term, source, unit) run, POST_HOLE elsewhere.  Unit and weight sit side by side so that the pass that makes the
// slots writes each with one 8-byte store and the scoring kernels read each with one 8-byte load.
using Slot = uint2;

// device error bits raised by kernels
// DEV_BIG_WEIGHT: some posting weight does not fit 16 bits (weight + 1 > 0xFFFF): scoring uses the general kernel
enum : uint32_t { DEV_ERR_RANGE = 1u, DEV_ERR_ORDER = 2u, DEV_NEED_FIXUP = 4u, DEV_BIG_WEIGHT = 8u };

struct RawRows {
    const uint32_t* term;
    const uint32_t* unit;
    const uint32_t* tf;
    const uint8_t* meta;
};

// grow-only device allocation
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 4) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            e = hipMalloc(&p, bytes);
            want = bytes;
        }
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

struct PhaseTimer {
    std::vector<const char*> names;
    std::vector<hipEvent_t> ev;  // 2 per phase
    std::vector<float> ms;
    bool enabled = false;
    size_t used = 0;
};

// sort tile geometry (shared by host planning and kernels)
constexpr int SORT_THREADS = 512;
constexpr int SORT_ITEMS = 8;
constexpr int SORT_WAVES = SORT_THREADS / 64;
constexpr int SORT_TILE = SORT_THREADS * SORT_ITEMS;
constexpr int SORT_MAX_BITS = 10;
constexpr int SORT_MAX_BINS = 1 << SORT_MAX_BITS;
constexpr int SORT_GROUP = 16;        // chunks per histogram workgroup

constexpr int RED_THREADS = 256;
constexpr int RED_ROUNDS = 4;                 // 64-row rounds a wave loads before it processes them
constexpr int RED_BATCH = 64 * RED_ROUNDS;
constexpr int POST_PAD = 4 * RED_BATCH;       // slack entries behind each posting slot array and the sorted rows
constexpr uint64_t MAX_ROWS = 0xFFFFF000ull;  // row indices plus a few batches of look-ahead stay 32-bit

struct BuildCtx {
    RowFmt fmt;
    uint32_t n_terms, n_docs, n_articles;
    uint32_t check_order;   // rows declared unit-major: verify
    uint32_t* doc_len;      // [n_docs] accumulators (zeroed)
    const uint8_t* doc_flags;  // [n_docs] SEBM25_DOC_* bits or null (every doc 'done'); read by the first sort pass
    uint32_t* err;          // device error word
};

// ---- launchers implemented in the .hip files (all asynchronous on `st`) ----
// sort_mode: 0 raw rows, digit taken from term_id alone; 1 raw rows, digit from the packed
// key; 2 packed rows.
void launch_sort_hist(int mode, RawRows raw, const uint64_t* packed, uint64_t R, uint64_t chunk_rows, uint32_t G,
                      int shift, int bits, const BuildCtx& ctx, unsigned short* prefix /* [G][bins] */,
                      uint32_t* group_tot /* [groups][bins] */, hipStream_t st);
size_t sort_offsets_tmp_words();
void launch_sort_offsets(uint32_t* group_tot, uint32_t n_groups, int bits, uint32_t* bin_tot /* [bins] */,
                         uint32_t* seg_tmp /* [sort_offsets_tmp_words()] */, hipStream_t st);
void launch_sort_scatter(int mode, RawRows raw, const uint64_t* packed, uint64_t* dst, uint64_t R, uint32_t G, int shift,
                         int bits, const BuildCtx& ctx, const unsigned short* prefix, const uint32_t* group_off,
                         hipStream_t st);
// the last pass of a two-pass sort on the term bits of unit-major rows, writing posting slots instead of rows:
// post [R + POST_PAD], news_ptr preset to 0xFFFFFFFF, dup zeroed (per term: web rows that are not
// run heads), totals [0] kept, [1] web heads, [5] kept web (zeroed), tmp scatter_post_tmp_words(G) words
void launch_sort_scatter_post(const uint64_t* packed, uint64_t R, uint32_t G, int shift, int bits, const BuildCtx& ctx,
                              const unsigned short* prefix, const uint32_t* group_off, Slot* post,
                              uint32_t* news_ptr, uint32_t* dup, uint32_t* tmp, uint64_t* totals, hipStream_t st);
size_t scatter_post_tmp_words(uint32_t G);
// row_ptr[0 .. n_terms] from the offsets of both passes (bin_start1: first input row of every low digit of the
// last pass's input `rows`; prefix / group_off / bin_start2: the last pass's)
void launch_first_slots(const uint64_t* rows, uint64_t R, uint32_t G, int shift, int bits1, int bits2, const uint32_t* bin_start1,
                        const unsigned short* prefix, const uint32_t* group_off, const uint32_t* bin_start2, uint32_t n_terms,
                        uint32_t* row_ptr, Slot* pad /* POST_PAD slots behind the last one <- holes */, hipStream_t st);
void launch_term_stats_dup(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows, uint32_t* df_dup,
                           hipStream_t st);
// in-place exclusive scan of n uint32 (tmp: at least scan_tmp_words(n) uint32)
size_t scan_tmp_words(size_t n);
void launch_exclusive_scan_u32(uint32_t* data, size_t n, uint32_t* tmp, hipStream_t st);

void launch_postings(const uint64_t* rows, uint64_t R, uint64_t chunk_rows, uint32_t n_chunks, RowFmt fmt, uint32_t n_terms,
                     Slot* post, uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* df /* zeroed */,
                     uint64_t* totals /* [0] kept, [1] web heads, [5] kept web; zeroed */,
                     uint32_t* tmp /* postings_tmp_words(n_chunks) words, 16-byte aligned */, uint32_t* err, hipStream_t st);
size_t postings_tmp_words(uint32_t n_chunks);
// fills row_ptr entries of terms without rows (0xFFFFFFFF) and sets row_ptr[n_terms] = R
size_t row_ptr_fill_tmp_words(uint32_t n_terms);
void launch_row_ptr_fill(uint32_t* row_ptr, uint32_t n_terms, uint64_t R, uint32_t* tmp, hipStream_t st);
void launch_postings_fixup(const uint64_t* rows, uint64_t R, RowFmt fmt, Slot* post, uint32_t* err, hipStream_t st);
void launch_term_stats(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows,
                       hipStream_t st);
void launch_doc_stats(const uint32_t* word_count, const uint32_t* doc_len, const uint8_t* doc_flags,
                      uint32_t n_docs, uint64_t* totals3 /* n_done, n_wc, sum_wc (zeroed) */, hipStream_t st);
// N from the host, or (n_done_dev != nullptr) from device memory: `n_done or 1`
void launch_idf(const uint32_t* df, uint32_t n_terms, uint64_t N, const uint64_t* n_done_dev, double* idf, double* idf_used,
                hipStream_t st);
void launch_idf_override(const double* idf, const uint8_t* present, uint32_t n_terms, double* idf_used,
                         hipStream_t st);

// incremental update (update_kernels.hip)
void launch_bitmap_set(const uint32_t* ids, uint32_t n_ids, uint32_t n, uint32_t shift, uint32_t* bm, hipStream_t st);
void launch_bitmap_or(uint32_t* dst, const uint32_t* src, uint32_t n_words, hipStream_t st);
void launch_tombstone(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* chg_web, const uint32_t* chg_news,
                      const uint32_t* coarse, uint32_t cw, uint32_t c_words, uint32_t shift, const uint32_t* dead_web, const uint32_t* dead_news, Slot* post, uint32_t* df,
                      uint32_t* term_rows, unsigned long long* counts /* [4] */, hipStream_t st);
void launch_check_listed(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* chg_web, const uint32_t* chg_news,
                         uint32_t* err, hipStream_t st);
void launch_doc_len_merge(uint32_t* doc_len, const uint32_t* new_len, const uint32_t* chg_web, uint32_t n_docs, hipStream_t st);
void launch_add_u32(uint32_t* dst, const uint32_t* src, uint32_t n, hipStream_t st);
// compaction by merging: live rows per tile of base rows, insertion points of the (sorted) update rows, the merge
uint32_t merge_tiles_for(uint64_t R);
void launch_live_tile_counts(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* dead_web, const uint32_t* dead_news,
                             uint32_t* counts, hipStream_t st);
void launch_insert_points(const uint64_t* upd, uint64_t n_upd, RowFmt ufmt, const uint64_t* base, uint64_t R, RowFmt bfmt, uint32_t* ip,
                          hipStream_t st);
void launch_merge_tiles(const uint64_t* base, uint64_t R, RowFmt bfmt, const uint32_t* dead_web, const uint32_t* dead_news,
                        const uint32_t* tile_off, const uint64_t* upd, uint64_t n_upd, const uint32_t* ip, RowFmt to, uint64_t* out,
                        hipStream_t st);
void launch_gather_live(const uint64_t* rows, uint64_t R, RowFmt from, RowFmt to, const uint32_t* dead_web,
                        const uint32_t* dead_news, uint64_t* out, unsigned long long* n_out, hipStream_t st);

// scoring
constexpr uint32_t SCORE_FAST_T = 8;  // queries of up to this many lists take the stepwise kernels
struct QItem {
    uint32_t q;         // query index
    uint32_t term_off;  // into the flattened term list
    uint32_t n_terms;   // 0..255
    uint32_t unit_lo, unit_hi;
    uint32_t out_slot;  // result list index (direct output row or temporary slice list)
    uint32_t direct;    // 1: writes the final result row
    uint32_t order;     // stepwise kernels: 3 bits per term position as the caller gave it = its place in the list;
                        // items of more than SCORE_REG_LISTS lists: offset (words) of the item's cursors in wide_cur
};
constexpr uint32_t SCORE_REG_LISTS = 256;  // lists whose cursors the general kernel keeps in registers
struct ScoreArgs {
    const uint32_t* row_ptr;
    const uint32_t* news_ptr;
    const Slot* post;  // unit and weight (tenths) of every slot
    const double* idf_used;
    const uint32_t* url_rank;  // may be null
    // tie-break key of unit u when url_rank is set: u < rank_base ? (u + 1) << rank_shift : url_rank[u - rank_base]
    // (rank_base = 0: url_rank holds a key for every unit)
    uint32_t rank_base, rank_shift;
    uint32_t unit_base;
    int news;  // 0 web, 1 news
    const QItem* items;
    const uint32_t* q_terms;  // flattened, deduped, valid
    uint32_t* wide_cur;       // cursor and end of every list of the items with more than SCORE_REG_LISTS lists
    uint32_t n_items;
    uint32_t k;
    // outputs: final rows [n_queries][k] and temporary slice lists [n_slices][k]
    uint32_t *out_unit, *out_matched, *out_tie, *out_count;
    double* out_score;
    uint32_t *tmp_unit, *tmp_matched, *tmp_tie, *tmp_count;
    double* tmp_score;
};
// n_by_t[i], i < score_fast_max_terms(): items with i + 1 lists; n_by_t[score_fast_max_terms()]: the rest
void launch_score(const ScoreArgs& a, const uint32_t* n_by_t, hipStream_t st);
uint32_t score_fast_max_terms();
struct MergeArgs {
    uint32_t n_queries, k;
    // per query: lists [first[q], first[q] + n[q]) of the input set; n[q] == 0 -> leave output alone
    const uint32_t* first;
    const uint32_t* n;
    const uint32_t *in_unit, *in_matched, *in_tie, *in_count;
    const double* in_score;
    uint32_t list_stride;  // distance (in lists) between consecutive lists of one query
    uint32_t *out_unit, *out_matched, *out_tie, *out_count;
    double* out_score;
};
void launch_merge(const MergeArgs& a, hipStream_t st);

// synthetic corpus
uint32_t gen_max_draws();
void launch_gen_count(const sebm25_corpus_spec& spec, const uint32_t* cdf, uint32_t* unit_rows, hipStream_t st);
void launch_gen_emit(const sebm25_corpus_spec& spec, const uint32_t* cdf, const uint32_t* row_off,
                     uint32_t* term_id, uint32_t* unit_id, uint32_t* tf, uint8_t* meta, uint32_t* word_count,
                     hipStream_t st);
void host_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out);
int host_gen_queries(uint64_t seed, uint32_t n_terms, const uint32_t* cdf, uint32_t min_rank, uint32_t n_queries,
                     uint32_t tpq, uint32_t* out);

}  // namespace sebm25
