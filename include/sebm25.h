/*
 * sebm25.h — C ABI of the MI355X (gfx950) BM25 index-build and top-k scoring library.
 *
 * The reference project has no FFI: its hot path is SQL text executed by PostgreSQL from two
 * Python callables.  Each entry point below names the reference statement(s) it replaces
 * (paths are relative to the reference checkout).  A Python host binds these with ctypes; the
 * signatures use plain pointers and sizes only.
 *
 * Conventions
 *  - Every function returns 0 (SEBM25_OK) or a positive SEBM25_E_* code and never throws.
 *    sebm25_last_error() returns a human-readable message for the last failing call on the
 *    handle (or, with a NULL handle, for the last failing call on the calling thread).
 *  - Array arguments may point to host memory or to device memory of the handle's GPU; the
 *    library detects which (hipPointerGetAttributes).  Host arrays are staged through HBM by
 *    the library; device arrays are used in place.
 *  - Calls are blocking: results are complete when the call returns.  A handle may be used
 *    from one thread at a time (calls on one handle are serialised by an internal mutex).
 *  - "unit" means a web document id (documents.id) for source 1 or a dense news-article index
 *    for source 2; the two id spaces are separate.
 */
#ifndef SEBM25_H
#define SEBM25_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sebm25_index sebm25_index;

enum {
    SEBM25_OK = 0,
    SEBM25_E_INVALID = 1,  /* bad argument (NULL pointer, zero vocabulary, unknown flag, ...)   */
    SEBM25_E_RANGE = 2,    /* a value does not fit: term_id >= n_terms, unit >= n_units,        */
                           /* invalid meta byte, tf or ids too wide for the packed row          */
    SEBM25_E_HIP = 3,      /* a HIP runtime call failed                                         */
    SEBM25_E_NOMEM = 4,    /* device or host allocation failed                                  */
    SEBM25_E_STATE = 5,    /* call order violated (e.g. search before build)                    */
    SEBM25_E_ORDER = 6,    /* rows were declared unit-major but are not                         */
    SEBM25_E_INTERNAL = 7
};

/* tokens.source_type values (alembic/versions/0007_unify_news_tokens.py:19-27). */
#define SEBM25_SRC_WEB 1u
#define SEBM25_SRC_NEWS 2u
/* tokens.field values written by the crawler (app/crawler/worker.py:230-234). */
#define SEBM25_FIELD_TITLE 1u
#define SEBM25_FIELD_DESCRIPTION 2u
#define SEBM25_FIELD_BODY 4u
/* meta byte of a row: source_type | field << 2. */
#define SEBM25_META(source_type, field) ((uint8_t)((source_type) | ((field) << 2)))

/* sebm25_rows.flags */
#define SEBM25_ROWS_UNIT_MAJOR 1u /* rows are ordered by (source_type, unit): all web rows first in
                                     ascending doc id, then all news rows in ascending article
                                     index (any order inside one unit).  Lets the build sort on
                                     term_id alone.  Verified on device; a violation returns
                                     SEBM25_E_ORDER.  Without the flag the build sorts on
                                     (term, source, unit). */
/* doc_flags bits */
#define SEBM25_DOC_DONE 1u    /* documents.status = 'done'                                       */
#define SEBM25_DOC_WC_NULL 2u /* documents.word_count IS NULL (skipped by AVG)                   */

/*
 * The `tokens` table as structure-of-arrays (alembic/versions/0001_initial_schema.py:65-74,
 * 0007_unify_news_tokens.py:19-27; writers app/crawler/worker.py:229-239 and
 * app/batch/news_fetcher.py:303-311), plus the two `documents` columns the job reads
 * (0001_initial_schema.py:42,45).  The TEXT term <-> term_id dictionary stays on the host.
 */
typedef struct sebm25_rows {
    const uint32_t* term_id;    /* [n_rows] dense term index, < n_terms                          */
    const uint32_t* unit_id;    /* [n_rows] doc id (web rows) or article index (news rows)       */
    const uint32_t* tf;         /* [n_rows] tokens.frequency                                     */
    const uint8_t* meta;        /* [n_rows] SEBM25_META(source_type, field)                      */
    uint64_t n_rows;            /* R, at most 2^32 - 4096                                        */
    uint32_t n_terms;           /* V >= 1                                                        */
    uint32_t n_docs;            /* size of the web doc id space (rows of `documents`)            */
    uint32_t n_articles;        /* size of the news article index space                          */
    const uint32_t* word_count; /* [n_docs] documents.word_count, or NULL: use the per-doc sum
                                   of tf over body rows computed by the build                    */
    const uint8_t* doc_flags;   /* [n_docs] SEBM25_DOC_* bits, or NULL: every doc is 'done' and
                                   has a word_count                                              */
    uint32_t flags;             /* SEBM25_ROWS_*                                                 */
} sebm25_rows;

/* Local (this shard's) counts produced by sebm25_build_local and consumed by sebm25_finalize.
 * All fields are plain sums over documents, so shards combine by addition. */
typedef struct sebm25_doc_totals {
    uint64_t n_done;    /* COUNT(*) WHERE status='done'            (app/batch/bm25_stats.py:12)  */
    uint64_t n_wc;      /* done docs with non-NULL word_count      (denominator of AVG, :9)      */
    uint64_t sum_wc;    /* SUM(word_count) over those docs         (numerator of AVG, :9)        */
    uint64_t n_postings;/* unique (term, unit) pairs kept in this shard's index (web + news)     */
} sebm25_doc_totals;

/* ---- lifetime --------------------------------------------------------------------------- */

/* Creates an empty index bound to GPU `device`.  No counterpart in the reference (its state
 * is the PostgreSQL database reached through app/common/db.py:22-32). */
int sebm25_create(int device, sebm25_index** out);
void sebm25_destroy(sebm25_index* ix);
const char* sebm25_last_error(const sebm25_index* ix);
/* All work of this handle is enqueued on `hip_stream` (a hipStream_t; NULL = the default
 * stream, which is also PyTorch's default on ROCm).  Every call returns after its work on that
 * stream has finished, so its outputs may be read from any stream.  Device arrays passed IN are
 * read on that stream: a caller that fills them on another stream waits for that stream first. */
int sebm25_set_stream(sebm25_index* ix, void* hip_stream);

/* ---- index build: replaces app/batch/bm25_stats.py:9-27 ---------------------------------- */

/*
 * Sorts the rows by (term, source, unit), collapses the up-to-three field rows of one
 * (term, unit) into one posting and builds the CSR inverted index; computes this shard's
 *   df[t]   = COUNT(DISTINCT doc_id) over web rows of t            (bm25_stats.py:20,23-24)
 *   totals  = the three document aggregates of bm25_stats.py:9,12
 *   doc_len = per web doc, sum of tf over its body (field 4) rows.
 * Posting weight (search_service.py:19-27 and :84-87): web  w = sum_f tf_f * W[f],
 * W = {1: 3.2, 2: 1.7, else 1.0}; news w = sum tf.  Stored exactly, in tenths, as a 32-bit
 * integer (32 tf_title + 17 tf_description + 10 tf_other; 10 tf for news): a (term, unit) whose
 * weight exceeds 2^32 - 1 tenths returns SEBM25_E_RANGE.
 * Postings of web docs that are not 'done' are counted in df (bm25_stats.py has no status
 * filter on tokens) but left out of the index (search_service.py:32).
 */
int sebm25_build_local(sebm25_index* ix, const sebm25_rows* rows);

/* Copies this shard's df (uint32[n_terms]) and totals out, e.g. into buffers that a
 * collective then sums across shards. */
int sebm25_local_stats(sebm25_index* ix, uint32_t* df_out, sebm25_doc_totals* totals_out);

/*
 * Completes the statistics from corpus-wide sums (NULL = use this shard's own):
 *   N     = n_done, or 1 when n_done = 0                           (bm25_stats.py:13)
 *   avgdl = sum_wc / n_wc, or 0.0 when n_wc = 0                    (bm25_stats.py:9-10)
 *   idf_t = LN((N - df_t + 0.5) / (df_t + 0.5) + 1)                (bm25_stats.py:21)
 * evaluated in double as log1p((N - df + 0.5) / (df + 0.5)).
 */
int sebm25_finalize(sebm25_index* ix, const uint32_t* df_global, const sebm25_doc_totals* totals_global);

/* sebm25_build_local followed by sebm25_finalize(ix, NULL, NULL): the whole single-GPU job. */
int sebm25_build(sebm25_index* ix, const sebm25_rows* rows);

/*
 * The content of `term_statistics` (0001_initial_schema.py:110-115; INSERT at
 * bm25_stats.py:18-26): per term doc_frequency and idf, plus the scalar N and avg_doc_len that
 * every row repeats.  term_rows[t] = number of input rows of term t; the reference inserts a
 * row only for terms with term_rows > 0.  Any output pointer may be NULL.
 */
int sebm25_get_stats(sebm25_index* ix, uint32_t* df_out, double* idf_out, uint32_t* term_rows_out,
                     uint64_t* n_docs_out, double* avgdl_out);

/* Per web doc sum of tf over body rows (uint32[n_docs]). */
int sebm25_get_doc_len(sebm25_index* ix, uint32_t* doc_len_out);

/*
 * Replaces the idf used for scoring: idf_used[t] = present[t] ? idf[t] : 1.0, which is
 * COALESCE(ts.idf, 1.0) of search_service.py:21,47,86 for a term_statistics table that may
 * lag behind `tokens`.  present == NULL means every term is present.
 */
int sebm25_set_idf(sebm25_index* ix, const double* idf, const uint8_t* present);

/* Tie-break key per unit, ascending = better: the rank of the unit's url in the ORDER BY
 * url ASC collation (search_service.py:38,99).  NULL restores the default, the global unit id.
 * `source` is SEBM25_SRC_WEB or SEBM25_SRC_NEWS; rank has n_docs or n_articles entries. */
int sebm25_set_url_rank(sebm25_index* ix, int source, const uint32_t* rank);

/* The same keys without an array for the units that kept their place in url order (no counterpart in the
 * reference, whose ORDER BY reads the urls): for an index whose first n_base units of `source` were numbered in
 * url order (as the export does at a rebuild) and whose later units were appended by updates.  Unit u < n_base gets
 * the key (u + 1) << shift, unit n_base + i the key keys[i] (host or device array, n_keys = units - n_base entries):
 * a unit appended between the urls of base units j and j + 1 takes a key in ((j + 1) << shift, (j + 2) << shift),
 * one before the first base unit a key in [1, 1 << shift).  Searches then read a key from memory only for the
 * appended units.  (n_base + 1) << shift must fit 32 bits, shift <= 16; SEBM25_E_RANGE otherwise. */
int sebm25_set_url_rank_appended(sebm25_index* ix, int source, uint32_t n_base, uint32_t shift, const uint32_t* keys,
                                 uint32_t n_keys);

/* Added to local unit ids in search results (document-sharded indexes). */
int sebm25_set_unit_base(sebm25_index* ix, uint32_t web_base, uint32_t news_base);

/* ---- incremental update: what the reference does row by row in `tokens` --------------------
 *
 * The crawler replaces a document by deleting its rows and inserting the new ones
 * (app/crawler/worker.py:229-239; news: app/batch/news_fetcher.py:303-311), and the statistics job then
 * recomputes everything from the whole table every BATCH_INTERVAL_S seconds (.env:11,
 * app/batch/bm25_stats.py:15-27).  sebm25_update brings an index that is already built to the state a
 * full build of the changed table would give, reading only the rows of the units that changed:
 *   - `changed_docs` / `changed_articles` list every unit whose rows were replaced, removed or added
 *     (also a doc whose status changed); all rows the index holds for them are dropped;
 *   - `rows` holds the rows these units have now, all fields (nothing for a unit that was removed), with
 *     rows->n_terms / n_docs / n_articles the sizes of the id spaces now (they can only grow) and
 *     rows->word_count / doc_flags the full per-doc arrays as they are now.  A row of a unit that is
 *     not listed fails with SEBM25_E_INVALID and leaves the index as it was.
 * Afterwards df, term_rows, N, avgdl, idf, doc lengths, the postings and every search result equal
 * those of sebm25_build over the final table, bit for bit.  The new rows are held in an update segment beside
 * the base segment (an update folds its rows and the live rows of the update segment that is there into one new
 * update segment, so a search scores two segments at most and merges); sebm25_compact merges the update
 * segment into the base.  url ranks must be set again when an update adds units; an idf override
 * (sebm25_set_idf) is replaced by the recomputed idf.  sebm25_update_local leaves the statistics to a
 * following sebm25_local_stats / sebm25_finalize (document-sharded indexes), as sebm25_build_local does.
 */
int sebm25_update(sebm25_index* ix, const sebm25_rows* rows, const uint32_t* changed_docs, uint32_t n_changed_docs,
                  const uint32_t* changed_articles, uint32_t n_changed_articles);
int sebm25_update_local(sebm25_index* ix, const sebm25_rows* rows, const uint32_t* changed_docs, uint32_t n_changed_docs,
                        const uint32_t* changed_articles, uint32_t n_changed_articles);
/* Makes one segment of the live rows of the base segment and the update segment (both sorted: a merge), on the
 * device (no rows cross the host).  Results do not change.  The statistics stay finalized. */
int sebm25_compact(sebm25_index* ix);

/* Sizes of the built index. */
typedef struct sebm25_index_info {
    uint64_t n_rows;
    uint64_t n_postings;      /* kept postings, web + news */
    uint64_t n_web_postings;
    uint32_t n_terms, n_docs, n_articles;
    uint32_t term_bits, unit_bits, tf_bits; /* packed sort row layout of the base segment */
    uint32_t sort_passes;
    uint32_t n_segments;      /* 1, or 2 while an update segment is held (since the last build or compaction) */
    uint64_t n_dead_rows;     /* rows of replaced or removed units that the segments still carry */
    uint64_t n_update_rows;   /* rows of the update segment: every update sorts them again, so compact before they are many */
    uint32_t fused_postings;  /* 1: the base segment's posting slots were written by the last sort pass (unit-major rows,
                                 1025 .. 2^20 terms, no run the pass leaves to the separate posting pass); 0: by the
                                 posting pass over the sorted rows.  The results are the same. */
} sebm25_index_info;
int sebm25_get_info(sebm25_index* ix, sebm25_index_info* out);

/*
 * Copies the CSR out (any pointer may be NULL): for term t, web postings are
 * [row_ptr[t], news_ptr[t]) and news postings [news_ptr[t], row_ptr[t+1]), each in ascending
 * unit order.  row_ptr has n_terms + 1 entries, news_ptr n_terms, post_* n_postings;
 * post_weight_tenths is the posting weight times ten (an exact integer).  row_ptr and news_ptr
 * must be host arrays.  This is an export call: the index itself keeps one slot per sorted row,
 * and the compaction into this CSR runs on the host (O(n_rows) time and host memory).
 */
int sebm25_get_postings(sebm25_index* ix, uint64_t* row_ptr, uint64_t* news_ptr, uint32_t* post_unit,
                        uint32_t* post_weight_tenths);

/* ---- scoring + top-k: replaces the SQL executed at app/api/search_service.py:319-320 and
 *      :351-352 (SEARCH_SQL :13-40, FALLBACK_SEARCH_SQL :42-66, NEWS_SEARCH_SQL :68-101) ------ */

/*
 * Batched scoring.  Query q has term ids q_terms[q_offsets[q] .. q_offsets[q+1]) (host arrays), any number
 * of them; duplicate ids count once and ids >= n_terms match nothing (`t.term = ANY($1)`).  Queries of up to
 * eight distinct matching terms take the fastest kernels, up to 256 the general one, and beyond that the
 * general one with its list cursors in device memory (8 bytes per term and work item, allocated by the call).
 * For each unit with at least one matching posting of the requested source:
 *   score   = sum over query terms (in the order given) of (idf_used[t] / 10) * w10(t, unit)
 *             with w10 the integer weight in tenths; double arithmetic, no fused multiply-add
 *   matched = COUNT(DISTINCT t.term)
 * Results are ordered score DESC, url rank ASC and cut at k (`LIMIT $2`), k <= 2048.
 * Outputs (host or device, caller-allocated): out_unit/out_score/out_matched/out_tie hold
 * n_queries * k entries, row q filled for the first out_count[q] slots; the rest is padded
 * with unit 0xFFFFFFFF, score 0, matched 0.  out_tie (the tie-break key of each result, needed
 * to merge shards) may be NULL.
 */
int sebm25_search(sebm25_index* ix, int source, const uint32_t* q_terms, const uint64_t* q_offsets,
                  uint32_t n_queries, uint32_t k, uint32_t* out_unit, double* out_score,
                  uint32_t* out_matched, uint32_t* out_tie, uint32_t* out_count);

/*
 * sebm25_search for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc,
 * code that runs beside a graph capture on another stream.  It has no counterpart in the reference.  (The
 * call itself is not capturable: it plans on the host, copies the plan and waits for its stream.)  Same
 * arguments, results and ordering as sebm25_search, but a successful call allocates nothing on the device or
 * the host (a failing one builds its error text; with timing enabled the first call creates its events): the work
 * items, the deduplicated query terms and the temporary result lists of queries that are split by
 * unit range live in one fixed device array of about 2.8 MB per GPU (held from sebm25_create of the
 * first handle on that GPU until the process ends), and the host-side plan in pinned storage of the
 * handle.  Limits: n_queries <= 1024, k <= 256, at most 4096 distinct matching query terms per call and 256
 * per query (SEBM25_E_RANGE beyond them); all output arrays must be device memory (out_tie may be NULL).  A
 * batch whose split queries would need more than 131072 temporary entries (lists x k) or more than
 * 4096 work items is scored without splitting, which changes the speed and never the result.  All
 * calls on one GPU share that GPU's array: within a process they are serialised by the library;
 * calls from different handles must not be relied upon to overlap, and nothing in the array has to
 * be cleared between calls (every entry read by a call is written by that call first).
 */
int sebm25_search_noalloc(sebm25_index* ix, int source, const uint32_t* q_terms, const uint64_t* q_offsets,
                          uint32_t n_queries, uint32_t k, uint32_t* out_unit, double* out_score,
                          uint32_t* out_matched, uint32_t* out_tie, uint32_t* out_count);

/*
 * Merges n_lists per-shard result sets (the layout sebm25_search writes, concatenated list by
 * list: in_x[(l * n_queries + q) * k + i], in_count[l * n_queries + q]) into one top-k per
 * query with the same ordering.  Device or host pointers.  The reference's counterpart is the
 * single shared database doing ORDER BY ... LIMIT over all rows (search_service.py:38-39).
 */
int sebm25_merge_topk(sebm25_index* ix, uint32_t n_lists, uint32_t n_queries, uint32_t k,
                      const uint32_t* in_unit, const double* in_score, const uint32_t* in_matched,
                      const uint32_t* in_tie, const uint32_t* in_count, uint32_t* out_unit,
                      double* out_score, uint32_t* out_matched, uint32_t* out_tie, uint32_t* out_count);

/* ---- measurement ------------------------------------------------------------------------- */

/* Device time of each phase of the last build / search on this handle, from HIP events on the
 * handle's stream.  names_out receives pointers to static strings.  Returns the number of
 * phases through *n_out (at most cap are written). */
int sebm25_get_timings(sebm25_index* ix, const char** names_out, float* ms_out, uint32_t cap, uint32_t* n_out);
/* Timing events are recorded only while enabled (default off; they add a few us per phase). */
int sebm25_enable_timing(sebm25_index* ix, int on);

/* ---- synthetic corpus (measurement and tests; no counterpart in the reference) ----------- */

typedef struct sebm25_corpus_spec {
    uint64_t seed;
    uint32_t n_terms;      /* V                                                                  */
    uint32_t doc_begin;    /* web docs [doc_begin, doc_end) of the global corpus are generated,  */
    uint32_t doc_end;      /* with unit ids local to the shard (doc - doc_begin)                 */
    uint32_t article_begin;/* news articles [article_begin, article_end)                         */
    uint32_t article_end;
    double zipf_s;         /* term rank r drawn with p(r) ~ r^-zipf_s; term_id = r - 1           */
    uint32_t body_min;     /* body length L ~ U{body_min .. body_max} draws                      */
    uint32_t body_max;
    uint32_t title_len;    /* draws in the title field                                           */
    uint32_t desc_len;     /* draws in the description field                                     */
} sebm25_corpus_spec;

/* Fills cdf[n_terms] (host) with the 32-bit fixed-point cumulative Zipf distribution the
 * generators sample from. */
int sebm25_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out);

/*
 * Generates the shard's rows directly in HBM.  First call with all output pointers NULL to
 * get *n_rows_out; then call again with device buffers of that many entries (word_count:
 * doc_end - doc_begin entries).  Rows come out unit-major: web docs ascending, then articles;
 * inside a unit by field (1, 2, 4) then ascending term id.
 */
int sebm25_gen_corpus(sebm25_index* ix, const sebm25_corpus_spec* spec, uint64_t* n_rows_out,
                      uint32_t* term_id, uint32_t* unit_id, uint32_t* tf, uint8_t* meta,
                      uint32_t* word_count);

/* Generates n_queries queries of terms_per_query distinct term ids drawn from the same Zipf
 * law restricted to ranks >= min_rank (host output, n_queries * terms_per_query entries). */
int sebm25_gen_queries(uint64_t seed, uint32_t n_terms, double zipf_s, uint32_t min_rank,
                       uint32_t n_queries, uint32_t terms_per_query, uint32_t* q_terms_out);

#ifdef __cplusplus
}
#endif
#endif /* SEBM25_H */
