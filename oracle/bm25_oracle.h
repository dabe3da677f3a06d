/*
 * bm25_oracle.h — CPU restatement of the reference's BM25 statistics job and scoring SQL.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may include, link or call this; it is
 * the checker used by tests/, by the smoke test and by the cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors for this path and
 * its arithmetic runs inside PostgreSQL (not available here), so this restatement follows the
 * SQL text and is cross-checked against sqlite3 executing equivalent statements
 * (tests/test_oracle_sqlite.py) and against hand-worked cases, not against reference outputs.
 */
#ifndef BM25_ORACLE_H
#define BM25_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_index orc_index;

typedef struct orc_totals {
    uint64_t n_done, n_wc, sum_wc, n_postings;
} orc_totals;

typedef struct orc_corpus_spec {
    uint64_t seed;
    uint32_t n_terms, doc_begin, doc_end, article_begin, article_end;
    double zipf_s;
    uint32_t body_min, body_max, title_len, desc_len;
} orc_corpus_spec;

/* synthetic corpus (same definition as the GPU generator, implemented independently) */
void orc_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out);
uint64_t orc_gen_corpus(const orc_corpus_spec* spec, const uint32_t* cdf, uint32_t* term_id,
                        uint32_t* unit_id, uint32_t* tf, uint8_t* meta, uint32_t* word_count);
int orc_gen_queries(uint64_t seed, uint32_t n_terms, const uint32_t* cdf, uint32_t min_rank,
                    uint32_t n_queries, uint32_t terms_per_query, uint32_t* out);

/* statistics job + inverted index.  Returns NULL when a row is out of range. */
orc_index* orc_build(const uint32_t* term_id, const uint32_t* unit_id, const uint32_t* tf,
                     const uint8_t* meta, uint64_t n_rows, uint32_t n_terms, uint32_t n_docs,
                     uint32_t n_articles, const uint32_t* word_count, const uint8_t* doc_flags);
void orc_free(orc_index* ix);
/* recompute N, avgdl, idf from corpus-wide sums (document-sharded use) */
void orc_finalize(orc_index* ix, const uint32_t* df_global, const orc_totals* totals_global);

const uint32_t* orc_df(const orc_index* ix);
const double* orc_idf(const orc_index* ix);
const uint32_t* orc_term_rows(const orc_index* ix);
const uint32_t* orc_doc_len(const orc_index* ix);
const uint64_t* orc_row_ptr(const orc_index* ix);
const uint64_t* orc_news_ptr(const orc_index* ix);
const uint32_t* orc_post_unit(const orc_index* ix);
const uint32_t* orc_post_weight(const orc_index* ix); /* weights in tenths (exact integers) */
void orc_get_totals(const orc_index* ix, orc_totals* out);
uint64_t orc_n_docs_stat(const orc_index* ix); /* N of bm25_stats.py:12-13 */
double orc_avgdl(const orc_index* ix);

void orc_set_idf(orc_index* ix, const double* idf, const uint8_t* present);

/* scoring straight from the token rows with the SQL's formula; returns the result count */
uint32_t orc_search_rows(orc_index* ix, int source, const uint32_t* terms, uint32_t n_terms_q,
                         uint32_t k, const uint32_t* url_rank, uint32_t unit_base,
                         uint32_t* out_unit, double* out_score, uint32_t* out_matched);
/* scoring from the merged integer postings (the arithmetic the GPU index performs):
 * score = sum over terms of (idf / 10.0) * weight_tenths */
uint32_t orc_search_csr(orc_index* ix, int source, const uint32_t* terms, uint32_t n_terms_q,
                        uint32_t k, const uint32_t* url_rank, uint32_t unit_base,
                        uint32_t* out_unit, double* out_score, uint32_t* out_matched);

#ifdef __cplusplus
}
#endif
#endif
