/*
 * bm25_oracle.c — CPU restatement (plain C) of the reference's BM25 statistics job and
 * scoring SQL.  TEST INFRASTRUCTURE ONLY (see bm25_oracle.h): the product never links this.
 * PARITY UNPINNED against reference outputs (the reference has none); pinned instead by the
 * sqlite3 cross-check and hand-worked cases in tests/.
 *
 * Reference statements followed (paths relative to the reference checkout):
 *   app/batch/bm25_stats.py:9-10   avgdl = AVG(word_count) over status='done', NULL -> 0.0
 *   app/batch/bm25_stats.py:12-13  N = COUNT(*) over status='done', 0 -> 1
 *   app/batch/bm25_stats.py:19-24  per term: df = COUNT(DISTINCT doc_id) (NULL doc_id of news
 *                                  rows not counted), idf = LN((N - df + 0.5)/(df + 0.5) + 1)
 *   app/api/search_service.py:19-39   web score = SUM(tf * COALESCE(idf,1.0) * W[field]),
 *                                     matched = COUNT(DISTINCT term), status='done' docs only,
 *                                     ORDER BY score DESC, url ASC LIMIT k
 *   app/api/search_service.py:84-100  news score = SUM(tf * COALESCE(idf,1.0)), source_type=2
 */
#include "bm25_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ synthetic corpus --- */

static inline uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
/* counter-based generator: any (stream, unit, slot) can be drawn independently */
static inline uint64_t unit_key(uint64_t seed, uint64_t stream, uint64_t unit) {
    uint64_t h = mix64(seed + stream * 0x9E3779B97F4A7C15ULL);
    return mix64(h ^ (unit * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
}
static inline uint32_t draw32(uint64_t key, uint64_t slot) {
    return (uint32_t)(mix64(key + slot * 0x2545F4914F6CDD1DULL) >> 32);
}
static inline uint32_t scale32(uint32_t r, uint32_t n) { return (uint32_t)(((uint64_t)r * n) >> 32); }

void orc_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out) {
    double total = 0.0;
    for (uint32_t r = 1; r <= n_terms; ++r) total += (zipf_s == 1.0) ? 1.0 / (double)r : pow((double)r, -zipf_s);
    double cum = 0.0;
    for (uint32_t r = 1; r <= n_terms; ++r) {
        cum += (zipf_s == 1.0) ? 1.0 / (double)r : pow((double)r, -zipf_s);
        double x = floor(cum / total * 4294967296.0);
        cdf_out[r - 1] = x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x;
    }
    cdf_out[n_terms - 1] = 0xFFFFFFFFu;
}

/* first index i with cdf[i] >= r */
static inline uint32_t zipf_term(const uint32_t* cdf, uint32_t n, uint32_t r) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        uint32_t mid = lo + (hi - lo) / 2;
        if (cdf[mid] >= r) hi = mid; else lo = mid + 1;
    }
    return lo;
}

static int cmp_u32(const void* a, const void* b) {
    uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
    return x < y ? -1 : x > y;
}

/* draws `n` terms for slots [slot0, slot0+n), emits one row per distinct term (ascending) */
static uint64_t emit_field(uint64_t key, uint32_t slot0, uint32_t n, const uint32_t* cdf, uint32_t V,
                           uint32_t* scratch, uint32_t unit, uint8_t meta, uint64_t pos,
                           uint32_t* term_id, uint32_t* unit_id, uint32_t* tf, uint8_t* m) {
    for (uint32_t i = 0; i < n; ++i) scratch[i] = zipf_term(cdf, V, draw32(key, slot0 + i));
    qsort(scratch, n, sizeof(uint32_t), cmp_u32);
    for (uint32_t i = 0; i < n;) {
        uint32_t j = i;
        while (j < n && scratch[j] == scratch[i]) ++j;
        if (term_id) {
            term_id[pos] = scratch[i];
            unit_id[pos] = unit;
            tf[pos] = j - i;
            m[pos] = meta;
        }
        ++pos;
        i = j;
    }
    return pos;
}

uint64_t orc_gen_corpus(const orc_corpus_spec* s, const uint32_t* cdf, uint32_t* term_id, uint32_t* unit_id,
                        uint32_t* tf, uint8_t* meta, uint32_t* word_count) {
    uint32_t span = s->body_max - s->body_min + 1;
    uint32_t maxn = s->title_len + s->desc_len + s->body_max + 1;
    uint32_t* scratch = (uint32_t*)malloc(sizeof(uint32_t) * maxn);
    uint64_t pos = 0;
    for (uint32_t d = s->doc_begin; d < s->doc_end; ++d) {
        uint64_t key = unit_key(s->seed, 1, d);
        uint32_t L = s->body_min + scale32(draw32(key, 0), span);
        uint32_t u = d - s->doc_begin;
        pos = emit_field(key, 1, s->title_len, cdf, s->n_terms, scratch, u, (uint8_t)(1 | (1 << 2)), pos, term_id, unit_id, tf, meta);
        pos = emit_field(key, 1 + s->title_len, s->desc_len, cdf, s->n_terms, scratch, u, (uint8_t)(1 | (2 << 2)), pos, term_id, unit_id, tf, meta);
        pos = emit_field(key, 1 + s->title_len + s->desc_len, L, cdf, s->n_terms, scratch, u, (uint8_t)(1 | (4 << 2)), pos, term_id, unit_id, tf, meta);
        if (word_count) word_count[u] = L;
    }
    for (uint32_t a = s->article_begin; a < s->article_end; ++a) {
        uint64_t key = unit_key(s->seed, 2, a);
        uint32_t L = s->body_min + scale32(draw32(key, 0), span);
        uint32_t u = a - s->article_begin;
        /* one tokenize() over title + description + content (news_fetcher.py:298-311) */
        pos = emit_field(key, 1, s->title_len + s->desc_len + L, cdf, s->n_terms, scratch, u, (uint8_t)(2 | (4 << 2)), pos, term_id, unit_id, tf, meta);
    }
    free(scratch);
    return pos;
}

int orc_gen_queries(uint64_t seed, uint32_t n_terms, const uint32_t* cdf, uint32_t min_rank, uint32_t n_queries,
                    uint32_t tpq, uint32_t* out) {
    if (min_rank < 1) min_rank = 1;
    if ((uint64_t)n_terms < (uint64_t)min_rank - 1 + tpq) return 1;
    uint64_t lo = min_rank >= 2 ? (uint64_t)cdf[min_rank - 2] + 1 : 0;
    uint64_t range = 4294967296ULL - lo;
    for (uint32_t q = 0; q < n_queries; ++q) {
        uint64_t key = unit_key(seed, 3, q);
        uint32_t got = 0;
        for (uint64_t slot = 0; got < tpq; ++slot) {
            uint32_t r = (uint32_t)(lo + (((uint64_t)draw32(key, slot) * range) >> 32));
            uint32_t t = zipf_term(cdf, n_terms, r);
            if (t + 1 < min_rank) t = min_rank - 1;
            int dup = 0;
            for (uint32_t i = 0; i < got; ++i) dup |= out[(uint64_t)q * tpq + i] == t;
            if (!dup) out[(uint64_t)q * tpq + got++] = t;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ statistics + index --- */

struct orc_index {
    uint64_t R;
    uint32_t V, n_docs, n_articles;
    /* copies of the rows in (term, source, unit, input order) order */
    uint32_t *s_unit, *s_tf;
    uint8_t* s_meta;
    uint64_t* term_start; /* [V+1] into the sorted rows */
    uint8_t* doc_done;    /* [n_docs] */
    uint32_t *df, *term_rows, *doc_len;
    double *idf, *idf_used;
    uint64_t *row_ptr, *news_ptr;
    uint32_t* post_unit;
    uint32_t* post_w; /* posting weight in tenths: 32*tf_title + 17*tf_description + 10*tf_other (web), 10*tf (news) */
    orc_totals totals;
    uint64_t N;
    double avgdl;
    /* scoring scratch */
    double* acc;
    uint32_t *cnt, *last_term, *touched;
};

static void radix_pass(const uint64_t* in, uint64_t* out, uint64_t n, const uint32_t* key, int shift, uint64_t* count) {
    memset(count, 0, sizeof(uint64_t) * 65537);
    for (uint64_t i = 0; i < n; ++i) count[((key[in[i]] >> shift) & 0xFFFF) + 1]++;
    for (int b = 0; b < 65536; ++b) count[b + 1] += count[b];
    for (uint64_t i = 0; i < n; ++i) out[count[(key[in[i]] >> shift) & 0xFFFF]++] = in[i];
}

static inline double field_weight(uint32_t field) { return field == 1 ? 3.2 : field == 2 ? 1.7 : 1.0; }

static void compute_idf(orc_index* ix, const uint32_t* df) {
    /* LN((N - df + 0.5) / (df + 0.5) + 1), bm25_stats.py:21.  PostgreSQL evaluates this in
     * NUMERIC with >= 16 significant digits at every step; log1p keeps the double evaluation
     * that accurate when df is close to N. */
    for (uint32_t t = 0; t < ix->V; ++t) {
        double q = ((double)ix->N - (double)df[t] + 0.5) / ((double)df[t] + 0.5);
        ix->idf[t] = log1p(q);
        ix->idf_used[t] = ix->idf[t];
    }
}

void orc_finalize(orc_index* ix, const uint32_t* df_global, const orc_totals* tg) {
    const orc_totals* t = tg ? tg : &ix->totals;
    ix->N = t->n_done ? t->n_done : 1;                                   /* bm25_stats.py:13 */
    ix->avgdl = t->n_wc ? (double)t->sum_wc / (double)t->n_wc : 0.0;     /* bm25_stats.py:9-10 */
    compute_idf(ix, df_global ? df_global : ix->df);
}

orc_index* orc_build(const uint32_t* term_id, const uint32_t* unit_id, const uint32_t* tf, const uint8_t* meta,
                     uint64_t R, uint32_t V, uint32_t n_docs, uint32_t n_articles, const uint32_t* word_count,
                     const uint8_t* doc_flags) {
    if (V == 0) return NULL;
    for (uint64_t i = 0; i < R; ++i) {
        uint32_t st = meta[i] & 3u;
        if (term_id[i] >= V || (st != 1 && st != 2)) return NULL;
        if (unit_id[i] >= (st == 1 ? n_docs : n_articles)) return NULL;
    }
    orc_index* ix = (orc_index*)calloc(1, sizeof(orc_index));
    ix->R = R; ix->V = V; ix->n_docs = n_docs; ix->n_articles = n_articles;

    /* stable LSD sort of row indices by (term, source, unit) */
    uint64_t* a = (uint64_t*)calloc(R ? R : 1, sizeof(uint64_t));
    uint64_t* b = (uint64_t*)malloc(sizeof(uint64_t) * (R ? R : 1));
    uint32_t* srckey = (uint32_t*)calloc(R ? R : 1, sizeof(uint32_t));
    uint64_t* count = (uint64_t*)malloc(sizeof(uint64_t) * 65537);
    uint32_t max_unit = 0, max_term = 0;
    for (uint64_t i = 0; i < R; ++i) {
        a[i] = i;
        srckey[i] = (meta[i] & 3u) - 1u;
        if (unit_id[i] > max_unit) max_unit = unit_id[i];
        if (term_id[i] > max_term) max_term = term_id[i];
    }
    radix_pass(a, b, R, unit_id, 0, count);
    uint64_t *cur = b, *oth = a, *tmp;
    if (max_unit >> 16) { radix_pass(cur, oth, R, unit_id, 16, count); tmp = cur; cur = oth; oth = tmp; }
    radix_pass(cur, oth, R, srckey, 0, count); tmp = cur; cur = oth; oth = tmp;
    radix_pass(cur, oth, R, term_id, 0, count); tmp = cur; cur = oth; oth = tmp;
    if (max_term >> 16) { radix_pass(cur, oth, R, term_id, 16, count); tmp = cur; cur = oth; oth = tmp; }

    ix->s_unit = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    ix->s_tf = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    ix->s_meta = (uint8_t*)malloc(R ? R : 1);
    ix->term_start = (uint64_t*)calloc((size_t)V + 1, sizeof(uint64_t));
    ix->term_rows = (uint32_t*)calloc(V, sizeof(uint32_t));
    for (uint64_t i = 0; i < R; ++i) {
        uint64_t r = cur[i];
        ix->s_unit[i] = unit_id[r]; ix->s_tf[i] = tf[r]; ix->s_meta[i] = meta[r];
        ix->term_rows[term_id[r]]++;
    }
    for (uint32_t t = 0; t < V; ++t) ix->term_start[t + 1] = ix->term_start[t] + ix->term_rows[t];
    free(a); free(b); free(srckey); free(count);

    /* documents: status, word_count */
    ix->doc_done = (uint8_t*)malloc(n_docs ? n_docs : 1);
    for (uint32_t d = 0; d < n_docs; ++d) ix->doc_done[d] = doc_flags ? (doc_flags[d] & 1u) : 1;
    ix->doc_len = (uint32_t*)calloc(n_docs ? n_docs : 1, sizeof(uint32_t));
    for (uint64_t i = 0; i < R; ++i)
        if ((meta[i] & 3u) == 1 && (meta[i] >> 2) == 4) ix->doc_len[unit_id[i]] += tf[i];
    for (uint32_t d = 0; d < n_docs; ++d) {
        if (!ix->doc_done[d]) continue;
        ix->totals.n_done++;
        if (doc_flags && (doc_flags[d] & 2u)) continue; /* word_count IS NULL: AVG skips it */
        ix->totals.n_wc++;
        ix->totals.sum_wc += word_count ? word_count[d] : ix->doc_len[d];
    }

    /* collapse rows of one (term, source, unit) into a posting */
    ix->df = (uint32_t*)calloc(V, sizeof(uint32_t));
    ix->row_ptr = (uint64_t*)calloc((size_t)V + 1, sizeof(uint64_t));
    ix->news_ptr = (uint64_t*)calloc(V, sizeof(uint64_t));
    ix->post_unit = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    ix->post_w = (uint32_t*)malloc(sizeof(uint32_t) * (R ? R : 1));
    uint64_t p = 0;
    for (uint32_t t = 0; t < V; ++t) {
        ix->row_ptr[t] = p;
        int seen_news = 0;
        ix->news_ptr[t] = 0;
        for (uint64_t i = ix->term_start[t]; i < ix->term_start[t + 1];) {
            uint32_t st = ix->s_meta[i] & 3u, u = ix->s_unit[i];
            uint64_t j = i;
            uint64_t w = 0; /* exact: W = {3.2, 1.7, 1.0} are 32, 17, 10 tenths */
            while (j < ix->term_start[t + 1] && (ix->s_meta[j] & 3u) == st && ix->s_unit[j] == u) {
                uint32_t field = ix->s_meta[j] >> 2;
                w += (uint64_t)ix->s_tf[j] * (st == 1 ? (field == 1 ? 32u : field == 2 ? 17u : 10u) : 10u);
                ++j;
            }
            if (w > 0xFFFFFFFFull) w = 0xFFFFFFFFull;
            if (st == 2 && !seen_news) { ix->news_ptr[t] = p; seen_news = 1; }
            if (st == 1) ix->df[t]++; /* COUNT(DISTINCT doc_id): any status, NULL (news) not counted */
            if (st == 2 || ix->doc_done[u]) { ix->post_unit[p] = u; ix->post_w[p] = (uint32_t)w; ++p; }
            i = j;
        }
        if (!seen_news) ix->news_ptr[t] = p;
    }
    ix->row_ptr[V] = p;
    ix->totals.n_postings = p;

    ix->idf = (double*)malloc(sizeof(double) * V);
    ix->idf_used = (double*)malloc(sizeof(double) * V);
    orc_finalize(ix, NULL, NULL);

    uint32_t nu = n_docs > n_articles ? n_docs : n_articles;
    if (!nu) nu = 1;
    ix->acc = (double*)calloc(nu, sizeof(double));
    ix->cnt = (uint32_t*)calloc(nu, sizeof(uint32_t));
    ix->last_term = (uint32_t*)calloc(nu, sizeof(uint32_t));
    ix->touched = (uint32_t*)malloc(sizeof(uint32_t) * nu);
    return ix;
}

void orc_free(orc_index* ix) {
    if (!ix) return;
    free(ix->s_unit); free(ix->s_tf); free(ix->s_meta); free(ix->term_start); free(ix->doc_done);
    free(ix->df); free(ix->term_rows); free(ix->doc_len); free(ix->idf); free(ix->idf_used);
    free(ix->row_ptr); free(ix->news_ptr); free(ix->post_unit); free(ix->post_w);
    free(ix->acc); free(ix->cnt); free(ix->last_term); free(ix->touched);
    free(ix);
}

const uint32_t* orc_df(const orc_index* ix) { return ix->df; }
const double* orc_idf(const orc_index* ix) { return ix->idf; }
const uint32_t* orc_term_rows(const orc_index* ix) { return ix->term_rows; }
const uint32_t* orc_doc_len(const orc_index* ix) { return ix->doc_len; }
const uint64_t* orc_row_ptr(const orc_index* ix) { return ix->row_ptr; }
const uint64_t* orc_news_ptr(const orc_index* ix) { return ix->news_ptr; }
const uint32_t* orc_post_unit(const orc_index* ix) { return ix->post_unit; }
const uint32_t* orc_post_weight(const orc_index* ix) { return ix->post_w; }
void orc_get_totals(const orc_index* ix, orc_totals* out) { *out = ix->totals; }
uint64_t orc_n_docs_stat(const orc_index* ix) { return ix->N; }
double orc_avgdl(const orc_index* ix) { return ix->avgdl; }

void orc_set_idf(orc_index* ix, const double* idf, const uint8_t* present) {
    for (uint32_t t = 0; t < ix->V; ++t) ix->idf_used[t] = (!present || present[t]) ? idf[t] : 1.0; /* COALESCE(ts.idf, 1.0) */
}

/* ------------------------------------------------------------------ scoring + top-k ------ */

typedef struct cand { double score; uint32_t tie, unit, matched; } cand;
static int cmp_cand(const void* a, const void* b) {
    const cand *x = (const cand*)a, *y = (const cand*)b;
    if (x->score != y->score) return x->score > y->score ? -1 : 1; /* ORDER BY token_score DESC */
    if (x->tie != y->tie) return x->tie < y->tie ? -1 : 1;         /* , url ASC */
    return 0;
}

static uint32_t finish_topk(orc_index* ix, uint32_t n_touched, uint32_t k, const uint32_t* url_rank,
                            uint32_t unit_base, uint32_t* out_unit, double* out_score, uint32_t* out_matched) {
    cand* c = (cand*)malloc(sizeof(cand) * (n_touched ? n_touched : 1));
    for (uint32_t i = 0; i < n_touched; ++i) {
        uint32_t u = ix->touched[i];
        c[i].score = ix->acc[u];
        c[i].unit = u + unit_base;
        c[i].tie = url_rank ? url_rank[u] : u + unit_base;
        c[i].matched = ix->cnt[u];
        ix->acc[u] = 0.0; ix->cnt[u] = 0; ix->last_term[u] = 0;
    }
    qsort(c, n_touched, sizeof(cand), cmp_cand);
    uint32_t n = n_touched < k ? n_touched : k; /* LIMIT k */
    for (uint32_t i = 0; i < n; ++i) { out_unit[i] = c[i].unit; out_score[i] = c[i].score; out_matched[i] = c[i].matched; }
    free(c);
    return n;
}

static uint32_t dedupe_terms(const orc_index* ix, const uint32_t* terms, uint32_t n, uint32_t* out) {
    uint32_t m = 0;
    for (uint32_t i = 0; i < n; ++i) { /* t.term = ANY($1): a repeated term selects the same rows once */
        if (terms[i] >= ix->V) continue;
        int dup = 0;
        for (uint32_t j = 0; j < m; ++j) dup |= out[j] == terms[i];
        if (!dup) out[m++] = terms[i];
    }
    return m;
}

uint32_t orc_search_rows(orc_index* ix, int source, const uint32_t* terms, uint32_t nq, uint32_t k,
                         const uint32_t* url_rank, uint32_t unit_base, uint32_t* out_unit, double* out_score,
                         uint32_t* out_matched) {
    uint32_t* ts = (uint32_t*)malloc(sizeof(uint32_t) * (nq ? nq : 1));
    uint32_t m = dedupe_terms(ix, terms, nq, ts), n_touched = 0;
    for (uint32_t qi = 0; qi < m; ++qi) {
        uint32_t t = ts[qi];
        double idf = ix->idf_used[t];
        for (uint64_t i = ix->term_start[t]; i < ix->term_start[t + 1]; ++i) {
            uint32_t st = ix->s_meta[i] & 3u, u = ix->s_unit[i];
            if ((int)st != source) continue;
            if (st == 1 && !ix->doc_done[u]) continue; /* JOIN documents ... WHERE d.status='done' */
            double v = (double)ix->s_tf[i] * idf;       /* t.frequency * COALESCE(ts.idf, 1.0) */
            if (st == 1) v = v * field_weight(ix->s_meta[i] >> 2); /* * CASE t.field ... (web only) */
            if (ix->last_term[u] == 0) ix->touched[n_touched++] = u;
            if (ix->last_term[u] != qi + 1) { ix->cnt[u]++; ix->last_term[u] = qi + 1; }
            ix->acc[u] = ix->acc[u] + v;
        }
    }
    free(ts);
    return finish_topk(ix, n_touched, k, url_rank, unit_base, out_unit, out_score, out_matched);
}

uint32_t orc_search_csr(orc_index* ix, int source, const uint32_t* terms, uint32_t nq, uint32_t k,
                        const uint32_t* url_rank, uint32_t unit_base, uint32_t* out_unit, double* out_score,
                        uint32_t* out_matched) {
    uint32_t* ts = (uint32_t*)malloc(sizeof(uint32_t) * (nq ? nq : 1));
    uint32_t m = dedupe_terms(ix, terms, nq, ts), n_touched = 0;
    for (uint32_t qi = 0; qi < m; ++qi) {
        uint32_t t = ts[qi];
        double idf10 = ix->idf_used[t] / 10.0; /* one division per list, then one multiply per posting */
        uint64_t b = source == 1 ? ix->row_ptr[t] : ix->news_ptr[t];
        uint64_t e = source == 1 ? ix->news_ptr[t] : ix->row_ptr[t + 1];
        for (uint64_t i = b; i < e; ++i) {
            uint32_t u = ix->post_unit[i];
            if (ix->last_term[u] == 0) ix->touched[n_touched++] = u;
            ix->cnt[u]++; ix->last_term[u] = qi + 1;
            ix->acc[u] = ix->acc[u] + idf10 * (double)ix->post_w[i];
        }
    }
    free(ts);
    return finish_topk(ix, n_touched, k, url_rank, unit_base, out_unit, out_score, out_matched);
}
