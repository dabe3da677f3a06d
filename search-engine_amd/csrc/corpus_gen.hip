// Synthetic Zipfian corpus generated directly in HBM (measurement and tests).  The reference
// has no counterpart; the row format is the one its writers produce
// (app/crawler/worker.py:229-239, app/batch/news_fetcher.py:303-311): a web doc has one row per
// (field, term) with field in {1 title, 2 description, 4 body}; a news article has one row per
// term with field 4, source_type 2 and no doc id.
//
// Counter-based generator: every draw is a pure function of (seed, stream, unit, slot), so any
// shard can produce its own documents and a CPU implementation of the same definition yields
// identical rows.  Term sampling is an integer binary search in a 32-bit fixed-point CDF, so
// no floating point runs on the device.  One wavefront generates one unit.
#include "common.hpp"

namespace sebm25 {

namespace {

constexpr int GEN_MAX_DRAWS = 1024;  // draws of one field (title + description + body for news)
constexpr int GEN_WAVES = 4;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
__host__ __device__ __forceinline__ uint64_t unit_key(uint64_t seed, uint64_t stream, uint64_t unit) {
    const uint64_t h = mix64(seed + stream * 0x9E3779B97F4A7C15ULL);
    return mix64(h ^ (unit * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL));
}
__host__ __device__ __forceinline__ uint32_t draw32(uint64_t key, uint64_t slot) {
    return (uint32_t)(mix64(key + slot * 0x2545F4914F6CDD1DULL) >> 32);
}
__host__ __device__ __forceinline__ uint32_t scale32(uint32_t r, uint32_t n) {
    return (uint32_t)(((uint64_t)r * n) >> 32);
}
// first index i with cdf[i] >= r
__host__ __device__ __forceinline__ uint32_t zipf_term(const uint32_t* cdf, uint32_t n, uint32_t r) {
    uint32_t lo = 0, hi = n - 1;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] >= r) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// wave-wide ascending bitonic sort of p2 (power of two, >= 64) words in LDS
__device__ void wave_sort(uint32_t* v, uint32_t p2) {
    const uint32_t lane = lane_id();
    for (uint32_t size = 2; size <= p2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t e = lane; e < p2 / 2; e += 64) {
                const uint32_t lo = ((e & ~(stride - 1)) << 1) | (e & (stride - 1));
                const uint32_t hi = lo | stride;
                const bool asc = (lo & size) == 0;
                const uint32_t a = v[lo], b = v[hi];
                if ((a > b) == asc) {
                    v[lo] = b;
                    v[hi] = a;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Draws one field, sorts it, and either counts its distinct terms or writes one row per
// distinct term at out_pos.  Returns the number of distinct terms.
template <bool EMIT>
__device__ uint32_t gen_field(uint64_t key, uint32_t slot0, uint32_t n, const uint32_t* cdf, uint32_t V,
                              uint32_t* buf, uint32_t* heads, uint32_t unit, uint8_t meta, uint64_t out_pos,
                              uint32_t* term_id, uint32_t* unit_id, uint32_t* tf, uint8_t* m) {
    const uint32_t lane = lane_id();
    uint32_t p2 = 64;
    while (p2 < n) p2 <<= 1;
    for (uint32_t i = lane; i < p2; i += 64) buf[i] = i < n ? zipf_term(cdf, V, draw32(key, slot0 + i)) : 0xFFFFFFFFu;
    __builtin_amdgcn_wave_barrier();
    wave_sort(buf, p2);
    uint32_t n_heads = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        const bool head = i < n && (i == 0 || buf[i] != buf[i - 1]);
        const uint64_t hm = __ballot(head);
        if (EMIT && head) heads[n_heads + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull))] = i;
        n_heads += (uint32_t)__popcll(hm);
    }
    if (EMIT) {
        __builtin_amdgcn_wave_barrier();
        for (uint32_t h = lane; h < n_heads; h += 64) {
            const uint32_t i = heads[h];
            const uint32_t nxt = (h + 1 < n_heads) ? heads[h + 1] : n;
            term_id[out_pos + h] = buf[i];
            unit_id[out_pos + h] = unit;
            tf[out_pos + h] = nxt - i;
            m[out_pos + h] = meta;
        }
        __builtin_amdgcn_wave_barrier();
    }
    return n_heads;
}

template <bool EMIT>
__global__ __launch_bounds__(64 * GEN_WAVES) void k_gen(sebm25_corpus_spec s, const uint32_t* cdf,
                                                        uint32_t* unit_rows, const uint32_t* row_off,
                                                        uint32_t* term_id, uint32_t* unit_id, uint32_t* tf,
                                                        uint8_t* meta, uint32_t* word_count) {
    __shared__ uint32_t buf_s[GEN_WAVES][GEN_MAX_DRAWS];
    __shared__ uint32_t heads_s[GEN_WAVES][GEN_MAX_DRAWS];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t n_web = s.doc_end - s.doc_begin, n_news = s.article_end - s.article_begin;
    const uint64_t u = (uint64_t)blockIdx.x * GEN_WAVES + wave;
    if (u >= (uint64_t)n_web + n_news) return;
    uint32_t* buf = buf_s[wave];
    uint32_t* heads = heads_s[wave];
    const bool news = u >= n_web;
    const uint32_t local = news ? (uint32_t)(u - n_web) : (uint32_t)u;
    const uint64_t global_id = news ? (uint64_t)s.article_begin + local : (uint64_t)s.doc_begin + local;
    const uint64_t key = unit_key(s.seed, news ? 2 : 1, global_id);
    const uint32_t L = s.body_min + scale32(draw32(key, 0), s.body_max - s.body_min + 1);
    uint64_t pos = EMIT ? row_off[u] : 0;
    uint32_t rows = 0;
    if (!news) {
        uint32_t c;
        c = gen_field<EMIT>(key, 1, s.title_len, cdf, s.n_terms, buf, heads, local, SEBM25_META(1u, 1u), pos, term_id, unit_id, tf, meta);
        pos += c; rows += c;
        c = gen_field<EMIT>(key, 1 + s.title_len, s.desc_len, cdf, s.n_terms, buf, heads, local, SEBM25_META(1u, 2u), pos, term_id, unit_id, tf, meta);
        pos += c; rows += c;
        c = gen_field<EMIT>(key, 1 + s.title_len + s.desc_len, L, cdf, s.n_terms, buf, heads, local, SEBM25_META(1u, 4u), pos, term_id, unit_id, tf, meta);
        rows += c;
        if (EMIT && word_count && lane_id() == 0) word_count[local] = L;
    } else {
        rows = gen_field<EMIT>(key, 1, s.title_len + s.desc_len + L, cdf, s.n_terms, buf, heads, local, SEBM25_META(2u, 4u), pos, term_id, unit_id, tf, meta);
    }
    if (!EMIT && lane_id() == 0) unit_rows[u] = rows;
}

}  // namespace

uint32_t gen_max_draws() { return GEN_MAX_DRAWS; }

void launch_gen_count(const sebm25_corpus_spec& spec, const uint32_t* cdf, uint32_t* unit_rows, hipStream_t st) {
    const uint64_t n = (uint64_t)(spec.doc_end - spec.doc_begin) + (spec.article_end - spec.article_begin);
    if (n == 0) return;
    hipLaunchKernelGGL(k_gen<false>, dim3((unsigned)((n + GEN_WAVES - 1) / GEN_WAVES)), dim3(64 * GEN_WAVES), 0, st, spec, cdf,
                       unit_rows, (const uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                       (uint8_t*)nullptr, (uint32_t*)nullptr);
}

void launch_gen_emit(const sebm25_corpus_spec& spec, const uint32_t* cdf, const uint32_t* row_off, uint32_t* term_id,
                     uint32_t* unit_id, uint32_t* tf, uint8_t* meta, uint32_t* word_count, hipStream_t st) {
    const uint64_t n = (uint64_t)(spec.doc_end - spec.doc_begin) + (spec.article_end - spec.article_begin);
    if (n == 0) return;
    hipLaunchKernelGGL(k_gen<true>, dim3((unsigned)((n + GEN_WAVES - 1) / GEN_WAVES)), dim3(64 * GEN_WAVES), 0, st, spec, cdf,
                       (uint32_t*)nullptr, row_off, term_id, unit_id, tf, meta, word_count);
}

// host helpers with the same definitions (CDF and query generation run on the host)
void host_zipf_cdf(uint32_t n_terms, double zipf_s, uint32_t* cdf_out) {
    double total = 0.0;
    for (uint32_t r = 1; r <= n_terms; ++r) total += (zipf_s == 1.0) ? 1.0 / (double)r : pow((double)r, -zipf_s);
    double cum = 0.0;
    for (uint32_t r = 1; r <= n_terms; ++r) {
        cum += (zipf_s == 1.0) ? 1.0 / (double)r : pow((double)r, -zipf_s);
        const double x = floor(cum / total * 4294967296.0);
        cdf_out[r - 1] = x >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)x;
    }
    cdf_out[n_terms - 1] = 0xFFFFFFFFu;
}

int host_gen_queries(uint64_t seed, uint32_t n_terms, const uint32_t* cdf, uint32_t min_rank, uint32_t n_queries,
                     uint32_t tpq, uint32_t* out) {
    if (min_rank < 1) min_rank = 1;
    if ((uint64_t)n_terms < (uint64_t)min_rank - 1 + tpq) return 1;
    const uint64_t lo = min_rank >= 2 ? (uint64_t)cdf[min_rank - 2] + 1 : 0;
    const uint64_t range = 4294967296ULL - lo;
    for (uint32_t q = 0; q < n_queries; ++q) {
        const uint64_t key = unit_key(seed, 3, q);
        uint32_t got = 0;
        for (uint64_t slot = 0; got < tpq; ++slot) {
            const uint32_t r = (uint32_t)(lo + (((uint64_t)draw32(key, slot) * range) >> 32));
            uint32_t t = zipf_term(cdf, n_terms, r);
            if (t + 1 < min_rank) t = min_rank - 1;
            bool dup = false;
            for (uint32_t i = 0; i < got; ++i) dup |= out[(uint64_t)q * tpq + i] == t;
            if (!dup) out[(uint64_t)q * tpq + got++] = t;
        }
    }
    return 0;
}

}  // namespace sebm25
