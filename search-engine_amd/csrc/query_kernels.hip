// Batched scoring + top-k for gfx950.  Replaces the scoring statements the reference sends to
// PostgreSQL (app/api/search_service.py:13-40 web, :42-66 fallback, :68-101 news):
//   score(unit) = SUM over query terms of idf_used[t] * w(t, unit),  matched = COUNT(DISTINCT term),
//   ORDER BY score DESC, url ASC LIMIT k.
//
// One wavefront owns one work item = (query, range of unit ids).  A term's posting list is a
// range of slots sorted by unit (a few of them holes that carry no posting), so the wave walks all lists of the query in step through a window of WIN consecutive
// unit ids whose accumulators live in LDS: 64 coalesced postings per list per load, a
// "touched" bitmap (one 32-bit word per lane) instead of clearing accumulators, and a
// candidate buffer that is cut back to the best k by a wave-wide bitonic sort whenever it
// fills.  Lists are added in query-term order, so a score is the same double sum whatever the
// work split.  No workgroup barriers: a wave's LDS operations complete in issue order.
#include "common.hpp"

namespace sebm25 {

namespace {

constexpr int WIN = 2048;  // unit ids per window = 64 lanes x 32 bitmap bits

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// order-preserving map double -> uint64 (larger double = larger key); every finite value > 0
__device__ __forceinline__ uint64_t score_key(double s) {
    const uint64_t b = (uint64_t)__double_as_longlong(s);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_score(uint64_t k) {
    const uint64_t b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
// "a ranks before b": score DESC, then tie-break key ASC
__device__ __forceinline__ bool better(uint64_t ka, uint32_t ta, uint64_t kb, uint32_t tb) {
    return ka > kb || (ka == kb && ta < tb);
}

// Candidate buffer of one wave (LDS).  Slots [0, n) are live; after cut() they are sorted best
// first and n <= k.  offer() adds up to 64 candidates behind a cut, so CAP must be at least k + 64.
// HAS_TIE = false: the tie-break key of every candidate equals its unit (no url ranks were given), so
// it is not stored.
template <int CAP, bool HAS_TIE = true>
struct TopK {
    uint64_t* key;
    uint32_t* unit;
    uint32_t* tie_;          // only when HAS_TIE
    unsigned char* mat;      // matched terms (at most 64)
    uint32_t n;
    uint64_t thr_key;  // a candidate must rank before (thr_key, thr_tie) to matter
    uint32_t thr_tie;
    double thr_score;  // thr_key as a score (-inf while fewer than k candidates are held): a cheap first test

    __device__ void init(unsigned char* lds) {
        key = reinterpret_cast<uint64_t*>(lds);
        unit = reinterpret_cast<uint32_t*>(lds + (size_t)CAP * 8);
        tie_ = unit + CAP;
        mat = reinterpret_cast<unsigned char*>(unit + (HAS_TIE ? 2 : 1) * CAP);
        n = 0;
        thr_key = 0;
        thr_tie = 0xFFFFFFFFu;
        thr_score = -__builtin_huge_val();
    }
    static constexpr size_t bytes() { return ((size_t)CAP * (8 + 4 + (HAS_TIE ? 4 : 0) + 1) + 15) / 16 * 16; }
    __device__ uint32_t tie(uint32_t i) const { return HAS_TIE ? tie_[i] : unit[i]; }

    // one compare-exchange of the sorting network: the better entry goes to slot lo
    __device__ __forceinline__ void order(uint32_t lo, uint32_t hi) {
        const uint64_t klo = key[lo], khi = key[hi];
        const uint32_t ul = unit[lo], uh = unit[hi];
        const uint32_t tlo = HAS_TIE ? tie_[lo] : ul, thi = HAS_TIE ? tie_[hi] : uh;
        if (better(khi, thi, klo, tlo)) {
            key[lo] = khi;
            key[hi] = klo;
            if (HAS_TIE) {
                tie_[lo] = thi;
                tie_[hi] = tlo;
            }
            unit[lo] = uh;
            unit[hi] = ul;
            const unsigned char ml = mat[lo], mh = mat[hi];
            mat[lo] = mh;
            mat[hi] = ml;
        }
    }

    // Wave-wide bitonic sort of all CAP slots (dead slots hold the worst key), best first.  Every
    // comparator puts the better entry into the lower slot (a merge starts with a mirror stage
    // instead of sorting half the blocks the other way), so the network for the next power of two
    // sorts any CAP: comparators that reach past CAP would compare with a worst entry that stays
    // where it is, and are skipped.
    __device__ void cut(uint32_t k) {
        constexpr uint32_t P = CAP <= 64 ? 64 : CAP <= 128 ? 128 : CAP <= 256 ? 256 : CAP <= 512 ? 512 : CAP <= 1024 ? 1024
                               : CAP <= 2048 ? 2048 : 4096;
        static_assert(CAP <= 4096 && CAP % 64 == 0, "candidate buffer sizes are multiples of 64 up to 4096");
        const uint32_t lane = lane_id();
        for (uint32_t i = n + lane; i < CAP; i += 64) {
            key[i] = 0;
            if (HAS_TIE) tie_[i] = 0xFFFFFFFFu;
            unit[i] = 0xFFFFFFFFu;
            mat[i] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t lg = 1; (1u << lg) <= P; ++lg) {
            const uint32_t size = 1u << lg, half = size >> 1;
#pragma unroll 1
            for (uint32_t e = lane; e < P / 2; e += 64) {  // mirror stage
                const uint32_t blk = (e >> (lg - 1u)) << lg, i = e & (half - 1u);
                const uint32_t lo = blk + i, hi = blk + size - 1u - i;
                if (hi < CAP) order(lo, hi);
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t stride = half >> 1; stride > 0; stride >>= 1) {
#pragma unroll 1
                for (uint32_t e = lane; e < P / 2; e += 64) {
                    const uint32_t lo = ((e & ~(stride - 1)) << 1) | (e & (stride - 1));
                    const uint32_t hi = lo | stride;
                    if (hi < CAP) order(lo, hi);
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (n > k) n = k;
        if (n == k && k > 0) {
            thr_key = key[k - 1];
            thr_tie = tie(k - 1);
            thr_score = key_score(thr_key);
        }
    }

    // every lane may offer one candidate; wave-uniform control flow
    __device__ void offer(bool has, uint64_t ck, uint32_t ct, uint32_t cu, uint32_t cm, uint32_t k) {
        bool pass = has && better(ck, ct, thr_key, thr_tie);
        uint64_t pm = __ballot(pass);
        if (pm == 0) return;
        if (n + (uint32_t)__popcll(pm) > CAP) {
            cut(k);
            pass = has && better(ck, ct, thr_key, thr_tie);
            pm = __ballot(pass);
        }
        if (pass) {
            const uint32_t pos = n + (uint32_t)__popcll(pm & lanemask_lt());
            key[pos] = ck;
            if (HAS_TIE) tie_[pos] = ct;
            unit[pos] = cu;
            mat[pos] = (unsigned char)cm;
        }
        n += (uint32_t)__popcll(pm);
        __builtin_amdgcn_wave_barrier();
    }

    __device__ void write_out(uint32_t k, uint32_t* o_unit, double* o_score, uint32_t* o_mat, uint32_t* o_tie,
                              uint32_t* o_count) {
        cut(k);
        for (uint32_t i = lane_id(); i < k; i += 64) {
            const bool live = i < n;
            o_unit[i] = live ? unit[i] : 0xFFFFFFFFu;
            o_score[i] = live ? key_score(key[i]) : 0.0;
            o_mat[i] = live ? mat[i] : 0u;
            if (o_tie) o_tie[i] = live ? tie(i) : 0xFFFFFFFFu;
        }
        if (lane_id() == 0) *o_count = n;
    }
};

// first index in [b, e) whose unit is >= x (per-lane binary search in global memory)
__device__ __forceinline__ uint32_t lower_bound_unit(const uint32_t* post_unit, uint32_t b, uint32_t e, uint32_t x) {
    while (b < e) {
        const uint32_t mid = b + ((e - b) >> 1);
        if (post_unit[mid] < x) b = mid + 1; else e = mid;
    }
    return b;
}

template <int CAP>
__global__ __launch_bounds__(64) void k_score(ScoreArgs a) {
    __shared__ __align__(16) double acc[WIN];
    __shared__ unsigned char cnt[WIN];
    __shared__ uint32_t touched[WIN / 32];
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP>::bytes()];

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint32_t T = it.n_terms;

    // lane t < T keeps the cursor of list t
    uint32_t my_cur = 0, my_end = 0;
    double my_idf = 0.0;
    if (lane < T) {
        const uint32_t term = a.q_terms[it.term_off + lane];
        const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
        const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
        my_cur = lower_bound_unit(a.post_unit, b, e, it.unit_lo);
        my_end = (it.unit_hi == 0xFFFFFFFFu) ? e : lower_bound_unit(a.post_unit, my_cur, e, it.unit_hi);
        my_idf = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
    }
    touched[lane] = 0;
    TopK<CAP> tk;
    tk.init(topk_mem);
    __builtin_amdgcn_wave_barrier();

    while (true) {
        // window start: the smallest unit any list still has to deliver
        uint32_t nxt = (lane < T && my_cur < my_end) ? a.post_unit[my_cur] : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nxt = min(nxt, (uint32_t)__shfl_xor(nxt, off));
        if (nxt == 0xFFFFFFFFu) break;
        const uint32_t base = nxt & ~31u;
        const uint64_t wend = (uint64_t)base + WIN;

        for (uint32_t t = 0; t < T; ++t) {
            uint32_t c = __shfl(my_cur, (int)t);
            const uint32_t e = __shfl(my_end, (int)t);
            const double idf = __shfl(my_idf, (int)t);
            while (true) {
                const uint32_t i = c + lane;
                const bool valid = i < e;
                const uint32_t u = valid ? a.post_unit[i] : 0xFFFFFFFFu;
                const bool in = valid && (uint64_t)u < wend;
                const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
                const uint32_t w = in ? a.post_w[i] : POST_HOLE;
                if (w != POST_HOLE) {  // a hole only moves the cursor
                    const double v = __dmul_rn(idf, (double)w);  // idf holds idf / 10, weights are in tenths
                    const uint32_t idx = u - base;
                    const uint32_t bit = 1u << (idx & 31u);
                    const uint32_t old = atomicOr(&touched[idx >> 5], bit);
                    if (old & bit) {
                        acc[idx] = __dadd_rn(acc[idx], v);
                        cnt[idx] = (unsigned char)(cnt[idx] + 1);
                    } else {
                        acc[idx] = v;
                        cnt[idx] = 1;
                    }
                }
                c += n_in;
                if (n_in < 64) break;
            }
            if (lane == t) my_cur = c;
            __builtin_amdgcn_wave_barrier();
        }

        // harvest the window: lane l owns units [base + 32 l, base + 32 l + 32)
        uint32_t word = touched[lane];
        touched[lane] = 0;
        while (__ballot(word != 0)) {
            const bool has = word != 0;
            const uint32_t b = has ? (uint32_t)__ffs((int)word) - 1u : 0u;
            word &= word - 1u;
            const uint32_t idx = lane * 32u + b;
            double s = 0.0;
            uint32_t m = 0, unit = 0, tie = 0;
            if (has) {
                s = acc[idx];
                m = cnt[idx];
                unit = base + idx;
                tie = a.url_rank ? a.url_rank[unit] : unit + a.unit_base;
            }
            tk.offer(has, score_key(s), tie, unit + a.unit_base, m, a.k);
        }
    }

    const size_t o = (size_t)it.out_slot * a.k;
    if (it.direct)
        tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                     a.out_count + it.out_slot);
    else
        tk.write_out(a.k, a.tmp_unit + o, a.tmp_score + o, a.tmp_matched + o, a.tmp_tie + o, a.tmp_count + it.out_slot);
}

// Fast path for queries with at most FAST_T lists.  Each list is read in chunks of 64 slots
// aligned to 64 (one coalesced load of units, one of weights); three chunks per list sit in
// registers and the fourth is requested the moment the first is used up, two chunks before its
// data is needed, so a window normally touches no global memory on its critical path.  The doc
// ids first touched in a window are appended to a dense list, so the harvest offers 64
// candidates per step instead of scanning a sparse bitmap.
constexpr int FAST_T = 4;

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

template <int CAP, int WINSZ>
__global__ __launch_bounds__(64) void k_score_fast(ScoreArgs a) {
    __shared__ __align__(16) double acc[WINSZ];
    __shared__ unsigned char cnt[WINSZ];
    __shared__ uint32_t touched[WINSZ / 32];
    __shared__ unsigned short olist[WINSZ];
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP>::bytes()];

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t T = it.n_terms;
    constexpr uint32_t INF = 0xFFFFFFFFu;

    uint32_t my_cur = 0, my_end = 0;
    double my_idf = 0.0;
    if (lane < T) {
        const uint32_t term = a.q_terms[it.term_off + lane];
        const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
        const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
        my_cur = lower_bound_unit(a.post_unit, b, e, it.unit_lo);
        my_end = (it.unit_hi == INF) ? e : lower_bound_unit(a.post_unit, my_cur, e, it.unit_hi);
        my_idf = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
    }
    // per list: slot index of the current chunk's lane 0, first unconsumed lane of that chunk, end
    // of the list, and three chunks of units / weights (slots outside [cursor, end) read as INF)
    uint32_t pos[FAST_T], off[FAST_T], end[FAST_T];
    uint32_t u0[FAST_T], u1[FAST_T], u2[FAST_T], w0[FAST_T], w1[FAST_T], w2[FAST_T];
    double idf[FAST_T];
    auto load_chunk = [&](int t, uint32_t p, uint32_t lo, uint32_t& u, uint32_t& w) {
        const uint32_t i = p + lane;
        const bool ok = i >= lo && i < end[t];
        u = ok ? a.post_unit[i] : INF;
        w = ok ? a.post_w[i] : POST_HOLE;
    };
#pragma unroll
    for (int t = 0; t < FAST_T; ++t) {
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)my_cur, t);
        end[t] = (uint32_t)__builtin_amdgcn_readlane((int)my_end, t);
        idf[t] = readlane_f64(my_idf, t);
        pos[t] = cur & ~63u;
        off[t] = cur & 63u;
        if (cur >= end[t]) {  // empty list (or t >= T): nothing to read
            pos[t] = 0;
            off[t] = 0;
            end[t] = 0;
        }
        load_chunk(t, pos[t], cur, u0[t], w0[t]);
        load_chunk(t, pos[t] + 64u, cur, u1[t], w1[t]);
        load_chunk(t, pos[t] + 128u, cur, u2[t], w2[t]);
    }
    if (lane < WINSZ / 32) touched[lane] = 0;
    TopK<CAP> tk;
    tk.init(topk_mem);
    uint32_t n_o = 0;
    __builtin_amdgcn_wave_barrier();

    while (true) {
        // window start: the smallest unit any list still has to deliver
        uint32_t nxt = INF;
#pragma unroll
        for (int t = 0; t < FAST_T; ++t)
            if ((uint32_t)t < T) nxt = min(nxt, (uint32_t)__builtin_amdgcn_readlane((int)u0[t], (int)off[t]));
        if (nxt == INF) break;
        const uint32_t base = nxt & ~31u;
        const uint64_t wend = (uint64_t)base + WINSZ;

#pragma unroll
        for (int t = 0; t < FAST_T; ++t) {
            if ((uint32_t)t >= T) break;
            while (true) {
                // the unconsumed slots of the current chunk that fall into the window
                const bool in = lane >= off[t] && u0[t] != INF && (uint64_t)u0[t] < wend;
                const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
                bool first = false;
                uint32_t idx = 0;
                if (in && w0[t] != POST_HOLE) {  // a hole only moves the cursor
                    const double v = __dmul_rn(idf[t], (double)w0[t]);
                    idx = u0[t] - base;
                    const uint32_t bit = 1u << (idx & 31u);
                    const uint32_t old = atomicOr(&touched[idx >> 5], bit);
                    first = !(old & bit);
                    if (first) {
                        acc[idx] = v;
                        cnt[idx] = 1;
                    } else {
                        acc[idx] = __dadd_rn(acc[idx], v);
                        cnt[idx] = (unsigned char)(cnt[idx] + 1);
                    }
                }
                const uint64_t om = __ballot(first);
                if (first) olist[n_o + (uint32_t)__popcll(om & lt)] = (unsigned short)idx;
                n_o += (uint32_t)__popcll(om);
                off[t] += n_in;
                if (off[t] < 64u) break;
                // chunk used up: the next two are in registers already, request the one after them
                u0[t] = u1[t]; w0[t] = w1[t];
                u1[t] = u2[t]; w1[t] = w2[t];
                pos[t] += 64u;
                off[t] = 0;
                load_chunk(t, pos[t] + 128u, 0u, u2[t], w2[t]);
                __builtin_amdgcn_wave_barrier();
            }
            __builtin_amdgcn_wave_barrier();
        }

        // harvest: every doc id touched in this window, 64 per step
        for (uint32_t o0 = 0; o0 < n_o; o0 += 64) {
            const bool has = o0 + lane < n_o;
            double s = 0.0;
            uint32_t m = 0, unit = 0, tie = 0;
            if (has) {
                const uint32_t idx = olist[o0 + lane];
                s = acc[idx];
                m = cnt[idx];
                touched[idx >> 5] = 0;
                unit = base + idx;
                tie = a.url_rank ? a.url_rank[unit] : unit + a.unit_base;
            }
            tk.offer(has, score_key(s), tie, unit + a.unit_base, m, a.k);
        }
        n_o = 0;
        __builtin_amdgcn_wave_barrier();
    }

    const size_t o = (size_t)it.out_slot * a.k;
    if (it.direct)
        tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                     a.out_count + it.out_slot);
    else
        tk.write_out(a.k, a.tmp_unit + o, a.tmp_score + o, a.tmp_matched + o, a.tmp_tie + o, a.tmp_count + it.out_slot);
}

// Fast path in posting space.  Instead of windows of unit ids (whose lanes are mostly idle when the
// lists are sparse: a 512-unit window holds about 14 postings of a typical list), a step takes,
// from every list, all slots up to a frontier F = the smallest "last unit of the chunk in
// registers" over the lists.  Every posting of a unit <= F is then in this step, whichever list
// it is in, so the step's sums are final; the list that sets F uses all 64 lanes.  Units are
// accumulated in a small open-addressing hash table in LDS (at most FAST_T x 64 distinct units per
// step), lists in query-term order, so a score is the same double sum as on every other path.
constexpr int HASH_SLOTS = 512;
constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

// HAS_RANK: url ranks were given (a.url_rank != null); without them a unit's tie-break key is its id
// waves per SIMD the register budget is held to: what the kernel's LDS allows anyway
constexpr int hash_waves_per_simd(int cap) { return cap <= 128 ? 4 : cap <= 512 ? 3 : 1; }

template <int CAP, bool HAS_RANK>
__global__ __launch_bounds__(64, hash_waves_per_simd(CAP)) void k_score_hash(ScoreArgs a) {
    __shared__ __align__(16) double acc[HASH_SLOTS];
    __shared__ uint32_t tag[HASH_SLOTS];
    __shared__ unsigned char cnt[HASH_SLOTS];
    __shared__ unsigned short used[FAST_T * 64];
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP, HAS_RANK>::bytes()];

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t T = it.n_terms;
    constexpr uint32_t INF = 0xFFFFFFFFu;

    uint32_t my_cur = 0, my_end = 0;
    double my_idf = 0.0;
    if (lane < T) {
        const uint32_t term = a.q_terms[it.term_off + lane];
        const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
        const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
        my_cur = lower_bound_unit(a.post_unit, b, e, it.unit_lo);
        my_end = (it.unit_hi == INF) ? e : lower_bound_unit(a.post_unit, my_cur, e, it.unit_hi);
        my_idf = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
    }
    // per list: slot index of the current chunk's lane 0, first unconsumed lane of that chunk, end
    // of the list, and three aligned chunks of units / weights (slots outside [cursor, end) read INF)
    uint32_t pos[FAST_T], off[FAST_T], end[FAST_T];
    uint32_t u0[FAST_T], u1[FAST_T], u2[FAST_T], w0[FAST_T], w1[FAST_T], w2[FAST_T];
    double idf[FAST_T];
    auto load_chunk = [&](int t, uint32_t p, uint32_t lo, uint32_t& u, uint32_t& w) {
        const uint32_t i = p + lane;
        const bool ok = i >= lo && i < end[t];
        u = ok ? a.post_unit[i] : INF;
        w = ok ? a.post_w[i] : POST_HOLE;
    };
#pragma unroll
    for (int t = 0; t < FAST_T; ++t) {
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)my_cur, t);
        end[t] = (uint32_t)__builtin_amdgcn_readlane((int)my_end, t);
        idf[t] = readlane_f64(my_idf, t);
        pos[t] = cur & ~63u;
        off[t] = cur & 63u;
        if (cur >= end[t]) {  // empty list (or t >= T): nothing to read
            pos[t] = 0;
            off[t] = 0;
            end[t] = 0;
        }
        load_chunk(t, pos[t], cur, u0[t], w0[t]);
        load_chunk(t, pos[t] + 64u, cur, u1[t], w1[t]);
        load_chunk(t, pos[t] + 128u, cur, u2[t], w2[t]);
    }
    for (uint32_t x = lane; x < HASH_SLOTS; x += 64) tag[x] = HASH_EMPTY;
    TopK<CAP, HAS_RANK> tk;
    tk.init(topk_mem);
    __builtin_amdgcn_wave_barrier();

    while (true) {
        // frontier: the smallest last unit of the chunks at hand (INF for a list whose rest is in registers)
        uint32_t frontier = INF, any = INF;
#pragma unroll
        for (int t = 0; t < FAST_T; ++t) {
            if ((uint32_t)t < T) {
                any = min(any, (uint32_t)__builtin_amdgcn_readlane((int)u0[t], (int)off[t]));
                frontier = min(frontier, (uint32_t)__builtin_amdgcn_readlane((int)u0[t], 63));
            }
        }
        if (any == INF) break;  // every list is used up

        uint32_t n_used = 0;
#pragma unroll
        for (int t = 0; t < FAST_T; ++t) {
            if ((uint32_t)t >= T) break;
            const bool in = lane >= off[t] && u0[t] != INF && u0[t] <= frontier;
            const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
            bool fresh = false;
            uint32_t h = 0;
            if (in && w0[t] != POST_HOLE) {  // a hole only moves the cursor
                const uint32_t u = u0[t];
                const double v = __dmul_rn(idf[t], (double)w0[t]);
                h = (u * 2654435761u) >> 23;  // 9 bits: HASH_SLOTS = 512
                while (true) {
                    const uint32_t old = atomicCAS(&tag[h], HASH_EMPTY, u);
                    if (old == HASH_EMPTY) {
                        fresh = true;
                        break;
                    }
                    if (old == u) break;
                    h = (h + 1u) & (HASH_SLOTS - 1u);
                }
                if (fresh) {
                    acc[h] = v;
                    cnt[h] = 1;
                } else {
                    acc[h] = __dadd_rn(acc[h], v);
                    cnt[h] = (unsigned char)(cnt[h] + 1);
                }
            }
            const uint64_t fm = __ballot(fresh);
            if (fresh) used[n_used + (uint32_t)__popcll(fm & lt)] = (unsigned short)h;
            n_used += (uint32_t)__popcll(fm);
            off[t] += n_in;
            if (off[t] >= 64u) {
                // chunk used up: the next two are in registers already, request the one after them
                u0[t] = u1[t]; w0[t] = w1[t];
                u1[t] = u2[t]; w1[t] = w2[t];
                pos[t] += 64u;
                off[t] = 0;
                load_chunk(t, pos[t] + 128u, 0u, u2[t], w2[t]);
            }
            __builtin_amdgcn_wave_barrier();
        }

        // harvest: every unit of this step, 64 per round; the table is left empty again
        for (uint32_t o0 = 0; o0 < n_used; o0 += 64) {
            const bool has = o0 + lane < n_used;
            double s = 0.0;
            uint32_t h = 0, unit = 0;
            if (has) {
                h = used[o0 + lane];
                s = acc[h];
                unit = tag[h];
                tag[h] = HASH_EMPTY;
            }
            // most units cannot enter the top k any more: test the score alone before anything else
            const bool may = has && s >= tk.thr_score;
            if (__ballot(may)) {
                uint32_t m = 0, tie = 0;
                if (may) {
                    m = cnt[h];
                    tie = HAS_RANK ? a.url_rank[unit] : unit + a.unit_base;
                }
                tk.offer(may, score_key(s), tie, unit + a.unit_base, m, a.k);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    const size_t o = (size_t)it.out_slot * a.k;
    if (it.direct)
        tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                     a.out_count + it.out_slot);
    else
        tk.write_out(a.k, a.tmp_unit + o, a.tmp_score + o, a.tmp_matched + o, a.tmp_tie + o, a.tmp_count + it.out_slot);
}

// one wave per query: merges n[q] result lists (each best-first, <= k entries) into one
template <int CAP>
__global__ __launch_bounds__(64) void k_merge(MergeArgs a) {
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP>::bytes()];
    const uint32_t q = blockIdx.x;
    const uint32_t nl = a.n[q];
    if (nl == 0) return;
    TopK<CAP> tk;
    tk.init(topk_mem);
    const uint32_t lane = lane_id();
    for (uint32_t l = 0; l < nl; ++l) {
        const size_t list = (size_t)a.first[q] + (size_t)l * a.list_stride;
        const uint32_t c = min(a.in_count[list], a.k);
        for (uint32_t i0 = 0; i0 < c; i0 += 64) {
            const uint32_t i = i0 + lane;
            const bool has = i < c;
            const size_t o = list * a.k + i;
            const double s = has ? a.in_score[o] : 0.0;
            tk.offer(has, score_key(s), has ? a.in_tie[o] : 0u, has ? a.in_unit[o] : 0u, has ? a.in_matched[o] : 0u, a.k);
        }
    }
    const size_t o = (size_t)q * a.k;
    tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                 a.out_count + q);
}

}  // namespace

// items [0, n_fast) have at most FAST_T lists, the rest use the general kernel
void launch_score(const ScoreArgs& a, uint32_t n_fast, int window, hipStream_t st) {
    if (n_fast && window == 0) {  // posting-space steps with hashed accumulators
        dim3 grid(n_fast), block(64);
        if (a.url_rank) {
            if (a.k <= 64) hipLaunchKernelGGL((k_score_hash<128, true>), grid, block, 0, st, a);
            else if (a.k <= 256) hipLaunchKernelGGL((k_score_hash<320, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_score_hash<2112, true>), grid, block, 0, st, a);
        } else {
            if (a.k <= 64) hipLaunchKernelGGL((k_score_hash<128, false>), grid, block, 0, st, a);
            else if (a.k <= 256) hipLaunchKernelGGL((k_score_hash<320, false>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_score_hash<2112, false>), grid, block, 0, st, a);
        }
    } else if (n_fast) {
        dim3 grid(n_fast), block(64);
        if (a.k <= 64) {
            if (window <= 512) hipLaunchKernelGGL((k_score_fast<128, 512>), grid, block, 0, st, a);
            else if (window <= 1024) hipLaunchKernelGGL((k_score_fast<128, 1024>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_score_fast<128, 2048>), grid, block, 0, st, a);
        } else if (a.k <= 256) {
            if (window <= 1024) hipLaunchKernelGGL((k_score_fast<320, 1024>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_score_fast<320, 2048>), grid, block, 0, st, a);
        } else {
            hipLaunchKernelGGL((k_score_fast<2112, 1024>), grid, block, 0, st, a);
        }
    }
    if (a.n_items > n_fast) {
        ScoreArgs b = a;
        b.items = a.items + n_fast;
        b.n_items = a.n_items - n_fast;
        dim3 grid(b.n_items), block(64);
        if (a.k <= 64)
            hipLaunchKernelGGL(k_score<128>, grid, block, 0, st, b);
        else if (a.k <= 256)
            hipLaunchKernelGGL(k_score<320>, grid, block, 0, st, b);
        else
            hipLaunchKernelGGL(k_score<2112>, grid, block, 0, st, b);
    }
}

uint32_t score_fast_max_terms() { return FAST_T; }

void launch_merge(const MergeArgs& a, hipStream_t st) {
    if (a.n_queries == 0) return;
    dim3 grid(a.n_queries), block(64);
    if (a.k <= 64)
        hipLaunchKernelGGL(k_merge<128>, grid, block, 0, st, a);
    else if (a.k <= 256)
        hipLaunchKernelGGL(k_merge<320>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(k_merge<2112>, grid, block, 0, st, a);
}

}  // namespace sebm25
