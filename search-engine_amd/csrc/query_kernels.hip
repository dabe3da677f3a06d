// Batched scoring + top-k for gfx950.  Replaces the scoring statements the reference sends to
// PostgreSQL (app/api/search_service.py:13-40 web, :42-66 fallback, :68-101 news):
//   score(unit) = SUM over query terms of idf_used[t] * w(t, unit),  matched = COUNT(DISTINCT term),
//   ORDER BY score DESC, url ASC LIMIT k.
//
// One wavefront owns one work item = (query, range of unit ids).  A term's posting list is a range of
// slots sorted by unit (a few of them holes that carry no posting).  All lists of a query advance together
// to a common frontier, 64 coalesced slots per list per load.  Queries of two to four lists take the
// merge-join kernel (k_score_join): the lists' blocks sit in LDS as sorted arrays and a posting finds its
// unit's other postings by binary search.  Queries of one list and of five to eight lists take the table
// kernel (k_score_step): integer accumulators in a small LDS hash table.  Both keep the best k in registers
// (small k) or in an LDS buffer that is merged, not re-sorted, when it fills.  Longer queries take the
// general kernel (k_score), which walks windows of WIN consecutive unit ids with directly indexed
// accumulators.  Lists are always added in query-term order, so a score is the same double sum whatever the
// work split.  No workgroup barriers: a wave's LDS operations complete in issue order.
#include "common.hpp"

namespace sebm25 {

namespace {

constexpr int WIN = 2048;  // unit ids per window = 64 lanes x 32 bitmap bits

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// order-preserving map double -> uint64 (larger double = larger key); every finite value > 0
// (a zero of either sign maps to one key: the SQL comparison treats -0.0 and 0.0 as equal, so they tie)
__device__ __forceinline__ uint64_t score_key(double s) {
    const uint64_t b = s == 0.0 ? 0ull : (uint64_t)__double_as_longlong(s);
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

// Tie-break key of a unit when keys were given (sebm25_set_url_rank / sebm25_set_url_rank_appended): the first `base`
// units kept their place in url order and need no array (key (unit + 1) << shift), the others read theirs.
__device__ __forceinline__ uint32_t rank_key(const uint32_t* rank, uint32_t base, uint32_t shift, uint32_t unit) {
    return unit < base ? (unit + 1u) << shift : rank[unit - base];
}

// value of lane (l ^ MASK): lane-exchange forms that stay in the vector pipeline where the hardware has one
// (quad permutes, row mirrors), a swizzle for the 16 / 31 patterns, the LDS crossbar only across the wave halves
template <uint32_t MASK>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    auto dpp = [](uint32_t x, auto ctrl) { return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, decltype(ctrl)::value, 0xF, 0xF, false); };
    using std::integral_constant;
    if constexpr (MASK == 1) return dpp(v, integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
    else if constexpr (MASK == 2) return dpp(v, integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
    else if constexpr (MASK == 3) return dpp(v, integral_constant<int, 0x1B>{});   // quad_perm [3,2,1,0]
    else if constexpr (MASK == 7) return dpp(v, integral_constant<int, 0x141>{});  // row_half_mirror
    else if constexpr (MASK == 15) return dpp(v, integral_constant<int, 0x140>{}); // row_mirror
    else if constexpr (MASK == 4) return dpp(dpp(v, integral_constant<int, 0x141>{}), integral_constant<int, 0x1B>{});   // ^7 then ^3
    else if constexpr (MASK == 8) return dpp(dpp(v, integral_constant<int, 0x140>{}), integral_constant<int, 0x141>{});  // ^15 then ^7
    else if constexpr (MASK == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
    else if constexpr (MASK == 31) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x7C1F);
    else return (uint32_t)__shfl_xor((int)v, (int)MASK);
}

// Candidate buffer of one wave (LDS) of CAP = S + W entries, for k up to S (S = the largest power of two that
// leaves 32 <= W <= 64).  Slots [0, S) hold the best candidates seen so far, sorted best first (dead slots hold
// the worst key); slots [S, S + W) collect new candidates.  When those are full, cut() sorts them in registers
// (a 64-lane bitonic network of lane-exchange steps) and merges them into the sorted part: one mirror stage
// against the sorted part's last 64 entries, done from the registers (what loses there is dropped, so nothing
// goes back to the waiting slots), then the half-cleaner stages of a bitonic merge over [0, S).
// HAS_TIE = false: the tie-break key of every candidate equals its unit (no url ranks were given),
// so it is not stored.  LAZY_ONLY: tie-break keys are always read at cut time (set_lazy_ties must be called
// before the first offer), so the waiting slots hold none.
constexpr uint32_t topk_sorted_size(uint32_t cap) {
    uint32_t s = 64;
    while (s * 2 + 32 <= cap) s *= 2;
    return s;
}
// MAT: the type the count of matching terms is held in (8 bits for the kernels of up to eight lists; the general
// kernel and the merge of finished lists, where a query may hold any number of terms, keep 32).
template <int CAP, bool HAS_TIE = true, bool LAZY_ONLY = false, typename MAT = unsigned char>
struct TopK {
    static constexpr uint32_t S = topk_sorted_size(CAP), W = CAP - S;
    static_assert(W >= 32 && W <= 64 && (!LAZY_ONLY || HAS_TIE), "32 to 64 waiting slots");
    uint64_t* key;
    uint32_t* unit;
    uint32_t* tie_;          // only when HAS_TIE
    MAT* mat;                // matched terms
    uint32_t ns, np;         // live entries of the sorted part, entries waiting behind it
    uint64_t thr_key;  // a candidate must rank before (thr_key, thr_tie) to matter
    uint32_t thr_tie;
    double thr_score;  // thr_key as a score (-inf while fewer than k candidates are held): a cheap first test
    // Lazy ties: the tie-break key of a candidate is rank[unit - base], a random 4-byte read.  When set, offer()
    // takes no key (it lets every candidate in whose score reaches the k-th best) and cut() reads the keys of the
    // waiting candidates in one go: one memory latency per 64 candidates instead of one per offer.
    const uint32_t* lazy_rank;
    uint32_t lazy_base;                   // global id of the index's first unit
    uint32_t lazy_kbase, lazy_kshift;     // units that need no array (rank_key)
    static constexpr bool CAN_LAZY = HAS_TIE;
    __device__ void set_lazy_ties(const uint32_t* rank, uint32_t base, uint32_t kbase, uint32_t kshift) {
        lazy_rank = rank;
        lazy_base = base;
        lazy_kbase = kbase;
        lazy_kshift = kshift;
    }
    __device__ __forceinline__ uint32_t lazy_key(uint32_t global_unit) const {
        return rank_key(lazy_rank, lazy_kbase, lazy_kshift, global_unit - lazy_base);
    }

    __device__ void init(unsigned char* lds) {
        lazy_rank = nullptr;
        lazy_base = lazy_kbase = lazy_kshift = 0;
        key = reinterpret_cast<uint64_t*>(lds);
        unit = reinterpret_cast<uint32_t*>(lds + (size_t)CAP * 8);
        tie_ = unit + CAP;
        mat = reinterpret_cast<MAT*>(unit + CAP + TIES);
        ns = np = 0;
        thr_key = 0;
        thr_tie = 0xFFFFFFFFu;
        thr_score = -__builtin_huge_val();
        for (uint32_t i = lane_id(); i < S; i += 64) {
            key[i] = 0;
            if (HAS_TIE && !LAZY_ONLY) tie_[i] = 0xFFFFFFFFu;
            unit[i] = 0xFFFFFFFFu;
            mat[i] = 0;
        }
        __builtin_amdgcn_wave_barrier();
    }
    static constexpr uint32_t TIES = (HAS_TIE && !LAZY_ONLY) ? (uint32_t)CAP : 0u;  // tie-break keys held
    static constexpr size_t bytes() { return ((size_t)CAP * (8 + 4 + sizeof(MAT)) + (size_t)TIES * 4 + 15) / 16 * 16; }
#ifdef SEBM25_MIDF_LDS  // experiment builds only
    static constexpr bool FILTER_MIDDLE_LISTS = true;
#else
    static constexpr bool FILTER_MIDDLE_LISTS = false;  // measured: the lower, lazily refreshed k-th best score filters too little
#endif
    // LAZY_ONLY keeps no tie-break keys at all: a key is the rank of the entry's unit, read when two scores tie
    // (and once per cut for the k-th best); a dead entry ranks behind everything
    __device__ __forceinline__ uint32_t tie_of_unit(uint32_t u) const { return u == 0xFFFFFFFFu ? 0xFFFFFFFFu : lazy_key(u); }
    __device__ uint32_t tie(uint32_t i) const { return HAS_TIE ? (LAZY_ONLY ? tie_of_unit(unit[i]) : tie_[i]) : unit[i]; }

    // one compare-exchange of the merge network: the better entry goes to slot lo
    __device__ __forceinline__ void order(uint32_t lo, uint32_t hi) {
        const uint64_t klo = key[lo], khi = key[hi];
        const uint32_t ul = unit[lo], uh = unit[hi];
        uint32_t tlo = ul, thi = uh;
        if (HAS_TIE && !LAZY_ONLY) {
            tlo = tie_[lo];
            thi = tie_[hi];
        } else if (LAZY_ONLY && khi == klo) {
            tlo = tie_of_unit(ul);
            thi = tie_of_unit(uh);
        }
        if (better(khi, thi, klo, tlo)) {
            key[lo] = khi;
            key[hi] = klo;
            if (HAS_TIE && !LAZY_ONLY) {
                tie_[lo] = thi;
                tie_[hi] = tlo;
            }
            unit[lo] = uh;
            unit[hi] = ul;
            const MAT ml = mat[lo], mh = mat[hi];
            mat[lo] = mh;
            mat[hi] = ml;
        }
    }

    __device__ __forceinline__ void cut(uint32_t k) {
        const uint32_t lane = lane_id();
        if (np) {
            // ---- the waiting entries, one per lane, sorted best first across the lanes
            const bool live = lane < np;
            uint64_t kk = live ? key[S + lane] : 0ull;
            uint32_t uu = live ? unit[S + lane] : 0xFFFFFFFFu;
            uint32_t tt = uu;
            if (HAS_TIE) {
                tt = 0xFFFFFFFFu;
                if (live) tt = (LAZY_ONLY || lazy_rank) ? lazy_key(uu) : tie_[S + lane];
            }
            uint32_t mm = live ? mat[S + lane] : 0u;
            // one compare-exchange stage: partner = lane ^ MASK, TOP = highest bit of MASK
#define SEBM25_EXCHANGE(MASK, TOP)                                                                       \
            {                                                                                             \
                const uint32_t plo = lane_xor<MASK>((uint32_t)kk), phi = lane_xor<MASK>((uint32_t)(kk >> 32)); \
                const uint32_t pu = lane_xor<MASK>(uu);                                                   \
                const uint32_t pt = HAS_TIE ? lane_xor<MASK>(tt) : pu;                                    \
                const uint32_t pmat = lane_xor<MASK>(mm);                                                 \
                const uint64_t pk = ((uint64_t)phi << 32) | plo;                                          \
                const bool lower = (lane & (TOP)) == 0u;                                                  \
                const bool partner_better = better(pk, pt, kk, tt), mine_better = better(kk, tt, pk, pt); \
                if (lower ? partner_better : mine_better) { /* the lower lane keeps the better entry */    \
                    kk = pk;                                                                              \
                    uu = pu;                                                                              \
                    tt = pt;                                                                              \
                    mm = pmat;                                                                            \
                }                                                                                         \
            }
            // bitonic sort of 64 lanes: per level a mirror stage, then half-cleaners of falling stride
            SEBM25_EXCHANGE(1, 1)
            SEBM25_EXCHANGE(3, 2) SEBM25_EXCHANGE(1, 1)
            SEBM25_EXCHANGE(7, 4) SEBM25_EXCHANGE(2, 2) SEBM25_EXCHANGE(1, 1)
            SEBM25_EXCHANGE(15, 8) SEBM25_EXCHANGE(4, 4) SEBM25_EXCHANGE(2, 2) SEBM25_EXCHANGE(1, 1)
            SEBM25_EXCHANGE(31, 16) SEBM25_EXCHANGE(8, 8) SEBM25_EXCHANGE(4, 4) SEBM25_EXCHANGE(2, 2) SEBM25_EXCHANGE(1, 1)
            SEBM25_EXCHANGE(63, 32) SEBM25_EXCHANGE(16, 16) SEBM25_EXCHANGE(8, 8) SEBM25_EXCHANGE(4, 4) SEBM25_EXCHANGE(2, 2) SEBM25_EXCHANGE(1, 1)
#undef SEBM25_EXCHANGE
            // ---- merge: mirror stage (sorted entry S - 64 + lane against the waiting entry of lane 63 - lane; the
            // better one stays in the sorted part), then half-cleaners over the sorted part
            {
                const uint32_t rlo = lane_xor<63>((uint32_t)kk), rhi = lane_xor<63>((uint32_t)(kk >> 32));
                const uint32_t ru = lane_xor<63>(uu), rm = lane_xor<63>(mm);
                const uint32_t rt = HAS_TIE ? lane_xor<63>(tt) : ru;
                const uint64_t rk = ((uint64_t)rhi << 32) | rlo;
                const uint32_t i = S - 64u + lane;
                const uint64_t ki = key[i];
                if (rk > ki || (rk == ki && rt < tie(i))) {   // (the tie-break key of entry i is looked at on a tie only)
                    key[i] = rk;
                    unit[i] = ru;
                    if (HAS_TIE && !LAZY_ONLY) tie_[i] = rt;
                    mat[i] = (MAT)rm;
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t stride = S >> 1; stride > 0; stride >>= 1) {
#pragma unroll 1
                for (uint32_t e = lane; e < S / 2; e += 64) {
                    const uint32_t lo = ((e & ~(stride - 1)) << 1) | (e & (stride - 1));
                    order(lo, lo | stride);
                }
                __builtin_amdgcn_wave_barrier();
            }
            ns = min(ns + np, S);
            np = 0;
        }
        if (ns >= k && k > 0) {
            thr_key = key[k - 1];
            thr_tie = tie(k - 1);
            thr_score = key_score(thr_key);
        }
    }

    // the lanes with p set append their candidates to the waiting slots
    __device__ __forceinline__ void store(bool p, bool lazy, uint64_t ck, uint32_t ct, uint32_t cu, uint32_t cm) {
        const uint64_t m = __ballot(p);
        if (p) {
            const uint32_t pos = S + np + (uint32_t)__popcll(m & lanemask_lt());
            key[pos] = ck;
            if (HAS_TIE && !LAZY_ONLY && !lazy) tie_[pos] = ct;
            unit[pos] = cu;
            mat[pos] = (MAT)cm;
        }
        np += (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ bool passes(bool has, bool lazy, uint64_t ck, uint32_t ct) const {
        return has && (lazy ? ck >= thr_key : better(ck, ct, thr_key, thr_tie));
    }

    // every lane may offer one candidate; wave-uniform control flow
    __device__ __forceinline__ void offer(bool has, uint64_t ck, uint32_t ct, uint32_t cu, uint32_t cm, uint32_t k) {
        const bool lazy = LAZY_ONLY || (HAS_TIE && lazy_rank != nullptr);
        bool pass = passes(has, lazy, ck, ct);
        uint64_t pm = __ballot(pass);
        if (pm == 0) return;
        if (np + (uint32_t)__popcll(pm) > W) {
            cut(k);
            pass = passes(has, lazy, ck, ct);
            pm = __ballot(pass);
        }
        if (W < 64 && (uint32_t)__popcll(pm) > W) {  // more than the (now empty) waiting slots take: the lower lanes first
            store(pass && lane_id() < 32u, lazy, ck, ct, cu, cm);
            cut(k);
            pass = passes(has && lane_id() >= 32u, lazy, ck, ct);
        }
        store(pass, lazy, ck, ct, cu, cm);
    }

    __device__ void write_out(uint32_t k, uint32_t* o_unit, double* o_score, uint32_t* o_mat, uint32_t* o_tie,
                              uint32_t* o_count) {
        cut(k);
        const uint32_t n = min(ns, k);
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
__device__ __forceinline__ uint32_t lower_bound_unit(const Slot* post, uint32_t b, uint32_t e, uint32_t x) {
    while (b < e) {
        const uint32_t mid = b + ((e - b) >> 1);
        if (post[mid].x < x) b = mid + 1; else e = mid;
    }
    return b;
}

// The general form.  Up to 64 * SCORE_LIST_BLOCKS lists: lane l keeps the cursors of lists l, l + 64, l + 128, ...
// in registers; more lists: the cursors live in global memory (see the branch at the top).
constexpr int SCORE_LIST_BLOCKS = 4;
template <int CAP>
__global__ __launch_bounds__(64) void k_score(ScoreArgs a) {
    __shared__ __align__(16) double acc[WIN];
    __shared__ uint32_t cnt[WIN];
    __shared__ uint32_t touched[WIN / 32];
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP, true, false, uint32_t>::bytes()];

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint32_t T = it.n_terms;
    // one window's postings of one list into the accumulators; -> the list's cursor behind them
    auto take_list = [&](uint32_t c, uint32_t e, double idf, uint32_t base, uint64_t wend) {
        while (true) {
            const uint32_t i = c + lane;
            const bool valid = i < e;
            const Slot sl = valid ? a.post[i] : make_uint2(0xFFFFFFFFu, POST_HOLE);
            const uint32_t u = sl.x;
            const bool in = valid && (uint64_t)u < wend;
            const uint32_t n_in = (uint32_t)__popcll(__ballot(in));
            const uint32_t w = in ? sl.y : POST_HOLE;
            if (w != POST_HOLE) {  // a hole only moves the cursor
                const double v = __dmul_rn(idf, (double)w);  // idf holds idf / 10, weights are in tenths
                const uint32_t idx = u - base;
                const uint32_t bit = 1u << (idx & 31u);
                const uint32_t old = atomicOr(&touched[idx >> 5], bit);
                if (old & bit) {
                    acc[idx] = __dadd_rn(acc[idx], v);
                    cnt[idx] = cnt[idx] + 1u;
                } else {
                    acc[idx] = v;
                    cnt[idx] = 1;
                }
            }
            c += n_in;
            if (n_in < 64) break;
        }
        return c;
    };
    // the window's sums into the candidates: lane l owns units [base + 32 l, base + 32 l + 32)
    auto harvest = [&](TopK<CAP, true, false, uint32_t>& tk, uint32_t base) {
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
                tie = a.url_rank ? rank_key(a.url_rank, a.rank_base, a.rank_shift, unit) : unit + a.unit_base;
            }
            tk.offer(has, score_key(s), tie, unit + a.unit_base, m, a.k);
        }
    };
    auto write_result = [&](TopK<CAP, true, false, uint32_t>& tk) {
        const size_t o = (size_t)it.out_slot * a.k;
        if (it.direct)
            tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                         a.out_count + it.out_slot);
        else
            tk.write_out(a.k, a.tmp_unit + o, a.tmp_score + o, a.tmp_matched + o, a.tmp_tie + o, a.tmp_count + it.out_slot);
    };

    if (T > 64u * SCORE_LIST_BLOCKS) {
        // More lists than the lanes keep cursors for (`t.term = ANY($1)` takes any number of terms,
        // search_service.py:33): cursor and end of every list live in global memory, two words per list at
        // a.wide_cur + it.order, and are walked 64 lists at a time.  Same windows, same order of the sums.
        uint32_t* wc = a.wide_cur + it.order;
        for (uint32_t t = lane; t < T; t += 64u) {
            const uint32_t term = a.q_terms[it.term_off + t];
            const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
            const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
            const uint32_t c = lower_bound_unit(a.post, b, e, it.unit_lo);
            wc[2u * t] = c;
            wc[2u * t + 1u] = (it.unit_hi == 0xFFFFFFFFu) ? e : lower_bound_unit(a.post, c, e, it.unit_hi);
        }
        touched[lane] = 0;
        TopK<CAP, true, false, uint32_t> tk;
        tk.init(topk_mem);
        __builtin_amdgcn_wave_barrier();
        while (true) {
            uint32_t nxt = 0xFFFFFFFFu;
            for (uint32_t t = lane; t < T; t += 64u) {
                const uint32_t c = wc[2u * t], e = wc[2u * t + 1u];
                if (c < e) nxt = min(nxt, a.post[c].x);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nxt = min(nxt, (uint32_t)__shfl_xor(nxt, off));
            if (nxt == 0xFFFFFFFFu) break;
            const uint32_t base = nxt & ~31u;
            const uint64_t wend = (uint64_t)base + WIN;
            for (uint32_t t0 = 0; t0 < T; t0 += 64u) {
                const uint32_t t = t0 + lane;
                uint32_t cur = 0, end = 0;
                double idf_l = 0.0;
                if (t < T) {
                    cur = wc[2u * t];
                    end = wc[2u * t + 1u];
                    idf_l = __ddiv_rn(a.idf_used[a.q_terms[it.term_off + t]], 10.0);
                }
                const uint32_t nl = min(64u, T - t0);
                for (uint32_t l = 0; l < nl; ++l) {  // lists in query-term order
                    const uint32_t c = take_list(__shfl(cur, (int)l), __shfl(end, (int)l), __shfl(idf_l, (int)l), base, wend);
                    if (lane == l) cur = c;
                    __builtin_amdgcn_wave_barrier();
                }
                if (t < T) wc[2u * t] = cur;
            }
            harvest(tk, base);
        }
        write_result(tk);
        return;
    }

    uint32_t my_cur[SCORE_LIST_BLOCKS], my_end[SCORE_LIST_BLOCKS];
    double my_idf[SCORE_LIST_BLOCKS];
#pragma unroll
    for (int blk = 0; blk < SCORE_LIST_BLOCKS; ++blk) {
        my_cur[blk] = my_end[blk] = 0;
        my_idf[blk] = 0.0;
        const uint32_t t = (uint32_t)blk * 64u + lane;
        if (t < T) {
            const uint32_t term = a.q_terms[it.term_off + t];
            const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
            const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
            my_cur[blk] = lower_bound_unit(a.post, b, e, it.unit_lo);
            my_end[blk] = (it.unit_hi == 0xFFFFFFFFu) ? e : lower_bound_unit(a.post, my_cur[blk], e, it.unit_hi);
            my_idf[blk] = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
        }
    }
    touched[lane] = 0;
    TopK<CAP, true, false, uint32_t> tk;
    tk.init(topk_mem);
    __builtin_amdgcn_wave_barrier();

    while (true) {
        // window start: the smallest unit any list still has to deliver
        uint32_t nxt = 0xFFFFFFFFu;
#pragma unroll
        for (int blk = 0; blk < SCORE_LIST_BLOCKS; ++blk)
            if (my_cur[blk] < my_end[blk]) nxt = min(nxt, a.post[my_cur[blk]].x);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nxt = min(nxt, (uint32_t)__shfl_xor(nxt, off));
        if (nxt == 0xFFFFFFFFu) break;
        const uint32_t base = nxt & ~31u;
        const uint64_t wend = (uint64_t)base + WIN;

#pragma unroll
        for (int blk = 0; blk < SCORE_LIST_BLOCKS; ++blk) {
            if ((uint32_t)blk * 64u >= T) break;
            const uint32_t nl = min(64u, T - (uint32_t)blk * 64u);
            for (uint32_t l = 0; l < nl; ++l) {  // lists in query-term order
                const uint32_t c = take_list(__shfl(my_cur[blk], (int)l), __shfl(my_end[blk], (int)l), __shfl(my_idf[blk], (int)l), base, wend);
                if (lane == l) my_cur[blk] = c;
                __builtin_amdgcn_wave_barrier();
            }
        }

        harvest(tk, base);
    }
    write_result(tk);
}

// queries with at most FAST_T lists take the stepwise kernels below
constexpr int FAST_T = (int)SCORE_FAST_T;

__device__ __forceinline__ double readlane_f64(double v, int l) {
    const long long b = __double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------
// Stepwise scoring (the default for queries of at most FAST_T lists).
//
// A step takes, from every list, all slots up to a frontier F = the smallest "last unit of the chunk in
// registers" over the lists.  Every posting of a unit <= F is then in this step, whichever list it is in, so
// the step's sums are final; the list that sets F uses all 64 lanes.
//  * The accumulators hold the integer weights, one 16-bit field per list (weight + 1, 0 = no
//    posting).  A list only ever writes its own field, so adding a posting is a plain 2-byte store
//    (no read-modify-write), the number of matching lists is the number of non-zero fields, and the
//    score is formed at harvest as the same double sum in query-term order as on every other path.
//  * Units wait in the table until 64 of them can be harvested together.  No entry is looked up after
//    the step that created it, so removing entries at step ends keeps every probe sequence of the
//    running step intact.
//  * A slot that has been used is overwritten with INF in its register, so "still to do" needs no
//    per-list cursor: a list's part of a step is simply `unit <= F`.
//  * For small k (the launch code says how small) the best k sit in registers, one per lane, sorted; a candidate is placed with one
//    ballot and one lane shift, and the k-th best is exact at every moment.
//
// Tried and dropped: dropping postings that cannot reach the current k-th best score (a per-list
// minimum weight from the other lists' largest in-step weights), with and without a first pass over
// the short lists to learn the threshold early.  Exact, but half of the postings of the benchmark
// batch belong to queries with two or three long lists, whose 64-slot chunks nearly always hold a
// title hit: the bound stays loose, a third of the postings survive, and the bookkeeping costs more
// than the skipped insertions save (3.97 ms per batch against 3.66 without).

__device__ __forceinline__ uint32_t lane_from_below(uint32_t x) {  // lane i <- lane i-1, lane 0 keeps its own
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}

// The best k <= 64 candidates in registers: lane i holds the i-th best (empty lanes hold the worst key).
template <bool HAS_TIE>
struct RegTopK {
    uint64_t key;
    uint32_t unit, tie_, mat;
    uint32_t n;
    uint64_t thr_key;
    uint32_t thr_tie;
    double thr_score;
    static constexpr size_t bytes() { return 16; }
    // the k-th best score is high and exact at every moment: bounds against it are worth their arithmetic
    static constexpr bool FILTER_MIDDLE_LISTS = true;
    static constexpr bool CAN_LAZY = false;
    __device__ void set_lazy_ties(const uint32_t*, uint32_t, uint32_t, uint32_t) {}
    __device__ void init(unsigned char*) {
        key = 0;
        unit = 0xFFFFFFFFu;
        tie_ = 0xFFFFFFFFu;
        mat = 0;
        n = 0;
        thr_key = 0;
        thr_tie = 0xFFFFFFFFu;
        thr_score = -__builtin_huge_val();
    }
    __device__ __forceinline__ uint32_t tie() const { return HAS_TIE ? tie_ : unit; }
    // every lane may offer one candidate; wave-uniform control flow
    __device__ void offer(bool has, uint64_t ck, uint32_t ct, uint32_t cu, uint32_t cm, uint32_t k) {
        uint64_t pm = __ballot(has && better(ck, ct, thr_key, thr_tie));
        const uint32_t lane = lane_id();
        while (pm) {
            const int l = __ffsll((unsigned long long)pm) - 1;
            pm &= pm - 1ull;
            const uint64_t k1 = readlane_u64(ck, l);
            const uint32_t t1 = (uint32_t)__builtin_amdgcn_readlane((int)ct, l);
            const uint32_t u1 = (uint32_t)__builtin_amdgcn_readlane((int)cu, l);
            const uint32_t m1 = (uint32_t)__builtin_amdgcn_readlane((int)cm, l);
            if (!better(k1, t1, thr_key, thr_tie)) continue;  // the k-th best may have risen meanwhile
            // lanes whose entry ranks behind the candidate: a suffix of the wave, never empty here
            const uint64_t b = __ballot(better(k1, t1, key, tie()));
            const uint32_t pos = (uint32_t)__ffsll((unsigned long long)b) - 1u;
            if (pos >= k) continue;
            const uint32_t klo = lane_from_below((uint32_t)key), khi = lane_from_below((uint32_t)(key >> 32));
            const uint32_t us = lane_from_below(unit), ms = lane_from_below(mat);
            const uint32_t ts = HAS_TIE ? lane_from_below(tie_) : 0u;
            if (lane > pos) {
                key = ((uint64_t)khi << 32) | klo;
                unit = us;
                mat = ms;
                if (HAS_TIE) tie_ = ts;
            } else if (lane == pos) {
                key = k1;
                unit = u1;
                mat = m1;
                if (HAS_TIE) tie_ = t1;
            }
            if (n < k) ++n;
            if (n == k) {
                thr_key = readlane_u64(key, (int)k - 1);
                thr_tie = (uint32_t)__builtin_amdgcn_readlane((int)tie(), (int)k - 1);
                thr_score = key_score(thr_key);
            }
        }
    }
    __device__ void write_out(uint32_t k, uint32_t* o_unit, double* o_score, uint32_t* o_mat, uint32_t* o_tie,
                              uint32_t* o_count) {
        const uint32_t i = lane_id();
        if (i < k) {
            const bool live = i < n;
            o_unit[i] = live ? unit : 0xFFFFFFFFu;
            o_score[i] = live ? key_score(key) : 0.0;
            o_mat[i] = live ? mat : 0u;
            if (o_tie) o_tie[i] = live ? tie() : 0xFFFFFFFFu;
        }
        if (i == 0) *o_count = n;
    }
};

#ifndef SEBM25_STEP_BLOCK
#define SEBM25_STEP_BLOCK 2
#endif
#ifndef SEBM25_STEP_BLOCK_MID
#define SEBM25_STEP_BLOCK_MID 2
#endif
#ifndef SEBM25_STEP_WAVES_MID
#define SEBM25_STEP_WAVES_MID 3
#endif
// 64-slot chunks of a list held in registers (and as many in flight): for the variant with the candidates in
// registers, and for the variants with an LDS candidate buffer (smaller blocks: fewer registers)
#ifndef SEBM25_STEP_SLOTS
#define SEBM25_STEP_SLOTS 512
#endif
#ifndef SEBM25_STEP_WAVES
#define SEBM25_STEP_WAVES 6
#endif
constexpr int STEP_SLOTS = SEBM25_STEP_SLOTS, STEP_WAVES = SEBM25_STEP_WAVES;
#ifndef SEBM25_STEP_SLOTS_MID
#define SEBM25_STEP_SLOTS_MID 512
#endif
constexpr int STEP_BLOCK = SEBM25_STEP_BLOCK, STEP_BLOCK_MID = SEBM25_STEP_BLOCK_MID, STEP_WAVES_MID = SEBM25_STEP_WAVES_MID,
              STEP_SLOTS_MID = SEBM25_STEP_SLOTS_MID;

#ifdef SEBM25_EXP_STATS  // experiment builds only: steps, insert passes, postings inserted, harvest rounds, offers, items
__device__ unsigned long long g_step_stats[8];
#endif

__device__ __forceinline__ uint32_t sfirst(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Lists sit shortest first (the host orders them); the last one is the longest.
//  * Every list keeps a block of STEP_BLOCK chunks in registers and has the next block in flight, so a wave has
//    2 x T x STEP_BLOCK x 512 bytes of loads outstanding while it works.  A step's frontier is the smallest
//    "last unit of the block" over the lists, so the list that sets it delivers a whole block.
//  * The longest list is filtered: a unit that only this list holds scores idf_last * w, so its posting is kept
//    only when that alone reaches the k-th best score held so far (w_min = the smallest such weight, 0 while
//    fewer than k candidates are held).  Postings of units that a shorter list delivered in this step are found
//    by probing the table and add their field to the entry.
template <int T, class TK, bool HAS_RANK, int WAVES_PER_SIMD, int B, int SLOTS, int BL = B>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_score_step(ScoreArgs a) {
    static_assert(T >= 1 && T <= FAST_T, "one 16-bit accumulator field per list");
    // A step may bring at most LIMIT units into the table (besides up to 63 that wait from earlier steps); a
    // block-wide step that could bring more is replaced by a chunk-wide one (at most 64 slots per list).
    constexpr uint32_t STEP_SLOTS = SLOTS, STEP_LIMIT = SLOTS - 64;
    static_assert((SLOTS & (SLOTS - 1)) == 0 && (T - 1) * 64 <= (int)STEP_LIMIT, "the table never fills");
    __shared__ uint32_t tag[STEP_SLOTS];
#ifdef SEBM25_WF4
    constexpr uint32_t WF = 4;
#else
    constexpr uint32_t WF = T > 4 ? 8 : (T == 3 ? 4 : T);  // fields per slot: a slot's fields are read and cleared as one word (two for more than four lists)
#endif
    __shared__ __align__(16) unsigned short wfield[STEP_SLOTS * WF];  // per slot and list: weight in tenths + 1
    // the list of slots in use (at most STEP_LIMIT + 63 <= STEP_SLOTS entries).  With three lists a slot's fourth
    // field is free: entry i of the list lives there (field 3 of slot i), and the LDS saved is a wave more per CU.
    constexpr bool USED_IN_FIELD = T == 3 && WF == 4;
    __shared__ unsigned short used_mem[USED_IN_FIELD ? 1 : STEP_LIMIT + 64];
    auto used = [&](uint32_t i) -> unsigned short& { return USED_IN_FIELD ? wfield[i * 4u + 3u] : used_mem[i]; };
    __shared__ __align__(16) unsigned char topk_mem[TK::bytes()];
#ifdef SEBM25_EXP_PAD_LDS  // experiment builds only: more LDS per wave, i.e. fewer waves per CU
    __shared__ uint32_t lds_pad[SEBM25_EXP_PAD_LDS / 4];
    lds_pad[threadIdx.x] = a.k;
    if (lds_pad[63 - threadIdx.x] == 0xFFFFFFFFu) return;
#endif

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    constexpr uint32_t INF = 0xFFFFFFFFu;
    auto ballot = [](bool c) { return (uint64_t)__builtin_amdgcn_ballot_w64(c); };
    auto lanes = [](uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); };

    uint32_t my_cur = 0, my_end = 0;
    double my_idf = 0.0;
    if (lane < (uint32_t)T) {
        const uint32_t term = a.q_terms[it.term_off + lane];
        const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
        const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
        my_cur = lower_bound_unit(a.post, b, e, it.unit_lo);
        my_end = (it.unit_hi == INF) ? e : lower_bound_unit(a.post, my_cur, e, it.unit_hi);
        my_idf = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
    }
    // per list: slot index of the current block's first slot, end of the list, the current block (units and
    // weights of slot pos + 64 j + lane) and the next one.  A slot outside [cursor, end) and a slot already used
    // read as INF.  clast[t][j]: the unit in the last lane of chunk j as it was loaded.
    uint32_t pos[T], end[T];
    // chunks per block: BL for the longest list (it sets most frontiers: the longer its block, the fewer steps), B for
    // the others (what they bring in one step must fit the table)
    constexpr int BM = B > BL ? B : BL;
    auto nb = [](int t) constexpr { return t == T - 1 ? BL : B; };
    uint32_t u[T][BM], w[T][BM], nu[T][BM], nw[T][BM], clast[T][BM];
    double idf[T];
    auto load_block = [&](int t, uint32_t p, uint32_t lo, uint32_t (&uu)[BM], uint32_t (&ww)[BM]) {
        if (p >= lo && p + 64u * nb(t) <= end[t]) {  // inside the list: no masks
            const Slot* ps = a.post + p;
#pragma unroll
            for (int j = 0; j < BM; ++j) {
                if (j >= nb(t)) break;
                const Slot sl = ps[64u * j + lane];
                uu[j] = sl.x;
                ww[j] = sl.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < BM; ++j) {
                if (j >= nb(t)) break;
                const uint32_t i = p + 64u * (uint32_t)j + lane;
                const bool ok = i >= lo && i < end[t];
                const Slot sl = ok ? a.post[i] : make_uint2(INF, POST_HOLE);
                uu[j] = sl.x;
                ww[j] = sl.y;
            }
        }
    };
    // sh[j] / idf[j]: accumulator field and idf of the term the caller gave at position j
    uint32_t sh[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const uint32_t p = sfirst((it.order >> (3 * j)) & 7u);
        sh[j] = 16u * p;
        idf[j] = readlane_f64(my_idf, (int)p);
    }
    const double idf_last = readlane_f64(my_idf, T - 1);
    // idf of the lists as they sit, and whether the lists between the shortest and the longest may be filtered too:
    // bounds of the form "what the longer lists can add at most" need every idf to be positive
    double idf_p[T], inv_idf_p[T];
    bool all_positive = true;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        idf_p[t] = readlane_f64(my_idf, t);
        all_positive = all_positive && idf_p[t] > 0.0 && idf_p[t] < 1e300;
        inv_idf_p[t] = 1.0 / idf_p[t];
    }
    constexpr bool MIDF = T >= 2 && TK::FILTER_MIDDLE_LISTS;
    // the first list that is filtered: with two lists the shorter one (every insert of such a query is its own: 10M-doc
    // index, 10k two-term queries at k = 10: 10.9 -> 8.9 ms); with more lists the second, because filtering the
    // shortest as well needs the block maximum of the second list at every step and wins too little back (17.6 -> 18.1 ms)
    constexpr int FIRST_F = T == 2 ? 0 : 1;
    uint32_t maxw[T];       // largest weight in the list's current block
    uint32_t w_mid[T];      // lists FIRST_F .. T-2: smallest weight whose posting is inserted when no shorter list holds the unit
#pragma unroll
    for (int t = 0; t < T; ++t) w_mid[t] = 0;
    auto block_max = [&](int t) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < BM; ++j)
            if (j < nb(t)) v = max(v, w[t][j] == POST_HOLE ? 0u : w[t][j]);
        v = max(v, lane_xor<1>(v));
        v = max(v, lane_xor<2>(v));
        v = max(v, lane_xor<4>(v));
        v = max(v, lane_xor<8>(v));
        v = max(v, lane_xor<16>(v));
        maxw[t] = max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 32));
    };
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)my_cur, t);
        end[t] = (uint32_t)__builtin_amdgcn_readlane((int)my_end, t);
        pos[t] = cur & ~63u;
        if (cur >= end[t]) {  // empty list: nothing to read
            pos[t] = 0;
            end[t] = 0;
        }
        load_block(t, pos[t], cur, u[t], w[t]);
        load_block(t, pos[t] + 64u * nb(t), 0u, nu[t], nw[t]);
    }
    for (uint32_t x = lane; x < (uint32_t)STEP_SLOTS; x += 64) tag[x] = HASH_EMPTY;
    for (uint32_t x = lane; x < (uint32_t)STEP_SLOTS * WF; x += 64) wfield[x] = 0;
    TK tk;
    tk.init(topk_mem);
    constexpr bool LAZY = HAS_RANK && TK::CAN_LAZY;
    if (LAZY) tk.set_lazy_ties(a.url_rank, a.unit_base, a.rank_base, a.rank_shift);
    uint32_t n_used = 0;
    uint32_t w_min = 0;
    double thr_seen = tk.thr_score;
    uint32_t F_prev = 0;
    bool first_step = true;
#ifdef SEBM25_EXP_STATS
    uint32_t st_steps = 0, st_ins = 0, st_rounds = 0, st_offers = 0, st_passes = 0;
#endif
#pragma unroll
    for (int t = 0; t < T; ++t) {
#pragma unroll
        for (int j = 0; j < BM; ++j)
            if (j < nb(t)) clast[t][j] = (uint32_t)__builtin_amdgcn_readlane((int)u[t][j], 63);
        if (MIDF && t > FIRST_F) block_max(t);
    }
    __builtin_amdgcn_wave_barrier();

    while (true) {
        uint32_t F = clast[0][nb(0) - 1];
#pragma unroll
        for (int t = 1; t < T; ++t) F = min(F, clast[t][nb(t) - 1]);
        // how much the block-wide step would bring into the table (only the shorter lists insert)
        uint32_t n_in = 0;
        if ((T - 1) * B * 64 > (int)STEP_LIMIT) {
            const uint32_t Fb = min(F, INF - 1u);
#pragma unroll
            for (int t = 0; t + 1 < T; ++t)
#pragma unroll
                for (int j = 0; j < B; ++j) n_in += (uint32_t)__popcll(__ballot(u[t][j] <= Fb));  // (t < T - 1: B chunks)
        }
        if (n_in > STEP_LIMIT) {
            // chunk-wide step: the frontier is the smallest last unit over every list's first chunk that is not used up
            F = INF;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                uint32_t c = clast[t][nb(t) - 1];
#pragma unroll
                for (int j = BM - 2; j >= 0; --j)
                    if (j <= nb(t) - 2) c = (first_step || clast[t][j] > F_prev) ? clast[t][j] : c;
                F = min(F, c);
            }
        }
        first_step = false;
        const bool final_step = F == INF;  // every list is in its last block: take what is left
        const uint32_t Fc = min(F, INF - 1u);
        F_prev = Fc;

        // Lists FIRST_F .. T-2: a posting whose unit no shorter list delivered is inserted only
        // if its own part plus the most the longer lists can add (idf x the largest weight of their current blocks,
        // which hold every posting of this step) reaches the k-th best score.  A unit dropped here scores below that
        // score whatever the longer lists hold for it, and whatever part of it a longer list may still offer is
        // smaller still, so it is rejected there against the score of that moment (which only rises).
        if (MIDF) {
            const double thr = tk.thr_score;
            if (all_positive && thr > 0.0) {
                double ub = 0.0;
#pragma unroll
                for (int t = T - 2; t >= FIRST_F; --t) {
                    ub += idf_p[t + 1] * (double)maxw[t + 1];
                    const double x = (thr * (1.0 - 1e-12) - ub) * inv_idf_p[t];
                    w_mid[t] = x < 2.0 ? 0u : (x >= 65536.0 ? 0xFFFFu : (uint32_t)x - 1u);
                }
            }
        }
        bool probe = false;  // the shorter lists brought units into the table in this step
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const bool advance = clast[t][nb(t) - 1] <= Fc;  // the block is used up by this step
#pragma unroll
            for (int j = 0; j < BM; ++j) {
                if (j >= nb(t)) break;
                // Every yes/no per posting is a wave-wide lane mask in scalar registers: masks of plain compares cost
                // nothing beyond the compare, combining them is scalar work, and only selects and guarded accesses turn
                // one back into a per-lane condition.
                const uint64_t in_m = ballot(u[t][j] <= Fc);
                if (!in_m) continue;
                uint64_t s_m = in_m & ballot(w[t][j] != POST_HOLE);  // a hole only leaves its chunk
#ifdef SEBM25_EXP_NOINSERT
                if (t < T - 1) s_m = 0;
#endif
                uint64_t found_m = 0;
                if (t == T - 1 || (MIDF && t >= 1)) {
                    // the filtered lists: add to units that are in the table
                    if (probe && s_m) {
                        // nearly every lane is done after one look (its slot is empty or holds its unit)
                        const bool s = lanes(s_m);
                        const uint32_t un = u[t][j];
                        uint32_t h = (un ^ (un >> 9)) & (STEP_SLOTS - 1u);
                        const uint32_t tg0 = s ? tag[h] : HASH_EMPTY;
                        found_m = s_m & ballot(tg0 == un);
                        const uint64_t more_m = s_m & ~found_m & ballot(tg0 != HASH_EMPTY);
                        if (more_m) {
                            bool later = false;
                            if (lanes(more_m)) {
                                while (true) {
                                    h = (h + 1u) & (STEP_SLOTS - 1u);
                                    const uint32_t tg = tag[h];
                                    if (tg == un) {
                                        later = true;
                                        break;
                                    }
                                    if (tg == HASH_EMPTY) break;
                                }
                            }
                            found_m |= ballot(later);
                        }
                        if (lanes(found_m)) wfield[h * WF + (uint32_t)t] = (unsigned short)(w[t][j] + 1u);
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (t < T - 1) s_m &= ~found_m & ballot(w[t][j] >= w_mid[t]);  // what is left goes into the table below
                } else if (MIDF && FIRST_F == 0) {
                    s_m &= ballot(w[t][j] >= w_mid[t]);  // the shortest list: nothing to look up, the same bound
                }
                if (t == T - 1) {
                    // a unit that only the longest list holds has its final score already: straight to the candidates
                    const uint64_t alone_m = s_m & ~found_m & ballot(w[t][j] >= w_min);
                    if (alone_m) {
                        const double sc = __dmul_rn(idf_last, (double)w[t][j]);
                        const uint64_t may_m = alone_m & ballot(sc >= tk.thr_score);
                        if (may_m) {
                            const bool may = lanes(may_m);
                            uint32_t tie = 0;
                            if (may && !LAZY) tie = HAS_RANK ? rank_key(a.url_rank, a.rank_base, a.rank_shift, u[t][j]) : u[t][j] + a.unit_base;
                            tk.offer(may, score_key(sc), tie, u[t][j] + a.unit_base, 1u, a.k);
                        }
                    }
                    s_m = 0;  // the longest list never inserts
                }
#ifdef SEBM25_EXP_STATS
                st_ins += (uint32_t)__popcll(s_m);
                st_passes += s_m ? 1u : 0u;
#endif
                if (s_m) {
                    bool fresh = false;
                    uint32_t h = 0;
                    if (lanes(s_m)) {
                        const uint32_t un = u[t][j];
                        h = (un ^ (un >> 9)) & (STEP_SLOTS - 1u);
                        while (true) {
                            const uint32_t old = atomicCAS(&tag[h], HASH_EMPTY, un);
                            if (old == HASH_EMPTY) {
                                fresh = true;
                                break;
                            }
                            if (old == un) break;
                            h = (h + 1u) & (STEP_SLOTS - 1u);
                        }
                        wfield[h * WF + (uint32_t)t] = (unsigned short)(w[t][j] + 1u);
                    }
                    const uint64_t fm = ballot(fresh);
                    if (fresh) used(n_used + (uint32_t)__popcll(fm & lt)) = (unsigned short)h;
                    n_used += (uint32_t)__popcll(fm);
#ifndef SEBM25_EXP_NOPROBE
                    if (t < T - 1) probe = true;
#endif
                    __builtin_amdgcn_wave_barrier();
                }
                if (!advance) u[t][j] = lanes(in_m) ? INF : u[t][j];  // used slots leave their chunk
            }
            if (advance) {  // the block in flight becomes the current one, the one behind it is requested
#pragma unroll
                for (int j = 0; j < BM; ++j) {
                    if (j >= nb(t)) break;
                    u[t][j] = nu[t][j];
                    w[t][j] = nw[t][j];
                }
                pos[t] += 64u * nb(t);
                load_block(t, pos[t] + 64u * nb(t), 0u, nu[t], nw[t]);
#pragma unroll
                for (int j = 0; j < BM; ++j)
                    if (j < nb(t)) clast[t][j] = (uint32_t)__builtin_amdgcn_readlane((int)u[t][j], 63);
                if (MIDF && t > FIRST_F) block_max(t);
            }
        }
#ifdef SEBM25_EXP_STATS
        ++st_steps;
#endif

        // harvest 64 units at a time from the back of the list of used slots
        while (n_used >= 64u || (final_step && n_used)) {
            const uint32_t take = min(n_used, 64u);
            n_used -= take;
            const bool has = lane < take;
            double s = 0.0;
            uint32_t m = 0, unit = 0;
            if (has) {
                const uint32_t h = used(n_used + lane);
                unit = tag[h];
                uint64_t f, f_hi = 0;
                if constexpr (WF == 8) {
                    f = *reinterpret_cast<const uint64_t*>(&wfield[h * 8u]);
                    f_hi = *reinterpret_cast<const uint64_t*>(&wfield[h * 8u + 4u]);
                    *reinterpret_cast<uint64_t*>(&wfield[h * 8u]) = 0ull;
                    *reinterpret_cast<uint64_t*>(&wfield[h * 8u + 4u]) = 0ull;
                } else if constexpr (WF == 4) {
                    f = *reinterpret_cast<const uint64_t*>(&wfield[h * 4u]);
                    // (a list entry in the fourth field stays: nothing writes list entries during a harvest)
                    *reinterpret_cast<uint64_t*>(&wfield[h * 4u]) = USED_IN_FIELD ? (f & 0xFFFF000000000000ull) : 0ull;
                } else if constexpr (WF == 2) {
                    f = *reinterpret_cast<const uint32_t*>(&wfield[h * 2u]);
                    *reinterpret_cast<uint32_t*>(&wfield[h * 2u]) = 0u;
                } else {
                    f = wfield[h];
                    wfield[h] = 0;
                }
                tag[h] = HASH_EMPTY;
#pragma unroll
                for (int j = 0; j < T; ++j) {  // the order the caller gave the terms in
                    const uint32_t ft = (uint32_t)((WF == 8 && sh[j] >= 64u ? f_hi : f) >> (sh[j] & 63u)) & 0xFFFFu;
                    if (ft) {
                        const double v = __dmul_rn(idf[j], (double)(ft - 1u));
                        s = m ? __dadd_rn(s, v) : v;
                        ++m;
                    }
                }
            }
            // most units cannot enter the top k any more: test the score alone before anything else
            const uint64_t may_m = ballot(s >= tk.thr_score) & (take >= 64u ? ~0ull : (1ull << take) - 1ull);
#ifdef SEBM25_EXP_STATS
            ++st_rounds;
            st_offers += (uint32_t)__popcll(may_m);
#endif
            if (may_m) {
                const bool may = lanes(may_m);
                uint32_t tie = 0;
                if (may && !LAZY) tie = HAS_RANK ? rank_key(a.url_rank, a.rank_base, a.rank_shift, unit) : unit + a.unit_base;
                tk.offer(may, score_key(s), tie, unit + a.unit_base, m, a.k);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (final_step) break;
        if (tk.thr_score != thr_seen) {  // the k-th best score rose: what the longest list must now bring alone
            thr_seen = tk.thr_score;
            w_min = 0;
            if (idf_last > 0.0 && thr_seen > 0.0) {
                const double x = thr_seen / idf_last;
                if (x >= 65535.0) {
                    w_min = 0xFFFFu;  // above every 16-bit weight
                } else {
                    w_min = (uint32_t)x;
                    w_min = w_min > 1u ? w_min - 2u : 0u;
                    while (__dmul_rn(idf_last, (double)w_min) < thr_seen) ++w_min;  // exact: the product the harvest forms
                }
            }
        }
    }
#ifdef SEBM25_EXP_STATS
    if (lane == 0) {
        atomicAdd(&g_step_stats[0], (unsigned long long)st_steps);
        atomicAdd(&g_step_stats[1], (unsigned long long)st_passes);
        atomicAdd(&g_step_stats[2], (unsigned long long)st_ins);
        atomicAdd(&g_step_stats[3], (unsigned long long)st_rounds);
        atomicAdd(&g_step_stats[4], (unsigned long long)st_offers);
        atomicAdd(&g_step_stats[5], 1ull);
    }
#endif

    const size_t o = (size_t)it.out_slot * a.k;
    if (it.direct)
        tk.write_out(a.k, a.out_unit + o, a.out_score + o, a.out_matched + o, a.out_tie ? a.out_tie + o : nullptr,
                     a.out_count + it.out_slot);
    else
        tk.write_out(a.k, a.tmp_unit + o, a.tmp_score + o, a.tmp_matched + o, a.tmp_tie + o, a.tmp_count + it.out_slot);
}

// ------------------------------------------------------------------------------------------
// Stepwise scoring by merge join (queries of two to four lists).
//
// The same steps as k_score_step (every list holds a block of 64-slot chunks and has the next block in flight;
// a step takes every slot up to the smallest "last unit of the block" over the lists), but the blocks sit in
// LDS as sorted arrays of slots and no table is filled.  The lists are taken shortest first.  A posting of the
// step looks its unit up in the blocks of the longer lists (the chunk by compares with the chunks' last units,
// which are wave-uniform, then a binary search over its 64 slots: six dependent LDS reads), 64 postings per
// pass, and has the unit's score at once, summed in the order the caller gave the terms in.  Every slot found
// that way is turned into a hole in LDS, so the longer list's own turn skips it: a unit is scored once, by
// the shortest list that holds it, and nothing ever searches a shorter list.
//  * Lists other than the shortest (of three or four) are filtered before they search: a posting whose unit no
//    shorter list held is worth a search only if its own part plus the most the longer lists can add (idf x
//    the largest weight of their current blocks, which hold every posting of the step) reaches the k-th best
//    score.  What passes is compacted into a work list first, so the passes run with most lanes busy.  A unit
//    dropped there scores below the k-th best whatever the longer lists hold for it; the part of it that a
//    longer list may offer later is smaller still and is rejected against the score of that moment, which only
//    rises.  This needs every idf of the query to be positive and is switched off otherwise.
//  * The longest list, most of the postings, is only tested against the smallest weight with which a posting
//    that no other list shares could reach the k-th best score.
//  * A block keeps the slots before the list's cursor as {0, hole} and the slots behind its end as
//    {0xFFFFFFFF, hole}: it stays sorted, and first_idx (non-zero in a list's first block only) keeps a search
//    for unit 0 off the former.  A hole never matches: the first slot of a unit's run carries the weight
//    when the run has one.
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int T, class TK, bool HAS_RANK, int WAVES_PER_SIMD, int B, int BL>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_score_join(ScoreArgs a) {
    static_assert(T >= 2 && T <= 4, "two to four lists");
    constexpr uint32_t INF = 0xFFFFFFFFu;
    constexpr int BM = B > BL ? B : BL;
    auto nb = [](int t) constexpr { return t == T - 1 ? BL : B; };          // chunks per block
    auto off = [](int t) constexpr { return (uint32_t)(64 * B * t); };       // first slot of the list's block
    __shared__ __align__(16) Slot blk[64 * (B * (T - 1) + BL)];
    __shared__ unsigned short wk[64 * B * (T >= 4 ? T - 1 : 1)];             // slots of the shorter lists' blocks that take part in the step
    __shared__ __align__(16) unsigned char topk_mem[TK::bytes()];

    const QItem it = a.items[blockIdx.x];
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    auto ballot = [](bool c) { return (uint64_t)__builtin_amdgcn_ballot_w64(c); };
    auto lanes = [](uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); };

    uint32_t my_cur = 0, my_end = 0;
    double my_idf = 0.0;
    if (lane < (uint32_t)T) {
        const uint32_t term = a.q_terms[it.term_off + lane];
        const uint32_t b = a.news ? a.news_ptr[term] : a.row_ptr[term];
        const uint32_t e = a.news ? a.row_ptr[term + 1] : a.news_ptr[term];
        my_cur = lower_bound_unit(a.post, b, e, it.unit_lo);
        my_end = (it.unit_hi == INF) ? e : lower_bound_unit(a.post, my_cur, e, it.unit_hi);
        my_idf = __ddiv_rn(a.idf_used[term], 10.0);  // weights are stored in tenths
    }
    uint32_t pos[T], end[T];       // slot index of the current block's first slot, end of the list
    uint32_t s[T], first_idx[T];   // first slot of the block not yet taken; first slot of the block inside the list's range
    uint32_t cl[T][BM];            // unit of the last slot of every chunk of the block
    uint32_t maxw[T];              // largest weight in the block
    uint32_t nu[T][BM], nw[T][BM]; // the block in flight
    auto load_block = [&](auto tc, uint32_t p, uint32_t lo) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        if (p >= lo && p + 64u * nb(t) <= end[t]) {  // inside the list: no masks
            const Slot* ps = a.post + p;
#pragma unroll
            for (int j = 0; j < nb(t); ++j) {
                const Slot sl = ps[64u * j + lane];
                nu[t][j] = sl.x;
                nw[t][j] = sl.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < nb(t); ++j) {
                const uint32_t i = p + 64u * (uint32_t)j + lane;
                const bool behind = i >= end[t], ok = i >= lo && !behind;
                const Slot sl = ok ? a.post[i] : make_uint2(behind ? INF : 0u, POST_HOLE);
                nu[t][j] = sl.x;
                nw[t][j] = sl.y;
            }
        }
    };
    // the first list that is filtered: with more than two lists the shortest one searches for every posting (filtering
    // it as well changed nothing: 12.9 -> 13.2 ms per 10k three-term queries over the 10M-doc index)
    constexpr int FIRST_F = T == 2 ? 0 : 1;
    // the block in flight becomes the current one
    auto take_block = [&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        int32_t mx = 0;
#pragma unroll
        for (int j = 0; j < nb(t); ++j) {
            blk[off(t) + 64u * (uint32_t)j + lane] = make_uint2(nu[t][j], nw[t][j]);
            cl[t][j] = (uint32_t)__builtin_amdgcn_readlane((int)nu[t][j], 63);
            if (t > FIRST_F) mx = max(mx, (int32_t)nw[t][j]);  // a hole reads as -1 (weights here are below 2^31)
        }
        if (t > FIRST_F) {
            uint32_t v = (uint32_t)mx;
            v = max(v, lane_xor<1>(v));
            v = max(v, lane_xor<2>(v));
            v = max(v, lane_xor<4>(v));
            v = max(v, lane_xor<8>(v));
            v = max(v, lane_xor<16>(v));
            maxw[t] = max((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 32));
        }
    };
    // idf[j], pl[j]: idf of the term the caller gave at position j and the list it became
    double idf[T];
    uint32_t pl[T];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        pl[j] = sfirst((it.order >> (3 * j)) & 7u);
        idf[j] = readlane_f64(my_idf, (int)pl[j]);
    }
    const double idf_last = readlane_f64(my_idf, T - 1);
    double idf_p[T], inv_idf_p[T];  // idf of the lists as they sit
    bool all_positive = true;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        idf_p[t] = readlane_f64(my_idf, t);
        all_positive = all_positive && idf_p[t] > 0.0 && idf_p[t] < 1e300;
        inv_idf_p[t] = 1.0 / idf_p[t];
        maxw[t] = 0;
    }
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)my_cur, t);
        end[t] = (uint32_t)__builtin_amdgcn_readlane((int)my_end, t);
        pos[t] = cur & ~63u;
        if (cur >= end[t]) {  // empty list: nothing to read
            pos[t] = 0;
            end[t] = 0;
        }
        s[t] = first_idx[t] = cur >= end[t] ? 0u : cur - pos[t];
        load_block(tc, pos[t], cur);
    });
    static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        take_block(tc);
        load_block(tc, pos[t] + 64u * nb(t), 0u);
    });
    TK tk;
    tk.init(topk_mem);
    constexpr bool LAZY = HAS_RANK && TK::CAN_LAZY;
    if (LAZY) tk.set_lazy_ties(a.url_rank, a.unit_base, a.rank_base, a.rank_shift);
    uint32_t w_min = 0;
    double thr_seen = tk.thr_score;
    uint32_t F_prev = 0;
    bool first_step = true;
    __builtin_amdgcn_wave_barrier();
#ifdef SEBM25_EXP_STATS
    uint32_t st_steps = 0, st_ins = 0, st_rounds = 0, st_offers = 0, st_passes = 0;
#endif

    auto offer = [&](uint64_t may_m, double sc, uint32_t unit, uint32_t m) __attribute__((always_inline)) {
        const bool may = lanes(may_m);
        uint32_t tie = 0;
        if (may && !LAZY) tie = HAS_RANK ? rank_key(a.url_rank, a.rank_base, a.rank_shift, unit) : unit + a.unit_base;
        tk.offer(may, score_key(sc), tie, unit + a.unit_base, m, a.k);
    };

    while (true) {
        uint32_t F = cl[0][nb(0) - 1];
#pragma unroll
        for (int t = 1; t < T; ++t) F = min(F, cl[t][nb(t) - 1]);
        const bool final_step = F == INF;  // every list is in its last block: take what is left
        const uint32_t Fc = min(F, INF - 1u);

        // smallest weight with which a posting of a filtered list is looked at
        uint32_t w_mid[T];
#pragma unroll
        for (int t = 0; t < T; ++t) w_mid[t] = 0;
        {
            const double thr = tk.thr_score;
            if (all_positive && thr > 0.0) {
                double ub = 0.0;
#pragma unroll
                for (int t = T - 2; t >= FIRST_F; --t) {
                    ub += idf_p[t + 1] * (double)maxw[t + 1];
                    const double x = (thr * (1.0 - 1e-12) - ub) * inv_idf_p[t];
                    w_mid[t] = x < 2.0 ? 0u : (x >= 2147483648.0 ? 0x7FFFFFFFu : (uint32_t)x - 1u);
                }
            }
        }

        // Three shorter lists (four-list queries): one work list for all of them and passes that take 64 of its postings
        // whatever their lists, every lane searching every list but the shortest (10k queries, k = 210: 34.9 against 41.8 ms
        // per batch with a list at a time).  Two shorter lists: that form was slower than a list at a time (k = 210: 18.9
        // against 17.7 ms: the second list's passes then search two lists instead of one); what pays is to let postings of the
        // second list ride in the lanes that the shortest list's last pass leaves free (16.2 ms; k = 10: 11.6 -> 11.3 ms).
        if constexpr (T >= 4) {
            // ---- the lists but the longest: the slots of the step that still hold a posting worth a search, compacted
            //      (shortest list first)
            uint32_t n = 0;
            static_for<0, T - 1>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
#pragma unroll
                for (int c = 0; c < nb(t); ++c) {
                    if (s[t] >= 64u * (uint32_t)(c + 1)) continue;  // taken by earlier steps
                    const uint32_t idx = 64u * (uint32_t)c + lane;
                    const Slot me = blk[off(t) + idx];
                    const uint64_t in_m = ballot(me.x <= Fc);
                    uint64_t keep = in_m & ballot(me.y + 1u > w_mid[t]);  // (a hole wraps to 0)
                    if (s[t] > 64u * (uint32_t)c) keep &= ~0ull << (s[t] - 64u * (uint32_t)c);
                    if (keep) {
                        if (lanes(keep)) wk[n + (uint32_t)__popcll(keep & lt)] = (unsigned short)(off(t) + idx);
                        n += (uint32_t)__popcll(keep);
                    }
                    s[t] = 64u * (uint32_t)c + (uint32_t)__popcll(in_m);
                    if (cl[t][c] > Fc) break;
                }
            });
            __builtin_amdgcn_wave_barrier();
#ifdef SEBM25_EXP_STATS
            st_ins += n;
#endif
            // ---- 64 of them per pass, whatever their lists.  Every lane searches every list but the shortest, side by side
            //      (each round issues its reads before the first compare), and uses what the lists longer than its own return.
            for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
#ifdef SEBM25_EXP_STATS
                ++st_passes;
#endif
                const bool has = i0 + lane < n;
                const uint32_t gi = has ? wk[i0 + lane] : 0u;
                const Slot me = blk[gi];
                uint32_t tl = 0;  // the list the lane's posting is in
#pragma unroll
                for (int t = 1; t + 1 < T; ++t) tl += gi >= off(t) ? 1u : 0u;
                bool act = has && me.y != POST_HOLE;  // (a pass before this one may have taken the posting)
                uint32_t sp[T];
                static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    uint32_t c0 = 0;  // the chunk: units beyond a chunk's last unit lie in a later one
#pragma unroll
                    for (int j = 0; j + 1 < nb(o); ++j) c0 += me.x > cl[o][j] ? 64u : 0u;
                    sp[o] = off(o) + c0;
                });
#pragma unroll
                for (int h = 32; h >= 1; h >>= 1) {  // lower bound over the chunk's 64 slots
                    uint32_t v[T];
                    static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        v[o] = blk[sp[o] + (uint32_t)(h - 1)].x;
                    });
                    static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        sp[o] += v[o] < me.x ? (uint32_t)h : 0u;
                    });
                }
                uint32_t rx[T], ry[T];  // (values, not slots: a select between members of two slots goes through memory)
                static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    sp[o] = max(sp[o], off(o) + first_idx[o]);
                    const Slot r = blk[sp[o]];
                    rx[o] = r.x;
                    ry[o] = r.y;
                });
                const uint32_t mx = me.x, my = me.y;
                uint32_t wt[T];
                wt[0] = tl == 0u ? my : POST_HOLE;
                static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    // a posting of a shorter list that finds its unit here takes the slot: the list's own postings skip it
                    const bool found = act && tl < (uint32_t)o && rx[o] == mx && ry[o] != POST_HOLE;
                    if (found) blk[sp[o]].y = POST_HOLE;
                    if constexpr (o < T - 1) {
                        // this pass's postings of list o: taken just now by a lane of a shorter list?
                        __builtin_amdgcn_wave_barrier();
                        const uint32_t now = blk[gi].y;
                        if (tl == (uint32_t)o && now == POST_HOLE) act = false;
                    }
                    const uint32_t other = found ? ry[o] : POST_HOLE;
                    wt[o] = tl == (uint32_t)o ? my : other;
                });
                double sc = 0.0;
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < T; ++j) {  // the order the caller gave the terms in
                    uint32_t w = wt[0];
#pragma unroll
                    for (int l = 1; l < T; ++l) w = pl[j] == (uint32_t)l ? wt[l] : w;
                    if (w != POST_HOLE) {
                        const double v = __dmul_rn(idf[j], (double)w);
                        sc = m ? __dadd_rn(sc, v) : v;
                        ++m;
                    }
                }
                const uint64_t may_m = ballot(act) & ballot(sc >= tk.thr_score);
#ifdef SEBM25_EXP_STATS
                st_offers += (uint32_t)__popcll(may_m);
#endif
                if (may_m) offer(may_m, sc, mx, m);
            }
            __builtin_amdgcn_wave_barrier();
        } else if constexpr (T == 3) {
            // ---- three lists: the postings of the shortest list search both longer lists; the lanes they leave free in their
            //      last pass take postings of the second list along (those only use what the longest list returns); what is
            //      left of the second list's postings then takes passes of its own, which search one list
            unsigned char* wkb = reinterpret_cast<unsigned char*>(wk);
            constexpr uint32_t HALF = 64u * (uint32_t)B;
            static_assert(HALF <= 256u, "slot indices of a block fit a byte");
            uint32_t nn[2] = {0, 0};
            static_for<0, 2>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
#pragma unroll
                for (int c = 0; c < nb(t); ++c) {
                    if (s[t] >= 64u * (uint32_t)(c + 1)) continue;  // taken by earlier steps
                    const uint32_t idx = 64u * (uint32_t)c + lane;
                    const Slot me = blk[off(t) + idx];
                    const uint64_t in_m = ballot(me.x <= Fc);
                    uint64_t keep = in_m & ballot(me.y + 1u > w_mid[t]);  // (a hole wraps to 0)
                    if (s[t] > 64u * (uint32_t)c) keep &= ~0ull << (s[t] - 64u * (uint32_t)c);
                    if (keep) {
                        if (lanes(keep)) wkb[(uint32_t)t * HALF + nn[t] + (uint32_t)__popcll(keep & lt)] = (unsigned char)idx;
                        nn[t] += (uint32_t)__popcll(keep);
                    }
                    s[t] = 64u * (uint32_t)c + (uint32_t)__popcll(in_m);
                    if (cl[t][c] > Fc) break;
                }
            });
            __builtin_amdgcn_wave_barrier();
            const uint32_t n0 = nn[0], n1 = nn[1];
#ifdef SEBM25_EXP_STATS
            st_ins += n0 + n1;
#endif
            uint32_t rode = 0;  // postings of the second list that went along with the shortest list's
            for (uint32_t i0 = 0; i0 < n0; i0 += 64u) {
#ifdef SEBM25_EXP_STATS
                ++st_passes;
#endif
                const uint32_t take0 = min(n0 - i0, 64u);
                const uint32_t riders = i0 + 64u >= n0 ? min(n1, 64u - take0) : 0u;
                const bool has = lane < take0 + riders;
                const uint32_t tl = lane < take0 ? 0u : 1u;
                const uint32_t at = tl == 0u ? i0 + lane : HALF + lane - take0;
                const uint32_t gi = has ? (tl == 0u ? off(0) : off(1)) + wkb[at] : 0u;
                const Slot me = blk[gi];
                const uint32_t mx = me.x, my = me.y;
                bool act = has && my != POST_HOLE;
                uint32_t sp[T];
                static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    uint32_t c0 = 0;  // the chunk: units beyond a chunk's last unit lie in a later one
#pragma unroll
                    for (int j = 0; j + 1 < nb(o); ++j) c0 += mx > cl[o][j] ? 64u : 0u;
                    sp[o] = off(o) + c0;
                });
#pragma unroll
                for (int h = 32; h >= 1; h >>= 1) {  // lower bound over the chunk's 64 slots, both lists side by side
                    uint32_t v[T];
                    static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        v[o] = blk[sp[o] + (uint32_t)(h - 1)].x;
                    });
                    static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        sp[o] += v[o] < mx ? (uint32_t)h : 0u;
                    });
                }
                uint32_t rx[T], ry[T];
                static_for<1, T>([&](auto oc) __attribute__((always_inline)) {
                    constexpr int o = decltype(oc)::value;
                    sp[o] = max(sp[o], off(o) + first_idx[o]);
                    const Slot r = blk[sp[o]];
                    rx[o] = r.x;
                    ry[o] = r.y;
                });
                const bool found1 = act && tl == 0u && rx[1] == mx && ry[1] != POST_HOLE;
                if (found1) blk[sp[1]].y = POST_HOLE;  // taken: the second list's postings skip it
                __builtin_amdgcn_wave_barrier();
                const uint32_t now = blk[gi].y;        // a posting of the second list that went along: taken just now?
                if (tl == 1u && now == POST_HOLE) act = false;
                const bool found2 = act && rx[2] == mx && ry[2] != POST_HOLE;
                if (found2) blk[sp[2]].y = POST_HOLE;  // taken: the longest list's own test skips it
                uint32_t wt[T];
                wt[0] = tl == 0u ? my : POST_HOLE;
                const uint32_t w1 = found1 ? ry[1] : POST_HOLE;
                wt[1] = tl == 1u ? my : w1;
                wt[2] = found2 ? ry[2] : POST_HOLE;
                double sc = 0.0;
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < T; ++j) {  // the order the caller gave the terms in
                    uint32_t w = wt[0];
#pragma unroll
                    for (int l = 1; l < T; ++l) w = pl[j] == (uint32_t)l ? wt[l] : w;
                    if (w != POST_HOLE) {
                        const double v = __dmul_rn(idf[j], (double)w);
                        sc = m ? __dadd_rn(sc, v) : v;
                        ++m;
                    }
                }
                const uint64_t may_m = ballot(act) & ballot(sc >= tk.thr_score);
#ifdef SEBM25_EXP_STATS
                st_offers += (uint32_t)__popcll(may_m);
#endif
                if (may_m) offer(may_m, sc, mx, m);
                rode += riders;
            }
            __builtin_amdgcn_wave_barrier();
            for (uint32_t i0 = rode; i0 < n1; i0 += 64u) {
#ifdef SEBM25_EXP_STATS
                ++st_passes;
#endif
                const bool has = i0 + lane < n1;
                const uint32_t gi = off(1) + (has ? wkb[HALF + i0 + lane] : 0u);
                const Slot me = blk[gi];
                const uint32_t mx = me.x, my = me.y;
                const bool act = has && my != POST_HOLE;  // (the shortest list's postings may have taken it meanwhile)
                constexpr int o = 2;
                uint32_t sp = off(o);
#pragma unroll
                for (int j = 0; j + 1 < nb(o); ++j) sp += mx > cl[o][j] ? 64u : 0u;
#pragma unroll
                for (int h = 32; h >= 1; h >>= 1) sp += blk[sp + (uint32_t)(h - 1)].x < mx ? (uint32_t)h : 0u;
                sp = max(sp, off(o) + first_idx[o]);
                const Slot r = blk[sp];
                const uint32_t rx = r.x, ry = r.y;
                const bool found = act && rx == mx && ry != POST_HOLE;
                if (found) blk[sp].y = POST_HOLE;
                uint32_t wt[T];
                wt[0] = POST_HOLE;
                wt[1] = my;
                wt[2] = found ? ry : POST_HOLE;
                double sc = 0.0;
                uint32_t m = 0;
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    uint32_t w = wt[0];
#pragma unroll
                    for (int l = 1; l < T; ++l) w = pl[j] == (uint32_t)l ? wt[l] : w;
                    if (w != POST_HOLE) {
                        const double v = __dmul_rn(idf[j], (double)w);
                        sc = m ? __dadd_rn(sc, v) : v;
                        ++m;
                    }
                }
                const uint64_t may_m = ballot(act) & ballot(sc >= tk.thr_score);
#ifdef SEBM25_EXP_STATS
                st_offers += (uint32_t)__popcll(may_m);
#endif
                if (may_m) offer(may_m, sc, mx, m);
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // ---- two lists: the postings of the shorter one, 64 per pass (the general form: a list at a time, shortest first)
            static_for<0, T - 1>([&](auto tc) __attribute__((always_inline)) {
                constexpr int t = decltype(tc)::value;
                // the slots of the step that still hold a posting worth a search, compacted
                uint32_t n = 0;
#pragma unroll
                for (int c = 0; c < nb(t); ++c) {
                    if (s[t] >= 64u * (uint32_t)(c + 1)) continue;  // taken by earlier steps
                    const uint32_t idx = 64u * (uint32_t)c + lane;
                    const Slot me = blk[off(t) + idx];
                    const uint64_t in_m = ballot(me.x <= Fc);
                    uint64_t keep = in_m & ballot(me.y + 1u > w_mid[t]);  // (a hole wraps to 0)
                    if (s[t] > 64u * (uint32_t)c) keep &= ~0ull << (s[t] - 64u * (uint32_t)c);
                    if (keep) {
                        if (lanes(keep)) wk[n + (uint32_t)__popcll(keep & lt)] = (unsigned short)idx;
                        n += (uint32_t)__popcll(keep);
                    }
                    s[t] = 64u * (uint32_t)c + (uint32_t)__popcll(in_m);
                    if (cl[t][c] > Fc) break;
                }
                __builtin_amdgcn_wave_barrier();
#ifdef SEBM25_EXP_STATS
                st_ins += n;
#endif
#ifdef SEBM25_EXP_NOPASS  // experiment builds only (wrong results): what the steps cost without the searches
                n = 0;
#endif
                for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
#ifdef SEBM25_EXP_STATS
                    ++st_passes;
#endif
                    const bool has = i0 + lane < n;
                    const uint32_t idx = has ? wk[i0 + lane] : 0u;
                    const Slot me = blk[off(t) + idx];
                    // the searches in the longer lists run side by side: every round issues its reads before the first compare
                    uint32_t wt[T], sp[T];
                    static_for<t + 1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        uint32_t c0 = 0;  // the chunk: units beyond a chunk's last unit lie in a later one
#pragma unroll
                        for (int j = 0; j + 1 < nb(o); ++j) c0 += me.x > cl[o][j] ? 64u : 0u;
                        sp[o] = off(o) + c0;
                    });
#pragma unroll
                    for (int h = 32; h >= 1; h >>= 1) {  // lower bound over the chunk's 64 slots
                        uint32_t v[T];
                        static_for<t + 1, T>([&](auto oc) __attribute__((always_inline)) {
                            constexpr int o = decltype(oc)::value;
                            v[o] = blk[sp[o] + (uint32_t)(h - 1)].x;
                        });
                        static_for<t + 1, T>([&](auto oc) __attribute__((always_inline)) {
                            constexpr int o = decltype(oc)::value;
                            sp[o] += v[o] < me.x ? (uint32_t)h : 0u;
                        });
                    }
                    Slot r[T];
                    static_for<t + 1, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        sp[o] = max(sp[o], off(o) + first_idx[o]);
                        r[o] = blk[sp[o]];
                    });
                    static_for<0, T>([&](auto oc) __attribute__((always_inline)) {
                        constexpr int o = decltype(oc)::value;
                        if constexpr (o < t) {
                            wt[o] = POST_HOLE;  // a shorter list that held the unit would have taken this slot
                        } else if constexpr (o == t) {
                            wt[o] = me.y;
                        } else {
                            const bool found = has && r[o].x == me.x && r[o].y != POST_HOLE;
                            wt[o] = found ? r[o].y : POST_HOLE;
                            if (found) blk[sp[o]].y = POST_HOLE;  // taken: the list's own turn skips it
                        }
                    });
                    double sc = 0.0;
                    uint32_t m = 0;
#pragma unroll
                    for (int j = 0; j < T; ++j) {  // the order the caller gave the terms in
                        uint32_t w = wt[0];
#pragma unroll
                        for (int l = 1; l < T; ++l) w = pl[j] == (uint32_t)l ? wt[l] : w;
                        if (w != POST_HOLE) {
                            const double v = __dmul_rn(idf[j], (double)w);
                            sc = m ? __dadd_rn(sc, v) : v;
                            ++m;
                        }
                    }
                    const uint64_t may_m = ballot(has) & ballot(sc >= tk.thr_score);
#ifdef SEBM25_EXP_STATS
                    st_offers += (uint32_t)__popcll(may_m);
#endif
                    if (may_m) offer(may_m, sc, me.x, m);
                }
                __builtin_amdgcn_wave_barrier();
            });
        }

        // ---- the longest list: what is left of it in this step belongs to units that no other list holds.  First the
        //      weights of the whole block against the smallest weight that could matter (nearly always none passes, and one
        //      scalar test ends the matter); which slots are in the step at all is only looked at for chunks that have one.
        {
            constexpr int t = T - 1;
            uint64_t wm[BL];
            uint64_t any_m = 0;
            {
                uint32_t wv[BL];
#pragma unroll
                for (int c = 0; c < BL; ++c) wv[c] = blk[off(t) + 64u * (uint32_t)c + lane].y;
#pragma unroll
                for (int c = 0; c < BL; ++c) {
                    wm[c] = ballot(wv[c] + 1u > w_min);  // (a hole wraps to 0)
                    any_m |= wm[c];
                }
            }
            if (any_m) {
#pragma unroll
                for (int c = 0; c < BL; ++c) {
                    if (!first_step && cl[t][c] <= F_prev) continue;  // taken by earlier steps
                    if (wm[c]) {
                        const Slot me = blk[off(t) + 64u * (uint32_t)c + lane];
                        uint64_t may_m = wm[c] & ballot(me.x <= Fc);
                        if (!first_step) may_m &= ~ballot(me.x <= F_prev);
                        const double sc = __dmul_rn(idf_last, (double)me.y);
                        may_m &= ballot(sc >= tk.thr_score);
#ifdef SEBM25_EXP_STATS
                        st_rounds += (uint32_t)__popcll(may_m);
#endif
                        if (may_m) offer(may_m, sc, me.x, 1u);
                    }
                    if (cl[t][c] > Fc) break;
                }
            }
        }
#ifdef SEBM25_EXP_STATS
        ++st_steps;
#endif
        if (final_step) break;
        first_step = false;
        F_prev = Fc;
        if (tk.thr_score != thr_seen) {  // the k-th best score rose: what the longest list must now bring alone
            thr_seen = tk.thr_score;
            w_min = 0;
            if (idf_last > 0.0 && thr_seen > 0.0) {
                const double x = thr_seen / idf_last;
                if (x >= 2147483648.0) {
                    w_min = 0x7FFFFFFFu;  // above every weight
                } else {
                    w_min = (uint32_t)x;
                    w_min = w_min > 1u ? w_min - 2u : 0u;
                    while (__dmul_rn(idf_last, (double)w_min) < thr_seen) ++w_min;  // exact: the product formed above
                }
            }
        }
        // ---- lists whose block is used up: the block in flight becomes the current one, the one behind it is requested
        static_for<0, T>([&](auto tc) __attribute__((always_inline)) {
            constexpr int t = decltype(tc)::value;
            if (cl[t][nb(t) - 1] <= Fc) {
                take_block(tc);
                pos[t] += 64u * nb(t);
                load_block(tc, pos[t] + 64u * nb(t), 0u);
                s[t] = 0;
                first_idx[t] = 0;
            }
        });
        __builtin_amdgcn_wave_barrier();
    }

#ifdef SEBM25_EXP_STATS  // steps, passes, postings that searched, offers of the longest list, other offers, items
    if (lane == 0) {
        atomicAdd(&g_step_stats[0], (unsigned long long)st_steps);
        atomicAdd(&g_step_stats[1], (unsigned long long)st_passes);
        atomicAdd(&g_step_stats[2], (unsigned long long)st_ins);
        atomicAdd(&g_step_stats[3], (unsigned long long)st_rounds);
        atomicAdd(&g_step_stats[4], (unsigned long long)st_offers);
        atomicAdd(&g_step_stats[5], 1ull);
    }
#endif
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
    __shared__ __align__(16) unsigned char topk_mem[TopK<CAP, true, false, uint32_t>::bytes()];
    const uint32_t q = blockIdx.x;
    const uint32_t nl = a.n[q];
    if (nl == 0) return;
    TopK<CAP, true, false, uint32_t> tk;
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

// Items are grouped by their number of lists: n_by_t[i] items with i + 1 lists for i < FAST_T (stepwise
// kernels, one instantiation per list count), then n_by_t[FAST_T] items for the general kernel.
template <int T, bool HAS_RANK>
static void launch_step_t(const ScoreArgs& b, hipStream_t st) {
    dim3 grid(b.n_items), block(64);
    // Chunks per block: two for the lists that insert (what they bring in a step must fit the table), four for the
    // longest list, which sets most frontiers (three with the 192-entry candidate buffer: measured).  More than four
    // lists: only three sizes of candidate storage.  The waves asked for leave every instantiation without spills.
    constexpr int BLK = STEP_BLOCK, BLK_L = 2 * STEP_BLOCK, BLK_MID = STEP_BLOCK_MID, BLK_192_L = 3, BLK_304_L = 4;
    constexpr int WAVES = T > 4 ? 3 : (T == 4 ? 5 : STEP_WAVES), WAVES_MID = T > 4 ? 2 : STEP_WAVES_MID;
    if (b.k <= 48) {
        hipLaunchKernelGGL((k_score_step<T, RegTopK<HAS_RANK>, HAS_RANK, WAVES, BLK, STEP_SLOTS, BLK_L>), grid, block, 0, st, b);
        return;
    }
    if constexpr (T <= 4) {
        if (b.k <= 128) {
            hipLaunchKernelGGL((k_score_step<T, TopK<192, HAS_RANK, HAS_RANK>, HAS_RANK, STEP_WAVES_MID, BLK_MID, STEP_SLOTS_MID, BLK_192_L>), grid, block, 0, st, b);
            return;
        }
    }
    if (b.k <= 256) {
        hipLaunchKernelGGL((k_score_step<T, TopK<304, HAS_RANK, HAS_RANK>, HAS_RANK, WAVES_MID, BLK_MID, STEP_SLOTS_MID, BLK_304_L>), grid, block, 0, st, b);
        return;
    }
    if constexpr (T <= 4) {  // candidate storage sized to k: the waves a CU holds fall with it (11, 7, 4 at three lists)
        if (b.k <= 512) {
            hipLaunchKernelGGL((k_score_step<T, TopK<576, HAS_RANK, HAS_RANK>, HAS_RANK, 2, BLK_MID, STEP_SLOTS_MID>), grid, block, 0, st, b);
            return;
        }
        if (b.k <= 1024) {
            hipLaunchKernelGGL((k_score_step<T, TopK<1088, HAS_RANK, HAS_RANK>, HAS_RANK, 2, BLK_MID, STEP_SLOTS_MID>), grid, block, 0, st, b);
            return;
        }
    }
    hipLaunchKernelGGL((k_score_step<T, TopK<2112, HAS_RANK, HAS_RANK>, HAS_RANK, 1, BLK_MID, 512>), grid, block, 0, st, b);
}
// Chunks per block of the merge-join kernels: (lists but the longest, longest list).  Longer blocks mean fewer steps
// and fuller passes, more LDS per wave means fewer waves.  10k three-term queries over the 10M-doc index, k = 10 /
// k = 210, before the searches of a pass ran side by side: (2, 4) 13.7 / 21.7 ms, (2, 6) 14.0 / 19.8, (2, 8) 13.1 / 20.5,
// (4, 8) 12.1 / 20.1, (2, 12) 13.9 / -, (2, 16) 15.7 / -; with them: (4, 8) 11.7 at k = 10, (3, 8) 18.0 and (2, 6) 18.7
// at k = 210 (four lists at k = 210: 42.8 against 41.2 ms per batch, so those keep (2, 6)).
#ifndef SEBM25_JOIN_BLOCK
#define SEBM25_JOIN_BLOCK 4
#endif
#ifndef SEBM25_JOIN_BLOCK_L
#define SEBM25_JOIN_BLOCK_L 8
#endif
#ifndef SEBM25_JOIN_WAVES
#define SEBM25_JOIN_WAVES 6
#endif
#ifndef SEBM25_JOIN_WAVES_MID
#define SEBM25_JOIN_WAVES_MID 4
#endif
template <int T, bool HAS_RANK>
static void launch_join_t(const ScoreArgs& b, hipStream_t st) {
    dim3 grid(b.n_items), block(64);
    constexpr int BLK = SEBM25_JOIN_BLOCK, BLK_L = SEBM25_JOIN_BLOCK_L, BM = T == 4 ? 2 : 3, BM_L = T == 4 ? 6 : 8;
    constexpr int W = SEBM25_JOIN_WAVES, WM = SEBM25_JOIN_WAVES_MID;
    if (b.k <= 48)
        hipLaunchKernelGGL((k_score_join<T, RegTopK<HAS_RANK>, HAS_RANK, W, BLK, BLK_L>), grid, block, 0, st, b);
    else if (b.k <= 128)
        hipLaunchKernelGGL((k_score_join<T, TopK<192, HAS_RANK, HAS_RANK>, HAS_RANK, WM, BM, BM_L>), grid, block, 0, st, b);
    else if (b.k <= 256)
        hipLaunchKernelGGL((k_score_join<T, TopK<304, HAS_RANK, HAS_RANK>, HAS_RANK, WM, BM, BM_L>), grid, block, 0, st, b);
    else if (b.k <= 512)
        hipLaunchKernelGGL((k_score_join<T, TopK<576, HAS_RANK, HAS_RANK>, HAS_RANK, 2, 2, 4>), grid, block, 0, st, b);
    else if (b.k <= 1024)
        hipLaunchKernelGGL((k_score_join<T, TopK<1088, HAS_RANK, HAS_RANK>, HAS_RANK, 2, 2, 4>), grid, block, 0, st, b);
    else
        hipLaunchKernelGGL((k_score_join<T, TopK<2112, HAS_RANK, HAS_RANK>, HAS_RANK, 1, 2, 4>), grid, block, 0, st, b);
}
// SEBM25_JOIN=0 in the environment sends queries of two to four lists to the table kernels (k_score_step) instead:
// for comparisons
static bool use_join() {
    static const bool on = [] {
        const char* e = getenv("SEBM25_JOIN");
        return !(e && e[0] == '0');
    }();
    return on;
}
template <int T>
static void launch_step(const ScoreArgs& b, hipStream_t st) {
    if constexpr (T >= 2 && T <= 4) {
        if (use_join()) {
            if (b.url_rank) launch_join_t<T, true>(b, st); else launch_join_t<T, false>(b, st);
            return;
        }
    }
    if (b.url_rank) launch_step_t<T, true>(b, st); else launch_step_t<T, false>(b, st);
}

void launch_score(const ScoreArgs& a, const uint32_t* n_by_t, hipStream_t st) {
    uint32_t first = 0;
    for (int g = 0; g <= FAST_T; ++g) {
        const uint32_t n = n_by_t[g];
        if (n == 0) continue;
        ScoreArgs b = a;
        b.items = a.items + first;
        b.n_items = n;
        first += n;
        dim3 grid(n), block(64);
        if (g == FAST_T) {
            if (a.k <= 64) hipLaunchKernelGGL(k_score<128>, grid, block, 0, st, b);
            else if (a.k <= 256) hipLaunchKernelGGL(k_score<320>, grid, block, 0, st, b);
            else hipLaunchKernelGGL(k_score<2112>, grid, block, 0, st, b);
        } else if (g == 0) launch_step<1>(b, st);
        else if (g == 1) launch_step<2>(b, st);
        else if (g == 2) launch_step<3>(b, st);
        else if (g == 3) launch_step<4>(b, st);
        else if (g == 4) launch_step<5>(b, st);
        else if (g == 5) launch_step<6>(b, st);
        else if (g == 6) launch_step<7>(b, st);
        else launch_step<8>(b, st);
    }
}

uint32_t score_fast_max_terms() { return FAST_T; }

#ifdef SEBM25_EXP_STATS
extern "C" int sebm25_exp_stats(unsigned long long* out8, int reset) {
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_step_stats), sizeof(z)) != hipSuccess) return 1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_step_stats), z, sizeof(z)) != hipSuccess) return 1;
    return 0;
}
#endif

void launch_merge(const MergeArgs& a, hipStream_t st) {
    if (a.n_queries == 0) return;
    dim3 grid(a.n_queries), block(64);
    if (a.k <= 64)
        hipLaunchKernelGGL(k_merge<128>, grid, block, 0, st, a);
    else if (a.k <= 256)
        hipLaunchKernelGGL(k_merge<320>, grid, block, 0, st, a);
    else if (a.k <= 512)
        hipLaunchKernelGGL(k_merge<576>, grid, block, 0, st, a);
    else if (a.k <= 1024)
        hipLaunchKernelGGL(k_merge<1088>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(k_merge<2112>, grid, block, 0, st, a);
}

}  // namespace sebm25
