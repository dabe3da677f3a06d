// Index-build kernels for gfx950 (wave64): stable LSD radix sort of token rows by term, the
// segmented reduction of sorted rows into postings, and the per-term / per-document
// statistics.  Replaces the GROUP BY t.term / COUNT(DISTINCT t.doc_id) / LN(...) statement of
// the reference (app/batch/bm25_stats.py:16-27) and the document aggregates at :9-13.
//
// Every kernel here is bandwidth-bound integer work: rows stream through in coalesced
// 64-lane accesses with all loads of a tile issued before the first use, digit ranking goes
// through per-wave LDS tables, and each workgroup owns a contiguous chunk of rows so that the
// runs it appends to a digit bin are adjacent in memory and merge in its XCD's L2.
#include "common.hpp"

namespace sebm25 {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Digit matching in the scatter kernel: which lanes of a wave hold the same digit.  Every lane
// ORs its lane bit into a per-wave LDS table slot chosen by the low MATCH_TAB_BITS digit bits and
// reads the slot back (all lanes that share those bits); remaining high bits are settled by
// ballots.  Table: 2 words (lanes 0-31, 32-63) per slot, all zero between uses.
constexpr int MATCH_TAB_BITS = 8;

__device__ __forceinline__ uint32_t hi32(uint64_t r) { return (uint32_t)(r >> 32); }
__device__ __forceinline__ uint32_t lo32(uint64_t r) { return (uint32_t)r; }
// low 32 bits of (r >> sh), sh wave-uniform
__device__ __forceinline__ uint32_t shr_lo32(uint64_t r, int sh) {
    const uint32_t a = __builtin_amdgcn_alignbit(hi32(r), lo32(r), (uint32_t)sh & 31u);
    const uint32_t b = hi32(r) >> ((uint32_t)sh & 31u);
    return sh < 32 ? a : b;
}

__device__ __forceinline__ uint32_t field_code(uint32_t field) { return field == 1u ? 1u : field == 2u ? 2u : 0u; }

// Cross-lane moves that stay in the vector ALU (DPP / readlane) instead of going through the LDS
// crossbar as __shfl does: lane i <- lane i-1 (lane 0 keeps its own value), lane i <- lane i+1
// (lane 63 keeps its own), and broadcasts of a fixed lane.
__device__ __forceinline__ uint32_t lane_up1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_dn1(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint64_t lane_up1(uint64_t x) {
    return ((uint64_t)lane_up1((uint32_t)(x >> 32)) << 32) | lane_up1((uint32_t)x);
}
__device__ __forceinline__ uint64_t lane_dn1(uint64_t x) {
    return ((uint64_t)lane_dn1((uint32_t)(x >> 32)) << 32) | lane_dn1((uint32_t)x);
}
// lane i <- lane i-1 / lane i+1 of x; the lane at the wave's edge takes `edge` instead
__device__ __forceinline__ uint32_t lane_up1_edge(uint32_t x, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_dn1_edge(uint32_t x, uint32_t edge) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)edge, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
// lane i <- lane i-1 of x, lane 0 <- lane 63 of prev: a rotation of prev puts its last lane into lane 0, the shift
// of x then fills every other lane (two DPP moves, no trip through a scalar register)
__device__ __forceinline__ uint32_t lane_up1_from(uint32_t x, uint32_t prev) {
    const int t = __builtin_amdgcn_update_dpp(0, (int)prev, 0x13C /* wave_ror:1 */, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
// lane i <- lane i+1 of x, lane 63 <- lane 0 of next
__device__ __forceinline__ uint32_t lane_dn1_from(uint32_t x, uint32_t next) {
    const int t = __builtin_amdgcn_update_dpp(0, (int)next, 0x134 /* wave_rol:1 */, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)x, 0x130 /* wave_shl:1 */, 0xF, 0xF, false);
}
template <int L>
__device__ __forceinline__ uint64_t lane_bcast(uint64_t x) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), L) << 32) |
           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, L);
}
// wave-wide sum, uniform result: quad swaps and row mirrors (DPP), then one readlane per row
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141 /* row_half_mirror */, 0xF, 0xF, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140 /* row_mirror */, 0xF, 0xF, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
           (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// ------------------------------------------------------------------ histogram (upsweep) ---
// Rows are cut into chunks of one tile (SORT_TILE rows), chunks into groups of SORT_GROUP.  A
// workgroup counts the digits of the chunks of one group, one wavefront per chunk, and writes
// for every chunk the counts of the group's earlier chunks (prefix[chunk][bin], 16 bits each) and at
// the end the group's totals (group_tot[group][bin]).  A chunk's write base for a bin is then
//   group_off[group][bin] + prefix[chunk][bin]
// with group_off = first slot of the bin (exclusive scan of the bin totals) + the exclusive scan of
// group_tot down the groups.  Chunks are small, so that the workgroups of one XCD, which take consecutive
// chunks, append to each bin next to one another and their runs merge into full lines in L2.
// MODE 0: raw rows, only term_id is read (the digit lies inside the term bits)
// MODE 1: raw rows, digit taken from the key packed from term / source / unit
// MODE 2: packed rows
constexpr int HIST_THREADS = 1024;
constexpr int HIST_UNROLL = 8;

// One count per lane.  Skewed or sorted input gives many lanes of a wave the same digit, and 64
// lanes adding to one LDS word serialise.  So the lanes that hold `hot` (the digit with the
// largest group seen lately, carried from call to call) are counted by one add, then the group
// of the first remaining lane by another (and becomes `hot` if it is larger); only what is left
// after that adds lane by lane.  Where the digits turn out evenly spread the grouping is
// skipped for the next few calls.
struct HistState {  // wave-uniform
    uint32_t hot = 0, plain_left = 0;
};
// hist holds two 16-bit counters per word (a chunk has fewer than 2^16 rows)
__device__ __forceinline__ void hist_bump(uint32_t* hist, uint32_t d, uint32_t n) {
    atomicAdd(&hist[d >> 1], n << ((d & 1u) * 16u));
}
__device__ __forceinline__ void hist_add(uint32_t* hist, uint32_t d, bool valid, HistState& st) {
    if (st.plain_left) {
        --st.plain_left;
        if (valid) hist_bump(hist, d, 1u);
        return;
    }
    const uint32_t lane = lane_id();
    const bool is_hot = valid && d == st.hot;
    const uint64_t hm = __ballot(is_hot);
    if (hm && lane == (uint32_t)__ffsll((unsigned long long)hm) - 1u) hist_bump(hist, st.hot, (uint32_t)__popcll(hm));
    const bool rest = valid && !is_hot;
    const uint64_t rm = __ballot(rest);
    if (rm == 0) return;
    const int first = __ffsll((unsigned long long)rm) - 1;
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, first);
    const bool in0 = rest && d == d0;
    const uint64_t m0 = __ballot(in0);
    if ((int)lane == first) hist_bump(hist, d0, (uint32_t)__popcll(m0));
    if (__popcll(m0) > __popcll(hm)) st.hot = d0;
    if (__popcll(m0) + __popcll(hm) < 12) st.plain_left = 15;
    if (rest && !in0) hist_bump(hist, d, 1u);
}

template <int MODE>
__global__ __launch_bounds__(HIST_THREADS) void k_sort_hist(RawRows raw, const uint64_t* packed, uint64_t R,
                                                            uint64_t chunk_rows, uint32_t G, int shift, int bits,
                                                            RowFmt fmt, uint32_t n_terms, unsigned short* prefix,
                                                            uint32_t* group_tot) {
    // one wavefront per chunk of the group, each with its own histogram: no barriers while counting
    __shared__ uint32_t hist[SORT_GROUP][SORT_MAX_BINS / 2];
    static_assert(HIST_THREADS == SORT_GROUP * 64, "one wavefront per chunk of a group");
    const uint32_t nb = 1u << bits, mask = nb - 1u;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    for (uint32_t x = threadIdx.x; x < SORT_GROUP * SORT_MAX_BINS / 2; x += HIST_THREADS) (&hist[0][0])[x] = 0;
    __syncthreads();
    const uint32_t g_begin = blockIdx.x * SORT_GROUP;
    const uint32_t g = g_begin + wave;
    if (g < G) {
        uint32_t* h = hist[wave];
        HistState hot;
        const uint64_t begin = (uint64_t)g * chunk_rows;
        const uint64_t end = min(R, begin + chunk_rows);
        const uint64_t n = end - begin;
        uint64_t done = 0;  // rows counted by the vector loop
        if (MODE == 0) {
            // chunk starts are multiples of the tile size, so 16-byte vector loads stay aligned
            const int sh = shift - fmt.term_shift();
            const uint4* p = reinterpret_cast<const uint4*>(raw.term + begin);
            const uint64_t nvec = n / 4;
            for (uint64_t v0 = 0; v0 < nvec; v0 += 64u * HIST_UNROLL) {
                uint4 x[HIST_UNROLL];
#pragma unroll
                for (int u = 0; u < HIST_UNROLL; ++u) {
                    const uint64_t v = v0 + (uint64_t)u * 64u + lane;
                    x[u] = v < nvec ? p[v] : make_uint4(0, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < HIST_UNROLL; ++u) {
                    const bool ok = v0 + (uint64_t)u * 64u + lane < nvec;
                    // term ids are clamped exactly as the scatter clamps them, so both agree on every digit
                    hist_add(h, (min(x[u].x, n_terms - 1u) >> sh) & mask, ok, hot);
                    hist_add(h, (min(x[u].y, n_terms - 1u) >> sh) & mask, ok, hot);
                    hist_add(h, (min(x[u].z, n_terms - 1u) >> sh) & mask, ok, hot);
                    hist_add(h, (min(x[u].w, n_terms - 1u) >> sh) & mask, ok, hot);
                }
            }
            done = nvec * 4;
        } else if (MODE == 2) {
            const ulonglong2* p = reinterpret_cast<const ulonglong2*>(packed + begin);
            const uint64_t nvec = n / 2;
            for (uint64_t v0 = 0; v0 < nvec; v0 += 64u * HIST_UNROLL) {
                ulonglong2 x[HIST_UNROLL];
#pragma unroll
                for (int u = 0; u < HIST_UNROLL; ++u) {
                    const uint64_t v = v0 + (uint64_t)u * 64u + lane;
                    x[u] = v < nvec ? p[v] : make_ulonglong2(0, 0);
                }
#pragma unroll
                for (int u = 0; u < HIST_UNROLL; ++u) {
                    const bool ok = v0 + (uint64_t)u * 64u + lane < nvec;
                    hist_add(h, shr_lo32(x[u].x, shift) & mask, ok, hot);
                    hist_add(h, shr_lo32(x[u].y, shift) & mask, ok, hot);
                }
            }
            done = nvec * 2;
        }
        // what the vector loop left (MODE 1: everything), one row per lane
        for (uint64_t i0 = begin + done; i0 < end; i0 += 64u) {
            const uint64_t i = i0 + lane;
            const bool ok = i < end;
            uint32_t d = 0;
            if (ok) {
                if (MODE == 0) {
                    d = (min(raw.term[i], n_terms - 1u) >> (shift - fmt.term_shift())) & mask;
                } else if (MODE == 2) {
                    d = shr_lo32(packed[i], shift) & mask;
                } else {
                    const uint32_t m = raw.meta[i];
                    const uint64_t key = fmt.pack(min(raw.term[i], n_terms - 1u), ((m & 3u) == SEBM25_SRC_NEWS) ? 1u : 0u,
                                                  raw.unit[i] & (uint32_t)((1ull << fmt.unit_bits) - 1ull), 0u, 0ull);
                    d = (uint32_t)(key >> shift) & mask;
                }
            }
            hist_add(h, d, ok, hot);
        }
    }
    __syncthreads();
    // counts of the group's earlier chunks, per chunk and bin (two 16-bit entries per word: a group
    // has at most SORT_GROUP * SORT_TILE = 65536 rows, so every count before its last chunk fits);
    // group totals in 32 bits
    static_assert(SORT_GROUP * SORT_TILE <= 65536, "prefix entries are 16 bits wide");
    uint32_t* prefix32 = reinterpret_cast<uint32_t*>(prefix);
    const uint32_t nw = max(nb / 2u, 1u);  // words per chunk
    for (uint32_t x = threadIdx.x; x < nw; x += HIST_THREADS) {
        uint32_t lo = 0, hi = 0;
#pragma unroll 4
        for (uint32_t w = 0; w < SORT_GROUP; ++w) {
            if (g_begin + w < G) {
                prefix32[(size_t)(g_begin + w) * nw + x] = lo | (hi << 16);
                const uint32_t word = hist[w][x];
                lo += word & 0xFFFFu;
                hi += word >> 16;
            }
        }
        group_tot[(size_t)blockIdx.x * nb + 2u * x] = lo;
        if (2u * x + 1u < nb) group_tot[(size_t)blockIdx.x * nb + 2u * x + 1u] = hi;
    }
}

// group_tot[group][bin] -> first slot of every (group, bin): an exclusive scan down the groups plus
// the first slot of the bin.  The groups are cut into SCAN_SEGS segments; lane = bin, so every access
// is a coalesced 256-byte piece of a row of the matrix.
//   k_sort_seg_sums : seg_tot[seg][bin] = sum of the segment's groups
//   k_sort_seg_scan : seg_tot -> exclusive scan down the segments (in place), bin_tot[bin] = total
//   (exclusive scan of bin_tot -> first slot of every bin)
//   k_sort_seg_apply: group_tot[group][bin] -> bin start + segment start + groups before it
constexpr uint32_t SCAN_SEGS = 64;

__global__ __launch_bounds__(256) void k_sort_seg_sums(const uint32_t* group_tot, uint32_t n_groups, uint32_t nb,
                                                       uint32_t seg_len, uint32_t* seg_tot) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);  // (bin block, segment)
    const uint32_t blocks = (nb + 63u) / 64u;
    const uint32_t seg = w / blocks, b = (w % blocks) * 64u + lane_id();
    if (seg >= SCAN_SEGS || b >= nb) return;
    const uint32_t g0 = seg * seg_len, g1 = min(n_groups, g0 + seg_len);
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t g = g0; g < g1; ++g) sum += group_tot[(size_t)g * nb + b];
    seg_tot[seg * nb + b] = sum;
}

__global__ __launch_bounds__(256) void k_sort_seg_scan(uint32_t* seg_tot, uint32_t nb, uint32_t* bin_tot) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= nb) return;
    uint32_t run = 0;
#pragma unroll 8
    for (uint32_t s = 0; s < SCAN_SEGS; ++s) {
        const uint32_t v = seg_tot[s * nb + b];
        seg_tot[s * nb + b] = run;
        run += v;
    }
    bin_tot[b] = run;
}

__global__ __launch_bounds__(256) void k_sort_seg_apply(uint32_t* group_tot, uint32_t n_groups, uint32_t nb,
                                                        uint32_t seg_len, const uint32_t* seg_off,
                                                        const uint32_t* bin_start) {
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t blocks = (nb + 63u) / 64u;
    const uint32_t seg = w / blocks, b = (w % blocks) * 64u + lane_id();
    if (seg >= SCAN_SEGS || b >= nb) return;
    const uint32_t g0 = seg * seg_len, g1 = min(n_groups, g0 + seg_len);
    uint32_t run = bin_start[b] + seg_off[seg * nb + b];
    constexpr uint32_t BATCH = 8;  // loads of a batch are independent: issued together
    for (uint32_t g = g0; g < g1; g += BATCH) {
        uint32_t v[BATCH];
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) v[k] = g + k < g1 ? group_tot[(size_t)(g + k) * nb + b] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) {
            if (g + k < g1) group_tot[(size_t)(g + k) * nb + b] = run;
            run += v[k];
        }
    }
}

// 32-bit views of the packed row's fields.  The 64-bit shifts and compares of RowFmt cost
// several times a 32-bit operation; here every test is a mask on one or both halves of the row
// and every extraction a funnel shift.  All members are wave-uniform.
struct RowBits {
    uint32_t key_hi, key_lo;  // (term, source, unit) bits
    uint32_t term_hi;         // term bits: always inside the high word (term ids are 32-bit)
    uint32_t src_hi, src_lo;  // the source bit
    uint32_t big_hi, big_lo;  // tf bits from 2^24 up
    uint32_t unit_mask, tf24_mask;
    int term_sh, unit_sh, fc_sh;
    __device__ explicit RowBits(const RowFmt& f) {
        const uint64_t key = ~0ull << f.key_shift(), src = 1ull << f.src_shift();
        const uint64_t tf = (1ull << f.tf_bits) - 1ull, big = tf & ~0xFFFFFFull;
        key_hi = (uint32_t)(key >> 32); key_lo = (uint32_t)key;
        term_hi = (uint32_t)((~0ull << f.term_shift()) >> 32);
        src_hi = (uint32_t)(src >> 32); src_lo = (uint32_t)src;
        big_hi = (uint32_t)(big >> 32); big_lo = (uint32_t)big;
        unit_mask = (uint32_t)((1ull << f.unit_bits) - 1ull);
        tf24_mask = (uint32_t)(tf & 0xFFFFFFull);
        term_sh = f.term_shift() - 32; unit_sh = f.unit_shift(); fc_sh = f.tf_bits;
    }
};
__device__ __forceinline__ bool key_differs(const RowBits& b, uint64_t x, uint64_t y) {
    return (((hi32(x) ^ hi32(y)) & b.key_hi) | ((lo32(x) ^ lo32(y)) & b.key_lo)) != 0u;
}
__device__ __forceinline__ bool is_news(const RowBits& b, uint64_t r) {
    return ((hi32(r) & b.src_hi) | (lo32(r) & b.src_lo)) != 0u;
}
__device__ __forceinline__ bool tf_big(const RowBits& b, uint64_t r) {
    return ((hi32(r) & b.big_hi) | (lo32(r) & b.big_lo)) != 0u;
}
// weight of one row in tenths, tf < 2^24: web tf * {32, 17, 10}[field code] (W = 3.2 / 1.7 / 1.0,
// search_service.py:19-27), news 10 * tf (:84-87)
__device__ __forceinline__ uint32_t tenths24(const RowBits& b, uint64_t r, bool news) {
    // the multiplier by table: byte fc of 0x0A11200A (10, 32, 17, 10), selected with the field code
    // taken from the row already multiplied by 8 (fc_sh >= 8 because tf has at least 8 bits)
    const uint32_t sh = shr_lo32(r, b.fc_sh - 3) & 0x18u;
    const uint32_t mult = news ? 10u : __builtin_amdgcn_ubfe(0x0A11200Au, sh, 8u);
    return __umul24(lo32(r) & b.tf24_mask, mult);
}

// ------------------------------------------------------------------ posting slots from the last pass ---
// Outputs of the last sort pass when it writes posting slots (k_sort_scatter_tile<2, true>).
struct PostOut {
    Slot* post;           // [R + POST_PAD] unit of every sorted row; weight (tenths) at the head of every kept run, POST_HOLE elsewhere
    uint32_t* news_ptr;   // [n_terms], preset to 0xFFFFFFFF: first news slot of the term (atomic minimum)
    uint32_t* dup;        // [n_terms], zeroed: web rows of the term that are not the head of their run
    uint4* partials;      // [chunks]: kept postings, web heads, kept web postings of the tile
};
constexpr uint32_t POST_NBR = 64;  // rows looked at on either side of a tile
// aliases the per-wave digit counters, which are dead once the tile is reordered
struct PostTileShared {
    uint64_t nbr[2 * POST_NBR];  // rows before the tile (nearest first), then rows behind it
    uint64_t first, last;        // first and last row of the tile in input order
    uint32_t n_tail, n_ahead;    // neighbours (counted from the tile) that share the first / last row's lower key bits
    uint32_t tot[3];
};
static_assert(sizeof(PostTileShared) <= SORT_WAVES * SORT_MAX_BINS * 2, "fits the digit counters");

// rows_s holds the tile ordered by the digit of this pass (stable), i.e. by term, and inside a term in input
// order: (source, unit) ascending, the field rows of one unit next to each other.  base[d] + p is the slot of
// the row at place p.  Wave w takes places [512 w, 512 w + 512) in rounds of 64, neighbours through lane
// shifts as in k_postings.  What the tile cannot see:
//  - whether the first row of a run continues a run of the tile before.  Rows of one (term, source, unit) agree
//    in every key bit below this pass's digit, and the input is ordered by those bits, so such a run can only
//    be one whose lower key bits equal those of the tile's first input row; it is a continuation iff one of
//    the rows before the tile that still share those bits (a handful: the rows of one unit whose terms share
//    the earlier digits) has the same full key.
//  - rows of a run behind the tile's end: the same with the tile's last input row and the rows behind it.
// More than POST_NBR such neighbours, runs of more than three rows and tf >= 2^24 raise DEV_NEED_FIXUP: the
// host then sorts the rows out and runs k_postings over them.
// df is not counted here: a term's web heads are its web rows minus the web rows that are not heads, and only
// the latter (about 3 % of the rows of a crawled corpus) cost an atomic.
template <bool LOWKEY>
__device__ __forceinline__ void post_tile(uint64_t* rows_s, const uint32_t* base, PostTileShared* ps, uint32_t n_tile,
                                          uint64_t tile0, uint64_t end, uint64_t R, uint32_t chunk, int shift, uint32_t mask,
                                          const RowFmt& fmt, uint64_t nbr, uint64_t edge_row, uint64_t row0, uint32_t* err,
                                          const PostOut& po) {
    // the wave index is the same for all lanes: tell the compiler, so that everything derived from it stays scalar
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    const RowBits rb(fmt);
    const uint64_t cut = (~0ull << fmt.key_shift()) & ~((uint64_t)mask << shift);  // key bits below this pass's digit
    const uint32_t cut_hi = hi32(cut), cut_lo = lo32(cut);
    auto cut_equal = [&](uint64_t a, uint64_t b) {
        return (((hi32(a) ^ hi32(b)) & cut_hi) | ((lo32(a) ^ lo32(b)) & cut_lo)) == 0u;
    };
    if (wave == 0u) {
        const uint64_t first = lane_bcast<0>(row0);
        const uint64_t mm = __ballot(tile0 > lane && cut_equal(nbr, first));
        const uint32_t n = mm == ~0ull ? POST_NBR : (uint32_t)__ffsll((unsigned long long)~mm) - 1u;
        ps->nbr[lane] = nbr;
        if (lane == 0) {
            ps->first = first;
            ps->n_tail = n;
            ps->tot[0] = ps->tot[1] = ps->tot[2] = 0;
            if (n == POST_NBR) atomicOr(err, DEV_NEED_FIXUP);
        }
    } else if (wave == 1u) {
        const uint64_t mm = __ballot(end + lane < R && cut_equal(nbr, edge_row));
        const uint32_t n = mm == ~0ull ? POST_NBR : (uint32_t)__ffsll((unsigned long long)~mm) - 1u;
        ps->nbr[POST_NBR + lane] = nbr;
        if (lane == 0) {
            ps->last = edge_row;
            ps->n_ahead = n;
            if (n == POST_NBR) atomicOr(err, DEV_NEED_FIXUP);
        }
    }
    if (n_tile < (uint32_t)SORT_TILE) {
        // a partial tile: what follows its last row in the tile buffer reads as a row of another unit
        const uint64_t sent = rows_s[n_tile - 1u] ^ (1ull << fmt.key_shift());
        for (uint32_t q = n_tile + threadIdx.x; q < (uint32_t)SORT_TILE; q += SORT_THREADS) rows_s[q] = sent;
    }
    __syncthreads();
    const uint32_t n_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)ps->n_tail);
    const uint32_t n_ahead = (uint32_t)__builtin_amdgcn_readfirstlane((int)ps->n_ahead);
    const uint64_t first = ps->first, last = ps->last;
    const uint32_t wb = wave * (64u * SORT_ITEMS);
    uint32_t n_kept = 0, n_wh = 0, n_kept_web = 0;
    if (wb < n_tile) {
        // Every yes/no per row is kept as a wave-wide lane mask in scalar registers: masks of plain compares
        // are free, combining them is scalar work, and only a select or a guarded store turns one back into a
        // per-lane condition.
        auto ballot = [](bool c) { return (uint64_t)__builtin_amdgcn_ballot_w64(c); };
        auto lanes = [](uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); };
        const uint32_t ts_hi = rb.term_hi | rb.src_hi, ts_lo = rb.src_lo;
        const bool check_big = (rb.big_hi | rb.big_lo) != 0u;  // tf can reach 2^24 at all
        const int dsh = shift - 32;                              // this pass's digit lies in the high word
        uint32_t seen_hi = 0, seen_lo = 0;
        // One round ahead of the round being written: per row its own weight in tenths, per round the rows whose key
        // differs from the row before (df), the news rows (nm) and the rows of web docs that are 'done' (ok: field
        // code 3 marks the others, set by the first sort pass).  xp: the round before (its last lane precedes lane 0).
        auto stage = [&](uint64_t xx, uint64_t xp, uint32_t& t10, uint64_t& df, uint64_t& nmask, uint64_t& ok) {
            const uint32_t dh = hi32(xx) ^ lane_up1_from(hi32(xx), hi32(xp));
            const uint32_t dl = lo32(xx) ^ lane_up1_from(lo32(xx), lo32(xp));
            const bool nw = (hi32(xx) & rb.src_hi) != 0u;  // at most 2^20 terms here: the source bit is in the high word
            df = ballot((LOWKEY ? (dl & rb.key_lo) | dh : dh & rb.key_hi) != 0u);
            nmask = ballot(nw);
            // the field code times 8: selects the multiplier's byte of 0x0A11200A (10, 32, 17, 10) and tells rows of docs not done
            const uint32_t fc8 = (LOWKEY ? lo32(xx) >> (rb.fc_sh - 3) : shr_lo32(xx, rb.fc_sh - 3)) & 0x18u;
            ok = ballot(fc8 != 0x18u);
            const uint32_t mult = nw ? 10u : __builtin_amdgcn_ubfe(0x0A11200Au, fc8, 8u);
            t10 = __umul24(lo32(xx) & rb.tf24_mask, mult);
            if (check_big) {
                seen_hi |= hi32(xx);
                seen_lo |= lo32(xx);
            }
        };
        // rows behind the tile read as rows of another unit, the row before the tile as one of another term
        const uint64_t sent = rows_s[n_tile - 1u] ^ (1ull << fmt.key_shift());
        const bool last_wave = wave == SORT_WAVES - 1u;
        uint64_t x0 = rows_s[wb + lane];
        uint64_t xp = wave ? rows_s[wb - 1u] : rows_s[0] ^ (1ull << 63);  // every lane: the row before the wave's first
        uint32_t t0w = 0;
        uint64_t df0 = 0, nm0 = 0, ok0 = 0;
        stage(x0, xp, t0w, df0, nm0, ok0);
        uint64_t fix_m = 0, big_m = 0;
        const int rows_here = (int)(n_tile - wb);  // rows of the tile from this wave's first on

        constexpr uint32_t NO_TERM = 0xFFFFFFFFu;
        uint32_t run_term = NO_TERM, run_cnt = 0;  // web rows that are no heads, of the term the last rounds ended with
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const uint32_t p = wb + (uint32_t)j * 64u + lane;
            // the next round (the last wave's ninth would lie behind the tile buffer)
            uint64_t x1;
            if (j + 1 < SORT_ITEMS) {
                x1 = rows_s[wb + (uint32_t)(j + 1) * 64u + lane];
            } else {
                x1 = rows_s[min(wb + (uint32_t)SORT_ITEMS * 64u + (lane & 3u), (uint32_t)SORT_TILE - 1u)];
                if (last_wave) x1 = sent;
            }
            uint32_t t1w = 0;
            uint64_t df1 = 0, nm1 = 0, ok1 = 0;
            stage(x1, x0, t1w, df1, nm1, ok1);

            const int left = rows_here - j * 64;
            const uint64_t vm = left >= 64 ? ~0ull : left <= 0 ? 0ull : (1ull << left) - 1ull;
            // weights of the next two rows: lane shifts; at the wave edge the first lanes of the next round
            const uint32_t n1 = lane_dn1_from(t0w, t1w);
            const uint32_t n2 = lane_dn1_edge(n1, (uint32_t)__builtin_amdgcn_readlane((int)t1w, 1));
            // rows whose next one / two / three rows continue their run
            const uint64_t s1 = ~((df0 >> 1) | (df1 << 63));
            const uint64_t s2 = s1 & ~((df0 >> 2) | (df1 << 62));
            const uint64_t s3 = s2 & ~((df0 >> 3) | (df1 << 61));
            const uint32_t term = hi32(x0) >> rb.term_sh;
            const uint32_t unit = LOWKEY ? __builtin_amdgcn_alignbit(hi32(x0), lo32(x0), (uint32_t)rb.unit_sh) & rb.unit_mask
                                         : (hi32(x0) >> (rb.unit_sh - 32)) & rb.unit_mask;
            const uint32_t slot = base[(hi32(x0) >> dsh) & mask] + p;
            uint64_t hm = df0 & vm;  // heads
            // the first row of a run that may have begun before the tile
#ifndef SEBM25_EXP_NONBR
            if (n_tail != 0u) {
                const uint64_t c0m = hm & ballot(cut_equal(x0, first));
                if (c0m != 0ull) {
                    bool cont = false;
                    if (lanes(c0m))
#pragma unroll 1
                        for (uint32_t i = 0; i < n_tail; ++i)
                            if (!key_differs(rb, ps->nbr[i], x0)) cont = true;
                    hm &= ~ballot(cont);
                }
            }
#endif
            const uint64_t km = hm & (nm0 | ok0);  // kept: the slot gets a weight
            const uint64_t wm = hm & ~nm0;         // web heads
            uint32_t w = t0w + (lanes(s1) ? n1 : 0u) + (lanes(s2) ? n2 : 0u);
            fix_m |= km & s3;  // more than three rows in one run: left to the exact pass
            // a run that may go on behind the tile: the rows there add to its weight, and count as merged rows
#ifndef SEBM25_EXP_NONBR
            if (n_ahead != 0u) {
                const uint64_t c1m = hm & ballot(cut_equal(x0, last));
                if (c1m != 0ull) {
                    const bool nwj = lanes(nm0);
                    const uint32_t here = 1u + (lanes(s1) ? 1u : 0u) + (lanes(s2) ? 1u : 0u);
                    uint32_t more = 0;
                    if (lanes(c1m))
#pragma unroll 1
                        for (uint32_t i = 0; i < n_ahead; ++i) {
                            const uint64_t y = ps->nbr[POST_NBR + i];
                            if (!key_differs(rb, y, x0)) {
                                w += tenths24(rb, y, nwj);
                                ++more;
                            }
                        }
                    fix_m |= km & ballot(here + more > 3u);
                    if (lanes(wm) && more != 0u) atomicAdd(&po.dup[term], more);
                }
            }
#endif
            if (lanes(vm)) po.post[slot] = make_uint2(unit, lanes(km) ? w : POST_HOLE);
            // weights of 0xFFFF tenths and more do not fit the 16-bit accumulator fields of the stepwise scoring kernels
            big_m |= km & ballot(w >= 0xFFFFu);
            n_kept += (uint32_t)__popcll(km);
            n_wh += (uint32_t)__popcll(wm);
            n_kept_web += (uint32_t)__popcll(km & wm);
            // the first news row of a term in this tile: candidate for the term's first news slot
            const uint64_t nvm = nm0 & vm;
            if (nvm != 0ull) {
                const uint32_t dh = hi32(x0) ^ lane_up1_from(hi32(x0), hi32(xp));
                const uint32_t dl = lo32(x0) ^ lane_up1_from(lo32(x0), lo32(xp));
                if (lanes(nvm & ballot(((dh & ts_hi) | (dl & ts_lo)) != 0u))) atomicMin(&po.news_ptr[term], slot);
            }

            // Web rows that are not the head of their run, per term: every web head brings the rows that continue
            // its run (one or two; longer runs go to the exact pass).  The tile is ordered by term, so a round whose
            // first and last row agree in it is a round of one term and only extends a running count; in a round of
            // several terms the first term's rows end the running count, the last term's start the next, and the
            // terms in between (short lists) add per head.
            const uint32_t t_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)term);
            const uint32_t t_last = (uint32_t)__builtin_amdgcn_readlane((int)term, left >= 64 ? 63 : left > 0 ? left - 1 : 0);
#ifndef SEBM25_EXP_NODUP
            const uint64_t d1 = wm & s1, d2 = wm & s2;
            if (d1 != 0ull) {
                if (t_first != run_term) {
                    if (lane == 0 && run_cnt) atomicAdd(&po.dup[run_term], run_cnt);
                    run_term = t_first;
                    run_cnt = 0;
                }
                if (t_first == t_last) {
                    run_cnt += (uint32_t)__popcll(d1) + (uint32_t)__popcll(d2);
                } else {
                    const uint64_t seg_first = ballot(term == t_first), seg_last = ballot(term == t_last);
                    run_cnt += (uint32_t)__popcll(d1 & seg_first) + (uint32_t)__popcll(d2 & seg_first);
                    if (lane == 0 && run_cnt) atomicAdd(&po.dup[run_term], run_cnt);
                    run_term = t_last;
                    run_cnt = (uint32_t)__popcll(d1 & seg_last) + (uint32_t)__popcll(d2 & seg_last);
                    const uint64_t mid = d1 & ~seg_first & ~seg_last;
                    if (mid != 0ull) {
                        if (lanes(mid)) atomicAdd(&po.dup[term], lanes(d2) ? 2u : 1u);
                    }
                }
            }
#endif
            xp = x0; x0 = x1; t0w = t1w; df0 = df1; nm0 = nm1; ok0 = ok1;
        }
        if (check_big) fix_m |= ballot(((seen_hi & rb.big_hi) | (seen_lo & rb.big_lo)) != 0u);
        if (lane == 0) {
            if (run_cnt) atomicAdd(&po.dup[run_term], run_cnt);
            if (fix_m) atomicOr(err, DEV_NEED_FIXUP);
            if (big_m) atomicOr(err, DEV_BIG_WEIGHT);
            atomicAdd(&ps->tot[0], n_kept);
            atomicAdd(&ps->tot[1], n_wh);
            atomicAdd(&ps->tot[2], n_kept_web);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) po.partials[chunk] = make_uint4(ps->tot[0], ps->tot[1], ps->tot[2], 0u);
}

constexpr int DL_SPAN = SORT_THREADS;  // doc ids covered by a tile's LDS body-length counters

// ------------------------------------------------------------------ scatter (downsweep) ---
// MODE 0: raw rows, digit inside the term bits (unit-major input: the usual first pass)
// MODE 1: raw rows, digit anywhere in the packed key (first pass of a (term, source, unit) sort)
// MODE 2: packed rows
// Raw rows are validated, packed and their body lengths accumulated on the way through.
//
// One tile (= one chunk) per workgroup: no tile loop, so the raw columns are only live until the
// rows are packed and the kernel fits 80 vector registers; the digit-match tables and the
// body-length counters share the LDS of the tile buffer (they are dead before the tile is
// reordered into it).  52 KB of LDS and 80 registers put three workgroups on a CU, which is what
// hides the latency of a tile's phases (load, rank, offsets, reorder, store); a looped form with
// the next tile's loads issued ahead of the stores, two workgroups per CU, was 45 % slower.
//
// POST (packed rows, last pass of a sort on the term bits of unit-major rows): the pass writes no rows.  Its
// tile, reordered by digit in LDS, holds every (term, source, unit) run of the final order in one piece
// except the runs of the one unit a tile boundary cuts, so each row's posting slot (unit; weight of the run
// at its head, a hole elsewhere) is written straight to the slot the row would have been scattered to.
// See the posting phase at the end of the kernel.
template <int MODE, bool POST>
__global__ __launch_bounds__(SORT_THREADS, 6) void k_sort_scatter_tile(RawRows raw, const uint64_t* packed, uint64_t* dst,
                                                                    uint64_t R, uint32_t G, int shift, int bits,
                                                                    BuildCtx ctx, const unsigned short* prefix,
                                                                    const uint32_t* group_off, PostOut po) {
    static_assert(!POST || MODE == 2, "posting slots are written from packed rows");
    __shared__ __align__(16) unsigned short whist[SORT_WAVES][SORT_MAX_BINS];  // per-wave digit counts, then local slots
    __shared__ __align__(16) uint64_t rows_s[SORT_TILE];                       // the tile, locally ordered by digit
    __shared__ __align__(8) uint32_t base[SORT_MAX_BINS];  // first the chunk's write base per digit, then global minus local slot
    __shared__ uint32_t wave_tot[SORT_WAVES];
    uint32_t* mtab = reinterpret_cast<uint32_t*>(rows_s);     // per wave: lane masks per folded digit (16 KB)
    uint32_t* dl_s = mtab + SORT_WAVES * (2u << MATCH_TAB_BITS);  // body length per doc of this tile (2 KB)
    static_assert(SORT_WAVES * (2u << MATCH_TAB_BITS) * 4 + DL_SPAN * 4 <= SORT_TILE * 8, "aliases fit the tile buffer");
    uint32_t* wh32 = reinterpret_cast<uint32_t*>(&whist[0][0]);  // two 16-bit counters per word

    const RowFmt fmt = ctx.fmt;
    const uint32_t nb = 1u << bits, mask = nb - 1u;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t t = threadIdx.x;
    // the last pass of a sort on the term bits takes its digit from the high word alone (term ids are 32-bit)
    auto digit_of = [&](uint64_t r) { return POST ? (hi32(r) >> (shift - 32)) & mask : shr_lo32(r, shift) & mask; };
    // workgroups b, b + 8, b + 16 ... run on one XCD: give them consecutive chunks, so that the runs
    // they append to a digit bin at about the same time are neighbours in that XCD's L2
    const uint32_t per_xcd = (G + 7u) / 8u;
#ifdef SEBM25_EXP_LINEAR_CHUNKS  // experiment builds only: chunks in dispatch order (what a ticket-ordered one-sweep pass would see)
    const uint32_t chunk = MODE == 2 ? blockIdx.x : (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
#elif defined(SEBM25_EXP_RUN)  // experiment builds only: the last pass takes runs of SEBM25_EXP_RUN chunks per XCD, interleaved
    uint32_t chunk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (POST) {
        const uint32_t blk = 8u * SEBM25_EXP_RUN, b = blockIdx.x;
        chunk = b < (G / blk) * blk ? (b / blk) * blk + (b & 7u) * SEBM25_EXP_RUN + ((b % blk) >> 3) : b;
    }
#else
    const uint32_t chunk = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
#endif
    if (chunk >= G) return;
    const uint64_t tile0 = (uint64_t)chunk * SORT_TILE;
    const uint64_t end = min(R, tile0 + SORT_TILE);
    const uint64_t wave0 = tile0 + (uint64_t)wave * 64u * SORT_ITEMS;  // wave w owns 8 rounds of 64 rows

    // the chunk's write base of digits 2t and 2t + 1: one 8-byte and one 4-byte load per thread from
    // clamped addresses, no loop (loads retire in issue order, a loop would wait for the rows too)
    static_assert(SORT_MAX_BINS == 2 * SORT_THREADS, "two digits per thread");
    uint2 base_g = make_uint2(0u, 0u);
    uint32_t base_p = 0;
    auto load_bases = [&]() {
        const uint32_t d = min(2u * t, nb - 2u);
        base_g = *reinterpret_cast<const uint2*>(group_off + (size_t)(chunk / SORT_GROUP) * nb + d);
        base_p = *reinterpret_cast<const uint32_t*>(prefix + (size_t)chunk * nb + d);
    };
    const uint32_t n_tile = (uint32_t)(end - tile0);
    const uint32_t l0 = wave * 64u * SORT_ITEMS + lane;  // position in the tile of this lane's row of round 0
    const uint64_t* packed_t = packed + tile0;
    const uint32_t* term_t = raw.term + tile0;
    const uint32_t* unit_t = raw.unit + tile0;
    const uint32_t* tf_t = raw.tf + tile0;
    const uint8_t* meta_t = raw.meta + tile0;
    uint64_t row[SORT_ITEMS];
    uint32_t term[SORT_ITEMS], unit[SORT_ITEMS], tfv[SORT_ITEMS], m[SORT_ITEMS];  // raw columns (MODE 0, 1)
    uint32_t prev_u = 0, prev_m = 0, first_u = 0, first_m = 0;  // row before this wave's rows; first row of the tile
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        const uint32_t i = min(l0 + (uint32_t)j * 64u, n_tile - 1u);  // 32-bit offsets from the tile's uniform base
        if (MODE == 2) {
            row[j] = packed_t[i];
        } else {
            term[j] = term_t[i];
            unit[j] = unit_t[i];
            tfv[j] = tf_t[i];
            m[j] = meta_t[i];
        }
    }
    if (MODE != 2) {
        const uint64_t ip = min(wave0 ? wave0 - 1 : 0, end - 1);
        prev_u = raw.unit[ip];
        prev_m = raw.meta[ip];
        first_u = raw.unit[tile0];
        first_m = raw.meta[tile0];
    }
    // POST: the 64 rows before the tile (wave 0, nearest first) and the 64 rows behind it (wave 1, with the
    // tile's last row), from clamped addresses
    uint64_t nbr = 0, edge_row = 0;
    if (POST && wave < 2u) {
        const uint64_t i = wave == 0u ? (tile0 > lane ? tile0 - 1u - lane : 0ull) : min(end + lane, R - 1u);
        nbr = packed[i];
        edge_row = packed[end - 1u];
    }
    asm volatile("" ::: "memory");  // keeps the loads below from moving up between the row loads
    load_bases();
    // every load above is issued before anything waits for the bases (the operands tie the adds below to this point)
    asm volatile("" : "+v"(base_g.x), "+v"(base_g.y), "+v"(base_p)::"memory");
    // entries of digits >= nb hold a clamped copy and are never read
    *reinterpret_cast<uint2*>(&base[2u * t]) = make_uint2(base_g.x + (base_p & 0xFFFFu), base_g.y + (base_p >> 16));
    for (uint32_t x = t; x < SORT_WAVES * SORT_MAX_BINS / 2; x += SORT_THREADS) wh32[x] = 0;
    for (uint32_t x = t; x < SORT_WAVES * (2u << MATCH_TAB_BITS) + DL_SPAN; x += SORT_THREADS) mtab[x] = 0;
    __syncthreads();

    uint32_t tile_u_first = 0;
    bool tile_web = false;
    if (MODE == 2) {
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j)
            if (l0 + (uint32_t)j * 64u >= n_tile) row[j] = ~0ull;
    } else {
        // validate and pack the raw rows, check unit-major order, accumulate body lengths
        tile_u_first = first_u;
        tile_web = (first_m & 3u) == SEBM25_SRC_WEB;
        // (source, unit) of the row before this wave's first row
        uint64_t carry = wave0 > 0 ? (((uint64_t)(((prev_m & 3u) == SEBM25_SRC_NEWS) ? 1u : 0u) << 32) | prev_u) : 0ull;
        // the packed row in 32-bit halves: term and source only reach the high word (bit 31 of the
        // low word for the source when term ids fill 32 bits)
        const int term_sh = 32 - fmt.term_bits;
        const uint32_t src_hi = fmt.term_bits < 32 ? 1u << (31 - fmt.term_bits) : 0u;
        const uint32_t src_lo = fmt.term_bits < 32 ? 0u : 0x80000000u;
        const bool check_tf = fmt.tf_bits < 32;
        const int tf_sh = fmt.tf_bits & 31;
        // Lanes behind the tile's end hold copies of its last row (clamped loads): they pass every
        // check the last row passes and continue its (source, unit) run, so the loop below needs no
        // validity test; their tf is zeroed here and their row replaced after the loop.
        if (n_tile < (uint32_t)SORT_TILE) {
    #pragma unroll
            for (int j = 0; j < SORT_ITEMS; ++j)
                if (l0 + (uint32_t)j * 64u >= n_tile) tfv[j] = 0;
        }
        // status of the row's document (web rows; rows are unit-major or at least clustered, so this gather
        // stays in cache): a doc that is not 'done' gets field code 3, which the posting pass turns into a hole
        uint32_t dfl[SORT_ITEMS];
        if (ctx.doc_flags) {
    #pragma unroll
            for (int j = 0; j < SORT_ITEMS; ++j) dfl[j] = ctx.doc_flags[min(unit[j], ctx.n_docs - 1u)];
        } else {
    #pragma unroll
            for (int j = 0; j < SORT_ITEMS; ++j) dfl[j] = SEBM25_DOC_DONE;
        }
        uint32_t carry_u = lo32(carry), carry_s = hi32(carry);
        auto ballot = [](bool c) { return (uint64_t)__builtin_amdgcn_ballot_w64(c); };
    #pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const uint32_t st = m[j] & 3u, field = m[j] >> 2;
            const bool news = st == SEBM25_SRC_NEWS;
            // bitwise, not short-circuit: four compares and mask logic instead of nested branches
            const bool bad_t = term[j] >= ctx.n_terms, bad_s = st - 1u >= 2u, bad_u = unit[j] >= (news ? ctx.n_articles : ctx.n_docs),
                       bad_f = check_tf & ((tfv[j] >> tf_sh) != 0u);
            const bool bad = bad_t | bad_s | bad_u | bad_f;
            // masks of plain compares, combined in scalar registers
            if ((ballot(bad_t) | ballot(bad_s) | ballot(bad_u) | ballot(bad_f)) != 0ull) {  // one scalar branch per round on the good path
                if (bad) {
                    atomicOr(ctx.err, DEV_ERR_RANGE);
                    // the same key bits as the histogram saw, every later index in range
                    term[j] = min(term[j], ctx.n_terms - 1u);
                    unit[j] &= (uint32_t)((1ull << fmt.unit_bits) - 1ull);
                    tfv[j] = 0;
                }
            }
            // same bits as RowFmt::pack(term, news, unit, field code, tf); field code: 1, 2, else 0
            const uint32_t fc = (!news & ((dfl[j] & SEBM25_DOC_DONE) == 0u)) ? 3u : field < 3u ? field : 0u;
            const uint64_t low = ((((uint64_t)(unit[j] >> 30) << 32) | ((unit[j] << 2) | fc)) << fmt.tf_bits) | tfv[j];
            const uint32_t hi = hi32(low) | (term[j] << term_sh) | (news ? src_hi : 0u);
            const uint32_t lo = lo32(low) | (news ? src_lo : 0u);
            row[j] = ((uint64_t)hi << 32) | lo;
            // unit-major check: (source, unit) must not decrease from one row to the next
            const uint32_t s = news ? 1u : 0u;
            const uint32_t prev_u = lane_up1_edge(unit[j], carry_u), prev_s = lane_up1_edge(s, carry_s);
            carry_u = (uint32_t)__builtin_amdgcn_readlane((int)unit[j], 63);
            carry_s = (uint32_t)__builtin_amdgcn_readlane((int)s, 63);
            const uint64_t eq_s = ballot(s == prev_s), same_m = ballot(unit[j] == prev_u) & eq_s;
            const uint64_t less_m = ballot(s < prev_s) | (eq_s & ballot(unit[j] < prev_u));
            if (ctx.check_order && less_m != 0ull) {
                if (lane == 0) atomicOr(ctx.err, DEV_ERR_ORDER);
            }
            // body length of web docs: sum of tf over field-4 rows.  A round usually lies inside
            // one doc (every lane but the first has its predecessor's key): then one lane adds the
            // wave's sum to the tile's LDS counter of that doc.
            const uint32_t body = (!news & (field == SEBM25_FIELD_BODY) & (unit[j] < ctx.n_docs)) ? tfv[j] : 0u;
            const bool uniform = (~same_m & ~1ull) == 0ull;
            uint32_t add = body;
            bool do_add = body != 0u;
            if (uniform) {
                add = wave_sum_u32(body);
                do_add = lane == 0 && add != 0u;
            }
            if (do_add) {
                const uint32_t off = unit[j] - tile_u_first;  // wraps to a large value for a doc before the tile's first
                if (tile_web && off < (uint32_t)DL_SPAN)
                    atomicAdd(&dl_s[off], add);
                else
                    atomicAdd(&ctx.doc_len[unit[j]], add);
            }
        }
        if (n_tile < (uint32_t)SORT_TILE) {
    #pragma unroll
            for (int j = 0; j < SORT_ITEMS; ++j)
                if (l0 + (uint32_t)j * 64u >= n_tile) row[j] = ~0ull;
        }

    }

    uint32_t rk[SORT_ITEMS];
    {
        uint32_t* tab = mtab + wave * (2u << MATCH_TAB_BITS);
        uint32_t dg[SORT_ITEMS];
        uint64_t cand[SORT_ITEMS], m0[SORT_ITEMS];
        uint32_t d0s[SORT_ITEMS];  // wave-uniform: the digit of the round's first valid lane
        const uint32_t rows_left = n_tile > l0 ? n_tile - l0 : 0u;  // round j's row is valid iff 64 j < rows_left
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const bool valid = l0 + (uint32_t)j * 64u < n_tile;
            const uint32_t d = digit_of(row[j]);
            dg[j] = d;
            const uint64_t vm = __ballot(valid);
            // the group of the first valid lane (the hot digit of skewed or sorted rows) by one compare
            const uint32_t d0 = (uint32_t)__builtin_amdgcn_readlane((int)d, vm ? __ffsll((unsigned long long)vm) - 1 : 0);
            d0s[j] = d0;
            const bool in0 = valid && d == d0;
            m0[j] = __ballot(in0);
            const bool rest = valid && !in0;
            const uint32_t slot = (d & ((1u << MATCH_TAB_BITS) - 1u)) * 2u;
            cand[j] = 0;
            if (rest) atomicOr(&tab[slot + (lane >> 5)], 1u << (lane & 31u));
            __builtin_amdgcn_wave_barrier();
            if (rest) cand[j] = *reinterpret_cast<const uint64_t*>(&tab[slot]);
            __builtin_amdgcn_wave_barrier();
            if (rest) *reinterpret_cast<uint64_t*>(&tab[slot]) = 0ull;
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const bool valid = (uint32_t)j * 64u < rows_left;  // recomputed: cheaper than carrying a bit per round
            uint64_t peers = cand[j];  // lanes that share the low digit bits: settle the high bits
            for (int b = MATCH_TAB_BITS; b < bits; ++b) {
                const bool one = (dg[j] >> b) & 1u;
                const uint64_t bal = __ballot(one);
                peers &= one ? bal : ~bal;
            }
            if (valid && dg[j] == d0s[j]) peers = m0[j];
            // every lane of a digit group reads the group's running count (one address: a broadcast),
            // then its first lane adds the group's size
            uint32_t old = 0;
            if (valid) old = whist[wave][dg[j]];
            __builtin_amdgcn_wave_barrier();
            if (valid && (peers & lanemask_lt()) == 0) whist[wave][dg[j]] = (unsigned short)(old + (uint32_t)__popcll(peers));
            rk[j] = old + (uint32_t)__popcll(peers & lanemask_lt());
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    // ---- per digit pair (2t, 2t+1): exclusive offsets over the waves and the tile counts, two
    //      16-bit counters per LDS word; then a block scan of the counts held in registers
    uint32_t c0 = 0, c1 = 0;
    if (2 * t < nb) {
        uint32_t c[SORT_WAVES];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) c[w] = wh32[w * (SORT_MAX_BINS / 2) + t];
        uint32_t run = 0;  // packed: neither half can carry (a tile has at most 4096 rows)
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const uint32_t x = c[w];
            c[w] = run;
            run += x;
        }
        c0 = run & 0xFFFFu;
        c1 = run >> 16;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wh32[w * (SORT_MAX_BINS / 2) + t] = c[w];
    }
    if (MODE != 2) {
        const uint32_t v = dl_s[t];
        if (v) atomicAdd(&ctx.doc_len[tile_u_first + t], v);
    }
    uint32_t inc = c0 + c1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(inc, off);
        if ((int)lane >= off) inc += x;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    if (2 * t < nb) {
        uint32_t ts0 = inc - (c0 + c1);
        for (uint32_t w = 0; w < wave; ++w) ts0 += wave_tot[w];
        const uint32_t ts1 = ts0 + c0;
        // global slot of local slot p of digit d: base[d] + p (32-bit wrap-around arithmetic)
        base[2 * t] -= ts0;
        base[2 * t + 1] -= ts1;
        const uint32_t add = ts0 | (ts1 << 16);  // per-wave offsets become local slots
        uint32_t c[SORT_WAVES];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) c[w] = wh32[w * (SORT_MAX_BINS / 2) + t];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) wh32[w * (SORT_MAX_BINS / 2) + t] = c[w] + add;
    }
    __syncthreads();

    // ---- reorder the tile by digit inside LDS (stable); the match tables and body-length counters
    //      that shared this memory are dead by now
#pragma unroll
    for (int j = 0; j < SORT_ITEMS; ++j) {
        if (l0 + (uint32_t)j * 64u < n_tile) {
            const uint32_t d = digit_of(row[j]);
            rows_s[whist[wave][d] + rk[j]] = row[j];
        }
    }
    __syncthreads();

    if constexpr (!POST) {
        // ---- write out: consecutive threads hold consecutive slots of a digit's run
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const uint32_t p = t + (uint32_t)j * SORT_THREADS;
            if (p < n_tile) {
                const uint64_t r = rows_s[p];
                const uint32_t d = shr_lo32(r, shift) & mask;
                dst[(uint32_t)(base[d] + p)] = r;
            }
        }
    } else {
        // unit ids reach into the low word of the row (large id spaces: the usual case) or lie in the high word
        if (fmt.key_shift() < 32)
            post_tile<true>(rows_s, base, reinterpret_cast<PostTileShared*>(&whist[0][0]), n_tile, tile0, end, R, chunk, shift, mask,
                            fmt, nbr, edge_row, row[0], ctx.err, po);
        else
            post_tile<false>(rows_s, base, reinterpret_cast<PostTileShared*>(&whist[0][0]), n_tile, tile0, end, R, chunk, shift, mask,
                             fmt, nbr, edge_row, row[0], ctx.err, po);
    }
}

void launch_sort_hist(int mode, RawRows raw, const uint64_t* packed, uint64_t R, uint64_t chunk_rows, uint32_t G,
                      int shift, int bits, const BuildCtx& ctx, unsigned short* prefix, uint32_t* group_tot, hipStream_t st) {
    dim3 grid((G + SORT_GROUP - 1) / SORT_GROUP), block(HIST_THREADS);
    if (mode == 0)
        hipLaunchKernelGGL(k_sort_hist<0>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, ctx.n_terms, prefix, group_tot);
    else if (mode == 1)
        hipLaunchKernelGGL(k_sort_hist<1>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, ctx.n_terms, prefix, group_tot);
    else
        hipLaunchKernelGGL(k_sort_hist<2>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, ctx.n_terms, prefix, group_tot);
}

__global__ void k_scan_single(uint32_t* sums, size_t n);

size_t sort_offsets_tmp_words() { return (size_t)SCAN_SEGS * SORT_MAX_BINS; }

// group_tot -> first slot of every (group, bin), in place; bin_tot [bins] and seg_tmp
// [sort_offsets_tmp_words()] are scratch
void launch_sort_offsets(uint32_t* group_tot, uint32_t n_groups, int bits, uint32_t* bin_tot, uint32_t* seg_tmp,
                         hipStream_t st) {
    const uint32_t nb = 1u << bits;
    const uint32_t seg_len = (n_groups + SCAN_SEGS - 1u) / SCAN_SEGS;
    const uint32_t waves = ((nb + 63u) / 64u) * SCAN_SEGS;
    hipLaunchKernelGGL(k_sort_seg_sums, dim3((waves + 3u) / 4u), dim3(256), 0, st, group_tot, n_groups, nb, seg_len, seg_tmp);
    hipLaunchKernelGGL(k_sort_seg_scan, dim3((nb + 255u) / 256u), dim3(256), 0, st, seg_tmp, nb, bin_tot);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, bin_tot, (size_t)nb);
    hipLaunchKernelGGL(k_sort_seg_apply, dim3((waves + 3u) / 4u), dim3(256), 0, st, group_tot, n_groups, nb, seg_len,
                       seg_tmp, bin_tot);
}

void launch_sort_scatter(int mode, RawRows raw, const uint64_t* packed, uint64_t* dst, uint64_t R, uint32_t G, int shift,
                         int bits, const BuildCtx& ctx, const unsigned short* prefix, const uint32_t* group_off,
                         hipStream_t st) {
    dim3 grid((G + 7u) / 8u * 8u), block(SORT_THREADS);
    const PostOut none{};
    if (mode == 0)
        hipLaunchKernelGGL((k_sort_scatter_tile<0, false>), grid, block, 0, st, raw, packed, dst, R, G, shift, bits, ctx, prefix, group_off, none);
    else if (mode == 1)
        hipLaunchKernelGGL((k_sort_scatter_tile<1, false>), grid, block, 0, st, raw, packed, dst, R, G, shift, bits, ctx, prefix, group_off, none);
    else
        hipLaunchKernelGGL((k_sort_scatter_tile<2, false>), grid, block, 0, st, raw, packed, dst, R, G, shift, bits, ctx, prefix, group_off, none);
}

__global__ __launch_bounds__(1024) void k_postings_totals(const uint4* partials, uint32_t n_chunks, unsigned long long* totals);

void launch_sort_scatter_post(const uint64_t* packed, uint64_t R, uint32_t G, int shift, int bits, const BuildCtx& ctx,
                              const unsigned short* prefix, const uint32_t* group_off, Slot* post,
                              uint32_t* news_ptr, uint32_t* dup, uint32_t* tmp, uint64_t* totals, hipStream_t st) {
    dim3 grid((G + 7u) / 8u * 8u), block(SORT_THREADS);
    const PostOut po{post, news_ptr, dup, reinterpret_cast<uint4*>(tmp)};
    hipLaunchKernelGGL((k_sort_scatter_tile<2, true>), grid, block, 0, st, RawRows{}, packed, (uint64_t*)nullptr, R, G, shift, bits,
                       ctx, prefix, group_off, po);
    hipLaunchKernelGGL(k_postings_totals, dim3(1), dim3(1024), 0, st, po.partials, G, reinterpret_cast<unsigned long long*>(totals));
}

size_t scatter_post_tmp_words(uint32_t G) { return (size_t)G * 4; }

// First slot of every term from the offsets of a two-pass sort on the term bits (no pass over the sorted rows
// is needed for it).  The last pass's input is ordered by the low term bits L, its output by (high bits h, L),
// so term (h, L) starts where the last pass puts the first row of digit h at or behind the first input row of L:
// the write base of that row's chunk for h plus the rows of digit h the chunk holds before it.  One workgroup
// per L counts those rows (fewer than one tile).  Terms without rows get the first slot of the next term.
__global__ __launch_bounds__(256) void k_first_slots(const uint64_t* rows, uint32_t R, uint32_t G, int shift, int bits1, int bits2,
                                                     const uint32_t* bin_start1, const unsigned short* prefix,
                                                     const uint32_t* group_off, const uint32_t* bin_start2, uint32_t n_terms,
                                                     uint32_t* row_ptr, Slot* pad) {
    __shared__ uint32_t hist[SORT_MAX_BINS];
    const uint32_t L = blockIdx.x, nb2 = 1u << bits2;
    for (uint32_t x = threadIdx.x; x < nb2; x += 256u) hist[x] = 0;
    __syncthreads();
    const uint32_t b = bin_start1[L];  // first input row whose low term bits are L or more (R: none)
    const uint32_t c = b / (uint32_t)SORT_TILE;
    for (uint32_t i = c * (uint32_t)SORT_TILE + threadIdx.x; i < b; i += 256u) atomicAdd(&hist[shr_lo32(rows[i], shift) & (nb2 - 1u)], 1u);
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < nb2; h += 256u) {
        const uint64_t t = ((uint64_t)h << bits1) | L;
        if (t >= n_terms) continue;
        uint32_t at;
        if (c < G)
            at = group_off[(size_t)(c / SORT_GROUP) * nb2 + h] + prefix[(size_t)c * nb2 + h] + hist[h];
        else  // behind the last row: the end of digit h's rows
            at = h + 1u < nb2 ? bin_start2[h + 1u] : R;
        row_ptr[t] = at;
    }
    if (L == 0) {
        if (threadIdx.x == 0) row_ptr[n_terms] = R;
        // what lies behind the last slot reads as holes
        for (uint32_t i = threadIdx.x; i < (uint32_t)POST_PAD; i += 256u) pad[i] = make_uint2(0xFFFFFFFFu, POST_HOLE);
    }
}

void launch_first_slots(const uint64_t* rows, uint64_t R, uint32_t G, int shift, int bits1, int bits2, const uint32_t* bin_start1,
                        const unsigned short* prefix, const uint32_t* group_off, const uint32_t* bin_start2, uint32_t n_terms,
                        uint32_t* row_ptr, Slot* pad, hipStream_t st) {
    hipLaunchKernelGGL(k_first_slots, dim3(1u << bits1), dim3(256), 0, st, rows, (uint32_t)R, G, shift, bits1, bits2, bin_start1, prefix,
                       group_off, bin_start2, n_terms, row_ptr, pad);
}

// ------------------------------------------------------------------ exclusive scan --------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint32_t block_reduce_u32(uint32_t v, uint32_t* sh) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (lane_id() == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    uint32_t tot = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) tot += sh[w];
    __syncthreads();
    return tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(const uint32_t* data, size_t n, uint32_t* block_sums) {
    __shared__ uint32_t sh[SCAN_THREADS / 64];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE;
    uint32_t s = 0;
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        const size_t i = base + (size_t)j * SCAN_THREADS + threadIdx.x;
        if (i < n) s += data[i];
    }
    const uint32_t tot = block_reduce_u32(s, sh);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of `n` words in place (any count)
__global__ __launch_bounds__(1024) void k_scan_single(uint32_t* sums, size_t n) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < n; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < n ? sums[i] : 0u;
        uint32_t inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t x = __shfl_up(inc, off);
            if ((int)lane_id() >= off) inc += x;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t wbase = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) wbase += wtot[w];
        const uint32_t carry = carry_s;
        if (i < n) sums[i] = carry + wbase + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wbase + inc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(uint32_t* data, size_t n, const uint32_t* block_off) {
    __shared__ uint32_t wtot[SCAN_THREADS / 64];
    // blocked arrangement: thread t owns SCAN_ITEMS consecutive words
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        v[j] = (base + j < n) ? data[base + j] : 0u;
        s += v[j];
    }
    uint32_t inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(inc, off);
        if ((int)lane_id() >= off) inc += x;
    }
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t run = block_off[blockIdx.x] + inc - s;
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) run += wtot[w];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) {
        if (base + j < n) data[base + j] = run;
        run += v[j];
    }
}

size_t scan_tmp_words(size_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

void launch_exclusive_scan_u32(uint32_t* data, size_t n, uint32_t* tmp, hipStream_t st) {
    if (n == 0) return;
    if (n <= 8192) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, data, n);
        return;
    }
    const size_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, data, n, tmp);
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, st, tmp, nblocks);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, data, n, tmp);
}

// ------------------------------------------------------------------ postings ----------------
// Sorted rows -> postings, one slot per sorted row and in the same place: slot i holds the unit
// of row i and, when row i is the first row of a kept (term, source, unit) run, the run's
// weight; every other slot is a hole (POST_HOLE) that scoring skips.  Nothing has to be counted
// or prefix-summed first, so the rows are read once.  Each wavefront streams its own contiguous
// range of rows (no workgroup barriers): 64 consecutive rows per load, neighbours compared
// through lane shifts, coalesced slot stores.

// row_ptr[t] = first slot of term t (left at 0xFFFFFFFF for terms without rows), news_ptr[t] =
// first news slot of term t (left at 0xFFFFFFFF for terms without news rows), df[t] += web heads of term t (zeroed by
// the caller), partials[range] = (kept postings, web heads, kept web postings) of the range, summed into
// totals[0], [1], [5] by k_postings_totals.
// Row flags:
//   head    : first row of its (term, source, unit) run
//   kept    : head of a news unit, or of a web doc whose status is 'done' (search_service.py:32)
//   webhead : head of a web doc of any status -> counts toward df (bm25_stats.py:20)
//
// Vector memory operations retire in issue order, so waiting for any load also waits for every
// store issued before it.  The loop is therefore a three-stage pipeline in which loads are only
// ever issued ahead of a batch's stores: while batch b is processed, the rows of batch b + 1 are
// already in registers and the rows of batch b + 2 are loaded.  Loads and slot stores are unconditional and fixed in number (clamped load addresses;
// the slot arrays are padded by POST_PAD entries, which take the stores of lanes behind the last
// row), which lets the wait at the end of a batch leave the batch's own stores in flight.
//
// Per row the kernel computes one 32-bit word v = tenths(row) | differs-from-predecessor << 31;
// what a run head needs to know about the rows behind it (do they continue the run, what do they
// weigh) then comes from three one-lane shifts of v instead of three more rows to decode.
__global__ __launch_bounds__(RED_THREADS) void k_postings(const uint64_t* rows, uint32_t R, uint32_t chunk_rows,
                                                          uint32_t n_chunks, RowFmt fmt, uint32_t n_terms, Slot* post,
                                                          uint32_t* row_ptr, uint32_t* news_ptr,
                                                          uint32_t* df, uint4* partials, uint32_t* err) {
    // the range index is the same for the whole wave: tell the compiler, so that everything derived
    // from it (bounds, base addresses, loop control) lives in scalar registers
    const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * (RED_THREADS / 64) + (threadIdx.x >> 6)));
    if (c >= n_chunks) return;
    const RowBits rb(fmt);
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t begin = c * chunk_rows;  // < R; the host keeps R + POST_PAD below 2^32
    const uint32_t end = R - begin < chunk_rows ? R : begin + chunk_rows;
    const uint64_t key_lsb = 1ull << fmt.key_shift();
    // rows behind the last one read as a row of another key; the row before row 0 as one of another term
    const uint64_t behind = rows[R - 1] ^ key_lsb;
    uint64_t carry = begin ? rows[begin - 1] : rows[0] ^ (1ull << 63);
    uint32_t n_kept = 0, n_wh = 0, n_kept_web = 0;
    // web heads of the term the last rounds consisted of, not yet added to df
    constexpr uint32_t NO_TERM = 0xFFFFFFFFu;
    uint32_t run_term = NO_TERM, run_wh = 0;
    // (term, source) bits of a row
    const uint32_t ts_hi = rb.term_hi | rb.src_hi, ts_lo = rb.src_lo;

    uint64_t ra[RED_ROUNDS], rb1[RED_ROUNDS], rc[RED_ROUNDS];  // rows of batch b, b + 1, b + 2
    // the row array is padded like the slot arrays: loads run up to 3 batches past the last row
    // without a clamp (what they bring is replaced below before it is looked at)
    auto load_rows = [&](uint32_t b0, uint64_t (&dst)[RED_ROUNDS]) {
        const uint64_t* p = rows + b0;
#pragma unroll
        for (int j = 0; j < RED_ROUNDS; ++j) dst[j] = p[(uint32_t)j * 64u + lane];
    };
    load_rows(begin, ra);
    load_rows(begin + RED_BATCH, rb1);
    // as many stores (into the padding) behind these loads as a batch issues behind its loads;
    // the compiler must not move them ahead of the loads
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < RED_ROUNDS; ++j) {
        post[R + (uint32_t)j * 64u + lane] = make_uint2(0xFFFFFFFFu, POST_HOLE);
    }
    asm volatile("" ::: "memory");

    for (uint32_t b0 = begin; b0 < end; b0 += RED_BATCH) {
        load_rows(b0 + 2 * RED_BATCH, rc);

        // ---- phase 1: per row the word v, whether its term / its (term, source) differ from the
        //      row before; v also of the first round behind the batch
        uint32_t v[RED_ROUNDS + 1];
        bool news[RED_ROUNDS], term_new[RED_ROUNDS], ts_new[RED_ROUNDS];
        uint32_t seen_hi = 0, seen_lo = 0;  // per lane: OR of the batch's rows (for the tf bits from 2^24 up)
        if (b0 + 2u * RED_BATCH > R) {      // only the last batches reach behind the last row
#pragma unroll
            for (int j = 0; j < RED_ROUNDS; ++j)
                if (b0 + (uint32_t)j * 64u + lane >= R) ra[j] = behind;
            if (b0 + RED_BATCH + lane >= R) rb1[0] = behind;
        }
        uint64_t before = carry;
#pragma unroll
        for (int j = 0; j <= RED_ROUNDS; ++j) {
            const uint64_t x = j < RED_ROUNDS ? ra[j] : rb1[0];
            const uint32_t dh = hi32(x) ^ lane_up1_edge(hi32(x), hi32(before));
            const uint32_t dl = lo32(x) ^ lane_up1_edge(lo32(x), lo32(before));
            const bool nw = is_news(rb, x);
            const bool differs = ((dh & rb.key_hi) | (dl & rb.key_lo)) != 0u;
            v[j] = tenths24(rb, x, nw) | (differs ? 0x80000000u : 0u);
            if (j < RED_ROUNDS) {
                news[j] = nw;
                term_new[j] = (dh & rb.term_hi) != 0u;
                ts_new[j] = ((dh & ts_hi) | (dl & ts_lo)) != 0u;
                seen_hi |= hi32(x);
                seen_lo |= lo32(x);
                before = lane_bcast<63>(x);
            }
        }
        carry = before;
        // a tf of 2^24 or more anywhere in the batch (rows of a neighbouring range included): exact weights later
        if (__ballot(((seen_hi & rb.big_hi) | (seen_lo & rb.big_lo)) != 0u) && lane == 0) atomicOr(err, DEV_NEED_FIXUP);

        // ---- phase 2: slots, term boundaries, df
        Slot* ps = post + b0;
        const uint32_t left = end - b0;  // rows of the range from this batch on
#pragma unroll
        for (int j = 0; j < RED_ROUNDS; ++j) {
            const uint32_t li = (uint32_t)j * 64u + lane;
            const bool valid = li < left;
            const uint64_t vm = __ballot(valid);
            const uint64_t x0 = ra[j];
            // words of rows i+1 .. i+3: lane shifts; at the wave edge the first lanes of the next round
            const uint32_t n1 = lane_dn1_edge(v[j], (uint32_t)__builtin_amdgcn_readlane((int)v[j + 1], 0));
            const uint32_t n2 = lane_dn1_edge(n1, (uint32_t)__builtin_amdgcn_readlane((int)v[j + 1], 1));
            const uint32_t n3 = lane_dn1_edge(n2, (uint32_t)__builtin_amdgcn_readlane((int)v[j + 1], 2));

            const bool head = valid & ((v[j] >> 31) != 0u);
            const uint32_t term = hi32(x0) >> rb.term_sh;
            const uint32_t unit = shr_lo32(x0, rb.unit_sh) & rb.unit_mask;
            const bool webhead = head & !news[j];
            // field code 3 marks the rows of a web doc whose status is not 'done' (set by the first sort pass)
            const bool kept = head & (news[j] | ((shr_lo32(x0, rb.fc_sh) & 3u) != 3u));
            const uint64_t km = __ballot(kept), wm = __ballot(webhead);
            // a row whose term differs from the row before starts the term's slots (terms without rows
            // are filled in by k_row_ptr_fill); its first news row starts the term's news slots
            if (valid & term_new[j]) row_ptr[term] = b0 + li;
            if (valid & news[j] & ts_new[j]) news_ptr[term] = b0 + li;
            // a run of up to three rows (title, description, body) with tf < 2^24 each cannot overflow;
            // longer runs (not produced by the reference's writers) are left to k_postings_fixup
            const bool same1 = !(n1 >> 31), same2 = same1 & !(n2 >> 31), same3 = same2 & !(n3 >> 31);
            if (__ballot(kept & same3) && lane == 0) atomicOr(err, DEV_NEED_FIXUP);
            const uint32_t w = (v[j] & 0x7FFFFFFFu) + (same1 ? n1 & 0x7FFFFFFFu : 0u) + (same2 ? n2 & 0x7FFFFFFFu : 0u);
            ps[li] = make_uint2(unit, kept ? w : POST_HOLE);
            // weights of 0xFFFF tenths and more do not fit the 16-bit accumulator fields of the stepwise scoring kernels
            if (__ballot(kept & (w >= 0xFFFFu)) && lane == 0) atomicOr(err, DEV_BIG_WEIGHT);
            n_kept += (uint32_t)__popcll(km);
            n_wh += (uint32_t)__popcll(wm);
            n_kept_web += (uint32_t)__popcll(km & wm);

            // df: web heads per term.  Rounds of a single term (the long lists) only extend a running
            // count; a round with several terms adds one count per term segment.
            const uint64_t starts = (__ballot(term_new[j]) | 1ull) & vm;  // first lanes of the round's term segments
            if (vm == 0) {
                // nothing of this round lies inside the range
            } else if (starts == 1ull) {
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)term);
                if (t0 != run_term) {
                    if (lane == 0 && run_wh) atomicAdd(&df[run_term], run_wh);
                    run_term = t0;
                    run_wh = 0;
                }
                run_wh += (uint32_t)__popcll(wm);
            } else {
                if (lane == 0 && run_wh) atomicAdd(&df[run_term], run_wh);
                run_term = NO_TERM;
                run_wh = 0;
                // the last lane of every segment adds the segment's web heads
                const uint64_t ends = vm & ((starts >> 1) | ~(vm >> 1));
                if ((ends >> lane) & 1ull) {
                    const uint64_t upto = lt | (1ull << lane);
                    const int s0 = 63 - __clzll((long long)(starts & upto));  // first lane of this lane's term
                    const uint32_t cnt = (uint32_t)__popcll(wm & upto & ~((1ull << s0) - 1ull));
                    if (cnt) atomicAdd(&df[term], cnt);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RED_ROUNDS; ++j) {
            ra[j] = rb1[j];
            rb1[j] = rc[j];
        }
    }
    if (lane == 0) {
        if (run_wh) atomicAdd(&df[run_term], run_wh);
        // the range's counts go to a slot of their own: thousands of atomics on one address would
        // queue up at one memory channel (34 ns each, measured) and hold back the traffic through it
        partials[c] = make_uint4(n_kept, n_wh, n_kept_web, 0u);
    }
}

// totals[0] += kept postings, totals[1] += web heads, totals[5] += kept web postings over all ranges
__global__ __launch_bounds__(1024) void k_postings_totals(const uint4* partials, uint32_t n_chunks, unsigned long long* totals) {
    __shared__ unsigned long long sm[3][16];
    unsigned long long a = 0, b = 0, c = 0;
    for (uint32_t i = threadIdx.x; i < n_chunks; i += 1024u) {
        const uint4 p = partials[i];
        a += p.x;
        b += p.y;
        c += p.z;
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
        c += __shfl_down(c, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        sm[0][threadIdx.x >> 6] = a;
        sm[1][threadIdx.x >> 6] = b;
        sm[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (int w = 0; w < 16; ++w) t += sm[threadIdx.x][w];
        if (t) atomicAdd(&totals[threadIdx.x == 2 ? 5 : threadIdx.x], t);
    }
}

// Exact 64-bit weights for the runs k_postings does not handle in its fast path: more than three
// rows of one (term, source, unit), or a tf of 2^24 or more.  One thread per row; runs only when
// k_postings raised DEV_NEED_FIXUP.
__global__ void k_postings_fixup(const uint64_t* rows, uint64_t R, RowFmt fmt, Slot* post, uint32_t* err) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R || post[i].y == POST_HOLE) return;  // only kept heads carry a weight
    const uint64_t x0 = rows[i];
    const bool news = fmt.src(x0) != 0u;
    uint64_t w = 0, n = 0;
    bool big = false;
    for (uint64_t ii = i; ii < R; ++ii, ++n) {
        const uint64_t x = rows[ii];
        if (fmt.key(x) != fmt.key(x0)) break;
        const uint32_t fc = fmt.fc(x);
        const uint64_t tf = fmt.tf(x);
        big |= tf >= (1ull << 24);
        const uint64_t add = tf * (uint64_t)(news ? 10u : (fc == 1u ? 32u : fc == 2u ? 17u : 10u));
        w = (w + add < w || w + add >= (uint64_t)POST_HOLE) ? (uint64_t)POST_HOLE : w + add;
    }
    if (n <= 3 && !big) return;
    if (w >= (uint64_t)POST_HOLE) {
        atomicOr(err, DEV_ERR_RANGE);
        w = POST_HOLE - 1u;
    }
    post[i].y = (uint32_t)w;
    if (w >= 0xFFFFull) atomicOr(err, DEV_BIG_WEIGHT);
}

void launch_postings_fixup(const uint64_t* rows, uint64_t R, RowFmt fmt, Slot* post, uint32_t* err, hipStream_t st) {
    if (R == 0) return;
    hipLaunchKernelGGL(k_postings_fixup, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, rows, R, fmt, post, err);
}

void launch_postings(const uint64_t* rows, uint64_t R, uint64_t chunk_rows, uint32_t n_chunks, RowFmt fmt, uint32_t n_terms,
                     Slot* post, uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* df,
                     uint64_t* totals, uint32_t* tmp, uint32_t* err, hipStream_t st) {
    if (R == 0) return;
    uint4* partials = reinterpret_cast<uint4*>(tmp);  // postings_tmp_words(n_chunks) words
    const uint32_t wpb = RED_THREADS / 64;
    const dim3 grid((n_chunks + wpb - 1) / wpb), block(RED_THREADS);
    unsigned long long* tot = reinterpret_cast<unsigned long long*>(totals);
    hipLaunchKernelGGL(k_postings, grid, block, 0, st, rows, (uint32_t)R, (uint32_t)chunk_rows, n_chunks, fmt, n_terms,
                       post, row_ptr, news_ptr, df, partials, err);
    hipLaunchKernelGGL(k_postings_totals, dim3(1), dim3(1024), 0, st, partials, n_chunks, tot);
}

size_t postings_tmp_words(uint32_t n_chunks) { return (size_t)n_chunks * 4; }

// Terms without rows: row_ptr[t] = row_ptr of the next term that has rows, row_ptr[n_terms] = R.
// First a backward minimum scan inside blocks of 1024 entries, then entries whose block holds
// nothing behind them take the first value of the next non-empty block.
constexpr int FILL_THREADS = 1024;
__global__ __launch_bounds__(FILL_THREADS) void k_row_ptr_fill_blocks(uint32_t* row_ptr, uint32_t n_terms, uint32_t R,
                                                                      uint32_t* block_first) {
    __shared__ uint32_t sm[2][FILL_THREADS];
    const uint32_t t = blockIdx.x * FILL_THREADS + threadIdx.x;
    uint32_t v = t < n_terms ? row_ptr[t] : (t == n_terms ? R : 0xFFFFFFFFu);
    int cur = 0;
    sm[0][threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < FILL_THREADS; off <<= 1) {
        if (threadIdx.x + off < FILL_THREADS) v = min(v, sm[cur][threadIdx.x + off]);
        cur ^= 1;
        sm[cur][threadIdx.x] = v;
        __syncthreads();
    }
    if (t <= n_terms) row_ptr[t] = v;
    if (threadIdx.x == 0) block_first[blockIdx.x] = v;
}
__global__ __launch_bounds__(256) void k_row_ptr_fill_rest(uint32_t* row_ptr, uint32_t n_terms, const uint32_t* block_first) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t > n_terms || row_ptr[t] != 0xFFFFFFFFu) return;
    uint32_t b = t / FILL_THREADS + 1u;  // the last block holds row_ptr[n_terms] = R, so this ends
    while (block_first[b] == 0xFFFFFFFFu) ++b;
    row_ptr[t] = block_first[b];
}

size_t row_ptr_fill_tmp_words(uint32_t n_terms) { return (size_t)n_terms / FILL_THREADS + 2; }

void launch_row_ptr_fill(uint32_t* row_ptr, uint32_t n_terms, uint64_t R, uint32_t* tmp, hipStream_t st) {
    const uint32_t n = n_terms + 1u;
    hipLaunchKernelGGL(k_row_ptr_fill_blocks, dim3((n + FILL_THREADS - 1) / FILL_THREADS), dim3(FILL_THREADS), 0, st,
                       row_ptr, n_terms, (uint32_t)R, tmp);
    hipLaunchKernelGGL(k_row_ptr_fill_rest, dim3((n + 255u) / 256u), dim3(256), 0, st, row_ptr, n_terms, tmp);
}

// ------------------------------------------------------------------ statistics ------------
// term_rows[t] = input rows of term t; terms without news rows get news_ptr[t] = row_ptr[t + 1]
// (their web segment is the whole list).
__global__ void k_term_stats(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    term_rows[t] = row_ptr[t + 1] - row_ptr[t];
    if (news_ptr[t] == 0xFFFFFFFFu) news_ptr[t] = row_ptr[t + 1];
}

// the same when the last sort pass wrote the posting slots: df_dup holds, per term, the web rows that are not the
// head of their (term, unit) run and becomes df = web rows - those (bm25_stats.py:20: COUNT(DISTINCT doc_id))
__global__ void k_term_stats_dup(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows,
                                 uint32_t* df_dup) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const uint32_t b = row_ptr[t], e = row_ptr[t + 1];
    term_rows[t] = e - b;
    uint32_t np = news_ptr[t];
    if (np == 0xFFFFFFFFu) news_ptr[t] = np = e;
    df_dup[t] = np - b - df_dup[t];
}

void launch_term_stats_dup(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows, uint32_t* df_dup,
                           hipStream_t st) {
    hipLaunchKernelGGL(k_term_stats_dup, dim3((n_terms + 255) / 256), dim3(256), 0, st, n_terms, row_ptr, news_ptr, term_rows, df_dup);
}

void launch_term_stats(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* term_rows,
                       hipStream_t st) {
    hipLaunchKernelGGL(k_term_stats, dim3((n_terms + 255) / 256), dim3(256), 0, st, n_terms, row_ptr, news_ptr, term_rows);
}

// N, and the numerator / denominator of AVG(word_count), over documents with status 'done'
// (bm25_stats.py:9-13).  totals3 = {n_done, n_wc, sum_wc}, zeroed by the caller.
// Few, large workgroups: every workgroup ends with three atomics on the same three words, and
// same-address atomics are served one after the other.
__global__ __launch_bounds__(1024) void k_doc_stats(const uint32_t* word_count, const uint32_t* doc_len,
                                                   const uint8_t* doc_flags, uint32_t n_docs,
                                                   unsigned long long* totals3) {
    __shared__ unsigned long long sh[3][16];
    unsigned long long nd = 0, nw = 0, sw = 0;
    for (uint32_t d = blockIdx.x * blockDim.x + threadIdx.x; d < n_docs; d += gridDim.x * blockDim.x) {
        const uint32_t fl = doc_flags ? doc_flags[d] : SEBM25_DOC_DONE;
        if (!(fl & SEBM25_DOC_DONE)) continue;
        nd += 1;
        if (fl & SEBM25_DOC_WC_NULL) continue;
        nw += 1;
        sw += word_count ? word_count[d] : doc_len[d];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        nd += __shfl_down(nd, off);
        nw += __shfl_down(nw, off);
        sw += __shfl_down(sw, off);
    }
    if (lane_id() == 0) {
        sh[0][threadIdx.x >> 6] = nd;
        sh[1][threadIdx.x >> 6] = nw;
        sh[2][threadIdx.x >> 6] = sw;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long s = 0;
        for (uint32_t w = 0; w < blockDim.x / 64u; ++w) s += sh[threadIdx.x][w];
        if (s) atomicAdd(&totals3[threadIdx.x], s);
    }
}

void launch_doc_stats(const uint32_t* word_count, const uint32_t* doc_len, const uint8_t* doc_flags, uint32_t n_docs,
                      uint64_t* totals3, hipStream_t st) {
    if (n_docs == 0) return;
    const uint32_t blocks = min((n_docs + 1023u) / 1024u, 256u);
    hipLaunchKernelGGL(k_doc_stats, dim3(blocks), dim3(1024), 0, st, word_count, doc_len, doc_flags, n_docs,
                       reinterpret_cast<unsigned long long*>(totals3));
}

// idf = LN((N - df + 0.5) / (df + 0.5) + 1) (bm25_stats.py:21) as log1p of the quotient, which
// stays accurate to an ulp when df is close to N (PostgreSQL computes it in NUMERIC).
// n_done_dev: when given, N is read from device memory (the count of 'done' docs a kernel queued before this one
// left there; 0 reads as 1, bm25_stats.py:13), so that the statistics need no trip to the host in between
__global__ void k_idf(const uint32_t* df, uint32_t n_terms, double N, const unsigned long long* n_done_dev, double* idf,
                      double* idf_used) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    if (n_done_dev) {
        const unsigned long long n = *n_done_dev;
        N = (double)(n ? n : 1ull);
    }
    const double d = (double)df[t];
    const double q = __ddiv_rn(__dadd_rn(__dadd_rn(N, -d), 0.5), __dadd_rn(d, 0.5));
    const double v = log1p(q);
    idf[t] = v;
    idf_used[t] = v;
}

void launch_idf(const uint32_t* df, uint32_t n_terms, uint64_t N, const uint64_t* n_done_dev, double* idf, double* idf_used,
                hipStream_t st) {
    hipLaunchKernelGGL(k_idf, dim3((n_terms + 255) / 256), dim3(256), 0, st, df, n_terms, (double)N,
                       reinterpret_cast<const unsigned long long*>(n_done_dev), idf, idf_used);
}

// idf_used[t] = COALESCE(ts.idf, 1.0) (search_service.py:21,47,86)
__global__ void k_idf_override(const double* idf, const uint8_t* present, uint32_t n_terms, double* idf_used) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    idf_used[t] = (!present || present[t]) ? idf[t] : 1.0;
}

void launch_idf_override(const double* idf, const uint8_t* present, uint32_t n_terms, double* idf_used,
                         hipStream_t st) {
    hipLaunchKernelGGL(k_idf_override, dim3((n_terms + 255) / 256), dim3(256), 0, st, idf, present, n_terms, idf_used);
}

}  // namespace sebm25
