// Kernels of the incremental update (sebm25_update / sebm25_compact).
//
// The reference keeps `tokens` current row by row: the crawler deletes a document's rows and inserts the new
// ones (app/crawler/worker.py:229-239, app/batch/news_fetcher.py:303-311) and the statistics job recomputes
// everything every BATCH_INTERVAL_S seconds (.env:11, app/batch/bm25_stats.py:15).  Here an update leaves the
// index segments that exist alone except for marking, in each of them, the rows of the units that changed as
// dead (their posting slots become holes, their terms' df / row counts go down), and builds the new rows of
// those units into a segment of their own.  A unit's live rows are therefore always in exactly one segment.
#include "common.hpp"

namespace sebm25 {

namespace {

__device__ __forceinline__ bool bit_of(const uint32_t* bm, uint32_t i) { return (bm[i >> 5] >> (i & 31u)) & 1u; }

// bits of the listed ids (ids >= n are ignored: a removed unit beyond the id space has no rows anywhere)
// (with shift > 0: one bit per group of 2^shift ids, set when any id of the group is listed)
__global__ void k_bitmap_set(const uint32_t* ids, uint32_t n_ids, uint32_t n, uint32_t shift, uint32_t* bm) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ids && ids[i] < n) atomicOr(&bm[(ids[i] >> shift) >> 5], 1u << ((ids[i] >> shift) & 31u));
}

__global__ void k_bitmap_or(uint32_t* dst, const uint32_t* src, uint32_t n_words) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) dst[i] |= src[i];
}

// One thread per sorted row of a segment, TOMB_BATCH rows per thread and round with every load of a round issued
// before the first is used.  The test of a row is a dependent pair of reads (the row, then its unit's bit), and
// the second is a gather that the texture path serves far below the rate rows stream in at; so each block keeps
// coarse bitmaps in LDS (one bit per 2^shift units: "some unit of this group changed"; cw words for web docs,
// then the words for articles) and goes to the exact bitmap in memory only for rows of a flagged group.  A row
// whose unit changed in this update (and was alive here until now) dies: its term loses a row; if it is the first
// row of its (term, source, unit) run and a web row, the term loses a document (df counts web docs of any
// status); if its slot carried a posting, the slot becomes a hole.
// counts: [0] postings removed, [1] web postings removed, [2] web (term, doc) pairs removed, [3] rows.
constexpr int TOMB_BATCH = 8;
__global__ __launch_bounds__(256) void k_tombstone(const uint64_t* rows, uint32_t R, RowFmt fmt, const uint32_t* chg_web,
                                                   const uint32_t* chg_news, const uint32_t* coarse_g, uint32_t cw,
                                                   uint32_t c_words, uint32_t shift, const uint32_t* dead_web,
                                                   const uint32_t* dead_news, Slot* post, uint32_t* df,
                                                   uint32_t* term_rows, unsigned long long* counts) {
    extern __shared__ uint32_t coarse[];
    for (uint32_t i = threadIdx.x; i < c_words; i += blockDim.x) coarse[i] = coarse_g[i];
    __syncthreads();
    uint32_t c_post = 0, c_web = 0, c_head = 0, c_rows = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x * TOMB_BATCH;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x * TOMB_BATCH + threadIdx.x; i0 < R; i0 += stride) {
        uint64_t r[TOMB_BATCH];
        uint32_t word[TOMB_BATCH];
#pragma unroll
        for (int j = 0; j < TOMB_BATCH; ++j) {
            const uint64_t i = i0 + (uint64_t)j * blockDim.x;
            r[j] = i < R ? rows[i] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < TOMB_BATCH; ++j) {
            const uint32_t unit = fmt.unit(r[j]), src = fmt.src(r[j]), g = unit >> shift;
            word[j] = (coarse[(src ? cw : 0u) + (g >> 5)] >> (g & 31u)) & 1u;
        }
#pragma unroll
        for (int j = 0; j < TOMB_BATCH; ++j) {
            const uint32_t unit = fmt.unit(r[j]);
            if (word[j]) word[j] = (fmt.src(r[j]) ? chg_news : chg_web)[unit >> 5] >> (unit & 31u);
        }
#pragma unroll
        for (int j = 0; j < TOMB_BATCH; ++j) {
            const uint64_t i = i0 + (uint64_t)j * blockDim.x;
            const uint32_t src = fmt.src(r[j]), unit = fmt.unit(r[j]), term = fmt.term(r[j]);
            bool dies = (word[j] & 1u) && i < R;
            if (!__ballot(dies)) continue;
            bool head = false;
            if (dies) dies = !bit_of(src ? dead_news : dead_web, unit);
            if (dies) {
                head = !src && (i == 0 || fmt.key(rows[i - 1]) != fmt.key(r[j]));
                if (post[i].y != POST_HOLE) {
                    post[i].y = POST_HOLE;
                    ++c_post;
                    c_web += src ? 0u : 1u;
                }
                ++c_rows;
                c_head += head ? 1u : 0u;
            }
            // Units that change together are neighbours in every long list, so the rows that die in one wavefront
            // mostly share a term: one atomic per (wavefront, term) instead of one per row (atomics on one address,
            // and on the addresses of one cache line, are served one at a time)
            for (uint64_t m = __ballot(dies); m;) {
                const int l = __ffsll((unsigned long long)m) - 1;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)term, l);
                const uint64_t same = __ballot(dies && term == t0);
                const uint64_t heads = __ballot(dies && term == t0 && head);
                if ((int)(threadIdx.x & 63u) == l) {
                    atomicSub(&term_rows[t0], (uint32_t)__popcll(same));
                    if (heads) atomicSub(&df[t0], (uint32_t)__popcll(heads));
                }
                m &= ~same;
            }
        }
    }
    // one set of atomics per block: every thread that saw a dying row adding to the same four words would queue
    // hundreds of thousands of atomics on one cache line
    __shared__ uint32_t block_sum[4];
    if (threadIdx.x < 4) block_sum[threadIdx.x] = 0;
    __syncthreads();
    uint32_t c[4] = {c_post, c_web, c_head, c_rows};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c[q] += (uint32_t)__shfl_xor((int)c[q], o);
        if ((threadIdx.x & 63u) == 0 && c[q]) atomicAdd(&block_sum[q], c[q]);
    }
    __syncthreads();
    if (threadIdx.x < 4 && block_sum[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)block_sum[threadIdx.x]);
}

// every row of the new segment must belong to a unit that the update lists as changed
__global__ __launch_bounds__(256) void k_check_listed(const uint64_t* rows, uint32_t R, RowFmt fmt, const uint32_t* chg_web,
                                                      const uint32_t* chg_news, uint32_t* err) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = rows[i];
        const uint32_t unit = fmt.unit(r);
        if (!bit_of(fmt.src(r) ? chg_news : chg_web, unit)) atomicOr(err, DEV_ERR_RANGE);
    }
}

// body length per doc after an update: the new segment's value for changed docs, the old value otherwise
__global__ void k_doc_len_merge(uint32_t* doc_len, const uint32_t* new_len, const uint32_t* chg_web, uint32_t n_docs) {
    const uint32_t d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n_docs && bit_of(chg_web, d)) doc_len[d] = new_len[d];
}

__global__ void k_add_u32(uint32_t* dst, const uint32_t* src, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// compaction: the live rows of one segment, repacked to the common row format, appended in any order.  One
// wavefront takes spans of GATHER_SPAN consecutive rows: it counts the live ones, reserves their places with one
// atomic (appends to one counter are served one at a time, so they must be few), then reads the span again and
// writes.
constexpr uint32_t GATHER_SPAN = 64 * 32;
__global__ __launch_bounds__(256) void k_gather_live(const uint64_t* rows, uint32_t R, RowFmt from, RowFmt to,
                                                     const uint32_t* dead_web, const uint32_t* dead_news, uint64_t* out,
                                                     unsigned long long* n_out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t s0 = wave * GATHER_SPAN; s0 < R; s0 += n_waves * GATHER_SPAN) {
        uint32_t n_live = 0;
        for (uint32_t j = 0; j < GATHER_SPAN; j += 64) {
            const uint64_t i = s0 + j + lane;
            bool live = false;
            if (i < R) {
                const uint64_t r = rows[i];
                live = !bit_of(from.src(r) ? dead_news : dead_web, from.unit(r));
            }
            n_live += (uint32_t)__popcll(__ballot(live));
        }
        if (n_live == 0) continue;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(n_out, (unsigned long long)n_live);
        base = ((unsigned long long)(uint32_t)__shfl((int)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0);
        for (uint32_t j = 0; j < GATHER_SPAN; j += 64) {
            const uint64_t i = s0 + j + lane;
            uint64_t r = 0;
            bool live = false;
            if (i < R) {
                r = rows[i];
                live = !bit_of(from.src(r) ? dead_news : dead_web, from.unit(r));
            }
            const uint64_t m = __ballot(live);
            if (live) out[base + __popcll(m & ((1ull << lane) - 1ull))] = to.pack(from.term(r), from.src(r), from.unit(r), from.fc(r), from.tf(r));
            base += (unsigned long long)__popcll(m);
        }
    }
}

// ---- compaction by merging --------------------------------------------------------------------------------
// The base segment's rows are sorted and so are (after a sort of their own) the few live rows of the update
// segments; a unit's live rows sit in one segment only, so no live row of one side compares equal to a live row
// of the other.  The merged position of a base row is therefore
//   (live base rows before it) + (update rows whose key is smaller),
// and of update row j:  j + (live base rows before its insertion point).  Rows compare on (term, source, unit),
// the sort key of every segment, decoded from whichever row format the segment has.
constexpr uint32_t MERGE_TILE = 4096, MERGE_THREADS = 1024, MERGE_ITEMS = MERGE_TILE / MERGE_THREADS;

__device__ __forceinline__ uint64_t merge_key(RowFmt f, uint64_t r) {
    return ((uint64_t)((f.term(r) << 1) | f.src(r)) << 32) | f.unit(r);
}

// live rows per tile of MERGE_TILE base rows (tiles run to index R inclusive: a row that does not exist is dead)
__global__ __launch_bounds__(MERGE_THREADS) void k_live_tile_counts(const uint64_t* rows, uint32_t R, RowFmt fmt, const uint32_t* dead_web,
                                                                    const uint32_t* dead_news, uint32_t* counts) {
    __shared__ uint32_t wsum[MERGE_THREADS / 64];
    uint32_t n = 0;
#pragma unroll
    for (uint32_t k = 0; k < MERGE_ITEMS; ++k) {  // any order will do for a count: consecutive lanes, consecutive rows
        const uint64_t i = (uint64_t)blockIdx.x * MERGE_TILE + k * MERGE_THREADS + threadIdx.x;
        if (i < R) {
            const uint64_t r = rows[i];
            n += bit_of(fmt.src(r) ? dead_news : dead_web, fmt.unit(r)) ? 0u : 1u;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (uint32_t w = 0; w < MERGE_THREADS / 64; ++w) t += wsum[w];
        counts[blockIdx.x] = t;
    }
}

// insertion point of every update row: the first base index whose key is not smaller (dead base rows keep their
// keys and their places, so the search runs over all R of them)
__global__ __launch_bounds__(256) void k_insert_points(const uint64_t* upd, uint32_t n_upd, RowFmt ufmt, const uint64_t* base, uint32_t R,
                                                       RowFmt bfmt, uint32_t* ip) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_upd) return;
    const uint64_t key = merge_key(ufmt, upd[j]);
    uint32_t lo = 0, hi = R;
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (merge_key(bfmt, base[mid]) < key) lo = mid + 1; else hi = mid;
    }
    ip[j] = lo;
}

// one workgroup per tile of base rows: writes the tile's live rows (repacked) and the update rows that are
// inserted inside the tile to their merged places.  tile_off: live base rows before the tile (exclusive scan of
// k_live_tile_counts).
__global__ __launch_bounds__(MERGE_THREADS) void k_merge_tiles(const uint64_t* base, uint32_t R, RowFmt bfmt, const uint32_t* dead_web,
                                                               const uint32_t* dead_news, const uint32_t* tile_off, const uint64_t* upd,
                                                               uint32_t n_upd, const uint32_t* ip, RowFmt to, uint64_t* out) {
    __shared__ uint32_t mark[MERGE_TILE];          // update rows inserted at each base index of the tile
    __shared__ unsigned short live_before[MERGE_TILE];
    __shared__ uint32_t wl[MERGE_THREADS / 64], wd[MERGE_THREADS / 64];
    __shared__ uint32_t range[2];
    const uint64_t a = (uint64_t)blockIdx.x * MERGE_TILE;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t < 2) {  // update rows with an insertion point in [a, a + MERGE_TILE): rows range[0] .. range[1] - 1
        const uint64_t x = a + (t ? MERGE_TILE : 0u);
        uint32_t lo = 0, hi = n_upd;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            if ((uint64_t)ip[mid] < x) lo = mid + 1; else hi = mid;
        }
        range[t] = lo;
    }
    for (uint32_t x = t; x < MERGE_TILE; x += MERGE_THREADS) mark[x] = 0;
    __syncthreads();
    const uint32_t c0 = range[0], c1 = range[1];
    for (uint32_t j = c0 + t; j < c1; j += MERGE_THREADS) atomicAdd(&mark[ip[j] - (uint32_t)a], 1u);
    __syncthreads();
    // thread t owns base rows a + MERGE_ITEMS t ... (consecutive, so prefix sums run in row order)
    uint64_t r[MERGE_ITEMS];
    bool live[MERGE_ITEMS];
    uint32_t m[MERGE_ITEMS], sl = 0, sd = 0;
    static_assert(MERGE_ITEMS == 4, "two 16-byte loads per thread");
    const uint64_t i_first = a + (uint64_t)t * MERGE_ITEMS;
    if (i_first + MERGE_ITEMS <= R) {  // (the rows start at a multiple of 32 bytes: the buffers are 256-byte aligned)
        const ulonglong2 lo = *reinterpret_cast<const ulonglong2*>(base + i_first), hi = *reinterpret_cast<const ulonglong2*>(base + i_first + 2);
        r[0] = lo.x; r[1] = lo.y; r[2] = hi.x; r[3] = hi.y;
    } else {
#pragma unroll
        for (uint32_t k = 0; k < MERGE_ITEMS; ++k) r[k] = i_first + k < R ? base[i_first + k] : 0ull;
    }
#pragma unroll
    for (uint32_t k = 0; k < MERGE_ITEMS; ++k) {
        live[k] = i_first + k < R && !bit_of(bfmt.src(r[k]) ? dead_news : dead_web, bfmt.unit(r[k]));
        m[k] = mark[t * MERGE_ITEMS + k];
        sl += live[k] ? 1u : 0u;
        sd += m[k];
    }
    uint32_t il = sl, id = sd;  // inclusive scans over the wave, then over the waves
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t xl = (uint32_t)__shfl_up((int)il, o), xd = (uint32_t)__shfl_up((int)id, o);
        if ((int)lane >= o) {
            il += xl;
            id += xd;
        }
    }
    if (lane == 63) {
        wl[wave] = il;
        wd[wave] = id;
    }
    __syncthreads();
    uint32_t el = il - sl, ed = id - sd;  // exclusive, within the tile
    for (uint32_t w = 0; w < wave; ++w) {
        el += wl[w];
        ed += wd[w];
    }
    const uint32_t off = tile_off[blockIdx.x];
#pragma unroll
    for (uint32_t k = 0; k < MERGE_ITEMS; ++k) {
        live_before[t * MERGE_ITEMS + k] = (unsigned short)el;
        ed += m[k];  // update rows inserted at or before this row come first
        if (live[k]) {
            out[(uint64_t)off + el + c0 + ed] = to.pack(bfmt.term(r[k]), bfmt.src(r[k]), bfmt.unit(r[k]), bfmt.fc(r[k]), bfmt.tf(r[k]));
            ++el;
        }
    }
    __syncthreads();
    for (uint32_t j = c0 + t; j < c1; j += MERGE_THREADS) out[(uint64_t)j + off + live_before[ip[j] - (uint32_t)a]] = upd[j];
}

}  // namespace

static inline dim3 grid_for(uint64_t n, int threads, int cap = 8192) {
    uint64_t g = (n + threads - 1) / threads;
    return dim3((unsigned)(g < 1 ? 1 : (g > (uint64_t)cap ? (uint64_t)cap : g)));
}

void launch_bitmap_set(const uint32_t* ids, uint32_t n_ids, uint32_t n, uint32_t shift, uint32_t* bm, hipStream_t st) {
    if (n_ids == 0) return;
    hipLaunchKernelGGL(k_bitmap_set, dim3((n_ids + 255) / 256), dim3(256), 0, st, ids, n_ids, n, shift, bm);
}
void launch_bitmap_or(uint32_t* dst, const uint32_t* src, uint32_t n_words, hipStream_t st) {
    if (n_words == 0) return;
    hipLaunchKernelGGL(k_bitmap_or, dim3((n_words + 255) / 256), dim3(256), 0, st, dst, src, n_words);
}
void launch_tombstone(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* chg_web, const uint32_t* chg_news,
                      const uint32_t* coarse, uint32_t cw, uint32_t c_words, uint32_t shift, const uint32_t* dead_web, const uint32_t* dead_news, Slot* post, uint32_t* df,
                      uint32_t* term_rows, unsigned long long* counts, hipStream_t st) {
    if (R == 0) return;
    hipLaunchKernelGGL(k_tombstone, grid_for((R + TOMB_BATCH - 1) / TOMB_BATCH, 256, 4096), dim3(256), (size_t)c_words * 4, st, rows, (uint32_t)R, fmt, chg_web, chg_news, coarse, cw, c_words, shift,
                       dead_web, dead_news, post, df, term_rows, counts);
}
void launch_check_listed(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* chg_web, const uint32_t* chg_news,
                         uint32_t* err, hipStream_t st) {
    if (R == 0) return;
    hipLaunchKernelGGL(k_check_listed, grid_for(R, 256), dim3(256), 0, st, rows, (uint32_t)R, fmt, chg_web, chg_news, err);
}
void launch_doc_len_merge(uint32_t* doc_len, const uint32_t* new_len, const uint32_t* chg_web, uint32_t n_docs, hipStream_t st) {
    if (n_docs == 0) return;
    hipLaunchKernelGGL(k_doc_len_merge, dim3((n_docs + 255) / 256), dim3(256), 0, st, doc_len, new_len, chg_web, n_docs);
}
void launch_add_u32(uint32_t* dst, const uint32_t* src, uint32_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_add_u32, dim3((n + 255) / 256), dim3(256), 0, st, dst, src, n);
}
void launch_gather_live(const uint64_t* rows, uint64_t R, RowFmt from, RowFmt to, const uint32_t* dead_web,
                        const uint32_t* dead_news, uint64_t* out, unsigned long long* n_out, hipStream_t st) {
    if (R == 0) return;
    hipLaunchKernelGGL(k_gather_live, grid_for(R, 256), dim3(256), 0, st, rows, (uint32_t)R, from, to, dead_web, dead_news, out,
                       n_out);
}

uint32_t merge_tiles_for(uint64_t R) { return (uint32_t)((R + MERGE_TILE) / MERGE_TILE); }  // index R included
void launch_live_tile_counts(const uint64_t* rows, uint64_t R, RowFmt fmt, const uint32_t* dead_web, const uint32_t* dead_news,
                             uint32_t* counts, hipStream_t st) {
    hipLaunchKernelGGL(k_live_tile_counts, dim3(merge_tiles_for(R)), dim3(MERGE_THREADS), 0, st, rows, (uint32_t)R, fmt, dead_web, dead_news, counts);
}
void launch_insert_points(const uint64_t* upd, uint64_t n_upd, RowFmt ufmt, const uint64_t* base, uint64_t R, RowFmt bfmt, uint32_t* ip,
                          hipStream_t st) {
    if (n_upd == 0) return;
    hipLaunchKernelGGL(k_insert_points, dim3((unsigned)((n_upd + 255) / 256)), dim3(256), 0, st, upd, (uint32_t)n_upd, ufmt, base, (uint32_t)R, bfmt, ip);
}
void launch_merge_tiles(const uint64_t* base, uint64_t R, RowFmt bfmt, const uint32_t* dead_web, const uint32_t* dead_news,
                        const uint32_t* tile_off, const uint64_t* upd, uint64_t n_upd, const uint32_t* ip, RowFmt to, uint64_t* out,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_merge_tiles, dim3(merge_tiles_for(R)), dim3(MERGE_THREADS), 0, st, base, (uint32_t)R, bfmt, dead_web, dead_news, tile_off,
                       upd, (uint32_t)n_upd, ip, to, out);
}

}  // namespace sebm25
