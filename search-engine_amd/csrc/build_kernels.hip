// Index-build kernels for gfx950 (wave64): stable LSD radix sort of token rows by term, the
// segmented reduction of sorted rows into postings, and the per-term / per-document
// statistics.  Replaces the GROUP BY t.term / COUNT(DISTINCT t.doc_id) / LN(...) statement of
// the reference (app/batch/bm25_stats.py:16-27) and the document aggregates at :9-13.
//
// Every kernel here is bandwidth-bound integer work: rows stream through in coalesced
// 64-lane accesses, digit ranking uses wave ballots, and each workgroup owns a contiguous
// chunk of rows so that the runs it appends to a digit bin are adjacent in memory and merge
// in its XCD's L2 before they reach HBM.
#include "common.hpp"

namespace sebm25 {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// Lanes of the wave that hold the same `bits`-bit digit as this lane (only among `valid`).
__device__ __forceinline__ uint64_t match_digit(uint32_t d, int bits, bool valid) {
    uint64_t peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
        const bool one = (d >> b) & 1u;
        const uint64_t bal = __ballot(one);
        peers &= one ? bal : ~bal;
    }
    return peers;
}

__device__ __forceinline__ uint32_t field_code(uint32_t field) { return field == 1u ? 1u : field == 2u ? 2u : 0u; }

// MODE 0: raw rows, only term_id is read (digit inside the term bits)
// MODE 1: raw rows, key packed from term/unit/meta
// MODE 2: packed rows
template <int MODE>
__device__ __forceinline__ uint64_t load_key(const RawRows& raw, const uint64_t* packed, uint64_t i, const RowFmt& f) {
    if (MODE == 0) return (uint64_t)raw.term[i] << f.term_shift();
    if (MODE == 1) {
        const uint32_t m = raw.meta[i];
        return f.pack(raw.term[i], ((m & 3u) == SEBM25_SRC_NEWS) ? 1u : 0u, raw.unit[i], 0u, 0ull);
    }
    return packed[i];
}

// ------------------------------------------------------------------ histogram (upsweep) ---
// Workgroup g counts the digits of its chunk of rows; counts are stored bin-major
// (chunk_hist[bin * G + g]) so that one flat exclusive scan yields every chunk's write base.
template <int MODE>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_hist(RawRows raw, const uint64_t* packed, uint64_t R,
                                                            uint64_t chunk_rows, uint32_t G, int shift, int bits,
                                                            RowFmt fmt, uint32_t* chunk_hist) {
    __shared__ uint32_t hist[SORT_MAX_BINS];
    const uint32_t nb = 1u << bits, mask = nb - 1u;
    for (uint32_t d = threadIdx.x; d < nb; d += SORT_THREADS) hist[d] = 0;
    __syncthreads();
    const uint64_t begin = (uint64_t)blockIdx.x * chunk_rows;
    const uint64_t end = min(R, begin + chunk_rows);
    for (uint64_t i0 = begin; i0 < end; i0 += SORT_THREADS) {
        const uint64_t i = i0 + threadIdx.x;
        const bool valid = i < end;
        uint32_t d = 0;
        if (valid) d = (uint32_t)(load_key<MODE>(raw, packed, i, fmt) >> shift) & mask;
        const uint64_t peers = match_digit(d, bits, valid);
        if (valid && (peers & lanemask_lt()) == 0) atomicAdd(&hist[d], (uint32_t)__popcll(peers));
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < nb; d += SORT_THREADS) chunk_hist[(size_t)d * G + blockIdx.x] = hist[d];
}

// ------------------------------------------------------------------ scatter (downsweep) ---
// Exclusive scan over `n` (<= 2 * SORT_THREADS) LDS words, two consecutive words per thread.
__device__ __forceinline__ void block_exclusive_scan_1024(uint32_t* v, uint32_t n, uint32_t* wave_tot) {
    const uint32_t t = threadIdx.x;
    const uint32_t a = (2 * t < n) ? v[2 * t] : 0u;
    const uint32_t b = (2 * t + 1 < n) ? v[2 * t + 1] : 0u;
    uint32_t inc = a + b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t x = __shfl_up(inc, off);
        if ((int)lane_id() >= off) inc += x;
    }
    if (lane_id() == 63) wave_tot[t >> 6] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (t >> 6); ++w) base += wave_tot[w];
    const uint32_t excl = base + inc - (a + b);
    if (2 * t < n) v[2 * t] = excl;
    if (2 * t + 1 < n) v[2 * t + 1] = excl + a;
    __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter(RawRows raw, const uint64_t* packed, uint64_t* dst,
                                                               uint64_t R, uint64_t chunk_rows, uint32_t G, int shift,
                                                               int bits, BuildCtx ctx, const uint32_t* chunk_off) {
    __shared__ uint32_t whist[SORT_WAVES][SORT_MAX_BINS];  // per-wave digit counts, then per-wave offsets
    __shared__ uint64_t rows_s[SORT_TILE];                 // the tile, locally ordered by digit
    __shared__ uint32_t tile_start[SORT_MAX_BINS];         // first local slot of each digit
    __shared__ uint32_t bin_base[SORT_MAX_BINS];           // next global slot of each digit for this chunk
    __shared__ uint32_t wave_tot[SORT_WAVES];

    const RowFmt fmt = ctx.fmt;
    const uint32_t nb = 1u << bits, mask = nb - 1u;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint64_t begin = (uint64_t)blockIdx.x * chunk_rows;
    const uint64_t end = min(R, begin + chunk_rows);
    for (uint32_t d = threadIdx.x; d < nb; d += SORT_THREADS) bin_base[d] = chunk_off[(size_t)d * G + blockIdx.x];

    for (uint64_t tile0 = begin; tile0 < end; tile0 += SORT_TILE) {
        for (uint32_t x = threadIdx.x; x < SORT_WAVES * nb; x += SORT_THREADS) whist[x / nb][x % nb] = 0;
        __syncthreads();

        // ---- load + rank: wave w owns rows [tile0 + w*64*ITEMS, +64*ITEMS), 64 consecutive rows per round
        uint64_t row[SORT_ITEMS];
        uint32_t rk[SORT_ITEMS];
        const uint64_t wave0 = tile0 + (uint64_t)wave * 64u * SORT_ITEMS;
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const uint64_t i = wave0 + (uint64_t)j * 64u + lane;
            const bool valid = i < end;
            uint64_t r = ~0ull;
            if (MODE == 2) {
                if (valid) r = packed[i];
            } else {
                uint32_t term = 0, unit = 0, tfv = 0, m = SEBM25_SRC_WEB;
                if (valid) {
                    term = raw.term[i];
                    unit = raw.unit[i];
                    tfv = raw.tf[i];
                    m = raw.meta[i];
                }
                const uint32_t st = m & 3u, field = m >> 2;
                const uint32_t src = (st == SEBM25_SRC_NEWS) ? 1u : 0u;
                if (valid) {
                    const bool bad = term >= ctx.n_terms || (st != SEBM25_SRC_WEB && st != SEBM25_SRC_NEWS) ||
                                     unit >= (src ? ctx.n_articles : ctx.n_docs) ||
                                     (fmt.tf_bits < 32 && (tfv >> fmt.tf_bits) != 0u);
                    if (bad) {
                        atomicOr(ctx.err, DEV_ERR_RANGE);
                        term = min(term, ctx.n_terms - 1u);  // keep every later index in range
                        unit = 0;
                        tfv = 0;
                    }
                    r = fmt.pack(term, src, unit, field_code(field), tfv);
                }
                // unit-major check and per-document body length (both on runs of equal (src, unit))
                const uint64_t ukey = valid ? (((uint64_t)src << 32) | unit) : ~0ull;
                uint64_t prev = __shfl_up(ukey, 1);
                if (lane == 0) {
                    prev = 0;
                    if (valid && i > 0) {
                        const uint32_t pm = raw.meta[i - 1];
                        prev = ((uint64_t)(((pm & 3u) == SEBM25_SRC_NEWS) ? 1u : 0u) << 32) | raw.unit[i - 1];
                    }
                }
                if (valid && ctx.check_order && ukey < prev) atomicOr(ctx.err, DEV_ERR_ORDER);
                const bool head = (lane == 0) || (ukey != prev);
                const uint64_t heads = __ballot(head);
                uint32_t inc = (valid && src == 0u && field == SEBM25_FIELD_BODY && unit < ctx.n_docs) ? tfv : 0u;
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t x = __shfl_up(inc, off);
                    if ((int)lane >= off) inc += x;
                }
                const uint32_t head_lane = 63u - (uint32_t)__clzll(heads & (~0ull >> (63u - lane)));
                const uint32_t before = __shfl(inc, (int)((head_lane + 63u) & 63u));
                const bool tail = (lane == 63u) || ((heads >> (lane + 1u)) & 1ull);
                if (tail) {
                    const uint32_t sum = inc - (head_lane ? before : 0u);
                    if (sum) atomicAdd(&ctx.doc_len[unit], sum);
                }
            }
            row[j] = r;
            const uint32_t d = (uint32_t)(r >> shift) & mask;
            const uint64_t peers = match_digit(d, bits, valid);
            const int leader = __ffsll((unsigned long long)peers) - 1;
            uint32_t old = 0;
            if (valid && (int)lane == leader) {
                old = whist[wave][d];
                whist[wave][d] = old + (uint32_t)__popcll(peers);
            }
            old = __shfl(old, leader < 0 ? 0 : leader);
            rk[j] = old + (uint32_t)__popcll(peers & lanemask_lt());
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        // ---- per digit: exclusive offsets over the waves, tile count
        for (uint32_t d = threadIdx.x; d < nb; d += SORT_THREADS) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < SORT_WAVES; ++w) {
                const uint32_t c = whist[w][d];
                whist[w][d] = run;
                run += c;
            }
            tile_start[d] = run;
        }
        __syncthreads();
        block_exclusive_scan_1024(tile_start, nb, wave_tot);

        // ---- reorder the tile by digit inside LDS (stable)
        const uint32_t n_tile = (uint32_t)min((uint64_t)SORT_TILE, end - tile0);
#pragma unroll
        for (int j = 0; j < SORT_ITEMS; ++j) {
            const uint64_t i = wave0 + (uint64_t)j * 64u + lane;
            if (i < end) {
                const uint32_t d = (uint32_t)(row[j] >> shift) & mask;
                rows_s[tile_start[d] + whist[wave][d] + rk[j]] = row[j];
            }
        }
        __syncthreads();

        // ---- write out: consecutive threads hold consecutive slots of a digit's run
        for (uint32_t p = threadIdx.x; p < n_tile; p += SORT_THREADS) {
            const uint64_t r = rows_s[p];
            const uint32_t d = (uint32_t)(r >> shift) & mask;
            dst[(uint64_t)bin_base[d] + (p - tile_start[d])] = r;
        }
        __syncthreads();
        for (uint32_t d = threadIdx.x; d < nb; d += SORT_THREADS) {
            const uint32_t nxt = (d + 1 < nb) ? tile_start[d + 1] : n_tile;
            bin_base[d] += nxt - tile_start[d];
        }
        // the barrier after zeroing whist at the top of the next tile orders these updates
    }
}

void launch_sort_hist(int mode, RawRows raw, const uint64_t* packed, uint64_t R, uint64_t chunk_rows, uint32_t G,
                      int shift, int bits, const BuildCtx& ctx, uint32_t* chunk_hist, hipStream_t st) {
    dim3 grid(G), block(SORT_THREADS);
    if (mode == 0)
        hipLaunchKernelGGL(k_sort_hist<0>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, chunk_hist);
    else if (mode == 1)
        hipLaunchKernelGGL(k_sort_hist<1>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, chunk_hist);
    else
        hipLaunchKernelGGL(k_sort_hist<2>, grid, block, 0, st, raw, packed, R, chunk_rows, G, shift, bits, ctx.fmt, chunk_hist);
}

void launch_sort_scatter(int mode, RawRows raw, const uint64_t* packed, uint64_t* dst, uint64_t R,
                         uint64_t chunk_rows, uint32_t G, int shift, int bits, const BuildCtx& ctx,
                         const uint32_t* chunk_off, hipStream_t st) {
    dim3 grid(G), block(SORT_THREADS);
    if (mode == 2)
        hipLaunchKernelGGL(k_sort_scatter<2>, grid, block, 0, st, raw, packed, dst, R, chunk_rows, G, shift, bits, ctx, chunk_off);
    else  // raw rows are always fully packed by the scatter, whatever the digit source
        hipLaunchKernelGGL(k_sort_scatter<1>, grid, block, 0, st, raw, packed, dst, R, chunk_rows, G, shift, bits, ctx, chunk_off);
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

// one workgroup: exclusive scan of the block sums in place (any count)
__global__ __launch_bounds__(1024) void k_scan_block_sums(uint32_t* sums, size_t nblocks) {
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < nblocks; base += 1024) {
        const size_t i = base + threadIdx.x;
        const uint32_t v = i < nblocks ? sums[i] : 0u;
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
        if (i < nblocks) sums[i] = carry + wbase + inc - v;
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
    const size_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, data, n, tmp);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(1), dim3(1024), 0, st, tmp, nblocks);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, st, data, n, tmp);
}

// ------------------------------------------------------------------ segmented reduction ---
// Sorted rows -> one posting per (term, source, unit).  flags per row:
//   head    : first row of its (term, source, unit) run
//   kept    : head of a news unit, or of a web doc whose status is 'done' (search_service.py:32)
//   webhead : head of a web doc of any status -> counts toward df (bm25_stats.py:20)
struct RowFlags {
    bool head, kept, webhead;
};
__device__ __forceinline__ RowFlags row_flags(uint64_t r, uint64_t prev, bool first, const RowFmt& f,
                                              const uint8_t* doc_flags) {
    RowFlags x;
    x.head = first || f.key(r) != f.key(prev);
    const bool news = f.src(r) != 0u;
    x.webhead = x.head && !news;
    x.kept = x.head && (news || !doc_flags || (doc_flags[f.unit(r)] & SEBM25_DOC_DONE));
    return x;
}

__global__ __launch_bounds__(RED_THREADS) void k_reduce_count(const uint64_t* rows, uint64_t R, uint64_t chunk_rows,
                                                              RowFmt fmt, const uint8_t* doc_flags,
                                                              uint64_t* chunk_counts) {
    __shared__ uint32_t sh[RED_THREADS / 64];
    const uint64_t begin = (uint64_t)blockIdx.x * chunk_rows;
    const uint64_t end = min(R, begin + chunk_rows);
    uint32_t kept = 0, wh = 0;
    for (uint64_t i = begin + threadIdx.x; i < end; i += RED_THREADS) {
        const uint64_t r = rows[i];
        const uint64_t prev = i ? rows[i - 1] : 0ull;
        const RowFlags fl = row_flags(r, prev, i == 0, fmt, doc_flags);
        kept += fl.kept;
        wh += fl.webhead;
    }
    const uint32_t tk = block_reduce_u32(kept, sh);
    const uint32_t tw = block_reduce_u32(wh, sh);
    if (threadIdx.x == 0) chunk_counts[blockIdx.x] = (uint64_t)tk | ((uint64_t)tw << 32);
}

__global__ __launch_bounds__(RED_THREADS) void k_reduce_emit(const uint64_t* rows, uint64_t R, uint64_t chunk_rows,
                                                             uint32_t G, RowFmt fmt, const uint8_t* doc_flags,
                                                             const uint64_t* chunk_counts, uint32_t n_terms,
                                                             uint32_t* post_unit, float* post_w, uint32_t* row_ptr,
                                                             uint32_t* news_ptr, uint32_t* wh_ptr, uint32_t* rs_ptr,
                                                             uint64_t* totals_out) {
    __shared__ uint64_t sh64[RED_THREADS / 64];
    __shared__ uint64_t run_s;
    const uint32_t t = threadIdx.x;
    // exclusive prefix of the chunk counts: low word kept postings, high word web heads
    uint64_t part = 0;
    for (uint32_t g = t; g < blockIdx.x; g += RED_THREADS) part += chunk_counts[g];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
    if (lane_id() == 0) sh64[t >> 6] = part;
    __syncthreads();
    if (t == 0) {
        uint64_t s = 0;
        for (int w = 0; w < RED_THREADS / 64; ++w) s += sh64[w];
        run_s = s;
    }
    __syncthreads();

    const uint64_t begin = (uint64_t)blockIdx.x * chunk_rows;
    const uint64_t end = min(R, begin + chunk_rows);
    for (uint64_t tile0 = begin; tile0 < end; tile0 += RED_TILE) {
        const uint64_t i0 = tile0 + (uint64_t)t * RED_ITEMS;
        uint64_t r[RED_ITEMS];
        uint64_t prev = (i0 > 0 && i0 < end) ? rows[i0 - 1] : 0ull;
        const uint64_t prev0 = prev;
        uint64_t cnt = 0;
        uint32_t headbits = 0, keptbits = 0, whbits = 0;
#pragma unroll
        for (int j = 0; j < RED_ITEMS; ++j) {
            const uint64_t i = i0 + j;
            r[j] = 0;
            if (i < end) {
                r[j] = rows[i];
                const RowFlags fl = row_flags(r[j], prev, i == 0, fmt, doc_flags);
                headbits |= (uint32_t)fl.head << j;
                keptbits |= (uint32_t)fl.kept << j;
                whbits |= (uint32_t)fl.webhead << j;
                cnt += (uint64_t)fl.kept | ((uint64_t)fl.webhead << 32);
                prev = r[j];
            }
        }
        // block exclusive scan of the packed counts
        uint64_t inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t x = __shfl_up(inc, off);
            if ((int)lane_id() >= off) inc += x;
        }
        if (lane_id() == 63) sh64[t >> 6] = inc;
        __syncthreads();
        uint64_t run = run_s + inc - cnt;
        for (uint32_t w = 0; w < (t >> 6); ++w) run += sh64[w];
        uint64_t tile_total = 0;
        for (int w = 0; w < RED_THREADS / 64; ++w) tile_total += sh64[w];

        prev = prev0;
#pragma unroll
        for (int j = 0; j < RED_ITEMS; ++j) {
            const uint64_t i = i0 + j;
            if (i < end) {
                const uint32_t p = (uint32_t)run, wh = (uint32_t)(run >> 32);
                const uint32_t term = fmt.term(r[j]);
                if ((headbits >> j) & 1u) {
                    const bool first = (i == 0);
                    const uint32_t pterm = first ? 0u : fmt.term(prev);
                    if (first || term != pterm) {
                        // this row opens term `term`; terms without rows before it are empty
                        for (uint32_t x = first ? 0u : pterm + 1u; x <= term; ++x) {
                            row_ptr[x] = p;
                            wh_ptr[x] = wh;
                            rs_ptr[x] = (uint32_t)i;
                        }
                    }
                    if (fmt.src(r[j]) && (first || term != pterm || fmt.src(prev) == 0u)) news_ptr[term] = p;
                    if ((keptbits >> j) & 1u) {
                        // weight: web sum_f tf_f * W[f] (search_service.py:19-27), news sum tf (:84-87);
                        // double, mul then add in row order, rounded once to float
                        const uint64_t key = fmt.key(r[j]);
                        const bool news = fmt.src(r[j]) != 0u;
                        double w = 0.0;
                        uint64_t ii = i;
                        uint64_t rr = r[j];
                        while (true) {
                            const double x = (double)fmt.tf(rr);
                            const uint32_t fc = fmt.fc(rr);
                            const double wf = news ? 1.0 : (fc == 1u ? 3.2 : fc == 2u ? 1.7 : 1.0);
                            w = __dadd_rn(w, news ? x : __dmul_rn(x, wf));
                            if (++ii >= R) break;
                            rr = rows[ii];
                            if (fmt.key(rr) != key) break;
                        }
                        post_unit[p] = fmt.unit(r[j]);
                        post_w[p] = (float)w;
                    }
                }
                run += (uint64_t)((keptbits >> j) & 1u) | ((uint64_t)((whbits >> j) & 1u) << 32);
                if (i == R - 1) {
                    for (uint32_t x = term + 1u; x <= n_terms; ++x) {
                        row_ptr[x] = (uint32_t)run;
                        wh_ptr[x] = (uint32_t)(run >> 32);
                        rs_ptr[x] = (uint32_t)R;
                    }
                    totals_out[0] = run & 0xFFFFFFFFull;  // kept postings
                    totals_out[1] = run >> 32;            // web heads (sum of df)
                }
                prev = r[j];
            }
        }
        __syncthreads();
        if (t == 0) run_s += tile_total;
        __syncthreads();
    }
}

void launch_reduce_count(const uint64_t* rows, uint64_t R, uint64_t chunk_rows, uint32_t G, RowFmt fmt,
                         const uint8_t* doc_flags, uint64_t* chunk_counts, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_count, dim3(G), dim3(RED_THREADS), 0, st, rows, R, chunk_rows, fmt, doc_flags, chunk_counts);
}

void launch_reduce_emit(const uint64_t* rows, uint64_t R, uint64_t chunk_rows, uint32_t G, RowFmt fmt,
                        const uint8_t* doc_flags, const uint64_t* chunk_counts, uint32_t n_terms, uint32_t* post_unit,
                        float* post_w, uint32_t* row_ptr, uint32_t* news_ptr, uint32_t* wh_ptr, uint32_t* rs_ptr,
                        uint64_t* totals_out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_emit, dim3(G), dim3(RED_THREADS), 0, st, rows, R, chunk_rows, G, fmt, doc_flags,
                       chunk_counts, n_terms, post_unit, post_w, row_ptr, news_ptr, wh_ptr, rs_ptr, totals_out);
}

// ------------------------------------------------------------------ statistics ------------
// df[t] = web heads of term t; term_rows[t] = input rows of term t; terms without news rows get
// news_ptr[t] = row_ptr[t + 1] (their web segment is the whole list).
__global__ void k_term_stats(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, const uint32_t* wh_ptr,
                             const uint32_t* rs_ptr, uint32_t* df, uint32_t* term_rows) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    df[t] = wh_ptr[t + 1] - wh_ptr[t];
    term_rows[t] = rs_ptr[t + 1] - rs_ptr[t];
    if (news_ptr[t] == 0xFFFFFFFFu) news_ptr[t] = row_ptr[t + 1];
}

void launch_term_stats(uint32_t n_terms, const uint32_t* row_ptr, uint32_t* news_ptr, const uint32_t* wh_ptr,
                       const uint32_t* rs_ptr, uint32_t* df, uint32_t* term_rows, hipStream_t st) {
    hipLaunchKernelGGL(k_term_stats, dim3((n_terms + 255) / 256), dim3(256), 0, st, n_terms, row_ptr, news_ptr, wh_ptr,
                       rs_ptr, df, term_rows);
}

// N, and the numerator / denominator of AVG(word_count), over documents with status 'done'
// (bm25_stats.py:9-13).  totals3 = {n_done, n_wc, sum_wc}, zeroed by the caller.
__global__ __launch_bounds__(256) void k_doc_stats(const uint32_t* word_count, const uint32_t* doc_len,
                                                   const uint8_t* doc_flags, uint32_t n_docs,
                                                   unsigned long long* totals3) {
    __shared__ unsigned long long sh[3][4];
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
        unsigned long long s = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
        if (s) atomicAdd(&totals3[threadIdx.x], s);
    }
}

void launch_doc_stats(const uint32_t* word_count, const uint32_t* doc_len, const uint8_t* doc_flags, uint32_t n_docs,
                      uint64_t* totals3, hipStream_t st) {
    if (n_docs == 0) return;
    const uint32_t blocks = min((n_docs + 255u) / 256u, 2048u);
    hipLaunchKernelGGL(k_doc_stats, dim3(blocks), dim3(256), 0, st, word_count, doc_len, doc_flags, n_docs,
                       reinterpret_cast<unsigned long long*>(totals3));
}

// idf = LN((N - df + 0.5) / (df + 0.5) + 1) (bm25_stats.py:21) as log1p of the quotient, which
// stays accurate to an ulp when df is close to N (PostgreSQL computes it in NUMERIC).
__global__ void k_idf(const uint32_t* df, uint32_t n_terms, double N, double* idf, double* idf_used) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_terms) return;
    const double d = (double)df[t];
    const double q = __ddiv_rn(__dadd_rn(__dadd_rn(N, -d), 0.5), __dadd_rn(d, 0.5));
    const double v = log1p(q);
    idf[t] = v;
    idf_used[t] = v;
}

void launch_idf(const uint32_t* df, uint32_t n_terms, uint64_t N, double* idf, double* idf_used, hipStream_t st) {
    hipLaunchKernelGGL(k_idf, dim3((n_terms + 255) / 256), dim3(256), 0, st, df, n_terms, (double)N, idf, idf_used);
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
