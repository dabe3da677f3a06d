// scratch: why is a one-bit-per-row test over 8-byte rows slow?  Variants of the dead-row scan.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
struct Fmt { int tb, ub, fb; };
// A: as in the library (64-bit field extraction, pointer select, one row per thread and round)
__global__ __launch_bounds__(256) void kA(const uint64_t* rows, uint32_t R, Fmt f, const uint32_t* bw, const uint32_t* bn, uint32_t* out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = rows[i];
        const uint32_t src = (uint32_t)(r >> (63 - f.tb)) & 1u;
        const uint32_t unit = (uint32_t)(r >> (f.fb + 2)) & (uint32_t)((1ull << f.ub) - 1ull);
        if (((src ? bn : bw)[unit >> 5] >> (unit & 31u)) & 1u) ++c;
    }
    if (c) atomicAdd(out, c);
}
// B: no bitmap at all (pure stream)
__global__ __launch_bounds__(256) void kB(const uint64_t* rows, uint32_t R, Fmt f, const uint32_t* bw, const uint32_t* bn, uint32_t* out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t r = rows[i];
        c += (uint32_t)(r >> 13) & 1u;
    }
    if (c == 0xFFFFFFFFu) atomicAdd(out, c);
}
// C: 16-byte loads (two rows per lane), 4 loads in flight, bitmap from global
__global__ __launch_bounds__(256) void kC(const uint64_t* rows, uint32_t R, Fmt f, const uint32_t* bw, const uint32_t* bn, uint32_t* out) {
    uint32_t c = 0;
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(rows);
    const uint32_t n2 = R / 2;
    const uint32_t ushift = f.fb + 2, umask = (uint32_t)((1ull << f.ub) - 1ull), sshift = 63 - f.tb;
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x * 4) {
        ulonglong2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j * 256 < n2) ? p[i + j * 256] : ulonglong2{0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t u0 = (uint32_t)(v[j].x >> ushift) & umask, u1 = (uint32_t)(v[j].y >> ushift) & umask;
            const uint32_t s0 = (uint32_t)(v[j].x >> sshift) & 1u, s1 = (uint32_t)(v[j].y >> sshift) & 1u;
            c += ((s0 ? bn : bw)[u0 >> 5] >> (u0 & 31u)) & 1u;
            c += ((s1 ? bn : bw)[u1 >> 5] >> (u1 & 31u)) & 1u;
        }
    }
    if (c) atomicAdd(out, c);
}
// D: like C, bitmap in LDS (web only)
__global__ __launch_bounds__(1024) void kD(const uint64_t* rows, uint32_t R, Fmt f, const uint32_t* bw, uint32_t words, uint32_t* out) {
    extern __shared__ uint32_t bm[];
    for (uint32_t x = threadIdx.x; x < words; x += blockDim.x) bm[x] = bw[x];
    __syncthreads();
    uint32_t c = 0;
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(rows);
    const uint32_t n2 = R / 2;
    const uint32_t ushift = f.fb + 2, umask = (uint32_t)((1ull << f.ub) - 1ull);
    for (uint64_t i = ((uint64_t)blockIdx.x * blockDim.x) * 4 + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x * 4) {
        ulonglong2 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (i + j * 1024 < n2) ? p[i + j * 1024] : ulonglong2{0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t u0 = (uint32_t)(v[j].x >> ushift) & umask, u1 = (uint32_t)(v[j].y >> ushift) & umask;
            c += (bm[u0 >> 5] >> (u0 & 31u)) & 1u;
            c += (bm[u1 >> 5] >> (u1 & 31u)) & 1u;
        }
    }
    if (c) atomicAdd(out, c);
}
int main() {
    const uint32_t R = 256914108, ND = 1000000;
    Fmt f{19, 20, 22};
    uint64_t* rows; uint32_t *bw, *bn, *out;
    hipMalloc(&rows, (size_t)R * 8 + 64); hipMalloc(&bw, ND / 8 + 64); hipMalloc(&bn, 64); hipMalloc(&out, 4);
    hipMemset(bw, 0, ND / 8 + 64); hipMemset(bn, 0, 64); hipMemset(out, 0, 4);
    // rows sorted by (term, unit): term t has ~ND*332/(13.7 (t+1)) rows with ascending pseudo-random units
    std::vector<uint64_t> h(R);
    size_t n = 0; uint64_t seed = 1;
    for (uint32_t t = 0; t < 500000 && n < R; ++t) {
        double df = 24.2e6 / (t + 1.0); if (df > ND * 0.98) df = ND * 0.98;
        const double step = ND / df; double u = 0;
        for (; u < ND && n < R; u += step * (0.5 + (double)((seed = seed * 6364136223846793005ull + 1442695040888963407ull) >> 40) / (1 << 24)))
            h[n++] = ((uint64_t)t << 45) | ((uint64_t)(uint32_t)u << 24) | 1;
    }
    for (; n < R; ++n) h[n] = h[n - 1];
    hipMemcpy(rows, h.data(), (size_t)R * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    auto timeit = [&](const char* name, auto launch) {
        launch(); hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); printf("%-40s %.3f ms  %.0f GB/s\n", name, ms / 3, 8.0 * R / (ms / 3 * 1e-3) / 1e9);
    };
    timeit("A library form", [&] { kA<<<8192, 256>>>(rows, R, f, bw, bn, out); });
    timeit("A library form, 65536 blocks", [&] { kA<<<65536, 256>>>(rows, R, f, bw, bn, out); });
    timeit("B pure stream", [&] { kB<<<8192, 256>>>(rows, R, f, bw, bn, out); });
    timeit("C 16-byte loads x4, global bitmap", [&] { kC<<<8192, 256>>>(rows, R, f, bw, bn, out); });
    timeit("D 16-byte loads x4, LDS bitmap", [&] { kD<<<256, 1024, ND / 8 + 64>>>(rows, R, f, bw, ND / 32 + 1, out); });
    timeit("D with 512 blocks", [&] { kD<<<512, 1024, 64 * 1024>>>(rows, R, f, bw, 16000, out); });
    return 0;
}
