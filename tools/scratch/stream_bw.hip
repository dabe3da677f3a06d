// scratch: what bandwidth do many independent wave-sized streams reach?  Each wavefront reads `len` consecutive
// elements of two arrays (units, weights) starting at its own offset, `UNROLL` 256-byte loads per array in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int UNROLL, int VEC>
__global__ __launch_bounds__(64) void k_stream(const uint32_t* a, const uint32_t* b, const uint32_t* start, uint32_t len, uint32_t* out) {
    const uint32_t lane = threadIdx.x;
    uint32_t acc = 0;
    const uint32_t s = start[blockIdx.x];
    for (uint32_t i = 0; i < len; i += 64 * UNROLL * VEC) {
        uint32_t x[UNROLL * VEC], y[UNROLL * VEC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (VEC == 1) {
                x[u] = a[s + i + 64 * u + lane];
                y[u] = b[s + i + 64 * u + lane];
            } else {
                const uint4 vx = *reinterpret_cast<const uint4*>(a + s + i + 256 * u + 4 * lane);
                const uint4 vy = *reinterpret_cast<const uint4*>(b + s + i + 256 * u + 4 * lane);
                x[4 * u] = vx.x; x[4 * u + 1] = vx.y; x[4 * u + 2] = vx.z; x[4 * u + 3] = vx.w;
                y[4 * u] = vy.x; y[4 * u + 1] = vy.y; y[4 * u + 2] = vy.z; y[4 * u + 3] = vy.w;
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL * VEC; ++u) acc += x[u] ^ y[u];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int UNROLL, int VEC>
void run(const char* name, const uint32_t* a, const uint32_t* b, const uint32_t* st, uint32_t n_waves, uint32_t len, uint32_t* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k_stream<UNROLL, VEC><<<n_waves, 64>>>(a, b, st, len, out);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) k_stream<UNROLL, VEC><<<n_waves, 64>>>(a, b, st, len, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %7.1f GB/s\n", name, ms / 3, 8.0 * n_waves * len / (ms / 3 * 1e-3) / 1e9);
}
int main() {
    const size_t N = 2500000000ull;  // 10 GB per array
    uint32_t *a, *b, *st, *out;
    hipMalloc(&a, N * 4); hipMalloc(&b, N * 4); hipMalloc(&out, 4);
    hipMemset(a, 1, N * 4); hipMemset(b, 2, N * 4);
    const uint32_t n_waves = 60000, len = 131072;
    std::vector<uint32_t> h(n_waves);
    srand(1);
    for (auto& x : h) x = (uint32_t)(((uint64_t)rand() * 977 + rand()) % (N - len - 4096)) & ~63u;
    hipMalloc(&st, n_waves * 4); hipMemcpy(st, h.data(), n_waves * 4, hipMemcpyHostToDevice);
    run<1, 1>("dword x1 in flight", a, b, st, n_waves, len, out);
    run<2, 1>("dword x2", a, b, st, n_waves, len, out);
    run<4, 1>("dword x4", a, b, st, n_waves, len, out);
    run<8, 1>("dword x8", a, b, st, n_waves, len, out);
    run<1, 4>("dwordx4 x1", a, b, st, n_waves, len, out);
    run<2, 4>("dwordx4 x2", a, b, st, n_waves, len, out);
    // sequential starts (waves side by side)
    for (uint32_t i = 0; i < n_waves; ++i) h[i] = i * len % (uint32_t)(N - len - 4096) & ~63u;
    hipMemcpy(st, h.data(), n_waves * 4, hipMemcpyHostToDevice);
    run<4, 1>("dword x4, adjacent ranges", a, b, st, n_waves, len, out);
    return 0;
}
