// What does one dependent trip to memory cost at the start of a kernel in a chain of short launches (the circuit sweeps of the
// sector path: 46 launches of 256 workgroups, each reading tables no earlier launch of the evaluation has touched)?
// Chain of L launches; workgroup t of launch l: s = small[l][t] (scalar) -> v = big[l][s + lane] (vector, dependent) -> store.
// Arrangements of the tables: (0) every table its own hipMalloc, (1) all tables carved out of ONE allocation,
// (2) every launch uses launch 0's tables (translations and lines stay warm).  depth = dependent trips (0..2).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/trip tools/micro/trip_latency.hip && /tmp/trip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_trip(const unsigned *__restrict__ small, const double *__restrict__ big, double *__restrict__ out,
                                               int depth, unsigned per_tile) {
    extern __shared__ double lds[];
    const unsigned t = blockIdx.x;
    double v = 1.0;
    if (per_tile == 1u) lds[threadIdx.x] = 1.0;   // (never: keeps the dynamic LDS allocation alive)
    if (depth >= 1) {
        const unsigned s = small[t];               // scalar trip
        if (depth >= 2) {
            v = big[(size_t)s + threadIdx.x];       // dependent vector trip
            if (depth >= 3) v = big[(size_t)s + (((unsigned)v + threadIdx.x * 7u) % per_tile)];   // and one more
        } else {
            v = (double)s;
        }
    }
    out[(size_t)t * 1024 + threadIdx.x] = v;
}

__global__ __launch_bounds__(256) void k_flush(const double4 *__restrict__ a, size_t n, double *__restrict__ out) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += a[i].x;
    if (acc == 12345.678) out[0] = acc;
}

int main(int argc, char **argv) {
    const int L = 46, T = 256, reps = 30;
    const size_t lds_bytes = argc > 1 ? (size_t)atoi(argv[1]) * 1024 : 0;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trip), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("dynamic LDS per workgroup: %zu KB\n", lds_bytes / 1024);
    // between two chains: stream 3 GB (what the <H> kernel of an evaluation does) — caches and translation caches are cold again
    const size_t flush_n = (size_t)3 << 30 >> 5;
    double4 *flush;
    CK(hipMalloc(&flush, flush_n * sizeof(double4)));
    CK(hipMemset(flush, 0, flush_n * sizeof(double4)));
    const unsigned per_tile = 8192;                 // doubles per tile: 64 KB
    const size_t big_bytes = (size_t)T * per_tile * sizeof(double), small_bytes = T * sizeof(unsigned);
    std::vector<unsigned> hs(T);
    for (int t = 0; t < T; ++t) hs[t] = (unsigned)t * per_tile;
    double *out;
    CK(hipMalloc(&out, (size_t)T * 1024 * sizeof(double)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int arr = 1; arr < 2; ++arr) {
        std::vector<unsigned *> sm(L);
        std::vector<double *> bg(L);
        void *arena = nullptr;
        if (arr == 1) {
            const size_t stride = ((big_bytes + small_bytes + 4095) / 4096) * 4096;
            CK(hipMalloc(&arena, stride * L));
            for (int l = 0; l < L; ++l) {
                bg[l] = (double *)((char *)arena + stride * l);
                sm[l] = (unsigned *)((char *)arena + stride * l + big_bytes);
            }
        } else {
            for (int l = 0; l < L; ++l) {
                CK(hipMalloc(&bg[l], big_bytes));
                CK(hipMalloc(&sm[l], small_bytes));
            }
        }
        for (int l = 0; l < L; ++l) {
            CK(hipMemcpy(sm[l], hs.data(), small_bytes, hipMemcpyHostToDevice));
            CK(hipMemset(bg[l], 0, big_bytes));
        }
        for (int depth = 0; depth <= 7; ++depth) {
            const bool cold = depth >= 4;
            float best = 1e9f;
            for (int rep = 0; rep < reps; ++rep) {
                if (cold) hipLaunchKernelGGL(k_flush, dim3(4096), dim3(256), 0, 0, flush, flush_n, out);
                CK(hipEventRecord(e0));
                for (int l = 0; l < L; ++l) {
                    const int u = arr == 2 ? 0 : l;
                    hipLaunchKernelGGL(k_trip, dim3(T), dim3(1024), lds_bytes, 0, sm[u], bg[u], out, depth & 3, per_tile);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep >= 5 && ms < best) best = ms;
            }
            printf("arrangement %d (%s) depth %d%s: %.2f us per launch\n", arr,
                   arr == 0 ? "own hipMalloc per table" : arr == 1 ? "one arena" : "same tables every launch", depth & 3,
                   cold ? " after a 3 GB stream" : "", 1e3f * best / L);
        }
        if (arena) CK(hipFree(arena));
        else
            for (int l = 0; l < L; ++l) {
                CK(hipFree(bg[l]));
                CK(hipFree(sm[l]));
            }
    }
    return 0;
}
