// Throughput of scattered LDS accesses per CU: 64-bit reads, 64-bit writes and f64 atomic adds (ds_add_f64, no return) on
// pseudo-random slots of a 4096-double tile, 512 threads per workgroup, one workgroup per CU.  k_sector_apply pays one such
// atomic per matrix element on top of k_sector_expect's two reads.  Prints ns per wave-instruction (64 lanes) per CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsat tools/micro/lds_atomic.hip && /tmp/ldsat
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE, bool SPREAD>
__global__ __launch_bounds__(512) void k_lds(double *out, int N) {
    __shared__ double tile[4096];
    for (int k = threadIdx.x; k < 4096; k += 512) tile[k] = 1.0 + k;
    __syncthreads();
    unsigned x = threadIdx.x * 2654435761u + 12345u;
    double acc = 0.0;
    for (int it = 0; it < N; it += 4) {
        unsigned s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x = x * 1664525u + 1013904223u;
            // SPREAD: the 32 lanes of a half wave fall into 32 distinct banks (slot = lane mod 32 + 32 * random)
            s[j] = SPREAD ? ((threadIdx.x & 31u) + 32u * ((x >> 12) & 127u)) : ((x >> 10) & 4095u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE == 0) acc += tile[s[j]];
            if (MODE == 1) tile[s[j]] = acc + (double)j;
            if (MODE == 2) __hip_atomic_fetch_add(&tile[s[j]], 0.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == 3) {   // expect-like: two reads and an fma; apply-like (MODE 4): plus the atomic
                acc += tile[s[j]] * tile[(s[j] * 7u + 1u) & 4095u];
            }
            if (MODE == 4) {
                const double v = tile[(s[j] * 7u + 1u) & 4095u];
                acc += tile[s[j]] * v;
                __hip_atomic_fetch_add(&tile[s[j]], 1e-9 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + tile[5];
}

template <int MODE, bool SPREAD>
int run(double *d_out, const char *what) {
    const int N = 1 << 14;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_lds<MODE, SPREAD>), dim3(256), dim3(512), 0, 0, d_out, N);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_lds<MODE, SPREAD>), dim3(256), dim3(512), 0, 0, d_out, N);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %-8s %7.3f ms  %6.2f ns per wave-instruction per CU\n", what, SPREAD ? "banked" : "random", ms, 1e6 * ms / ((double)N * 8.0));
    return 0;
}

int main() {
    double *d_out;
    CK(hipMalloc(&d_out, 4096 * sizeof(double)));
#define BOTH(M, W) if (run<M, false>(d_out, W) || run<M, true>(d_out, W)) return 1
    BOTH(0, "64-bit read");
    BOTH(1, "64-bit write");
    BOTH(2, "f64 atomic add (no return)");
    BOTH(3, "two reads + fma (k_sector_expect)");
    BOTH(4, "two reads + fma + atomic (k_sector_apply)");
    return 0;
}
