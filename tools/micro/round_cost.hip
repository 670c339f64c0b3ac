// What does one ROUND of the sector path's circuit sweeps cost — a third of the workgroup's threads rotate one pair of LDS
// amplitudes each, then the workgroup synchronises — as a function of the workgroup size and of what the round contains?
// One workgroup per CU, R rounds; modes: 0 barrier only, 1 LDS read-modify-write only (no barrier: wrong but timed),
// 2 both, 3 both with s_waitcnt lgkmcnt(0) replaced by nothing before the barrier (wrong), 4: __syncthreads().
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/round tools/micro/round_cost.hip && /tmp/round
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT, int MODE>
__global__ __launch_bounds__(NT) void k_round(double *out, int R, int active_div) {
    __shared__ double tile[8192];
    for (int k = threadIdx.x; k < 8192; k += NT) tile[k] = 1.0 + k;
    __syncthreads();
    const bool active = (threadIdx.x % active_div) == 0;
    unsigned si = (threadIdx.x * 37u) & 8191u, sj = (threadIdx.x * 37u + 4099u) & 8191u;
    const double c = 0.8, s = 0.6;
    const long long t0 = clock64();
    for (int q = 0; q < R; ++q) {
        if (MODE != 0 && active) {
            const double u = tile[si], v = tile[sj];
            tile[si] = c * u + s * v;
            tile[sj] = c * v - s * u;
            si = (si + 61u) & 8191u;
            sj = (sj + 61u) & 8191u;
        }
        if (MODE == 0 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (MODE == 3) asm volatile("s_barrier" ::: "memory");
        if (MODE == 4) __syncthreads();
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = (double)(t1 - t0) / R;
    if (threadIdx.x == 1) out[256 + blockIdx.x] = tile[si];
}

template <int NT, int MODE>
int run(double *d_out, int active_div) {
    const int R = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_round<NT, MODE>), dim3(256), dim3(NT), 0, 0, d_out, R, active_div);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_round<NT, MODE>), dim3(256), dim3(NT), 0, 0, d_out, R, active_div);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double cyc;
    CK(hipMemcpy(&cyc, d_out, sizeof(double), hipMemcpyDeviceToHost));
    printf("NT %4d mode %d active 1/%d: %.1f ns per round (%.0f clock64 ticks)\n", NT, MODE, active_div, 1e6 * ms / R, cyc);
    return 0;
}

// P independent pairs per thread and round: all reads first, then the rotations, then the writes
template <int NT, int P>
__global__ __launch_bounds__(NT) void k_round_multi(double *out, int R) {
    __shared__ double tile[8192];
    for (int k = threadIdx.x; k < 8192; k += NT) tile[k] = 1.0 + k;
    __syncthreads();
    unsigned si[P], sj[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        si[p] = (threadIdx.x * 37u + p * 1031u) & 8191u;
        sj[p] = (threadIdx.x * 37u + p * 1031u + 4099u) & 8191u;
    }
    const double c = 0.8, s = 0.6;
    for (int q = 0; q < R; ++q) {
        double u[P], v[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            u[p] = tile[si[p]];
            v[p] = tile[sj[p]];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            tile[si[p]] = c * u[p] + s * v[p];
            tile[sj[p]] = c * v[p] - s * u[p];
            si[p] = (si[p] + 61u) & 8191u;
            sj[p] = (sj[p] + 61u) & 8191u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (threadIdx.x == 1) out[256 + blockIdx.x] = tile[si[0]];
}
template <int NT, int P>
int run_multi(double *d_out) {
    const int R = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_round_multi<NT, P>), dim3(256), dim3(NT), 0, 0, d_out, R);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_round_multi<NT, P>), dim3(256), dim3(NT), 0, 0, d_out, R);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NT %4d, %d pairs per thread = %4d pairs per round: %.1f ns per round\n", NT, P, NT * P, 1e6 * ms / R);
    return 0;
}

// a FUSED round: an active thread holds a quad of amplitudes {a0, a1 = a0^xA, a2 = a0^xB, a3}, applies op A to (a0,a1) and (a2,a3),
// then op B to (a0,a2) and (a1,a3) — two ops of the circuit per barrier
template <int NT>
__global__ __launch_bounds__(NT) void k_round_quad(double *out, int R, int active_div) {
    __shared__ double tile[8192];
    for (int k = threadIdx.x; k < 8192; k += NT) tile[k] = 1.0 + k;
    __syncthreads();
    const bool active = (threadIdx.x % active_div) == 0;
    unsigned s0 = (threadIdx.x * 37u) & 8191u, s1 = (threadIdx.x * 37u + 4099u) & 8191u, s2 = (threadIdx.x * 37u + 2053u) & 8191u,
             s3 = (threadIdx.x * 37u + 6151u) & 8191u;
    const double c = 0.8, s = 0.6, c2 = 0.6, sb = 0.8;
    for (int q = 0; q < R; ++q) {
        if (active) {
            double t0 = tile[s0], t1 = tile[s1], t2 = tile[s2], t3 = tile[s3];
            const double u0 = c * t0 + s * t1, u1 = c * t1 - s * t0, u2 = c * t2 + s * t3, u3 = c * t3 - s * t2;
            tile[s0] = c2 * u0 + sb * u2;
            tile[s2] = c2 * u2 - sb * u0;
            tile[s1] = c2 * u1 + sb * u3;
            tile[s3] = c2 * u3 - sb * u1;
            s0 = (s0 + 61u) & 8191u; s1 = (s1 + 61u) & 8191u; s2 = (s2 + 61u) & 8191u; s3 = (s3 + 61u) & 8191u;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (threadIdx.x == 1) out[256 + blockIdx.x] = tile[s0];
}
template <int NT>
int run_quad(double *d_out, int active_div) {
    const int R = 2000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_round_quad<NT>), dim3(256), dim3(NT), 0, 0, d_out, R, active_div);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_round_quad<NT>), dim3(256), dim3(NT), 0, 0, d_out, R, active_div);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("NT %4d fused quad round, active 1/%d (%d quads): %.1f ns per round\n", NT, active_div, NT / active_div, 1e6 * ms / R);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1) {   // "quad": the fused rounds only
        double *d;
        CK(hipMalloc(&d, 1024 * sizeof(double)));
#define QUAD(NT) run_quad<NT>(d, 1); run_quad<NT>(d, 2); run_quad<NT>(d, 4); run_quad<NT>(d, 8);
        QUAD(128) QUAD(256) QUAD(512) QUAD(1024)
        return 0;
    }
    double *d_out;
    CK(hipMalloc(&d_out, 1024 * sizeof(double)));
#define ALL(NT) run<NT, 0>(d_out, 3); run<NT, 1>(d_out, 3); run<NT, 2>(d_out, 3); run<NT, 2>(d_out, 1); run<NT, 3>(d_out, 3); run<NT, 4>(d_out, 3);
    ALL(64) ALL(128) ALL(256) ALL(512) ALL(1024)
#define MULTI(NT) run_multi<NT, 1>(d_out); run_multi<NT, 2>(d_out); run_multi<NT, 4>(d_out); run_multi<NT, 8>(d_out);
    MULTI(64) MULTI(128) MULTI(256) MULTI(512) MULTI(1024)
    return 0;
}
