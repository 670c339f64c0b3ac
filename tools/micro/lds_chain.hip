// The dependent chain of the third sweep form: a wave reads two LDS amplitudes per lane, rotates them, writes them back, and the next
// row does the same on slots the row before may have written (the in-order LDS pipe is all that orders them).  Cycles and ns per row
// of one wave as a function of the waves of the workgroup that run such chains side by side, with the slots (a) the same every row
// (a true dependence through LDS), (b) fresh every row (pipelined: what the LDS pipe sustains).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_chain tools/micro/lds_chain.hip && /tmp/lds_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(1024) void k_chain(double *out, int R, int nwave) {
    __shared__ double tile[8192];
    for (int k = threadIdx.x; k < 8192; k += 1024) tile[k] = 1.0 + k;
    __syncthreads();
    const unsigned wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned si = (wv * 512u + lane * 2u) & 8191u, sj = (wv * 512u + lane * 2u + 1u) & 8191u;   // conflict-free, private to the wave
    const double c = 0.8, s = 0.6;
    long long t0 = 0, t1 = 0;
    if (wv < (unsigned)nwave) {
        t0 = __builtin_amdgcn_s_memtime();
        for (int q = 0; q < R; ++q) {
            const double u = tile[si], v = tile[sj];
            tile[si] = c * u + s * v;
            tile[sj] = c * v - s * u;
            if (MODE == 1) {   // other slots next row (of the wave's own 512): nothing to wait for but the pipe
                si = (wv * 512u + ((si + 128u) & 511u)) & 8191u;
                sj = (wv * 512u + ((sj + 128u) & 511u)) & 8191u;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    if (lane == 0 && wv < (unsigned)nwave) out[blockIdx.x * 16 + wv] = (double)(t1 - t0) / R;
    if (threadIdx.x == 1) out[4096 + blockIdx.x] = tile[si];
}

int main() {
    double *d_out;
    CK(hipMalloc(&d_out, 8192 * sizeof(double)));
    const int R = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int nw : {1, 2, 4, 8, 16}) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_chain<0>, dim3(256), dim3(1024), 0, 0, d_out, R, nw);
                else hipLaunchKernelGGL(k_chain<1>, dim3(256), dim3(1024), 0, 0, d_out, R, nw);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
            }
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            double h[16];
            CK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
            printf("%s, %2d waves: %.1f memtime ticks per row (100 MHz: %.1f ns), kernel %.1f ns per row\n", mode ? "fresh slots" : "same slots ", nw, h[0],
                   h[0] * 10.0, 1e6 * ms / R);
        }
    return 0;
}
