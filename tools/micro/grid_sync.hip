// cost of a cooperative-groups grid barrier on MI355X for the geometry of the sector sweeps (256 workgroups x 1024 threads,
// ~100 KB of LDS each): build with hipcc --offload-arch=gfx950 -O3 tools/micro/grid_sync.hip -o gpurun_out/grid_sync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ __launch_bounds__(1024) void k_sync(int n, double *out) {
    extern __shared__ double lds[];
    cg::grid_group g = cg::this_grid();
    double acc = threadIdx.x;
    for (int i = 0; i < n; ++i) {
        lds[threadIdx.x] = acc;
        g.sync();
        acc += lds[(threadIdx.x + 1) & 1023];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
int main() {
    double *out;
    hipMalloc(&out, 4096 * sizeof(double));
    for (int blocks : {128, 256, 512}) {
        for (int threads : {256, 1024}) {
            int n = 200;
            size_t smem = blocks <= 256 ? 100 * 1024 : 48 * 1024;
            hipFuncSetAttribute((const void *)k_sync, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            void *args[] = {&n, &out};
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipError_t e = hipLaunchCooperativeKernel((const void *)k_sync, dim3(blocks), dim3(threads), args, smem, 0);
            if (e != hipSuccess) { printf("blocks %d threads %d: %s\n", blocks, threads, hipGetErrorString(e)); continue; }
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchCooperativeKernel((const void *)k_sync, dim3(blocks), dim3(threads), args, smem, 0);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            printf("blocks %d threads %d: %.2f us per grid sync (%d syncs in %.3f ms)\n", blocks, threads, 1e3 * ms / n, n, ms);
        }
    }
    return 0;
}
