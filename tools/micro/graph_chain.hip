// A chain of 46 dependent small kernels (the shape of a sector evaluation's circuit sweeps: 256 workgroups x 1024 threads, a few
// microseconds of work each) launched one by one on a stream against the same chain captured once in a hipGraph and replayed:
// microseconds per kernel, for kernels of ~0, ~5 and ~20 us.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/graph_chain tools/micro/graph_chain.hip && /tmp/graph_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(1024) void k_work(const double *in, double *out, int spin) {
    const size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
    double v = in[i];
    for (int k = 0; k < spin; ++k) v = v * 1.0000001 + 1e-9;
    out[i] = v;
}

int main() {
    const int NK = 46, REPS = 200;
    double *a, *b;
    CK(hipMalloc(&a, 256 * 1024 * sizeof(double)));
    CK(hipMalloc(&b, 256 * 1024 * sizeof(double)));
    CK(hipMemset(a, 0, 256 * 1024 * sizeof(double)));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int spin : {0, 2000, 9000}) {
        auto chain = [&]() {
            for (int k = 0; k < NK; ++k) hipLaunchKernelGGL(k_work, dim3(256), dim3(1024), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, spin);
        };
        chain();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) chain();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms_stream = 0.f;
        CK(hipEventElapsedTime(&ms_stream, e0, e1));
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        chain();
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < REPS; ++r) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms_graph = 0.f;
        CK(hipEventElapsedTime(&ms_graph, e0, e1));
        printf("spin %5d: stream %.2f us per kernel, graph %.2f us per kernel\n", spin, 1e3 * ms_stream / (REPS * NK), 1e3 * ms_graph / (REPS * NK));
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
