// Probe: cost of a chain of 8 tiny dependent kernels launched (a) one by one, (b) as a captured hipGraph, per iteration.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void tiny(float* x, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
}

int main() {
    const int n = 1 << 16, chain = 8, iters = 2000;
    float* x; hipMalloc(&x, n * 4); hipMemset(x, 0, n * 4);
    hipStream_t s; hipStreamCreate(&s);
    auto run_plain = [&] { for (int k = 0; k < chain; ++k) tiny<<<n / 256, 256, 0, s>>>(x, n); };
    for (int i = 0; i < 50; ++i) run_plain();
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) run_plain();
    hipStreamSynchronize(s);
    double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;

    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    run_plain();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 50; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;

    // a graph of 10 iterations (80 nodes)
    hipGraph_t g2; hipGraphExec_t ge2;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int r = 0; r < 10; ++r) run_plain();
    hipStreamEndCapture(s, &g2);
    hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge2, s);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters / 10; ++i) hipGraphLaunch(ge2, s);
    hipStreamSynchronize(s);
    double us_graph10 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
    printf("chain of %d tiny kernels: plain %.1f us/iter, graph %.1f us/iter, 10-iteration graph %.1f us/iter\n", chain, us_plain, us_graph, us_graph10);
    return 0;
}
