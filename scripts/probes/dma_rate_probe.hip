// Probe: what rate does the L2 -> LDS path of a CU sustain for the staging pattern of the GEMM kernels?
// One 512-thread workgroup per CU (128 KB of LDS, like tg_bwd_kernel) copies, step after step, a 512-row x 128-byte operand slab
// into LDS with global_load_lds_dwordx4 (8 copies per thread and step, source addresses swizzled like tg_ktile_dma), from operand
// images of the cfg2 size (L2 / MALL resident after the first pass).  Variants:
//   layout 0: rows `pitch` bytes apart ([row][step][128 B], what the library uses), layout 1: step-major ([step][row][128 B]:
//             the 512 pieces of a step are contiguous);
//   reads 0/1: with 24 ds_read_b128 per wave and step beside the copies (the fragment traffic of the GEMM) or without;
//   wait: vmcnt(0) + barrier per step (like the GEMM), or only every 4th step.
// Reports bytes per clock and CU.  Build: hipcc --offload-arch=gfx950 -O3 -o build/dma_rate_probe scripts/probes/dma_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ (((row >> 1) & 7) ^ ((row >> 4) & 1)); }

template <int LAYOUT, int READS, int WAIT_EVERY>
__global__ void __launch_bounds__(512, 2) dma_kernel(const unsigned char* base, size_t pitch, int nrows_total, int nsteps, int iters,
                                                     unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    u32x4 acc = {0, 0, 0, 0};
    const int tile = blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        const size_t row0 = ((size_t)(tile * 7 + it * 13) * 512) % (size_t)(nrows_total - 512);
        for (int s = 0; s < nsteps; ++s) {
            u32x4* stage = (u32x4*)lds + (s & 1) * 4096;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = t + i * 512, row = idx >> 3;
                const int logical = swz(row, idx & 7);
                const unsigned char* src = LAYOUT == 0 ? base + (row0 + row) * pitch + (size_t)s * 128 + logical * 16
                                                       : base + ((size_t)s * nrows_total + row0 + row) * 128 + logical * 16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(stage + i * 512 + wave * 64), 16, 0, 0);
            }
            if (READS) {
                const u32x4* other = (const u32x4*)lds + ((s + 1) & 1) * 4096;
#pragma unroll
                for (int r = 0; r < 24; ++r) {
                    const u32x4 v = other[((lane & 15) + 16 * (r % 16)) * 8 + ((lane >> 4) ^ (r & 7))];
                    acc ^= v;
                }
            }
            if ((s % WAIT_EVERY) == WAIT_EVERY - 1) __syncthreads();
        }
        __syncthreads();
    }
    if (acc[0] == 0x12345678u) sink[0] = acc[1];
}

template <int LAYOUT, int READS, int WAIT_EVERY>
static void run(const char* name, const unsigned char* buf, size_t pitch, int nrows, int nsteps, unsigned* sink) {
    hipFuncSetAttribute((const void*)dma_kernel<LAYOUT, READS, WAIT_EVERY>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const int grid = 256, iters = 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    dma_kernel<LAYOUT, READS, WAIT_EVERY><<<grid, 512, 131072>>>(buf, pitch, nrows, nsteps, 2, sink);
    hipEventRecord(e0);
    dma_kernel<LAYOUT, READS, WAIT_EVERY><<<grid, 512, 131072>>>(buf, pitch, nrows, nsteps, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * iters * nsteps * 65536.0;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%-44s %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU at %.2f GHz (nominal)\n", name, ms, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / (clk_khz * 1e3),
           clk_khz / 1e6);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) printf("  HIP error %s\n", hipGetErrorString(e));
}

int main() {
    const int nrows = 40960, nsteps = 32;                 // 40 960 rows x 32 steps x 128 B = 168 MB (cfg2: S^k 123 MB + dGhat 42 MB)
    const size_t pitch = (size_t)nsteps * 128, bytes = (size_t)nrows * pitch;
    unsigned char* buf;
    unsigned* sink;
    hipMalloc(&buf, bytes);
    hipMalloc(&sink, 64);
    hipMemset(buf, 1, bytes);
    run<0, 0, 1>("row-pitch layout, copies only, sync/step", buf, pitch, nrows, nsteps, sink);
    run<1, 0, 1>("step-major layout, copies only, sync/step", buf, pitch, nrows, nsteps, sink);
    run<0, 1, 1>("row-pitch layout, + ds_reads, sync/step", buf, pitch, nrows, nsteps, sink);
    run<1, 1, 1>("step-major layout, + ds_reads, sync/step", buf, pitch, nrows, nsteps, sink);
    run<0, 0, 4>("row-pitch layout, copies only, sync/4 steps", buf, pitch, nrows, nsteps, sink);
    run<1, 0, 4>("step-major layout, copies only, sync/4 steps", buf, pitch, nrows, nsteps, sink);
    // the same copies out of a footprint that stays in every XCD's 4 MB L2 (1 024 rows = 4 MB): the ceiling of the L2 -> LDS path itself
    run<0, 0, 1>("L2-resident (4 MB), copies only, sync/step", buf, pitch, 1024, nsteps, sink);
    run<0, 1, 1>("L2-resident (4 MB), + ds_reads, sync/step", buf, pitch, 1024, nsteps, sink);
    run<0, 0, 4>("L2-resident (4 MB), copies only, sync/4 steps", buf, pitch, 1024, nsteps, sink);
    hipFree(buf); hipFree(sink);
    return 0;
}
