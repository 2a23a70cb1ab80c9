// Probe: which physical CUs (XCC id, SE, CU) does a stream created with hipExtStreamCreateWithCUMask run on, for
// several mask patterns; and the HBM bandwidth a streaming copy reaches on a masked stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <map>
#include <set>

__global__ void where_am_i(unsigned* out) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
    // spin a little so that blocks spread over all enabled CUs
    unsigned long long t0 = clock64();
    while (clock64() - t0 < 20000) {}
}

__global__ void stream_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

static void report(const char* name, hipStream_t s, unsigned* d_out, float4* a, float4* b, size_t n) {
    const int nb = 4096;
    std::vector<unsigned> h(2 * nb);
    hipMemsetAsync(d_out, 0xff, 2 * nb * 4, s);
    where_am_i<<<nb, 64, 0, s>>>(d_out);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d_out, 2 * nb * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> cus;
    for (int i = 0; i < nb; ++i) cus[h[2 * i] & 0xf].insert(h[2 * i + 1] & 0xffff0);   // drop wave id bits
    printf("%-28s:", name);
    size_t tot = 0;
    for (auto& kv : cus) { printf(" xcc%u=%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
    printf("  total=%zu", tot);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream_copy<<<8192, 256, 0, s>>>(a, b, n);
    hipEventRecord(e0, s);
    for (int i = 0; i < 5; ++i) stream_copy<<<8192, 256, 0, s>>>(a, b, n);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("  copy %.2f TB/s (r+w)\n", 5.0 * 2.0 * n * 16 / (ms * 1e-3) / 1e12);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs=%d\n", p.multiProcessorCount);
    unsigned* d_out; hipMalloc(&d_out, 2 * 4096 * 4);
    const size_t n = (size_t)1 << 26;     // 1 GiB per array
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16);
    hipStream_t s0; hipStreamCreate(&s0);
    report("unmasked", s0, d_out, a, b, n);
    struct Pat { const char* name; int kind; int arg; };
    Pat pats[] = {{"first 32 bits", 0, 32}, {"first 64 bits", 0, 64}, {"first 224 bits", 0, 224}, {"bits 224..255", 1, 224},
                  {"every 8th bit (i%8==0)", 2, 8}, {"i%32 < 4", 3, 4}, {"i%32 < 8", 3, 8}, {"i%32 >= 8", 4, 8}};
    for (auto& pt : pats) {
        uint32_t mask[8]; memset(mask, 0, sizeof(mask));
        for (int i = 0; i < 256; ++i) {
            bool on = false;
            if (pt.kind == 0) on = i < pt.arg;
            if (pt.kind == 1) on = i >= pt.arg;
            if (pt.kind == 2) on = (i % pt.arg) == 0;
            if (pt.kind == 3) on = (i % 32) < pt.arg;
            if (pt.kind == 4) on = (i % 32) >= pt.arg;
            if (on) mask[i / 32] |= 1u << (i % 32);
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("%-28s: create failed: %s\n", pt.name, hipGetErrorString(e)); continue; }
        report(pt.name, s, d_out, a, b, n);
        hipStreamDestroy(s);
    }
    return 0;
}
