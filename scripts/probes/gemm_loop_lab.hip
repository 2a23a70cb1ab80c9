// Probe: the main loop of the split-bf16 backward GEMM (256 x 256 tile, 8 waves, 32 contraction steps of 128-byte operand rows)
// in isolation, in variants -- which MFMA form, which operand-fragment schedule, what the LDS reads / the LDS-DMA / the barrier
// cost on top of the bare matrix-core time.  Operand images of the cfg2 size (dGhat 10 240 rows, S 30 208 rows, 4 KB per row),
// random split-bf16 content, the library's XCD-aware tile map, one workgroup per tile; the epilogue is a per-thread checksum
// (the X store is not what is probed).  Build (scripts/probes are not part of the product):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -o build/gemm_loop_lab scripts/probes/gemm_loop_lab.hip
// Variants (VAR):
//   0  library loop: tg_tile_mma (16x16x32 fragments), LDS-DMA trickled between the MFMA groups, barrier per step
//   1  16x16x32 MFMAs only (operands stay in registers; no LDS read, no DMA, no barrier)       -> ceiling of the form
//   2  32x32x16 MFMAs only                                                                     -> ceiling of the form
//   3  0 without DMA and barrier (LDS reads + MFMAs)
//   4  32x32x16 rolling-fragment loop, LDS reads + MFMAs only
//   5  32x32x16 rolling-fragment loop, DMA + barrier per step (the full loop)
//   6  0 with the step's barrier replaced by nothing (DMA kept; wrong data, timing only)
//   7  0 with ALL copies issued by waves 0-3 (16 each), waves 4-7 only multiply
//   8  0 with the copies of waves 0-3 in MFMA groups 0-1 and those of waves 4-7 in groups 2-3 (phase-shifted issue)
//   9  0 with ALL copies issued by the even waves
//  10  3 + barrier per step (no DMA)
//  11  1 + the copies (operands in registers, the DMA only writes LDS): issue cost of the copies without LDS-read contention
//  12  0 with all copies of a step issued in MFMA group 0
//  13  0 with the copies as buffer_load_dwordx4 ... offen lds (descriptor + 32-bit lane offsets fixed per tile, the step in soffset:
//      no address arithmetic in the loop), issued like 0
//  14  11 with those buffer copies
//  15  13 with all copies of a step in MFMA group 0
//  16  13 without the save / restore of M0 around each copy
// Every kernel also reports the shader-clock cycles wave 0 of workgroup 0 spent in the loop (s_memtime): effective clock.
#include "../../tangram_amd/csrc/tg_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef TgGeoLarge GE;
typedef PrecBF16x3 PR;

__device__ __forceinline__ f32x16 mma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(tg_bf16x8, a), __builtin_bit_cast(tg_bf16x8, b), c, 0, 0, 0);
}
// split-bf16 product on 32 x 32 x 16 fragments: a[0] hi, a[1] lo
__device__ __forceinline__ f32x16 mma32x3(const u32x4 (&a)[2], const u32x4 (&b)[2], f32x16 c) {
    c = mma32(a[1], b[0], c);
    c = mma32(a[0], b[1], c);
    c = mma32(a[0], b[0], c);
    return c;
}

// the global_load_lds form of the tile copy (what the library used before round 4; kept here as the probe's baseline)
template <int ROWS, int NT>
__device__ __forceinline__ void tg_ktile_dma(const unsigned char* base, size_t row0, size_t pitch_bytes, size_t step, u32x4* tile, int t, int wave,
                                             int part = 0, int nparts = 1) {
#pragma unroll
    for (int i = 0; i < ROWS * 8 / NT; ++i) {
        if (i % nparts != part) continue;
        const int idx = t + i * NT, row = idx >> 3;
        const int logical = tg_swz(row, idx & 7);
        tg_glds16_uncounted(base + (row0 + row) * pitch_bytes + step * 128 + logical * 16, (unsigned char*)(tile + i * NT + wave * 64));
    }
}
// LDS-DMA through a buffer descriptor, M0 written but NOT restored (VAR 16: is the save / restore pair worth anything?)
__device__ __forceinline__ void glds16_buf_nosave(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
// LDS-DMA through a buffer descriptor: source = descriptor base + lane offset (VGPR, fixed per tile) + soffset (SGPR: the step)
__device__ __forceinline__ void glds16_buf(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 r = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}

// One contraction step on 32 x 32 x 16 fragments (wave tile 128 x 64 = 4 x 2 fragments, two 16-deep k-slices per step).
// Lane (r = l & 31, g = l >> 5) feeds chunk p * 4 + 2 h + g of its row (p: hi / lo part, h: k-slice).  A fragments live in ONE
// register set and are re-loaded for the next slice right behind their last use (3 groups = 18 MFMAs ahead of the next use),
// B fragments are double-buffered across the slices.  hook(g) once per group of 6 MFMAs (8 groups per step).
template <class Hook>
__device__ __forceinline__ void tile_mma32(const u32x4* st, int wm, int wn, int lane, f32x16 (&acc)[4][2], Hook&& hook) {
    const int r = lane & 31, g = lane >> 5;
    const int ra = wm * 128 + r, rb = wn * 64 + r;
    const u32x4* sa = st + ra * 8;
    const u32x4* sb = st + GE::A_CHUNKS + rb * 8;
    const int xa = g ^ tg_swz(ra, 0), xb = g ^ tg_swz(rb, 0);      // (blocks start at multiples of 32: the swizzle term is the lane's)
    u32x4 a[4][2], b[2][2][2];
    auto ld_a = [&](int blk, int h) {
#pragma unroll
        for (int p = 0; p < 2; ++p) a[blk][p] = sa[blk * 256 + ((4 * p + 2 * h) ^ xa)];
    };
    auto ld_b = [&](int buf, int h) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[buf][blk][p] = sb[blk * 256 + ((4 * p + 2 * h) ^ xb)];
    };
    ld_b(0, 0);
    ld_a(0, 0);
    ld_a(1, 0);
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
        const int h = grp >> 2, blk = grp & 3;
        // reads issued in front of this group's MFMAs
        if (grp == 0) ld_a(2, 0);
        if (grp == 1) { ld_a(3, 0); ld_b(1, 1); }
        if (grp >= 2 && grp < 6) ld_a((grp - 2) & 3, 1);           // block (grp-2) of slice 1: its slice-0 fragments were used 2 groups ago
        hook(grp);
        TG_SCHED_FENCE();
        acc[blk][0] = mma32x3(a[blk], b[h][0], acc[blk][0]);
        acc[blk][1] = mma32x3(a[blk], b[h][1], acc[blk][1]);
        TG_SCHED_FENCE();
    }
}

template <int VAR>
__global__ void __launch_bounds__(512, 2) lab_kernel(const unsigned char* dG, const unsigned char* Sk, int nsteps, TgTileMap map, float* sink, unsigned long long* cyc) {
    TG_LDS_DECL;
    u32x4* lds = (u32x4*)tg_lds;
    const int t = threadIdx.x, lane = t & 63, wave = tg_uniform(t >> 6);
    const int wm = wave / GE::WN, wn = wave % GE::WN;
    int ct, vt;
    if (!tg_tilemap(map, blockIdx.x, ct, vt)) return;
    const int v0 = vt * GE::TM, c0 = ct * GE::TN;
    const size_t pitch = (size_t)nsteps * 128;
    constexpr bool F32 = (VAR == 2 || VAR == 4 || VAR == 5);
    constexpr bool BUF = (VAR == 13 || VAR == 14 || VAR == 15 || VAR == 16);
    constexpr bool DMA = BUF || (VAR == 0 || VAR == 5 || VAR == 6 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 11 || VAR == 12);
    constexpr bool BAR = VAR == 13 || VAR == 15 || VAR == 16 || (VAR == 0 || VAR == 5 || VAR == 7 || VAR == 8 || VAR == 9 || VAR == 10 || VAR == 12);
    f32x4 acc[GE::FM][GE::FN];
    f32x16 acc32[4][2];
#pragma unroll
    for (int i = 0; i < GE::FM; ++i)
#pragma unroll
        for (int j = 0; j < GE::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;

    // buffer copies: descriptors over the tile's operand rows, this lane's byte offset of each of its LA + LB copies
    const u32x4 rsA = make_rsrc(dG + (size_t)v0 * pitch, (unsigned)(GE::TM * pitch)), rsB = make_rsrc(Sk + (size_t)c0 * pitch, (unsigned)(GE::TN * pitch));
    unsigned voffA[GE::LA], voffB[GE::LB];
#pragma unroll
    for (int i = 0; i < GE::LA; ++i) { const int idx = t + i * GE::NT, row = idx >> 3; voffA[i] = (unsigned)(row * pitch) + tg_swz(row, idx & 7) * 16; }
#pragma unroll
    for (int i = 0; i < GE::LB; ++i) { const int idx = t + i * GE::NT, row = idx >> 3; voffB[i] = (unsigned)(row * pitch) + tg_swz(row, idx & 7) * 16; }
    const unsigned ldsbase = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)tg_lds;
    auto buf_copy = [&](int k, int step, int stage) {             // copy k (0 .. LA+LB-1) of `step` into `stage`
        const unsigned dst = ldsbase + stage * GE::STAGE_BYTES + ((k < GE::LA ? k : GE::LA + (k - GE::LA)) * GE::NT + wave * 64) * 16
                             + (k < GE::LA ? 0 : (GE::A_CHUNKS - GE::LA * GE::NT) * 16);
        if (VAR == 16) {
            if (k < GE::LA) glds16_buf_nosave(rsA, voffA[k], (unsigned)step * 128u, dst);
            else glds16_buf_nosave(rsB, voffB[k - GE::LA], (unsigned)step * 128u, dst);
            return;
        }
        if (k < GE::LA) glds16_buf(rsA, voffA[k], (unsigned)step * 128u, dst);
        else glds16_buf(rsB, voffB[k - GE::LA], (unsigned)step * 128u, dst);
    };
    tg_ktile_dma<GE::TM, GE::NT>(dG, (size_t)v0, pitch, 0, lds, t, wave);
    tg_ktile_dma<GE::TN, GE::NT>(Sk, (size_t)c0, pitch, 0, lds + GE::A_CHUNKS, t, wave);
    if (!DMA) {   // both stages filled once: the loop then only reads
        tg_ktile_dma<GE::TM, GE::NT>(dG, (size_t)v0, pitch, 1, lds + GE::STAGE_CHUNKS, t, wave);
        tg_ktile_dma<GE::TN, GE::NT>(Sk, (size_t)c0, pitch, 1, lds + GE::STAGE_CHUNKS + GE::A_CHUNKS, t, wave);
    }
    tg_dma_drain();
    __syncthreads();
    const unsigned long long t_begin = __builtin_readcyclecounter();

    if constexpr (VAR == 1 || VAR == 11 || VAR == 14) {       // operands fixed in registers
        const int r = lane & 15, g = lane >> 4;
        const u32x4* sa = lds + (wm * 128 + r) * 8;
        const u32x4* sb = lds + GE::A_CHUNKS + (wn * 64 + r) * 8;
        u32x4 a[8][2], b[4][2];
#pragma unroll
        for (int f = 0; f < 8; ++f)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[f][p] = sa[f * 128 + ((4 * p + g) ^ tg_swz(wm * 128 + r, 0) ^ (f & 1))];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[f][p] = sb[f * 128 + ((4 * p + g) ^ tg_swz(wn * 64 + r, 0) ^ (f & 1))];
        for (int s = 0; s < nsteps; ++s) {
            u32x4* nxt = lds + ((s + 1) & 1) * GE::STAGE_CHUNKS;
#pragma unroll
            for (int fi = 0; fi < 8; ++fi)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    if (VAR == 14 && h2 == 0 && s + 1 < nsteps) buf_copy(fi, s + 1, (s + 1) & 1);
                    if (VAR == 11 && h2 == 0 && s + 1 < nsteps) {              // one copy in front of 12 MFMAs, like the library's trickle
                        if (fi < GE::LA) tg_ktile_dma<GE::TM, GE::NT>(dG, (size_t)v0, pitch, (size_t)(s + 1), nxt, t, wave, fi, GE::LA);
                        else tg_ktile_dma<GE::TN, GE::NT>(Sk, (size_t)c0, pitch, (size_t)(s + 1), nxt + GE::A_CHUNKS, t, wave, fi - GE::LA, GE::LB);
                    }
#pragma unroll
                    for (int fj = 2 * h2; fj < 2 * h2 + 2; ++fj) acc[fi][fj] = PR::mma(a[fi], b[fj], acc[fi][fj]);
                    TG_SCHED_FENCE();
                }
            if (VAR == 11 || VAR == 14) tg_dma_drain();
        }
    } else if constexpr (VAR == 2) {
        const int r = lane & 31, g = lane >> 5;
        const u32x4* sa = lds + (wm * 128 + r) * 8;
        const u32x4* sb = lds + GE::A_CHUNKS + (wn * 64 + r) * 8;
        u32x4 a[4][2], b[2][2];
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[f][p] = sa[f * 256 + ((4 * p + g) ^ tg_swz(wm * 128 + r, 0))];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[f][p] = sb[f * 256 + ((4 * p + g) ^ tg_swz(wn * 64 + r, 0))];
        for (int s = 0; s < nsteps; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    acc32[blk][0] = mma32x3(a[blk], b[0], acc32[blk][0]);
                    acc32[blk][1] = mma32x3(a[blk], b[1], acc32[blk][1]);
                    TG_SCHED_FENCE();
                }
        }
    } else {
        for (int s = 0; s + 1 < nsteps; ++s) {
            u32x4* cur = lds + (s & 1) * GE::STAGE_CHUNKS;
            u32x4* nxt = lds + ((s + 1) & 1) * GE::STAGE_CHUNKS;
            auto dma_hook = [&](int i, int ngroups) {
                if (!DMA) return;
                constexpr int NIT = GE::LA + GE::LB;
                if constexpr (VAR == 7 || VAR == 9) {          // half of the waves issue everything: 2 * NIT copies each, 256-thread mapping
                    const bool loader = (VAR == 7) ? (wave < 4) : ((wave & 1) == 0);
                    if (!loader) return;
                    const int w2 = (VAR == 7) ? wave : (wave >> 1), t2 = w2 * 64 + lane;
                    const int NSP = (ngroups * 3) / 4;
#pragma unroll
                    for (int k = 0; k < 2 * NIT; ++k) {
                        if ((k * NSP) / (2 * NIT) != i) continue;
                        if (k < 2 * GE::LA) tg_ktile_dma<GE::TM, 256>(dG, (size_t)v0, pitch, (size_t)(s + 1), nxt, t2, w2, k, 2 * GE::LA);
                        else tg_ktile_dma<GE::TN, 256>(Sk, (size_t)c0, pitch, (size_t)(s + 1), nxt + GE::A_CHUNKS, t2, w2, k - 2 * GE::LA, 2 * GE::LB);
                    }
                    return;
                }
                const int NSP = (VAR == 12 || VAR == 15) ? 1 : (VAR == 8) ? ngroups / 2 : (ngroups * 3) / 4;
                const int ii = (VAR == 8 && wave >= 4) ? i - ngroups / 2 : i;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {
                    if ((k * NSP) / NIT != ii) continue;
                    if constexpr (BUF) { buf_copy(k, s + 1, (s + 1) & 1); continue; }
                    if (k < GE::LA) tg_ktile_dma<GE::TM, GE::NT>(dG, (size_t)v0, pitch, (size_t)(s + 1), nxt, t, wave, k, GE::LA);
                    else tg_ktile_dma<GE::TN, GE::NT>(Sk, (size_t)c0, pitch, (size_t)(s + 1), nxt + GE::A_CHUNKS, t, wave, k - GE::LA, GE::LB);
                }
            };
            if constexpr (F32) tile_mma32(cur, wm, wn, lane, acc32, [&](int i) { dma_hook(i, 8); });
            else tg_tile_mma<PR, GE>(cur, wm, wn, lane, acc, [&](int i) { dma_hook(i, TgMmaShape<PR, GE>::NG); });
            if (DMA) tg_dma_drain();
            if (BAR) __syncthreads();
        }
    }
    const unsigned long long t_end = __builtin_readcyclecounter();
    if (blockIdx.x == 0 && t == 0) cyc[0] = t_end - t_begin;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < GE::FM; ++i)
#pragma unroll
        for (int j = 0; j < GE::FN; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) sum += acc32[i][j][e];
    sink[(size_t)blockIdx.x * 512 + t] = sum;
}

__global__ void fill_kernel(unsigned* p, size_t n, unsigned seed) {       // operand images: hi = random bf16 in [-2, 2), lo = 2^-9 of that scale
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        const bool lo = ((i >> 2) & 4) != 0;                              // chunks 4-7 of a 128-byte step row are the lo parts
        auto bf = [&](unsigned h) { unsigned e = lo ? (118u + (h & 1)) : (125u + (h & 3)); return ((h >> 2) & 0x8000u) | (e << 7) | ((h >> 4) & 0x7fu); };
        p[i] = bf(x & 0xffffu) | (bf(x >> 16) << 16);
    }
}

template <int VAR>
static float run(const unsigned char* dG, const unsigned char* Sk, int nsteps, TgTileMap map, float* sink, int reps, unsigned long long* cyc) {
    hipFuncSetAttribute((const void*)lab_kernel<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, GE::LDS_BYTES);
    const int grid = tg_tilemap_grid(map);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) lab_kernel<VAR><<<grid, 512, GE::LDS_BYTES>>>(dG, Sk, nsteps, map, sink, cyc);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) lab_kernel<VAR><<<grid, 512, GE::LDS_BYTES>>>(dG, Sk, nsteps, map, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed VAR %d\n", VAR); exit(1); }
    return ms / reps;
}

int main() {
    const int V = 10000, C = 30000, nsteps = 32;
    const int nvt = (V + 255) / 256, nct = (C + 255) / 256;
    const size_t rowsA = (size_t)nvt * 256, rowsB = (size_t)nct * 256, pitch = (size_t)nsteps * 128;
    unsigned char *dG, *Sk; float* sink;
    hipMalloc(&dG, rowsA * pitch); hipMalloc(&Sk, rowsB * pitch);
    TgTileMap map{1, nct, nvt};
    hipMalloc(&sink, (size_t)tg_tilemap_grid(map) * 512 * 4);
    hipMemset(sink, 0, (size_t)tg_tilemap_grid(map) * 512 * 4);
    fill_kernel<<<2048, 256>>>((unsigned*)dG, rowsA * pitch / 4, 1u);
    fill_kernel<<<2048, 256>>>((unsigned*)Sk, rowsB * pitch / 4, 7u);
    hipDeviceSynchronize();
    const double flops = 2.0 * 3 * (double)rowsA * rowsB * (nsteps * 32);      // MFMA flops issued (3 products)
    unsigned long long* cyc;
    hipMalloc(&cyc, 8);
    const char* names[] = {"lib loop 16x16x32 (DMA+barrier)", "16x16x32 MFMA only", "32x32x16 MFMA only", "16x16x32 LDS reads + MFMA",
                           "32x32x16 rolling, LDS reads + MFMA", "32x32x16 rolling, DMA + barrier", "lib loop, no barrier (timing only)",
                           "lib loop, copies by waves 0-3", "lib loop, copies phase-shifted by half", "lib loop, copies by even waves",
                           "LDS reads + MFMA + barrier (no DMA)", "MFMA only + copies (no LDS reads)", "lib loop, copies in group 0",
                           "lib loop, buffer copies", "MFMA only + buffer copies", "lib loop, buffer copies in group 0",
                           "lib loop, buffer copies, M0 not restored"};
    std::vector<float> host((size_t)512);
    typedef float (*runner)(const unsigned char*, const unsigned char*, int, TgTileMap, float*, int, unsigned long long*);
    runner runs[] = {run<0>, run<1>, run<2>, run<3>, run<4>, run<5>, run<6>, run<7>, run<8>, run<9>, run<10>, run<11>, run<12>, run<13>, run<14>, run<15>, run<16>};
    for (int round = 0; round < 3; ++round) {
        for (int v = 0; v < 17; ++v) {
            const float ms = runs[v](dG, Sk, nsteps, map, sink, 10, cyc);
            hipMemcpy(host.data(), sink, 512 * 4, hipMemcpyDeviceToHost);
            double c = 0; for (float x : host) c += x;
            unsigned long long hc = 0;
            hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
            printf("round %d  VAR %2d  %-40s %.4f ms  %.3f of 2.5 PF  loop cycles (wg 0) %llu  checksum %.9e\n", round, v, names[v], ms,
                   flops / ms * 1e-9 / 2500.0, hc, c);
        }
    }
    return 0;
}
