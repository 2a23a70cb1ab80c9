// tg_device.h -- device-side vocabulary of the Tangram MI355X kernels (gfx950 / CDNA4 only).
//
// Two build modes:
//   (default) hipcc --offload-arch=gfx950: the real thing, MFMA builtins, 64-wide wavefronts.
//   -DTG_SIM  host clang + tests/hipsim/hipsim.h: test-only emulation used by the CPU test-suite
//             (the authoring container has no GPU).  Nothing in the product loads a TG_SIM build.
#pragma once
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#ifdef TG_SIM
// ------------------------------------------------------------------------------------------
#include "hipsim.h"
#define TG_KERNEL
#define TG_DEV static inline
#define TG_DEVM inline
#define TG_HD static inline
#define TG_LAUNCH_BOUNDS(n)
#define TG_LAUNCH_BOUNDS2(n, w)
#define TG_GLOBAL
#define threadIdx (hipsim::M().cur->tid)
#define blockIdx (hipsim::M().blockIdx)
#define blockDim (hipsim::M().blockDim)
#define gridDim (hipsim::M().gridDim)
#define TG_LDS_DECL unsigned char* tg_lds = hipsim::M().lds
#define TG_SCHED_FENCE() ((void)0)
#define __syncthreads() hipsim::block_barrier()
#ifdef TG_SIM_HWMATH
// EXPERIMENT build of the emulator (scripts/exp_rounding_drift.py; never used by the test-suite): the transcendental helpers as
// the HARDWARE evaluates them -- __expf(x) = v_exp_f32(x * log2(e)) with the product rounded to fp32, __logf(x) =
// v_log_f32(x) * ln 2, and v_exp_f32 / v_log_f32 themselves as 1-ulp approximations (modelled: the exact value rounded TOWARDS
// ZERO instead of to nearest) -- to measure how much of the hardware-vs-emulator trajectory difference these roundings explain.
// (TG_SIM_HWMATH = 1: towards zero; 2: away from zero; 3: one of the two neighbours picked by a hash of the argument -- three
//  different 1-ulp-accurate "implementations" of the same functions)
TG_DEV float tg_hw_rtz(double v) {
    float f = (float)v;
    if (f == 0.f || !(fabs(v) < 1e38)) return f;
    const float lo = (fabs((double)f) > fabs(v)) ? nextafterf(f, 0.f) : f;                    // neighbour towards zero
    const float hi = (fabs((double)lo) < fabs(v)) ? nextafterf(lo, lo > 0.f ? 3e38f : -3e38f) : lo;   // neighbour away from zero
    if (TG_SIM_HWMATH == 1) return lo;
    if (TG_SIM_HWMATH == 2) return hi;
    unsigned long long b; memcpy(&b, &v, 8); b ^= b >> 29; b *= 0x9E3779B97F4A7C15ull; b ^= b >> 32;
    return (b & 1) ? lo : hi;
}
TG_DEV float tg_exp2(float x) { return tg_hw_rtz(exp2((double)x)); }
TG_DEV float tg_exp(float x) { return tg_exp2(x * 1.4426950408889634f); }
TG_DEV float tg_log(float x) { return tg_hw_rtz(log2((double)x)) * 0.6931471805599453f; }
#else
TG_DEV float tg_exp(float x) { return expf(x); }
TG_DEV float tg_log(float x) { return logf(x); }
TG_DEV float tg_exp2(float x) { return exp2f(x); }
#endif
TG_DEV float tg_shfl_xor(float v, int mask) { return hipsim::shfl_idx(v, hipsim::lane_id() ^ mask); }
TG_DEV float tg_bfly(float v, int mask) { return tg_shfl_xor(v, mask); }      // (see the HIP build below)
// Adam's square root and its two divisions as IEEE evaluates them (the HIP build's cheap forms are correctly / faithfully rounded)
TG_DEV float tg_sqrt_cr(float x) { return sqrtf(x); }
TG_DEV float tg_div_by(float s, float b, float inv_b) { (void)inv_b; return s / b; }
TG_DEV float tg_div_fr(float a, float b) { return a / b; }
TG_DEV int tg_lane() { return hipsim::lane_id(); }
TG_DEV int tg_uniform(int x) { return x; }
TG_DEV unsigned tg_pack_bf16(float lo, float hi) {
    return unsigned(hipsim::f32_to_bf16(lo)) | (unsigned(hipsim::f32_to_bf16(hi)) << 16);
}
TG_DEV f32x4 tg_mma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    float t[4] = {c[0], c[1], c[2], c[3]};
    hipsim::mfma16_bf16(&a, &b, t);
    return f32x4{t[0], t[1], t[2], t[3]};
}
TG_DEV f32x4 tg_mma_f32(float a, float b, f32x4 c) {
    float t[4] = {c[0], c[1], c[2], c[3]};
    hipsim::mfma16_f32(a, b, t);
    return f32x4{t[0], t[1], t[2], t[3]};
}
// global -> LDS DMA of 16 bytes per lane: LDS destination = wave-uniform base + lane * 16
TG_DEV void tg_glds16(const unsigned char* src, unsigned char* lds_wave_base) {
    memcpy(lds_wave_base + 16 * hipsim::lane_id(), src, 16);
}
TG_DEV void tg_glds16_uncounted(const unsigned char* src, unsigned char* lds_wave_base) { tg_glds16(src, lds_wave_base); }
TG_DEV void tg_dma_drain() {}
TG_DEV void tg_flag_or(int* p, int v) { *p |= v; }
// buffer-descriptor form of the copy (see the HIP build below): descriptor = {base pointer, byte count}
struct TgRsrc { const unsigned char* base; unsigned bytes; };
TG_DEV TgRsrc tg_make_rsrc(const unsigned char* base, size_t bytes) { return TgRsrc{base, (unsigned)bytes}; }
TG_DEV void tg_glds16_buf(const TgRsrc& r, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    tg_glds16(r.base + (size_t)voff + soff, lds_wave_base);
}
#else
// ------------------------------------------------------------------------------------------
#include <hip/hip_runtime.h>
#define TG_KERNEL __global__
#define TG_DEV __device__ __forceinline__
#define TG_DEVM __device__ __forceinline__
#define TG_HD __host__ __device__ static inline
#define TG_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define TG_LAUNCH_BOUNDS2(n, w) __launch_bounds__(n, w)
// pointers read out of an argument array in memory are generic; a cast to the global address space lets uniform loads through them
// become scalar loads (s_load) and the others global_load instead of flat_load
#define TG_GLOBAL __attribute__((address_space(1)))
// all LDS of a kernel lives in ONE dynamic array whose base is 16-byte aligned
// (cdna_hip_programming.md Guideline 17; a second __shared__ object de-pipelines, section 5 trap 4a)
#define TG_LDS_DECL extern __shared__ __attribute__((aligned(16))) unsigned char tg_lds[]
// pins the instruction order at this point (the machine scheduler otherwise sinks ds_reads next to their first use)
#define TG_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
typedef __bf16 tg_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tg_bf16x2 __attribute__((ext_vector_type(2)));
TG_DEV float tg_exp(float x) { return __expf(x); }
TG_DEV float tg_log(float x) { return __logf(x); }
TG_DEV float tg_exp2(float x) { return __builtin_amdgcn_exp2f(x); }     // v_exp_f32
TG_DEV float tg_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
// One step of a BUTTERFLY all-reduce over the wave (steps in the order 1, 2, 4, 8, 16, 32): the partner's value.  Masks 1 - 8 are DPP
// modifiers of a v_mov (no LDS crossbar trip like ds_bpermute): quad_perm for 1 and 2; row_half_mirror (lane i <-> 7 - i = i ^ 7) for
// 4 and row_mirror (i ^ 15) for 8 -- inside a butterfly the lanes of a quad / of an 8-group already agree when those steps run, so
// the value read equals the one at lane ^ 4 / lane ^ 8 (the emulator reads exactly those: same bits).  16 and 32: ds_bpermute.
TG_DEV float tg_bfly(float v, int mask) {
    const int x = __builtin_bit_cast(int, v);
    int y;
    switch (mask) {
        case 1: y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); break;     // quad_perm [1, 0, 3, 2]
        case 2: y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); break;     // quad_perm [2, 3, 0, 1]
        case 4: y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true); break;    // row_half_mirror
        case 8: y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, true); break;    // row_mirror
        default: return __shfl_xor(v, mask, 64);
    }
    return __builtin_bit_cast(float, y);
}
// ---- Adam's square root and divisions (torch _single_tensor_adam: denom = sqrt(v) / bias_correction2_sqrt + eps; m / denom) without
// hipcc's IEEE sequences (sqrtf: denormal scaling + v_sqrt_f32 + two-sided ulp fix-up, ~19 issue slots; a division: v_div_scale x 2,
// v_rcp_f32, four fma, v_div_fmas, v_div_fixup, ~13 slots -- 45 of the update kernel's 89 VALU slots per element in round 4).
// Measured against IEEE on 2^21 arguments over 66 decades (tests/test_gpu_parity.py::test_adam_square_root_and_divisions_against_ieee):
// each of the three returns the correctly rounded result in all but 1.4e-5 / 2.6e-5 / 1.8e-5 of the cases and its neighbour (1 ulp)
// otherwise -- a perturbation of the update of <= 6e-8 relative on one element in 50 000, far below the 1-ulp freedom of exp / log.
// tg_sqrt_cr: v_rsq_f32 + one coupled Goldschmidt / Newton step with fma residuals (the sequence LLVM's own lowering uses when it need
// not keep denormals).  Zero, denormal and infinite x are returned as they are: IEEE gives sqrt(x) <= 1.1e-19 for a denormal x, which
// vanishes against eps in `sqrt(v) / bc + eps` for any eps >= 1e-11 (Adam's is 1e-8): the denominator is the same float.
TG_DEV float tg_sqrt_cr(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    return __builtin_amdgcn_classf(x, 0x260 | 0x090) ? x : g;      // +-0 (0x060), +inf (0x200), +-denormal (0x090)
}
// s / b for a wave-uniform b with its correctly rounded reciprocal inv_b: quotient estimate + one fma-residual correction (Markstein).
TG_DEV float tg_div_by(float s, float b, float inv_b) {
    const float q = s * inv_b;
    return __builtin_fmaf(__builtin_fmaf(-q, b, s), inv_b, q);
}
// a / b, b normal and positive (Adam's denominator, >= eps): v_rcp_f32 (1 ulp) + one residual correction: faithfully rounded.
TG_DEV float tg_div_fr(float a, float b) {
    const float y = __builtin_amdgcn_rcpf(b);
    const float q = a * y;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), y, q);
}
TG_DEV int tg_lane() { return threadIdx.x & 63; }
TG_DEV int tg_uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
TG_DEV unsigned tg_pack_bf16(float lo, float hi) {        // -> v_cvt_pk_bf16_f32 (RNE)
    tg_bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
// v_mfma_f32_16x16x32_bf16: lane l holds A[row=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][col=l&15];
// D: lane l, reg r -> row = 4*(l>>4)+r, col = l&15.
TG_DEV f32x4 tg_mma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tg_bf16x8, a), __builtin_bit_cast(tg_bf16x8, b),
                                                   c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: lane l holds A[row=l&15][k=l>>4], B[k=l>>4][col=l&15]; exact f32 fmaf chain.
TG_DEV f32x4 tg_mma_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// global_load_lds_dwordx4: asynchronous global -> LDS copy that bypasses the VGPRs; the LDS destination is
// M0 (wave-uniform base) + lane * 16, the global source address is per lane.  Completion is tracked by vmcnt;
// hipcc drains it before the next __syncthreads().
TG_DEV void tg_glds16(const unsigned char* src, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// The same copy issued from an asm statement, i.e. OUTSIDE hipcc's wait-count bookkeeping.  hipcc books the builtin as a
// flat access that may touch LDS, and from then on every wait it inserts for a ds_read is lgkmcnt(0) -- in the GEMM main
// loops: a full LDS drain in front of every MFMA group instead of the counted lgkmcnt(2 - 3) it emits without a copy in
// flight (cdna_hip_programming.md section 5.7).  The price: nothing waits for these copies but tg_dma_drain().
TG_DEV void tg_glds16_uncounted(const unsigned char* src, unsigned char* lds_wave_base) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// vmcnt(0) as the BUILTIN (simm16 0x0F70 = vmcnt 0, expcnt 7, lgkmcnt 15): hipcc keeps a user wait as it is and books it, so
// its own vector loads count as complete afterwards (an asm wait would leave them pending on some paths of its scoreboard,
// and the next write to one of their registers then gets a vmcnt(0) of its own in the middle of the MFMA stream).
TG_DEV void tg_dma_drain() { __builtin_amdgcn_s_waitcnt(0x0F70); }
TG_DEV void tg_flag_or(int* p, int v) { atomicOr(p, v); }
// The copy through a BUFFER DESCRIPTOR: buffer_load_dwordx4 ... offen lds.  Source = descriptor base (SGPRs) + this lane's byte
// offset (one VGPR, fixed for the whole tile) + soffset (an SGPR: the contraction step).  Against the global_load_lds form (64-bit
// lane addresses rebuilt by two VALU instructions per copy and step) the loop carries no address arithmetic at all, and the copy
// itself is cheaper to issue: the backward main loop -2.3 % (scripts/probes/gemm_loop_lab.hip, VAR 13 vs 0, profiles/r04/lab).
// Out-of-range offsets cannot fault: the hardware clamps buffer accesses to the descriptor's byte count.
typedef u32x4 TgRsrc;
TG_DEV TgRsrc tg_make_rsrc(const unsigned char* base, size_t bytes) {
    const unsigned long long b = (unsigned long long)base;
    TgRsrc r = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, (unsigned)bytes, 0x00020000u};
    r[0] = __builtin_amdgcn_readfirstlane(r[0]); r[1] = __builtin_amdgcn_readfirstlane(r[1]);
    r[2] = __builtin_amdgcn_readfirstlane(r[2]); r[3] = __builtin_amdgcn_readfirstlane(r[3]);
    return r;
}
TG_DEV void tg_glds16_buf(const TgRsrc& r, unsigned voff, unsigned soff, unsigned char* lds_wave_base) {
    const unsigned dst = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds_wave_base;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(r), "s"(dst), "s"(soff) : "memory");
}
#endif

TG_DEV float tg_bf16_lo_to_f32(unsigned packed) { return __builtin_bit_cast(float, packed << 16); }
TG_DEV float tg_bf16_hi_to_f32(unsigned packed) { return __builtin_bit_cast(float, packed & 0xffff0000u); }
TG_DEV float tg_fmax(float a, float b) { return a > b ? a : b; }
#define TG_LOG2E 1.4426950408889634f

// ----------------------------------------------------------------------------------------------
// GEMM operand precision policies.
//
// Operand format (both in HBM and in LDS): an operand row is a sequence of contraction STEPS of 128 bytes
// = 8 chunks of 16 bytes; a chunk is what one lane feeds to the matrix core (8 bf16 or 4 f32 along the
// contraction axis).  Lane (row = l&15, g = l>>4) reads chunk 4*(q+p)+g; the order of the contraction
// index inside a step is irrelevant as long as A and B agree, which they do.
//   PrecF32   : exact f32 MFMA (v_mfma_f32_16x16x4_f32).  step = 32 elements, chunks 0..7 = 8 k-chunks
//   PrecBF16  : operands rounded to bf16, f32 accumulate.  step = 64 elements, chunks 0..7 = 8 k-chunks
//   PrecBF16x3: split bf16: x = hi + lo, a*b ~ ah*bh + ah*bl + al*bh (f32 accumulate) -- fp32-parity.
//               step = 32 elements: chunks 0..3 = hi of the 4 k-chunks, chunks 4..7 = their lo parts.
//   KQ = k-chunk groups per step a lane walks through, NP = parts (hi, lo) per fragment.
// ----------------------------------------------------------------------------------------------
struct PrecF32 {
    static constexpr bool X16 = false;      // backward product X kept in fp32
    static constexpr int kId = 0, KQ = 2, NP = 1, CH = 4, BKE = 32, ESZ = 4, KCH = 8;
    static constexpr int NPB = NP, BRC = 8;     // parts of a B fragment; 16-byte chunks per B tile row (see PrecBF16x2S)
    TG_DEVM static void cvt(const float (&x)[4], u32x4& hi, u32x4& lo) {
        hi = u32x4{__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1]),
                   __builtin_bit_cast(unsigned, x[2]), __builtin_bit_cast(unsigned, x[3])};
        lo = hi;
    }
    TG_DEVM static f32x4 mma(const u32x4* a, const u32x4* b, f32x4 c) {
        const f32x4 af = __builtin_bit_cast(f32x4, a[0]), bf = __builtin_bit_cast(f32x4, b[0]);
        c = tg_mma_f32(af[0], bf[0], c);
        c = tg_mma_f32(af[1], bf[1], c);
        c = tg_mma_f32(af[2], bf[2], c);
        c = tg_mma_f32(af[3], bf[3], c);
        return c;
    }
};

struct PrecBF16 {
    static constexpr bool X16 = true;       // X = S dGhat^T is built from bf16 operands: keeping it in bf16 adds no new error class
    static constexpr int kId = 1, KQ = 2, NP = 1, CH = 8, BKE = 64, ESZ = 2, KCH = 8;
    static constexpr int NPB = NP, BRC = 8;
    TG_DEVM static void cvt(const float (&x)[8], u32x4& hi, u32x4& lo) {
        hi = u32x4{tg_pack_bf16(x[0], x[1]), tg_pack_bf16(x[2], x[3]), tg_pack_bf16(x[4], x[5]), tg_pack_bf16(x[6], x[7])};
        lo = hi;
    }
    TG_DEVM static f32x4 mma(const u32x4* a, const u32x4* b, f32x4 c) { return tg_mma_bf16(a[0], b[0], c); }
};

struct PrecBF16x3 {
    static constexpr bool X16 = false;
    static constexpr int kId = 2, KQ = 1, NP = 2, CH = 8, BKE = 32, ESZ = 4, KCH = 4;   // ESZ: bytes per element incl. lo
    static constexpr int NPB = NP, BRC = 8;
    TG_DEVM static void cvt(const float (&x)[8], u32x4& hi, u32x4& lo) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned p = tg_pack_bf16(x[2 * j], x[2 * j + 1]);
            hi[j] = p;
            r[2 * j] = x[2 * j] - tg_bf16_lo_to_f32(p);
            r[2 * j + 1] = x[2 * j + 1] - tg_bf16_hi_to_f32(p);
        }
        lo = u32x4{tg_pack_bf16(r[0], r[1]), tg_pack_bf16(r[2], r[3]), tg_pack_bf16(r[4], r[5]), tg_pack_bf16(r[6], r[7])};
    }
    TG_DEVM static f32x4 mma(const u32x4* a, const u32x4* b, f32x4 c) {
        c = tg_mma_bf16(a[1], b[0], c);      // small terms first
        c = tg_mma_bf16(a[0], b[1], c);
        c = tg_mma_bf16(a[0], b[0], c);
        return c;
    }
};

// PrecBF16x2S: split bf16 with an EXACT B operand.  In both GEMMs of the iteration the B operand is the constant S (forward:
// Ghat = P^T S, backward: X = S dGhat^T).  When every element of S (and of the augmentation columns riding with it) is exactly
// representable in bf16 -- raw counts below 256, one-hot cell types, the uniform density column -- its lo part is identically
// zero and the product a_hi * b_lo of the split-bf16 scheme adds exact zeros: it is skipped, a*b = a_lo*b_hi + a_hi*b_hi, the
// SAME value as PrecBF16x3 delivers (bit for bit up to the sign of an exact zero), with 2 matrix-core products instead of 3.
// The S images then hold hi parts only: B tile rows of 64 bytes (BRC = 4 chunks) instead of 128, half the copy traffic of S.
// The A operand (softmax(M) / dGhat) keeps the bf16x3 format.  Chosen by the library when it is asked to check S and finds it exact
// (tg_config.s_exact_mode = 1, tg_mapper_create); never requested by a caller directly.
struct PrecBF16x2S : PrecBF16x3 {
    static constexpr int kId = 3, NPB = 1, BRC = 4;
    TG_DEVM static f32x4 mma(const u32x4* a, const u32x4* b, f32x4 c) {
        c = tg_mma_bf16(a[1], b[0], c);      // small term first, like PrecBF16x3
        c = tg_mma_bf16(a[0], b[0], c);
        return c;
    }
};
// XOR swizzle of a COMPACT (64-byte, 4-chunk) B tile row: the 16-lane groups of a ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...)
// meet rows r, r+4, r+8, r+12 at chunk g or g^1 -- {0, 3, 2, 1}[(row >> 2) & 3] sends them to four different chunks, i.e. a
// group covers all 16 slots of the 256-byte bank row.  (Depends on (row >> 2) & 3 only: the same for every 16-row fragment.)
TG_DEV int tg_swz4(int row, int chunk) { return chunk ^ ((4 - ((row >> 2) & 3)) & 3); }

// Write the operand image of CH consecutive contraction elements (k-chunk `kc` of step `step`) of one operand row.
template <class PR>
TG_DEV void tg_store_operand_chunk(unsigned char* row_base, int step, int kc, const float (&x)[PR::CH]) {
    u32x4 hi, lo;
    PR::cvt(x, hi, lo);
    u32x4* dst = (u32x4*)(row_base + (size_t)step * 128);
    dst[kc] = hi;
    if (PR::NP == 2) dst[4 + kc] = lo;
}
// ... of the B operand S: the same image, or hi parts only in rows of BRC * 16 bytes per step (PrecBF16x2S)
template <class PR>
TG_DEV void tg_store_s_chunk(unsigned char* row_base, int step, int kc, const float (&x)[PR::CH]) {
    if constexpr (PR::BRC == 8) tg_store_operand_chunk<PR>(row_base, step, kc, x);
    else {
        u32x4 hi, lo;
        PR::cvt(x, hi, lo);
        ((u32x4*)(row_base + (size_t)step * (PR::BRC * 16)))[kc] = hi;
    }
}

// XOR swizzle of the 16-byte chunk index inside a 128-byte LDS tile row: ds_read_b128 of 16
// consecutive rows at one logical chunk touches all 16 slots of the two 256-byte bank rows.
// The extra (row >> 4) & 1 term is constant inside a 16-row fragment (reads unaffected) and makes the forward
// kernel's transposed ds_write_b128 (rows 4q + i of 8 consecutive lanes) conflict-free as well.
TG_DEV int tg_swz(int row, int chunk) { return chunk ^ (((row >> 1) & 7) ^ ((row >> 4) & 1)); }
