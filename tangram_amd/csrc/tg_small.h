// tg_small.h -- clusters mode (<= 32 rows of M): exact-fp32 matrix-core kernels that recompute P^T S instead of storing it.
// Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// Small-C path (clusters mode: C <= 32 "cells", e.g. 18 clusters x 250 genes x 9 852 spots, the unit of the reference's
// cross-validation, utils.py:576-600).  With so few rows the 128- / 256-wide GEMM tiles multiply mostly padding and the iteration
// is bound by its passes over the spot x gene matrices (Ghat partials, Ghat, dGhat image: ~80 MB per iteration at that shape).
// Here the contraction over the C clusters is 18 multiply-adds per element, so Ghat is RECOMPUTED where it is needed, not stored:
//   tg_sc_forward   per block of 64 spots x 256 genes, a wave per 64 genes: Ghat^T tiles (16 genes x 16 spots) on the matrix cores in
//                   exact fp32 (v_mfma_f32_16x16x4_f32; P from M through LDS, S from pre-laid-out operand images), G through an LDS transpose (one read
//                   of G), per-gene cosine partials, the density column, optionally the per-spot sums
//   tg_sc_backward  the same Ghat^T tiles again, dGhat = (alpha_k + va_v) G + (beta_k + vb_v) Ghat (second read of G) in the
//                   accumulator registers, which ARE the B operand of X_cv += sum_k S_ck dGhat_vk: no dGhat tile, no softmax image
// Two reads of G per iteration (20 MB at that shape) instead of ~80 MB; exact fp32 products whatever the GEMM precision of
// the handle (documented in DESIGN.md).  tg_gene_reduce, tg_adam_rowpass (+ the filter kernels) complete the iteration unchanged.
// ----------------------------------------------------------------------------------------------
#define TG_SC_MAXC 32
#define TG_SC_SB 64            // spots per block: four spot tiles of 16
#define TG_SC_KC 256           // genes per block and chunk: 64 per wave ...
#define TG_SC_KS 32            // ... staged through LDS 32 at a time
#define TG_SC_TILE (TG_SC_KS * TG_SC_SB)          // floats of one wave's G sub-tile, [gene quad][spot][4]
struct TgSmallArgs {
    const float* M; const float* rmax; const float* rmul;    // logits [C][Vp]; forward row constants (P f = exp2((M - max) log2e) * rmul)
    const float* Sa;           // S (with the augmentation column K, zero for c >= C and beyond K) in the operand layouts of the kernels,
    const float* Sx;           //   one contiguous run per (64 genes, lane): tg_prep_ssmall, tg_sc_load_ops
    const float* G;            // [Vr][Kp] fp32, zero padded
    float* Ghat;               // [Vr][Kp]: only the density column K is written (colsum_v)
    float* genepart;           // [spot blocks][2][Kp]
    float* voxstat;            // [chunks][2][Vr] when want_vox
    float* X;                  // [C][Vp] fp32 (backward)
    int C, CM, V, Vp, Vr, Kp, K, want_vox;      // CM: C rounded up to the kernels' compile-time cluster bound
    TgFinalizeArgs fin;        // backward: the reduced statistics the gradient coefficients are derived from
};
TG_HD int tg_sc_cm(int C) { return (C + 3) / 4 * 4; }
#define TG_SC_PP 80            // row pitch (floats) of the P tile in LDS [cluster][spot]: rows 16 banks apart
TG_HD int tg_sc_lds_fwd() { return (4 * TG_SC_TILE + TG_SC_MAXC * TG_SC_PP + 4 * 2 * TG_SC_SB) * 4; }
TG_HD int tg_sc_lds_bwd() { return (4 * TG_SC_TILE + TG_SC_MAXC * TG_SC_PP + 4 * 64 * 4 + 2 * TG_SC_SB) * 4; }

// Matrix-core layout of the small-C kernels (v_mfma_f32_16x16x4_f32, exact fp32 products): lane = (grp = lane / 16, ln = lane % 16).
// A tile of Ghat^T, 16 genes x 16 spots, is  sum_c St[gene][c] P[c][spot]:  A operand a_j = St[kt + ln][4 j + grp], B operand
// b_j = P[4 j + grp][spot ln], CM / 4 instructions; the result leaves lane (grp, ln) with spot ln and the FOUR CONSECUTIVE genes
// kt + 4 grp + r -- which is (i) a b128 read of the G tile staged [gene quad][spot][4] and (ii) exactly the B operand layout of the
// next product X[c][spot] += sum_genes S[c][gene] dGhat[gene][spot] (instruction r contracts genes kt + 4 grp' + r, grp' = 0..3),
// so dGhat never leaves the registers.  (Scalar-operand FMA versions of these loops were bound by the scalar cache: S is 20 KB.)
// P f of the block's 64 spots -> LDS [cluster][spot], ONCE per block: wave w takes clusters w, w + 4, ... (the row constants
// are wave-uniform, the logits a coalesced row segment); every wave then reads its B operands pb[st][j] = P[4 j + grp][16 st + ln]
template <int CM>
TG_DEV void tg_sc_p_tile(const TgSmallArgs& a, int v0, int wave, int lane, float* Pt) {
    const TG_GLOBAL float* M = (const TG_GLOBAL float*)a.M;
    const TG_GLOBAL float* rmax = (const TG_GLOBAL float*)a.rmax;
    const TG_GLOBAL float* rmul = (const TG_GLOBAL float*)a.rmul;
    const int v = v0 + lane;
#pragma unroll
    for (int j = 0; j < CM / 4; ++j) {
        const int c = wave + 4 * j;
        Pt[c * TG_SC_PP + lane] = (c < a.C && v < a.V) ? tg_exp2((M[(size_t)c * a.Vp + v] - rmax[c]) * TG_LOG2E) * rmul[c] : 0.f;
    }
}
template <int CM>
TG_DEV void tg_sc_p_operands(const float* Pt, int lane, float (&pb)[4][CM / 4]) {
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int j = 0; j < CM / 4; ++j) pb[st][j] = Pt[(4 * j + (lane >> 4)) * TG_SC_PP + 16 * st + (lane & 15)];
}
// S operands of a wave's 64 genes (four gene tiles gt), read as 16-byte pieces of the lane's contiguous run:
//   sa[gt][j]     = S[4 j + grp][kw + 16 gt + ln]             A operand of Ghat^T
//   sx[gt][r][cb] = S[16 cb + ln][kw + 16 gt + 4 grp + r]     A operand of the X product (backward only)
template <int CM, bool WITH_X>
struct TgScOps {
    float sa[4][CM / 4];
    float sx[4][4][WITH_X ? (CM + 15) / 16 : 1];
};
template <int CM, bool WITH_X>
TG_DEV void tg_sc_load_ops(const TG_GLOBAL float* Sa, const TG_GLOBAL float* Sx, int kw, int lane, TgScOps<CM, WITH_X>& o) {
    constexpr int NA = CM, NX = 16 * ((CM + 15) / 16);          // floats per (64 genes, lane)
    const TG_GLOBAL f32x4* pa = (const TG_GLOBAL f32x4*)(Sa + ((size_t)(kw >> 6) * 64 + lane) * NA);
    float* fa = &o.sa[0][0];
#pragma unroll
    for (int i = 0; i < NA / 4; ++i) { const f32x4 q = pa[i]; fa[4 * i] = q[0]; fa[4 * i + 1] = q[1]; fa[4 * i + 2] = q[2]; fa[4 * i + 3] = q[3]; }
    if constexpr (WITH_X) {
        const TG_GLOBAL f32x4* px = (const TG_GLOBAL f32x4*)(Sx + ((size_t)(kw >> 6) * 64 + lane) * NX);
        float* fx = &o.sx[0][0][0];
#pragma unroll
        for (int i = 0; i < NX / 4; ++i) { const f32x4 q = px[i]; fx[4 * i] = q[0]; fx[4 * i + 1] = q[1]; fx[4 * i + 2] = q[2]; fx[4 * i + 3] = q[3]; }
    }
}
template <int CM>
TG_DEV f32x4 tg_sc_ghat_tile(const float (&sa)[CM / 4], const float (&pb)[CM / 4]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < CM / 4; ++j) acc = tg_mma_f32(sa[j], pb[j], acc);
    return acc;
}
// this wave's next 32 genes of G for the block's 64 spots -> registers -> LDS [gene quad][spot][4]: 8 consecutive lanes load the
// 128 contiguous bytes of a row; conflict-free b128 writes and reads through the slot swizzle.  Split in two
// so that the loads of the NEXT sub-tile fly while the matrix cores work on the current one.
TG_DEV void tg_sc_load_g(const TG_GLOBAL float* G, int Kp, int v0, int k0, int lane, f32x4 (&g)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) g[i] = *(const TG_GLOBAL f32x4*)(G + (size_t)(v0 + (lane >> 3) + 8 * i) * Kp + k0 + 4 * (lane & 7));
}
// slot of (gene quad q, spot sp) in a wave's tile, in 16-byte units: the XOR spreads the 8 quads of a row over the 8 bank groups
TG_DEV int tg_sc_slot(int q, int sp) { return q * TG_SC_SB + (sp ^ q); }
TG_DEV void tg_sc_store_g(const f32x4 (&g)[8], int lane, float* Gt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) *(f32x4*)(Gt + tg_sc_slot(lane & 7, (lane >> 3) + 8 * i) * 4) = g[i];
}
// sums of N per-lane values over groups of G consecutive lanes, all at once: each of the log2(G) steps halves the values a lane
// carries (a lane keeps the half its bit selects and hands the other half to its partner at lane distance G/2, G/4, ... 1);
// lane L of a group ends with the sums of x[(N / G) L + i] in x[i], i < N / G
template <int H, int M, int N>
TG_DEV void tg_group_sum_step(float (&x)[N], int lane) {
    const bool up = (lane & M) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        const float keep = up ? x[i + H] : x[i], send = up ? x[i] : x[i + H];
        x[i] = keep + tg_shfl_xor(send, M);
    }
    if constexpr (M > 1) tg_group_sum_step<H / 2, M / 2, N>(x, lane);
}
template <int G, int N>
TG_DEV void tg_group_sum_to_lanes(float (&x)[N], int lane) {
    static_assert(N >= G && N % G == 0 && (N & (N - 1)) == 0 && (G & (G - 1)) == 0 && G <= 64, "powers of two, at least one value per lane");
    tg_group_sum_step<N / 2, G / 2, N>(x, lane);
}

// grid (blocks of 64 spots, gene chunks [, mappings]); wave w: genes 64 w .. 64 w + 63 of the chunk as four tiles of 16.
// Every load the first sub-tile needs is requested before anything is waited for; the second sub-tile's loads fly under the first's MFMAs.
template <int CM, bool VOX>
TG_DEV void tg_sc_forward_body(const TgSmallArgs& a) {
    constexpr int NJ = CM / 4;
    TG_LDS_DECL;
    const int t = threadIdx.x, lane = t & 63, wave = tg_uniform(t >> 6), grp = lane >> 4, ln = lane & 15;
    float* Gt = (float*)tg_lds + wave * TG_SC_TILE;
    float* Pt = (float*)tg_lds + 4 * TG_SC_TILE;                 // [CM][PP]
    float* red = Pt + TG_SC_MAXC * TG_SC_PP;                     // VOX: [4 waves][2][SB]
    const int v0 = blockIdx.x * TG_SC_SB, kw = blockIdx.y * TG_SC_KC + 64 * wave;
    const TG_GLOBAL float* G = (const TG_GLOBAL float*)a.G;
    const bool live0 = kw < a.Kp, live1 = kw + TG_SC_KS < a.Kp;  // (Kp is a multiple of 128: a sub-tile is inside or outside as a whole)
    f32x4 greg[8];
    TgScOps<CM, false> ops;
    if (live0) { tg_sc_load_g(G, a.Kp, v0, kw, lane, greg); tg_sc_load_ops<CM, false>((const TG_GLOBAL float*)a.Sa, nullptr, kw, lane, ops); }
    tg_sc_p_tile<CM>(a, v0, wave, lane, Pt);
    const int dK = a.K - kw;                                     // the density column K, if this wave has it: tile, lane group, register
    const int gtK = (dK >= 0 && dK < 64) ? dK >> 4 : -1, grpK = (dK & 15) >> 2, rK = dK & 3;
    float x[32];                                                 // [statistic][gene tile][r]: sums over the lane's four spots
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = 0.f;
    float vd[4] = {0.f, 0.f, 0.f, 0.f}, vn[4] = {0.f, 0.f, 0.f, 0.f}, colv[4] = {0.f, 0.f, 0.f, 0.f};
    if (live0) tg_sc_store_g(greg, lane, Gt);
    __syncthreads();
    float pb[4][NJ];
    tg_sc_p_operands<CM>(Pt, lane, pb);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const bool live = sub == 0 ? live0 : live1;
        if (sub == 0 && live1) tg_sc_load_g(G, a.Kp, v0, kw + TG_SC_KS, lane, greg);
        if (live) {
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int gt = 2 * sub + g2, kt = kw + 16 * gt;
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const f32x4 gh = tg_sc_ghat_tile<CM>(ops.sa[gt], pb[st]);
                    const f32x4 g4 = *(const f32x4*)(Gt + tg_sc_slot(4 * g2 + grp, 16 * st + ln) * 4);   // rows beyond V, columns beyond K of G are zero
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x[4 * gt + r] += gh[r] * g4[r];
                        x[16 + 4 * gt + r] += gh[r] * gh[r];
                        if constexpr (VOX) { vd[st] += gh[r] * g4[r]; vn[st] += (kt + 4 * grp + r < a.K) ? gh[r] * gh[r] : 0.f; }
                    }
                    if (gt == gtK) colv[st] = rK == 0 ? gh[0] : (rK == 1 ? gh[1] : (rK == 2 ? gh[2] : gh[3]));   // colsum_v (density term, :217)
                }
            }
        }
        if (sub == 0) {
            __syncthreads();
            if (live1) tg_sc_store_g(greg, lane, Gt);
            __syncthreads();
        }
    }
    // sums over the block's 64 spots: the 16 lanes of a group; lane ln ends with x[2 ln], x[2 ln + 1]
    tg_group_sum_to_lanes<16>(x, lane);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = 2 * ln + i, k = kw + 16 * ((idx & 15) >> 2) + 4 * grp + (idx & 3);
        if (k < a.Kp) a.genepart[((size_t)blockIdx.x * 2 + (idx >> 4)) * a.Kp + k] = x[i];
    }
    if (gtK >= 0 && grp == grpK)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int v = v0 + 16 * st + ln;
            if (v < a.V) a.Ghat[(size_t)v * a.Kp + a.K] = colv[st];
        }
    if constexpr (VOX) {                 // per-spot sums over this chunk's genes: the four lane groups, then the four waves in fixed order
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            vd[st] += tg_shfl_xor(vd[st], 16); vd[st] += tg_shfl_xor(vd[st], 32);
            vn[st] += tg_shfl_xor(vn[st], 16); vn[st] += tg_shfl_xor(vn[st], 32);
            if (grp == 0) { red[(wave * 2 + 0) * TG_SC_SB + 16 * st + ln] = vd[st]; red[(wave * 2 + 1) * TG_SC_SB + 16 * st + ln] = vn[st]; }
        }
        __syncthreads();
        if (t < TG_SC_SB && v0 + t < a.V) {
            float d = 0.f, n = 0.f;
            for (int w = 0; w < 4; ++w) { d += red[(w * 2 + 0) * TG_SC_SB + t]; n += red[(w * 2 + 1) * TG_SC_SB + t]; }
            a.voxstat[((size_t)blockIdx.y * 2 + 0) * a.Vr + v0 + t] = d;
            a.voxstat[((size_t)blockIdx.y * 2 + 1) * a.Vr + v0 + t] = n;
        }
    }
}

// gene coefficients (alpha, beta, mask) of gene k for the backward kernel
TG_DEV f32x4 tg_sc_gene_coef4(const TgSmallArgs& a, int k) {
    float al = 0.f, be = 0.f, cc = 0.f;
    if (k < a.K) tg_gene_coef(a.fin, a.fin.genestat, a.fin.gnorm2, a.fin.lambda_g1, k, al, be, cc);
    return f32x4{al, be, k < a.K ? 1.f : 0.f, 0.f};             // augmentation / padding columns carry no gradient
}

// grid (blocks of 64 spots [, 1, mappings]); the gene sub-tiles (32 genes per wave) are a loop: X_cv is a sum over all genes.
// Software pipeline: the G rows, S operands and gene coefficients of sub-tile s + 1 are requested before the MFMAs of sub-tile s.
template <int CM>
TG_DEV void tg_sc_backward_body(const TgSmallArgs& a) {
    constexpr int NJ = CM / 4, NCB = (CM + 15) / 16;
    TG_LDS_DECL;
    const int t = threadIdx.x, lane = t & 63, wave = tg_uniform(t >> 6), grp = lane >> 4, ln = lane & 15;
    float* Gt = (float*)tg_lds + wave * TG_SC_TILE;
    float* coef = (float*)tg_lds + 4 * TG_SC_TILE + wave * 64 * 4;      // [64 genes of the wave][alpha, beta, mask, -]
    float* cs = (float*)tg_lds + 4 * TG_SC_TILE + 4 * 64 * 4;           // [2][SB] va, vb
    float* Pt = cs + 2 * TG_SC_SB;                                      // [CM][PP]
    const int v0 = blockIdx.x * TG_SC_SB;
    const TG_GLOBAL float* Sa = (const TG_GLOBAL float*)a.Sa;
    const TG_GLOBAL float* Sx = (const TG_GLOBAL float*)a.Sx;
    const TG_GLOBAL float* G = (const TG_GLOBAL float*)a.G;
    const int nsub = 2 * ((a.Kp + TG_SC_KC - 1) / TG_SC_KC);
    f32x4 greg[8];
    TgScOps<CM, true> ops;
    bool live = 64 * wave < a.Kp;
    if (live) { tg_sc_load_g(G, a.Kp, v0, 64 * wave, lane, greg); tg_sc_load_ops<CM, true>(Sa, Sx, 64 * wave, lane, ops); }
    f32x4 cval = tg_sc_gene_coef4(a, 64 * wave + lane);
    float va_t = 0.f, vb_t = 0.f, av_t = 0.f;
    if (t < TG_SC_SB) {                                          // per-spot coefficients, like tg_dghat_emit<SELF>
        const int v = v0 + t;
        float c = 0.f, kl = 0.f;
        if (v < a.V) {
            float dot, n2a, n2b, colsum, dv;
            tg_spot_stats_load(a.fin, v, dot, n2a, n2b, colsum, dv);
            const float rho_scale = a.fin.fsum_dev ? 1.f / a.fin.fsum_dev[0] : a.fin.rho_scale;
            tg_spot_coef(a.fin, dot, n2a, n2b, colsum, dv, rho_scale, va_t, vb_t, av_t, c, kl);
        }
        cs[t] = va_t; cs[TG_SC_SB + t] = vb_t;
        // the spots' terms of the loss (voxel cosine, KL) summed over the block: the history workgroup adds the blocks up
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { c += tg_shfl_xor(c, m); kl += tg_shfl_xor(kl, m); }
        if (t == 0) { a.fin.spotpart[2 * blockIdx.x] = c; a.fin.spotpart[2 * blockIdx.x + 1] = kl; }
    }
    tg_sc_p_tile<CM>(a, v0, wave, lane, Pt);
    *(f32x4*)(coef + lane * 4) = cval;
    if (live) tg_sc_store_g(greg, lane, Gt);
    __syncthreads();
    float pb[4][NJ];
    tg_sc_p_operands<CM>(Pt, lane, pb);
    float va[4], vb[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) { va[st] = cs[16 * st + ln]; vb[st] = cs[TG_SC_SB + 16 * st + ln]; }
    f32x4 xacc[4][NCB];                                          // X[16 cb + 4 grp + i][spot 16 st + ln] over this wave's genes
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) xacc[st][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s2 = 0; s2 < nsub; s2 += 2) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int s = s2 + sub;
            const int kn = ((s + 1) >> 1) * TG_SC_KC + 64 * wave + TG_SC_KS * ((s + 1) & 1);     // first gene of the next sub-tile
            const bool nlive = s + 1 < nsub && kn < a.Kp, nchunk = sub == 1 && s + 1 < nsub;
            if (nlive) tg_sc_load_g(G, a.Kp, v0, kn, lane, greg);
            if (nchunk) cval = tg_sc_gene_coef4(a, kn + lane);
            if (live) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int gt = 2 * sub + g2;
                    f32x4 cf[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) cf[r] = *(const f32x4*)(coef + (16 * gt + 4 * grp + r) * 4);
#pragma unroll
                    for (int st = 0; st < 4; ++st) {
                        const f32x4 gh = tg_sc_ghat_tile<CM>(ops.sa[gt], pb[st]);
                        const f32x4 g4 = *(const f32x4*)(Gt + tg_sc_slot(4 * g2 + grp, 16 * st + ln) * 4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // (alpha_k + va_v) G + (beta_k + vb_v) Ghat (tg_dghat_emit)
                            const float d = ((cf[r][0] + va[st]) * g4[r] + (cf[r][1] + vb[st]) * gh[r]) * cf[r][2];
#pragma unroll
                            for (int cb = 0; cb < NCB; ++cb) xacc[st][cb] = tg_mma_f32(ops.sx[gt][r][cb], d, xacc[st][cb]);
                        }
                    }
                }
            }
            __syncthreads();                                     // every read of the G tile and the coefficient table is done
            if (nlive) tg_sc_store_g(greg, lane, Gt);
            if (nlive && sub == 1) tg_sc_load_ops<CM, true>(Sa, Sx, kn, lane, ops);     // (the operands cover the wave's 64 genes: once per chunk)
            if (nchunk) *(f32x4*)(coef + lane * 4) = cval;
            live = nlive;
            __syncthreads();
        }
    }
    constexpr int CR = 16 * NCB;
    float* red = (float*)tg_lds;                                 // [4 waves][CR][SB] over the G tiles (all reads of them are behind the barrier)
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[((wave * CR) + 16 * cb + 4 * grp + i) * TG_SC_SB + 16 * st + ln] = xacc[st][cb][i];
    __syncthreads();
    for (int o = t; o < CR * TG_SC_SB; o += TG_SC_KC) {
        const int c = o / TG_SC_SB, vv = o % TG_SC_SB;
        float x = 0.f;
        for (int w = 0; w < 4; ++w) x += red[(w * CR + c) * TG_SC_SB + vv];
        if (c < a.C && v0 + vv < a.V) a.X[(size_t)c * a.Vp + v0 + vv] = x;
    }
    if (t < TG_SC_SB && v0 + t < a.Vr) { a.fin.vcoef[v0 + t] = va_t; a.fin.vcoef[a.Vr + v0 + t] = vb_t; a.fin.vcoef[2 * a.Vr + v0 + t] = av_t; }   // a_v: read by the update kernel
}

template <int CM, bool VOX> TG_KERNEL void TG_LAUNCH_BOUNDS2(TG_SC_KC, 2) tg_sc_forward(TgSmallArgs a) { tg_sc_forward_body<CM, VOX>(a); }
template <int CM, bool VOX> TG_KERNEL void TG_LAUNCH_BOUNDS2(TG_SC_KC, 2) tg_sc_forward_b(const TgSmallArgs* argv) { const TgSmallArgs a = argv[blockIdx.z]; tg_sc_forward_body<CM, VOX>(a); }
template <int CM> TG_KERNEL void TG_LAUNCH_BOUNDS2(TG_SC_KC, 2) tg_sc_backward(TgSmallArgs a) { tg_sc_backward_body<CM>(a); }
template <int CM> TG_KERNEL void TG_LAUNCH_BOUNDS2(TG_SC_KC, 2) tg_sc_backward_b(const TgSmallArgs* argv) { const TgSmallArgs a = argv[blockIdx.z]; tg_sc_backward_body<CM>(a); }

// S with the augmentation column (k == K: ones or d_source; 0 beyond and for c >= C) in the operand layouts of tg_sc_load_ops:
// one thread per (64-gene block kb, lane, i): Sa[(kb 64 + lane) CM + i], i = gt CM/4 + j;  Sx[(kb 64 + lane) NX + i], i = (gt 4 + r) NCB + cb
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_prep_ssmall(const float* S, long long ldS, const float* aug, int C, int CM, int K, int Kp, float* Sa, float* Sx) {
    const int NCB = (CM + 15) / 16, NX = 16 * NCB, NJ = CM / 4;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (Kp / 64) * 64 * NX) return;
    const int i = idx % NX, lane = (idx / NX) % 64, kb = idx / (NX * 64), grp = lane >> 4, ln = lane & 15;
    auto sval = [&](int c, int k) { return (c >= C || k >= Kp) ? 0.f : (k < K ? S[(size_t)c * ldS + k] : (k == K ? (aug ? aug[c] : 1.f) : 0.f)); };
    if (i < CM) { const int gt = i / NJ, j = i % NJ; Sa[((size_t)kb * 64 + lane) * CM + i] = sval(4 * j + grp, 64 * kb + 16 * gt + ln); }
    const int cb = i % NCB, r = (i / NCB) % 4, gt = i / (4 * NCB);
    Sx[((size_t)kb * 64 + lane) * NX + i] = sval(16 * cb + ln, 64 * kb + 16 * gt + 4 * grp + r);
}
