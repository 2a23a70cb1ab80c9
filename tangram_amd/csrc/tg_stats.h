// tg_stats.h -- K2: the V x K element-wise stage between the GEMMs: sum of the forward partials, cosine statistics, loss scalars
// and gradient coefficients, dGhat in operand format.  Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// K2a: sum C-splits -> Ghat, per-gene partial sums over a block of spots, per-spot sums over genes
//   (cosine_similarity statistics, mapping_optimizer.py:205-206)
// ----------------------------------------------------------------------------------------------
#define TG_RB 16   // spots per block in the V x Kp elementwise kernels

struct TgGhatReduceArgs {
    const float* Gpart; int nsplit;
    int units, f_tm, f_nsteps;  // the forward kernel's decomposition (tg_fwd_nseg): slots to sum per spot tile
    const float* G;            // [Vr][Kp] fp32, zero padded
    float* Ghat;               // [Vr][Kp]
    float* genepart;           // [nrb][2][Kp]  (dot, |Ghat|^2)
    float* voxstat;            // [nky][2][Vr] (dot_v, |Ghat_v|^2) over the genes k < K of column block ky; written iff want_vox
    int V, Vr, Kp, K, want_vox;
};
#define TG_GH_COLS 256         // gene columns per workgroup: grid = (row blocks of TG_RB spots, ceil(Kp / TG_GH_COLS))

// One workgroup = 16 spots x 256 genes: wave w owns 4 of the spot rows, lane q one float4 of genes.  (The earlier layout,
// 16 rows x all genes per workgroup, left a V = 1250 spot shard with 79 workgroups to stream 12 partial copies of Ghat.)
TG_DEV void tg_ghat_reduce_body(const TgGhatReduceArgs& a) {
    TG_LDS_DECL;
    f32x4* red = (f32x4*)tg_lds;     // [4 row groups][64 lanes][2]
    const int t = threadIdx.x, q = t & 63, rg = t >> 6;
    const int rb = blockIdx.x, ky = blockIdx.y;
    const int vbeg = rb * TG_RB + rg * (TG_RB / 4);
    const int k = ky * TG_GH_COLS + 4 * q;
    const bool kok = k < a.Kp;
    f32x4 gd = {0, 0, 0, 0}, gn = {0, 0, 0, 0};
    float vd[TG_RB / 4], vn[TG_RB / 4];
    // partial slots of the forward tile this lane's elements belong to (the 16 spots of the block lie in one spot tile)
    const int nseg = tg_fwd_nseg(rb * TG_RB / a.f_tm, a.f_nsteps, (long long)(a.Vr / a.f_tm) * a.f_nsteps, a.units);
#pragma unroll
    for (int i = 0; i < TG_RB / 4; ++i) {
        vd[i] = vn[i] = 0.f;
        const int v = vbeg + i;
        if (kok && v < a.V) {
            const size_t off = (size_t)v * a.Kp + k;
            // the partial slots of this element, summed in slot order; requested four or eight at a time (a thin spot shard has 12
            // slots and ~1 workgroup per CU: one dependent load after the other made this kernel latency-bound, 24 us for 73 MB)
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            auto sum_slots = [&](auto width) {
                constexpr int W = decltype(width)::value;
                for (int p0 = 0; p0 < nseg; p0 += W) {
                    f32x4 part[W];
#pragma unroll
                    for (int q2 = 0; q2 < W; ++q2) {
                        const int p = (p0 + q2 < nseg) ? p0 + q2 : nseg - 1;      // (clamped: in bounds; the value is dropped below)
                        part[q2] = *(const f32x4*)(a.Gpart + (size_t)p * a.Vr * a.Kp + off);
                    }
#pragma unroll
                    for (int q2 = 0; q2 < W; ++q2)
                        if (p0 + q2 < nseg) s = (p0 + q2 == 0) ? part[q2] : s + part[q2];
                }
            };
            if (nseg <= 4) sum_slots(std::integral_constant<int, 4>()); else sum_slots(std::integral_constant<int, 8>());
            *(f32x4*)(a.Ghat + off) = s;
            const f32x4 g = *(const f32x4*)(a.G + off);
            gd += s * g;
            gn += s * s;
            if (a.want_vox) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < a.K) { vd[i] += s[e] * g[e]; vn[i] += s[e] * s[e]; }
            }
        }
    }
    red[(rg * 64 + q) * 2 + 0] = gd;
    red[(rg * 64 + q) * 2 + 1] = gn;
    __syncthreads();
    if (rg == 0 && kok) {
#pragma unroll
        for (int r = 1; r < 4; ++r) { gd += red[(r * 64 + q) * 2 + 0]; gn += red[(r * 64 + q) * 2 + 1]; }
        *(f32x4*)(a.genepart + ((size_t)rb * 2 + 0) * a.Kp + k) = gd;
        *(f32x4*)(a.genepart + ((size_t)rb * 2 + 1) * a.Kp + k) = gn;
    }
    if (a.want_vox) {
#pragma unroll
        for (int i = 0; i < TG_RB / 4; ++i) {
            float d = vd[i], n = vn[i];
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { d += tg_shfl_xor(d, m); n += tg_shfl_xor(n, m); }
            if (q == 0 && vbeg + i < a.V) {
                a.voxstat[((size_t)ky * 2 + 0) * a.Vr + vbeg + i] = d;
                a.voxstat[((size_t)ky * 2 + 1) * a.Vr + vbeg + i] = n;
            }
        }
    }
}

// K2b: second stage of the per-gene sums (fixed order => deterministic): KX genes x 1024 / KX partial groups per block.
// (A latency-bound kernel: every thread walks nrb / groups row blocks; with 4 groups it took 29 us at 600 row blocks.  KX = 64:
//  16 groups; KX = 16, for more than 512 row blocks: 64 groups and four times the workgroups -- 36 -> 12 us at 1 563 row blocks.)
#define TG_GR_GROUPS 16
// `link` (spot shard on the peer transport, round 6): the exchange of the statistics happens HERE -- the thread that owns gene k pushes its
// two sums into every rank's mailbox and stores the rank-order sum of the world's granules: what an all-reduce after this kernel delivers.
template <int KX>
TG_DEV void tg_gene_reduce_body(const float* genepart, int nrb, int Kp, float* genestat /*[2][Kp]*/, const TgPeerLink* link = nullptr) {
    constexpr int NG = 1024 / KX;
    TG_LDS_DECL;
    float* red = (float*)tg_lds;        // [NG][KX][2]
    const int kx = threadIdx.x % KX, grp = threadIdx.x / KX;
    const int k = blockIdx.x * KX + kx;
    float d0 = 0.f, n0 = 0.f, d1 = 0.f, n1 = 0.f;
    if (k < Kp) {
        int b = grp;
        for (; b + NG < nrb; b += 2 * NG) {
            d0 += genepart[((size_t)b * 2 + 0) * Kp + k];
            n0 += genepart[((size_t)b * 2 + 1) * Kp + k];
            d1 += genepart[((size_t)(b + NG) * 2 + 0) * Kp + k];
            n1 += genepart[((size_t)(b + NG) * 2 + 1) * Kp + k];
        }
        for (; b < nrb; b += NG) {
            d0 += genepart[((size_t)b * 2 + 0) * Kp + k];
            n0 += genepart[((size_t)b * 2 + 1) * Kp + k];
        }
    }
    red[(grp * KX + kx) * 2 + 0] = d0 + d1;
    red[(grp * KX + kx) * 2 + 1] = n0 + n1;
    __syncthreads();
    if (grp == 0 && k < Kp) {
        float d = 0.f, n = 0.f;
        for (int g = 0; g < NG; ++g) { d += red[(g * KX + kx) * 2 + 0]; n += red[(g * KX + kx) * 2 + 1]; }
        if (link) {
            tg_link_push(*link, link->e2 + k, d);
            tg_link_push(*link, link->e2 + Kp + k, n);
            tg_link_sum2(*link, link->e2 + k, link->e2 + Kp + k, d, n);
        }
        genestat[k] = d;
        genestat[Kp + k] = n;
    }
}

// ----------------------------------------------------------------------------------------------
// K2c: scalars + gradient coefficients.  One block of 1024 threads.
//   gv (:205,:208), vg (:206,:209), KL (:212-219), total (:266-270) -> history row
//   alpha_k, beta_k:  dGhat_vk (gene term)  = alpha_k G_vk + beta_k Ghat_vk
//   va_v, vb_v:       dGhat_vk (voxel term) = va_v G_vk + vb_v Ghat_vk
//   a_v = -lambda_d d_v / colsum_v   (dP_cv += a_v w_c)
// In a spot-sharded multi-GPU run genestat/gnorm2 hold globally reduced values while the per-spot
// sums are local; `nranks_v` and V_total make the means global.
// ----------------------------------------------------------------------------------------------
struct TgFinalizeArgs {
    const float* genestat;     // [2][Kp] (dot_k, |Ghat_k|^2) (global)
    const float* gnorm2;       // [Kp] |G_k|^2 (global)
    const float* Ghat;         // [Vr][Kp] (aug column K = colsum)
    const float* voxstat;      // [nky][2][Vr] partial over gene column blocks
    int nky;
    const float* vnorm2;       // [Vr] |G_v|^2 over genes
    const float* d;            // [Vr] density prior or null
    float* coef;               // [2][Kp] alpha, beta
    float* vcoef;              // [3][Vr] va, vb, a_v
    float* hist;               // history row [TGH_NTERMS]
    float lambda_g1, lambda_g2, lambda_d;
    float rho_scale;           // 1/C for a uniform source, 1 for d_source (rho_v = colsum_v * rho_scale)
    const float* fsum_dev;     // constrained mode: rho_v = colsum_v / sum_c f_c  (mapping_optimizer.py:512-513); else null
    int K, Kp, V, Vr, V_total, has_density;
    // spatial refinement terms (mapping_optimizer.py:234-248)
    const float* nbstat;       // [2][Kp] (dot(W Ghat, W G), |W Ghat|^2) per gene, or null
    const float* wgnorm2;      // [Kp] |W G|^2 per gene
    float* nbcoef;             // [2][Kp] -> d(loss)/d(W Ghat) = nbcoef0 * WG + nbcoef1 * WGhat
    const float* ctpart; int n_ctpart;   // per-spot sums of relu(D) (ct islands), or null
    float lambda_nb, lambda_ct; int T;
    int V_sp;                  // spots the spatial sums (ct islands) run over: V, or ALL spots on a spot shard (the spatial terms are
                               // evaluated on the gathered Ghat there, identically on every rank)
    float* part_out;           // spot shards: [0] = this rank's part of the voxel score (sum_v cos / V_total), [1] = of the KL sum; or null
    float* spotpart; int n_spotpart;     // [spot blocks][2] sums of the per-spot (cosine, KL) terms, left by the kernel that evaluates
                                         // tg_spot_coef anyway (tg_sc_backward, tg_dghat_emit<SELF>); null: tg_loss_scalars walks the spots itself
};

TG_DEV float tg_block_sum_1024(float x, float* red) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += tg_shfl_xor(x, m);
    __syncthreads();
    if (lane == 0) red[wave] = x;
    __syncthreads();
    float s = 0.f;
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
    return s;
}
// per-gene gradient coefficients and cosine term: dGhat_vk (gene term) = al G_vk + be Ghat_vk
TG_DEV void tg_gene_coef(const TgFinalizeArgs& a, const float* stat, const float* gn2, float lambda, int k, float& al, float& be, float& c) {
    const float dot = stat[k];
    const float na = tg_fmax(sqrtf(stat[a.Kp + k]), TG_COS_EPS);
    const float nb = tg_fmax(sqrtf(gn2[k]), TG_COS_EPS);
    c = dot / (na * nb);
    const float w = lambda / (float)a.K;
    al = -w / (na * nb);
    be = w * c / (na * na);
}
// per-spot coefficients from loaded statistics: voxel cosine term (va, vb, cosine c) and density term (a_v, KL summand)
TG_DEV void tg_spot_coef(const TgFinalizeArgs& a, float dot, float n2a, float n2b, float colsum, float dv, float rho_scale,
                         float& va, float& vb, float& av, float& c, float& kl) {
    va = vb = av = c = kl = 0.f;
    if (a.lambda_g2 != 0.f) {
        const float na = tg_fmax(sqrtf(n2a), TG_COS_EPS);
        const float nb = tg_fmax(sqrtf(n2b), TG_COS_EPS);
        c = dot / (na * nb);
        const float w = a.lambda_g2 / (float)a.V_total;
        va = -w / (na * nb);
        vb = w * c / (na * na);
    }
    if (a.has_density) {
        const float rho = colsum * rho_scale;
        if (dv != 0.f) kl = dv * (tg_log(dv) - tg_log(rho));   // KLDivLoss(sum): xlogy(d,d) - d*log(rho)
        av = -a.lambda_d * dv * rho_scale / rho;                // = -lambda_d d_v / colsum_v
    }
}
TG_DEV void tg_spot_stats_load(const TgFinalizeArgs& a, int v, float& dot, float& n2a, float& n2b, float& colsum, float& dv) {
    const bool in = v < a.V;
    dot = 0.f; n2a = (in && a.lambda_g2 != 0.f) ? 0.f : 1.f;
    if (in && a.lambda_g2 != 0.f)
        for (int y = 0; y < a.nky; ++y) { dot += a.voxstat[((size_t)y * 2 + 0) * a.Vr + v]; n2a += a.voxstat[((size_t)y * 2 + 1) * a.Vr + v]; }
    n2b = (in && a.lambda_g2 != 0.f) ? a.vnorm2[v] : 1.f;
    colsum = (in && a.has_density) ? a.Ghat[(size_t)v * a.Kp + a.K] : 1.f;
    dv = (in && a.has_density) ? a.d[v] : 0.f;
}

// N block sums with ONE pair of barriers, any block size that is a multiple of 64 (fixed summation order)
template <int N>
TG_DEV void tg_block_sums(float (&x)[N], float* red) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x[i] += tg_shfl_xor(x[i], m);
    __syncthreads();
    if (lane == 0)
#pragma unroll
        for (int i = 0; i < N; ++i) red[wave * N + i] = x[i];
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w * N + i];
        x[i] = s;
    }
}

// The scalars of one iteration -> history row; with WRITE also the gradient coefficient vectors (coef, nbcoef, vcoef).
// One workgroup of any size (multiple of 64); `red` needs (blockDim / 64) * 5 floats.
template <bool WRITE>
TG_DEV void tg_loss_scalars(const TgFinalizeArgs& a, float* red) {
    const int t = threadIdx.x, nt = blockDim.x;
    float cs = 0.f;
    for (int k = t; k < a.Kp; k += nt) {
        float al = 0.f, be = 0.f, c = 0.f;
        if (k < a.K) { tg_gene_coef(a, a.genestat, a.gnorm2, a.lambda_g1, k, al, be, c); cs += c; }
        if (WRITE) { a.coef[k] = al; a.coef[a.Kp + k] = be; }
    }
    float nbs = 0.f;
    if (a.nbstat) {
        for (int k = t; k < a.Kp; k += nt) {
            float al = 0.f, be = 0.f, c = 0.f;
            if (k < a.K) { tg_gene_coef(a, a.nbstat, a.wgnorm2, a.lambda_nb, k, al, be, c); nbs += c; }
            if (WRITE) { a.nbcoef[k] = al; a.nbcoef[a.Kp + k] = be; }
        }
    }
    float cts = 0.f;
    if (a.ctpart) for (int i = t; i < a.n_ctpart; i += nt) cts += a.ctpart[i];

    float vs = 0.f, kl = 0.f;
    const float rho_scale = a.fsum_dev ? 1.f / a.fsum_dev[0] : a.rho_scale;
    if (!WRITE && a.spotpart)
        for (int i = t; i < a.n_spotpart; i += nt) { vs += a.spotpart[2 * i]; kl += a.spotpart[2 * i + 1]; }
    // (4 spots per trip with all their loads issued first: this single-workgroup loop is pure memory latency)
    else for (int vb0 = t; vb0 < a.Vr; vb0 += 4 * nt) {
        float dot[4], n2a[4], n2b[4], colsum[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tg_spot_stats_load(a, vb0 + u * nt, dot[u], n2a[u], n2b[u], colsum[u], dv[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = vb0 + u * nt;
            if (v >= a.Vr) continue;
            float va = 0.f, vb = 0.f, av = 0.f, c = 0.f, klv = 0.f;
            if (v < a.V) {
                tg_spot_coef(a, dot[u], n2a[u], n2b[u], colsum[u], dv[u], rho_scale, va, vb, av, c, klv);
                vs += c; kl += klv;
            }
            if (WRITE) { a.vcoef[v] = va; a.vcoef[a.Vr + v] = vb; a.vcoef[2 * a.Vr + v] = av; }
        }
    }
    float sums[5] = {cs, nbs, cts, vs, kl};           // the five scalars share one block reduction
    tg_block_sums(sums, red);
    const float gv = sums[0] / (float)a.K, nbv = sums[1] / (float)a.K;
    const float isl = sums[2] / ((float)a.V_sp * (float)(a.T > 0 ? a.T : 1));
    const float vg = sums[3] / (float)a.V_total, klsum = sums[4];
    if (t == 0) {
        const float nanv = __builtin_nanf("");
        float total = -a.lambda_g1 * gv;
        if (!a.part_out) {          // (spot shard: the terms that are sums over spots join the total in tg_merge_stats, once they
                                    //  are global -- added in the same order on every rank, so the history is bit-identical everywhere)
            if (a.lambda_g2 != 0.f) total -= a.lambda_g2 * vg;
            if (a.has_density) total += a.lambda_d * klsum;
        }
        for (int i = 0; i < TGH_NTERMS; ++i) a.hist[i] = nanv;
        a.hist[TGH_TOTAL] = total;
        a.hist[TGH_MAIN] = gv;
        a.hist[TGH_VG] = (a.lambda_g2 != 0.f) ? vg : nanv;        // reference: 0*x/0 = nan (:209)
        a.hist[TGH_KL] = a.has_density ? klsum : nanv;
        if (a.nbstat) { a.hist[TGH_NB] = nbv; a.hist[TGH_TOTAL] -= a.lambda_nb * nbv; }
        if (a.ctpart) { a.hist[TGH_CT] = isl; a.hist[TGH_TOTAL] += a.lambda_ct * isl; }
        if (a.part_out) { a.part_out[0] = vg; a.part_out[1] = klsum; }
    }
}

TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_loss_finalize(TgFinalizeArgs a) {
    TG_LDS_DECL;
    tg_loss_scalars<true>(a, (float*)tg_lds);
}

// ----------------------------------------------------------------------------------------------
// K2d: dGhat in operand format (contraction axis = genes), rows = spots: [Vr][Kp/BKE steps][128 B]
// ----------------------------------------------------------------------------------------------
struct TgEmitArgs {
    const float* Ghat; const float* G; const float* coef; const float* vcoef;
    const float* extra;        // [Vr][Kp] additional d(loss)/dGhat (spatial terms; also feeds the augmentation columns) or null
    unsigned char* dG;
    int V, Vr, Kp, K, n_aug;   // columns K+1 .. K+n_aug-1 carry the cell-type gradient
    TgFinalizeArgs fin;        // SELF: the statistics the coefficients are derived from (coef / vcoef above are then unused)
};

// SELF: the workgroup derives the per-gene (alpha, beta) and its 16 per-spot (va, vb, a_v) coefficients itself, from the
// reduced statistics, into LDS -- they are purely local functions of them.  tg_loss_finalize (one workgroup, ~20 us of
// dependent latency) then no longer sits between the forward and the backward GEMM: the scalars of the history row are
// produced by one extra workgroup of the update kernel, off the critical path.  dynamic LDS: (2 Kp + 2 TG_RB) floats.
template <class PR, bool EXTRA, bool SELF>
TG_DEV void tg_dghat_emit_body(const TgEmitArgs& a) {
    TG_LDS_DECL;
    float* cf = (float*)tg_lds;                          // SELF: [2][Kp] alpha, beta; then [2][TG_RB] va, vb
    constexpr int NQ = PR::CH / 4;                       // float4 groups per operand chunk
    const int nch = a.Kp / PR::CH;
    // gridDim.y column blocks share a block of spots (round 6: a thin spot shard has 79 blocks of 16 spots -- a third of the CUs, each
    // thread a chain of eight dependent trips; the launcher cuts the gene axis until the grid fills the chip): this one takes the chunks
    // [ch0, ch1) of every row
    const int ch0 = (int)((long long)nch * blockIdx.y / gridDim.y), ch1 = (int)((long long)nch * (blockIdx.y + 1) / gridDim.y), nloc = ch1 - ch0;
    const int vbeg = blockIdx.x * TG_RB;
    const size_t pitch = (size_t)(a.Kp / PR::BKE) * 128;
    const float* coef = a.coef;
    if constexpr (SELF) {
        for (int k = ch0 * PR::CH + threadIdx.x; k < ch1 * PR::CH; k += 256) {
            float al = 0.f, be = 0.f, c = 0.f;
            if (k < a.K) tg_gene_coef(a.fin, a.fin.genestat, a.fin.gnorm2, a.fin.lambda_g1, k, al, be, c);
            cf[k] = al; cf[a.Kp + k] = be;
        }
        float c_blk = 0.f, kl_blk = 0.f;
        if (threadIdx.x < TG_RB) {
            const int v = vbeg + threadIdx.x;
            float va = 0.f, vb = 0.f, av = 0.f;
            if (v < a.V) {
                float dot, n2a, n2b, colsum, dv;
                tg_spot_stats_load(a.fin, v, dot, n2a, n2b, colsum, dv);
                const float rho_scale = a.fin.fsum_dev ? 1.f / a.fin.fsum_dev[0] : a.fin.rho_scale;
                tg_spot_coef(a.fin, dot, n2a, n2b, colsum, dv, rho_scale, va, vb, av, c_blk, kl_blk);
            }
            cf[2 * a.Kp + threadIdx.x] = va; cf[2 * a.Kp + TG_RB + threadIdx.x] = vb;
            if (v < a.Vr && blockIdx.y == 0) { a.fin.vcoef[v] = va; a.fin.vcoef[a.Vr + v] = vb; a.fin.vcoef[2 * a.Vr + v] = av; }   // a_v: read by the backward / update kernels
        }
        if (threadIdx.x < 64 && a.fin.spotpart && blockIdx.y == 0) {        // the spots' loss terms summed per block: the history workgroup adds the blocks up
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { c_blk += tg_shfl_xor(c_blk, m); kl_blk += tg_shfl_xor(kl_blk, m); }
            if (threadIdx.x == 0) { a.fin.spotpart[2 * blockIdx.x] = c_blk; a.fin.spotpart[2 * blockIdx.x + 1] = kl_blk; }
        }
        __syncthreads();
        coef = cf;
    }
    for (int idx = threadIdx.x; idx < nloc * TG_RB; idx += 256) {
        const int i = idx / nloc, ch = ch0 + idx % nloc;
        const int v = vbeg + i;
        if (v >= a.V) continue;
        const int k = ch * PR::CH;
        const float va = SELF ? cf[2 * a.Kp + i] : a.vcoef[v], vb = SELF ? cf[2 * a.Kp + TG_RB + i] : a.vcoef[a.Vr + v];
        float x[PR::CH];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const size_t off = (size_t)v * a.Kp + k + 4 * q;
            const f32x4 gh = *(const f32x4*)(a.Ghat + off), g = *(const f32x4*)(a.G + off);
            const f32x4 ca = *(const f32x4*)(coef + k + 4 * q), cb = *(const f32x4*)(coef + a.Kp + k + 4 * q);
            f32x4 ex = {0.f, 0.f, 0.f, 0.f};
            if (EXTRA) ex = *(const f32x4*)(a.extra + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = k + 4 * q + e;
                float val = (ca[e] + va) * g[e] + (cb[e] + vb) * gh[e] + ex[e];       // gene columns
                if (kk >= a.K) val = (EXTRA && kk > a.K && kk < a.K + a.n_aug) ? ex[e] : 0.f;   // augmentation / padding columns
                x[4 * q + e] = val;
            }
        }
        tg_store_operand_chunk<PR>(a.dG + (size_t)v * pitch, k / PR::BKE, (k % PR::BKE) / PR::CH, x);
    }
}
