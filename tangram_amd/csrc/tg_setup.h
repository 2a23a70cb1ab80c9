// tg_setup.h -- validation metrics, device-side initial logits, operand images of S and G, host pre-processing on the device.
// Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// Validation metrics of Mapper._val_loss_fn (mapping_optimizer.py:311-356; evaluated on the TRAINING split like the
// reference does, :321-322): gene score, voxel score, sparsity-weighted gene score, normalised row entropy.
// ----------------------------------------------------------------------------------------------
// one block per cell: rowent[c] = -sum_v P log P   (:333)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_row_entropy(const float* M, const float* rshift, const float* rinvz, int V, int Vp,
                                                    float* rowent) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float sh = rshift[c], iz = rinvz[c], liz = tg_log(iz);
    float s = 0.f;
    for (int v = t; v < V; v += 256) {
        const float z = M[(size_t)c * Vp + v] - sh;
        s += tg_exp(z) * iz * (z + liz);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += tg_shfl_xor(s, m);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (t == 0) rowent[c] = -(red[0] + red[1] + red[2] + red[3]);
}

struct TgValArgs {
    const float* genestat; const float* gnorm2; const float* gfrac;     // [2][Kp], [Kp], [Kp] (fraction of non-zero spots per gene)
    const float* voxstat; const float* vnorm2; const float* rowent;
    float* out;                                                          // [4]: gv + vg, gv, sparsity-weighted gv, entropy
    int K, Kp, V, Vr, C, nky;
    // spot shards: the sums over spots are taken per rank (`partial` = 1: part[0] = sum of the spot cosines, part[1] = sum of the row
    // entropies over this rank's spots, part[64 + k] = this rank's share of the non-zero fraction of gene k), all-reduced, and the
    // final call (`partial` = 0, `part` non-null) reads them back instead of summing itself.  Alone: part = null.
    int V_total, partial;
    float* part;
    float gfrac_scale;                                                   // V / V_total
};
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_val_finalize(TgValArgs a) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const int t = threadIdx.x;
    float vs = 0.f, es = 0.f;
    if (a.partial || !a.part) {
        for (int v = t; v < a.V; v += 1024) {
            float dot = 0.f, n2 = 0.f;
            for (int y = 0; y < a.nky; ++y) { dot += a.voxstat[((size_t)y * 2 + 0) * a.Vr + v]; n2 += a.voxstat[((size_t)y * 2 + 1) * a.Vr + v]; }
            const float na = tg_fmax(sqrtf(n2), TG_COS_EPS), nb = tg_fmax(sqrtf(a.vnorm2[v]), TG_COS_EPS);
            vs += dot / (na * nb);
        }
        vs = tg_block_sum_1024(vs, red);
        for (int c = t; c < a.C; c += 1024) es += a.rowent[c];
        es = tg_block_sum_1024(es, red);
    } else { vs = a.part[0]; es = a.part[1]; }
    if (a.partial) {
        if (t < 64) a.part[t] = (t == 0) ? vs : ((t == 1) ? es : 0.f);
        for (int k = t; k < a.Kp; k += 1024) a.part[64 + k] = (k < a.K) ? a.gfrac[k] * a.gfrac_scale : 0.f;
        return;
    }
    const float* gfrac = a.part ? a.part + 64 : a.gfrac;
    float cs = 0.f, ws = 0.f, wn = 0.f;
    for (int k = t; k < a.K; k += 1024) {
        const float na = tg_fmax(sqrtf(a.genestat[a.Kp + k]), TG_COS_EPS), nb = tg_fmax(sqrtf(a.gnorm2[k]), TG_COS_EPS);
        const float c = a.genestat[k] / (na * nb);
        cs += c;
        ws += c * gfrac[k];
        wn += gfrac[k];
    }
    const float gv = tg_block_sum_1024(cs, red) / (float)a.K;
    const float wsum = tg_block_sum_1024(ws, red), wnorm = tg_block_sum_1024(wn, red);
    const float vg = vs / (float)a.V_total;
    const float ent = es / ((float)a.C * logf((float)a.V_total));
    if (t == 0) { a.out[0] = gv + vg; a.out[1] = gv; a.out[2] = wsum / wnorm; a.out[3] = ent; }
}

// ----------------------------------------------------------------------------------------------
// Initial logits generated ON the device (opt-in replacement of `np.random.normal(0, 1, (n_cells, n_spots))`,
// mapping_optimizer.py:147-157, for problems whose C x V plane must never exist on the host: cfg4 holds 40 GB of logits).
// Counter-based: element (cell c, GLOBAL spot v) is a function of (seed, c * n_spots_total + v) alone -- a spot shard generates
// exactly the columns it owns and any partition of the spots yields the same logits.  One SplitMix64 finaliser per element
// gives two 32-bit uniforms, Box-Muller (cosine branch) the standard normal.  NOT NumPy's stream: parity runs keep the
// reference's generator (host_rng.py); SURVEY 7.3-7 allows a device generator where the CPU reference cannot run.
// ----------------------------------------------------------------------------------------------
TG_DEV float tg_counter_normal(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = idx * 0x9E3779B97F4A7C15ull + (seed ^ 0xD1B54A32D192ED03ull) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u1 = ((float)(unsigned)(z >> 40) + 0.5f) * (1.0f / 16777216.0f);      // 24 bits: (0, 1), never 0
    const float u2 = ((float)(unsigned)(z & 0xFFFFFFu) + 0.5f) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * tg_log(u1)) * cosf(6.283185307179586f * u2);
}
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_init_normal(float* out, long long n_rows, long long n_cols, long long ld, unsigned long long seed,
                                                    long long col0, long long n_cols_total) {
    const long long quads = (n_cols + 3) / 4, total = n_rows * quads;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long long)gridDim.x * 256) {
        const long long r = q / quads, c = 4 * (q % quads);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < n_cols) out[r * ld + c + e] = tg_counter_normal(seed, (unsigned long long)(r * n_cols_total + col0 + c + e));
    }
}

// ----------------------------------------------------------------------------------------------
// set-up kernels: operand images of S and the padded fp32 copy of G
// ----------------------------------------------------------------------------------------------
struct TgPrepSArgs {
    const float* S; int C, K;           // caller's [C][K] (row pitch ldS elements)
    long long ldS;
    const float* aug;                   // [C] values of the augmentation column K (null => 1)
    const float* ct; int T;             // [C][T] cell-type encoding -> columns K+1 .. K+T (ct-islands term), or null
    unsigned char* Sk; int Cr, Kp;      // [Cr][Kp/BKE][128 B]   (contraction over genes)
    unsigned char* St; int Cp;          // [Kp][Cp/BKE][128 B]   (contraction over cells)
};
TG_DEV float tg_s_aug(const TgPrepSArgs& a, int c, int k) {
    if (c >= a.C) return 0.f;
    if (k < a.K) return a.S[(size_t)c * a.ldS + k];
    if (k == a.K) return a.aug ? a.aug[c] : 1.f;
    if (a.ct && k - a.K - 1 < a.T) return a.ct[(size_t)c * a.T + (k - a.K - 1)];
    return 0.f;
}
template <class PR>
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_prep_sk(TgPrepSArgs a) {
    const int nch = a.Kp / PR::CH;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.Cr * nch) return;
    const int c = (int)(idx / nch), ch = (int)(idx % nch);
    const int k = ch * PR::CH;
    float x[PR::CH];
#pragma unroll
    for (int e = 0; e < PR::CH; ++e) x[e] = tg_s_aug(a, c, k + e);
    tg_store_s_chunk<PR>(a.Sk + (size_t)c * (a.Kp / PR::BKE) * (PR::BRC * 16), k / PR::BKE, (k % PR::BKE) / PR::CH, x);
}
template <class PR>
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_prep_st(TgPrepSArgs a) {
    const int nch = a.Cp / PR::CH;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)a.Kp * nch) return;
    const int k = (int)(idx % a.Kp), ch = (int)(idx / a.Kp);     // k fastest: coalesced reads of S rows
    const int c = ch * PR::CH;
    float x[PR::CH];
#pragma unroll
    for (int e = 0; e < PR::CH; ++e) x[e] = tg_s_aug(a, c + e, k);
    tg_store_s_chunk<PR>(a.St + (size_t)k * (a.Cp / PR::BKE) * (PR::BRC * 16), c / PR::BKE, (c % PR::BKE) / PR::CH, x);
}
// Is every element the S images are built from exactly representable in bf16 (then their lo parts are identically zero and
// PrecBF16x2S applies)?  *flag |= 1 otherwise.  (An integer OR: order-independent.)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_s_exact_check(TgPrepSArgs a, int* flag) {
    const size_t n = (size_t)a.C * (a.K + 1 + (a.ct ? a.T : 0));
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / (a.K + 1 + (a.ct ? a.T : 0))), k = (int)(i % (a.K + 1 + (a.ct ? a.T : 0)));
        const float x = tg_s_aug(a, c, k);
        bad |= tg_bf16_lo_to_f32(tg_pack_bf16(x, 0.f)) != x;
    }
    if (bad) tg_flag_or(flag, 1);
}

// Gp = zero-padded copy of G; vnorm2[v] = sum_k G^2; gnormpart[rb][k] = partial sum_v G^2
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_prep_g(const float* G, int V, int K, int Vr, int Kp, float* Gp, float* vnorm2,
                                               float* gnormpart /*[nrb][Kp]*/, float* gnnzpart /*[nrb][Kp]*/) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int vbeg = blockIdx.x * TG_RB;
    float vn[TG_RB];
#pragma unroll
    for (int i = 0; i < TG_RB; ++i) vn[i] = 0.f;
    for (int k = t; k < Kp; k += 256) {
        float gs = 0.f, nz = 0.f;
#pragma unroll
        for (int i = 0; i < TG_RB; ++i) {
            const int v = vbeg + i;
            float x = 0.f;
            if (v < V && k < K) x = G[(size_t)v * K + k];
            if (v < Vr) Gp[(size_t)v * Kp + k] = x;
            gs += x * x;
            nz += (x != 0.f) ? 1.f : 0.f;
            vn[i] += x * x;
        }
        gnormpart[(size_t)blockIdx.x * Kp + k] = gs;
        gnnzpart[(size_t)blockIdx.x * Kp + k] = nz;
    }
#pragma unroll
    for (int i = 0; i < TG_RB; ++i) {
        float n = vn[i];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) n += tg_shfl_xor(n, m);
        if (lane == 0) red[wave * TG_RB + i] = n;
    }
    __syncthreads();
    if (t < TG_RB && vbeg + t < Vr) vnorm2[vbeg + t] = red[t] + red[TG_RB + t] + red[2 * TG_RB + t] + red[3 * TG_RB + t];
}

// out[0] = sum_i x[i]  (one block of 1024 threads, fixed order): the density prior's total, used by the filter gradient (:512-515)
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_vec_sum(const float* x, int n, float* out) {
    TG_LDS_DECL;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) s += x[i];
    const float tot = tg_block_sum_1024(s, (float*)tg_lds);
    if (threadIdx.x == 0) out[0] = tot;
}

// out[k] = scale * sum_p part[p][k]: 64 columns per workgroup, 16 groups of rows p = g, g + 16, ... summed side by side, then the
// groups in fixed order.  (One thread per column walking all the parts -- 308 dependent loads at 9 852 spots -- took 141 us, as long
// as four training iterations of a clusters-mode problem, twice per mapper set-up.)
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_colsum_parts(const float* part, int nparts, int n, float* out, float scale) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;                                 // [16][64]
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6, k = blockIdx.x * 64 + c;
    float s = 0.f;
    if (k < n)
        for (int p = g; p < nparts; p += 16) s += part[(size_t)p * n + k];
    red[g * 64 + c] = s;
    __syncthreads();
    if (g == 0 && k < n) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i * 64 + c];
        out[k] = t * scale;
    }
}

// Dense block of gene columns [col0, col0 + ncols) of a CSR matrix (cells x genes, as AnnData keeps adata_sc.X): one workgroup
// per cell row; replaces `adata_sc.X.toarray()` on the host (utils.py:364-365, mapping_utils.py:259-266) for project_genes.
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_csr_cols_to_dense(const long long* indptr, const int* indices, const float* data, int col0,
                                                          int ncols, float* out, long long ld_out) {
    const long long row = blockIdx.x;
    float* o = out + row * ld_out;
    for (int k = threadIdx.x; k < ncols; k += 256) o[k] = 0.f;
    __syncthreads();
    const long long beg = indptr[row], end = indptr[row + 1];
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
        const int c = indices[i] - col0;
        if (c >= 0 && c < ncols) o[c] = data[i];           // (canonical CSR: one entry per (row, column))
    }
}

// ----------------------------------------------------------------------------------------------
// Host pre-processing on the device: what map_cells_to_space / pp_adatas do with NumPy before the first iteration
// ----------------------------------------------------------------------------------------------
// out[row][colmap[j]] = X[row][j] for the selected columns (colmap[j] >= 0) of a CSR matrix: the training-gene columns of
// adata_sc.X / adata_sp.X straight into the dense S / G of the mapper (mapping_utils.py:259-275: `adata[:, genes].X.toarray()`
// on the host).  One workgroup per row; values are copied, so the result is bit-identical to the host gather.
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_csr_gather_cols(const long long* indptr, const int* indices, const float* data, const int* colmap,
                                                        int ncols_out, float* out, long long ld_out) {
    const long long row = blockIdx.x;
    float* o = out + row * ld_out;
    for (int k = threadIdx.x; k < ncols_out; k += 256) o[k] = 0.f;
    __syncthreads();
    const long long beg = indptr[row], end = indptr[row + 1];
    for (long long i = beg + threadIdx.x; i < end; i += 256) {
        const int c = colmap[indices[i]];
        if (c >= 0) o[c] = data[i];                        // (canonical CSR: one entry per (row, column))
    }
}

// out[row] = sum of the row, accumulated in DOUBLE (one wave per row, 64 partial sums combined in lane order: fixed order) and
// rounded once: `adata_sp.X.sum(axis=1)` of pp_adatas (mapping_utils.py:88).  Dense (X, ld, ncols) or CSR (indptr, data).
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_row_sums(const float* X, long long ld, int ncols, const long long* indptr, const float* data,
                                                 long long nrows, float* out) {
    TG_LDS_DECL;
    double* red = (double*)tg_lds;                         // [4 waves][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long row = (long long)blockIdx.x * 4 + wave;
    double s = 0.0;
    if (row < nrows) {
        if (indptr) { for (long long i = indptr[row] + lane; i < indptr[row + 1]; i += 64) s += (double)data[i]; }
        else { for (int k = lane; k < ncols; k += 64) s += (double)X[row * ld + k]; }
    }
    red[wave * 64 + lane] = s;
    __syncthreads();
    if (lane == 0 && row < nrows) {
        double t = 0.0;
        for (int i = 0; i < 64; ++i) t += red[wave * 64 + i];
        out[row] = (float)t;
    }
}

// x[i] /= sum(x) with the total in double (one block, fixed order): rna_count_based_density (mapping_utils.py:89)
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_normalize_total(float* x, long long n) {
    TG_LDS_DECL;
    double* red = (double*)tg_lds;                         // [1024]
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += 1024) s += (double)x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const double tot = red[0];
    for (long long i = threadIdx.x; i < n; i += 1024) x[i] = (float)((double)x[i] / tot);
}

// out[cluster][k] = sum (or mean) over the member rows of X[.][k], accumulated in double in member order: adata_to_cluster_expression
// (mapping_utils.py:126-132).  members = CSR-like lists of row indices per cluster.  grid = (clusters, column blocks of 256).
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_cluster_sums(const float* X, long long ld, int ncols, const int* member_indptr, const int* member_rows,
                                                     int mean, float* out, long long ld_out) {
    const int cl = blockIdx.x, k = blockIdx.y * 256 + threadIdx.x;
    if (k >= ncols) return;
    const int b = member_indptr[cl], e = member_indptr[cl + 1];
    double s = 0.0;
    for (int i = b; i < e; ++i) s += (double)X[(long long)member_rows[i] * ld + k];
    if (mean) s /= (double)(e - b);                         // (an empty cluster yields NaN like NumPy's mean of nothing)
    out[(long long)cl * ld_out + k] = (float)s;
}

TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_fill(float* p, size_t n, float val) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = val;
}
