// tg_spatial.h -- spatial refinement terms on CSR spot graphs: neighbourhood, cell-type islands, Getis-Ord / Moran / Geary.
// Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// Spatial refinement terms (mapping_optimizer.py:234-248) on V x K matrices with CSR spot graphs (~6 nnz / row)
// instead of the reference's dense V x V products (spatial_weights.py:5-29).
// ----------------------------------------------------------------------------------------------
struct TgCsr { const int* indptr; const int* indices; const float* data; };

// Y[v][k] (op)= sum_j W[v][j] * src[j][k], k in [k_begin, k_end); src = A, or ca[k]*A + cb[k]*B when ca != null.
// One workgroup per spot row, threads along genes (coalesced).
struct TgSpmmArgs {
    TgCsr W; const float* A; const float* B; const float* ca; const float* cb;
    float* Y; int V, Kp, k_begin, k_end;
    int accumulate;            // Y += ... instead of Y = ...
    float* E;                  // optional [V][Kp]: sum_j W[v][j] (A[j][k] - A[v][k])^2   (local Geary sums, no cancellation)
    const float* addD;         // optional [V][Kp] addend
    const float* addc;         // optional [Kp]: subtracted per gene (centering constant)
};
// (4 genes per thread: float4 loads of the gathered rows -- a quarter of the load instructions of the one-gene-per-thread
//  version, 16 bytes per lane; the last, partial quad of [k_begin, k_end) is guarded per element)
// Round 5: (i) workgroup b runs on XCD b % 8, and the rows a spot gathers are its neighbours on the tissue, i.e. nearby rows: XCD x
// takes a CONTIGUOUS band of spots (rows x * V/8 ...), so that a band's gathered rows are shared through that XCD's L2 instead of every
// XCD streaming the whole matrix; (ii) the non-zeros of a row are taken eight at a time with every gathered row requested before the
// first is used (the one-at-a-time loop was a chain of ~7 dependent row loads).  Same sums in the same order: same bits.
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_spmm(TgSpmmArgs a) {
    constexpr int U = 8;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int q = nb >> 3, r = nb & 7;
    const int v = xcd * q + (xcd < r ? xcd : r) + j;            // (blocks with j == q exist for xcd < r only: every row exactly once)
    const int b = a.W.indptr[v], e = a.W.indptr[v + 1];
    for (int k = a.k_begin + 4 * threadIdx.x; k < a.k_end; k += 1024) {
        const size_t o = (size_t)v * a.Kp + k;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, ge = {0.f, 0.f, 0.f, 0.f};
        const f32x4 xv = a.E ? *(const f32x4*)(a.A + o) : s;
        f32x4 ca = {1.f, 1.f, 1.f, 1.f}, cb = s;
        if (a.ca) { ca = *(const f32x4*)(a.ca + k); cb = *(const f32x4*)(a.cb + k); }
        for (int i0 = b; i0 < e; i0 += U) {
            f32x4 x[U], y[U];
            float w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = i0 + u < e;
                const size_t off = (size_t)(ok ? a.W.indices[i0 + u] : v) * a.Kp + k;        // (beyond the row: its own row, unused)
                w[u] = ok ? a.W.data[i0 + u] : 0.f;
                x[u] = *(const f32x4*)(a.A + off);
                if (a.ca) y[u] = *(const f32x4*)(a.B + off);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (i0 + u >= e) continue;
                f32x4 xx = x[u];
                if (a.ca) xx = ca * xx + cb * y[u];
                s += w[u] * xx;
                if (a.E) { const f32x4 dx = xx - xv; ge += w[u] * dx * dx; }
            }
        }
        if (a.addD) s += *(const f32x4*)(a.addD + o);
        if (a.addc) s -= *(const f32x4*)(a.addc + k);
        if (k + 3 < a.k_end) {
            if (a.E) *(f32x4*)(a.E + o) = ge;
            *(f32x4*)(a.Y + o) = a.accumulate ? *(const f32x4*)(a.Y + o) + s : s;
        } else {
            for (int qq = 0; qq < 4 && k + qq < a.k_end; ++qq) {
                if (a.E) a.E[o + qq] = ge[qq];
                a.Y[o + qq] = a.accumulate ? a.Y[o + qq] + s[qq] : s[qq];
            }
        }
    }
}

// per-gene partial sums over a block of TG_RB spots: (sum A*B, sum A*A)  [second stage: tg_gene_reduce]
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_colstats(const float* A, const float* B, int V, int Kp, float* part /*[nrb][2][Kp]*/) {
    const int rb = blockIdx.x, vbeg = rb * TG_RB;
    for (int k = threadIdx.x; k < Kp; k += 256) {
        float d = 0.f, n = 0.f;
        for (int i = 0; i < TG_RB; ++i) {
            const int v = vbeg + i;
            if (v < V) { const float x = A[(size_t)v * Kp + k]; d += x * B[(size_t)v * Kp + k]; n += x * x; }
        }
        part[((size_t)rb * 2 + 0) * Kp + k] = d;
        part[((size_t)rb * 2 + 1) * Kp + k] = n;
    }
}

// cell-type islands (:242-248): ct = Ghat[:, K+1 : K+1+T]; D = ct - N ct; penalty = mean(max(D, 0));
// mask = 1[D > 0] / (V T)   (the reference's binary torch.max splits exact ties 0.5/0.5; ties have measure zero)
struct TgCtArgs {
    TgCsr N; const float* Ghat; float* mask /*[Vr][Tp]*/; float* ctpart /*[V]*/; float* extra;
    int V, Kp, K, T, Tp; float lambda_ct;
};
TG_KERNEL void TG_LAUNCH_BOUNDS(64) tg_ct_mask(TgCtArgs a) {
    const int v = blockIdx.x, b = a.N.indptr[v], e = a.N.indptr[v + 1];
    float part = 0.f;
    for (int t = threadIdx.x; t < a.T; t += 64) {
        const int col = a.K + 1 + t;
        float s = 0.f;
        for (int i = b; i < e; ++i) s += a.N.data[i] * a.Ghat[(size_t)a.N.indices[i] * a.Kp + col];
        const float D = a.Ghat[(size_t)v * a.Kp + col] - s;
        a.mask[(size_t)v * a.Tp + t] = (D > 0.f) ? 1.f / ((float)a.V * (float)a.T) : 0.f;
        part += (D > 0.f) ? D : 0.f;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += tg_shfl_xor(part, m);
    if (threadIdx.x == 0) a.ctpart[v] = part;
}
// d(penalty)/d(ct) = mask - N^T mask  -> augmentation columns of the extra gradient (a.N holds N^T here)
TG_KERNEL void TG_LAUNCH_BOUNDS(64) tg_ct_grad(TgCtArgs a) {
    const int v = blockIdx.x, b = a.N.indptr[v], e = a.N.indptr[v + 1];
    for (int t = threadIdx.x; t < a.T; t += 64) {
        float s = 0.f;
        for (int i = b; i < e; ++i) s += a.N.data[i] * a.mask[(size_t)a.N.indices[i] * a.Tp + t];
        a.extra[(size_t)v * a.Kp + a.K + 1 + t] = a.lambda_ct * (a.mask[(size_t)v * a.Tp + t] - s);
    }
}

// ----------------------------------------------------------------------------------------------
// Spatial autocorrelation terms (mapping_optimizer.py:159-187 indicators, :251-263 losses): Getis-Ord G*,
// Moran's I, Geary's C of Ghat compared by cosine similarity with the same indicators of G.  With x = Ghat[:,k],
// Y = Ws x, Z = Ws^T x, r/c = row/column sums of Ws, mu = mean(x), z = x - mu, u = Ws z = Y - mu r:
//   G*_v = Y_v / sum(x)                      (per-gene cosine is scale invariant => compare Y with the reference)
//   I_v  = V z_v u_v / sum(z^2)              (=> compare h = z u with the reference)
//   C    = sum_ij w_ij (x_j - x_i)^2 / (2 m2),  m2 = sum(z^2)/(V-1)            (one number per gene; K-vector cosine)
// Geary's double sum runs over the CSR non-zeros instead of the reference's V x V x K tensor (:182-185).
// ----------------------------------------------------------------------------------------------
enum { TGAC_S1 = 0, TGAC_S2, TGAC_S3, TGAC_S4, TGAC_GD, TGAC_GN, TGAC_NSTAT };   // sum x, x^2, sum_ij w_ij (x_j-x_i)^2, -, Y.Tg, Y^2
enum { TGAC_AG = 0, TGAC_BG, TGAC_AM, TGAC_BM, TGAC_GAM, TGAC_MU, TGAC_M2, TGAC_A, TGAC_Q, TGAC_NCOEF };

struct TgAcArgs {
    const float* X;            // [Vr][Kp] Ghat (or G at set-up)
    const float* Y; const float* Z;         // Ws X, Ws^T X
    const float* r; const float* rc;        // [Vr] row sums, row+column sums of Ws
    float* Tg; float* Tm; float* refp;      // references: [Vr][Kp], [Vr][Kp], [Kp]
    float* part; float* stat; float* stat2; // [nrb][nstat][Kp] partials, [TGAC_NSTAT][Kp] totals, [3][Kp] (h.Tm, h^2, sum z^2)
    float* coef;                            // [TGAC_NCOEF][Kp]
    float* B1; float* D;                    // SpMM source and direct gradient part
    float* cmpart; float* cm;               // centering constant of the Moran gradient: partials / per gene
    float* hist;
    int V, Vr, Kp, K, setup;
    float lam_getis, lam_moran, lam_geary;
};

TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_csr_rowsum(TgCsr W, int V, float* out, int accumulate) {
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    float s = accumulate ? out[v] : 0.f;
    for (int i = W.indptr[v]; i < W.indptr[v + 1]; ++i) s += W.data[i];
    out[v] = s;
}

// stage 1: per-gene partial sums over a block of TG_RB spots
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ac_stats1(TgAcArgs a) {
    const int rb = blockIdx.x, vbeg = rb * TG_RB;
    for (int k = threadIdx.x; k < a.Kp; k += 256) {
        float s[TGAC_NSTAT];
#pragma unroll
        for (int q = 0; q < TGAC_NSTAT; ++q) s[q] = 0.f;
        if (k < a.K)
            for (int i = 0; i < TG_RB; ++i) {
                const int v = vbeg + i;
                if (v >= a.V) break;
                const size_t o = (size_t)v * a.Kp + k;
                const float x = a.X[o], y = a.Y[o];
                s[TGAC_S1] += x; s[TGAC_S2] += x * x; s[TGAC_S3] += a.D[o];      // D holds the local Geary sums from tg_spmm
                s[TGAC_GN] += y * y;
                if (!a.setup && a.lam_getis > 0.f) s[TGAC_GD] += y * a.Tg[o];
            }
#pragma unroll
        for (int q = 0; q < TGAC_NSTAT; ++q) a.part[((size_t)rb * TGAC_NSTAT + q) * a.Kp + k] = s[q];
    }
}
// deterministic second stage of any [nparts][nstat][Kp] partial array
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_stat_reduce(const float* part, int nparts, int nstat, int Kp, float* out) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kp) return;
    for (int q = 0; q < nstat; ++q) {
        float s = 0.f;
        for (int p = 0; p < nparts; ++p) s += part[((size_t)p * nstat + q) * Kp + k];
        out[(size_t)q * Kp + k] = s;
    }
}
// stage 2: with mu from stage 1: q = sum (x - mu)^2 (two-pass, no cancellation) and, for Moran,
// h = (x - mu)(Y - mu r): partial sums of h.Tm and h^2
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ac_stats2(TgAcArgs a) {
    const int rb = blockIdx.x, vbeg = rb * TG_RB;
    for (int k = threadIdx.x; k < a.Kp; k += 256) {
        float d = 0.f, n = 0.f, qz = 0.f;
        if (k < a.K) {
            const float mu = a.stat[(size_t)TGAC_S1 * a.Kp + k] / (float)a.V;
            for (int i = 0; i < TG_RB; ++i) {
                const int v = vbeg + i;
                if (v >= a.V) break;
                const size_t o = (size_t)v * a.Kp + k;
                const float z = a.X[o] - mu;
                const float h = z * (a.Y[o] - mu * a.r[v]);
                n += h * h;
                qz += z * z;
                if (!a.setup && a.lam_moran > 0.f) d += h * a.Tm[o];
            }
        }
        a.part[((size_t)rb * 3 + 0) * a.Kp + k] = d;
        a.part[((size_t)rb * 3 + 1) * a.Kp + k] = n;
        a.part[((size_t)rb * 3 + 2) * a.Kp + k] = qz;
    }
}
// set-up: write the references computed from G.  Tg = Y / sum(x) (:171), Tm = V z u / sum z^2 (:175-176), refp = C (:185)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ac_refs(TgAcArgs a) {
    const int vbeg = blockIdx.x * TG_RB;
    for (int k = threadIdx.x; k < a.K; k += 256) {
        const float s1 = a.stat[(size_t)TGAC_S1 * a.Kp + k];
        const float mu = s1 / (float)a.V, q = a.stat2[2 * (size_t)a.Kp + k];
        for (int i = 0; i < TG_RB; ++i) {
            const int v = vbeg + i;
            if (v >= a.V) break;
            const size_t o = (size_t)v * a.Kp + k;
            a.Tg[o] = a.Y[o] / s1;
            a.Tm[o] = (float)a.V * (a.X[o] - mu) * (a.Y[o] - mu * a.r[v]) / q;
        }
        if (blockIdx.x == 0) {
            const float A = a.stat[(size_t)TGAC_S3 * a.Kp + k];
            a.refp[k] = A / (2.f * q / (float)(a.V - 1));
        }
    }
}
// per-gene coefficients and the three scalars (one block); stat2 = [2][Kp] (h.Tm, h^2), tnorm = [4][Kp] (rows 0 and 2: |Tg_k|^2, |Tm_k|^2)
struct TgAcFinArgs { TgAcArgs a; const float* tnorm; };
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_ac_finalize(TgAcFinArgs f) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const TgAcArgs& a = f.a;
    const int t = threadIdx.x;
    const float Vf = (float)a.V;
    float gs = 0.f, ms = 0.f, pd = 0.f, pn = 0.f, rn = 0.f;
    for (int k = t; k < a.Kp; k += 1024) {
        float ag = 0.f, bg = 0.f, am = 0.f, bm = 0.f, mu = 0.f, m2 = 1.f, A = 0.f, q = 1.f;
        if (k < a.K) {
            mu = a.stat[(size_t)TGAC_S1 * a.Kp + k] / Vf;
            q = a.stat2[2 * (size_t)a.Kp + k];
            m2 = q / (Vf - 1.f);
            A = a.stat[(size_t)TGAC_S3 * a.Kp + k];
            if (a.lam_getis > 0.f) {
                const float na = tg_fmax(sqrtf(a.stat[(size_t)TGAC_GN * a.Kp + k]), 1e-30f), nb = tg_fmax(sqrtf(f.tnorm[k]), 1e-30f);
                const float c = a.stat[(size_t)TGAC_GD * a.Kp + k] / (na * nb);
                gs += c;
                const float w = a.lam_getis / (float)a.K;
                ag = -w / (na * nb); bg = w * c / (na * na);
            }
            if (a.lam_moran > 0.f) {
                const float na = tg_fmax(sqrtf(a.stat2[a.Kp + k]), 1e-30f), nb = tg_fmax(sqrtf(f.tnorm[2 * (size_t)a.Kp + k]), 1e-30f);
                const float c = a.stat2[k] / (na * nb);
                ms += c;
                const float w = a.lam_moran / (float)a.K;
                am = -w / (na * nb); bm = w * c / (na * na);
            }
            if (a.lam_geary > 0.f) {
                const float p = A / (2.f * m2), rp = a.refp[k];
                pd += p * rp; pn += p * p; rn += rp * rp;
            }
        }
        a.coef[(size_t)TGAC_AG * a.Kp + k] = ag; a.coef[(size_t)TGAC_BG * a.Kp + k] = bg;
        a.coef[(size_t)TGAC_AM * a.Kp + k] = am; a.coef[(size_t)TGAC_BM * a.Kp + k] = bm;
        a.coef[(size_t)TGAC_MU * a.Kp + k] = mu; a.coef[(size_t)TGAC_M2 * a.Kp + k] = m2;
        a.coef[(size_t)TGAC_A * a.Kp + k] = A; a.coef[(size_t)TGAC_Q * a.Kp + k] = q;
    }
    const float getis = tg_block_sum_1024(gs, red) / (float)a.K, moran = tg_block_sum_1024(ms, red) / (float)a.K;
    const float dotp = tg_block_sum_1024(pd, red);
    const float npn = tg_fmax(sqrtf(tg_block_sum_1024(pn, red)), TG_COS_EPS), nrn = tg_fmax(sqrtf(tg_block_sum_1024(rn, red)), TG_COS_EPS);
    const float cosg = dotp / (npn * nrn);
    for (int k = t; k < a.Kp; k += 1024) {
        float gam = 0.f;
        if (k < a.K && a.lam_geary > 0.f) {
            const float p = a.coef[(size_t)TGAC_A * a.Kp + k] / (2.f * a.coef[(size_t)TGAC_M2 * a.Kp + k]);
            gam = -a.lam_geary * (a.refp[k] / (npn * nrn) - cosg * p / (npn * npn));
        }
        a.coef[(size_t)TGAC_GAM * a.Kp + k] = gam;
    }
    if (t == 0) {
        float total = a.hist[TGH_TOTAL];
        if (a.lam_getis > 0.f) { a.hist[TGH_GETIS] = getis; total -= a.lam_getis * getis; }
        if (a.lam_moran > 0.f) { a.hist[TGH_MORAN] = moran; total -= a.lam_moran * moran; }
        if (a.lam_geary > 0.f) { a.hist[TGH_GEARY] = cosg; total -= a.lam_geary * cosg; }
        a.hist[TGH_TOTAL] = total;
    }
}
// gradient assembly, elementwise part: B1 = source of the W^T SpMM (Getis + Moran), D = direct part (Moran + Geary),
// cmpart = partial sums of the Moran part's column mean (the centering Jacobian of z = x - mean(x))
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ac_grad(TgAcArgs a) {
    const int rb = blockIdx.x, vbeg = rb * TG_RB;
    const float Vf = (float)a.V;
    for (int k = threadIdx.x; k < a.Kp; k += 256) {
        float cms = 0.f;
        if (k < a.K) {
            const float ag = a.coef[(size_t)TGAC_AG * a.Kp + k], bg = a.coef[(size_t)TGAC_BG * a.Kp + k];
            const float am = a.coef[(size_t)TGAC_AM * a.Kp + k], bm = a.coef[(size_t)TGAC_BM * a.Kp + k];
            const float gam = a.coef[(size_t)TGAC_GAM * a.Kp + k], mu = a.coef[(size_t)TGAC_MU * a.Kp + k];
            const float m2 = a.coef[(size_t)TGAC_M2 * a.Kp + k], A = a.coef[(size_t)TGAC_A * a.Kp + k];
            for (int i = 0; i < TG_RB; ++i) {
                const int v = vbeg + i;
                if (v >= a.V) break;
                const size_t o = (size_t)v * a.Kp + k;
                const float x = a.X[o], y = a.Y[o], z = x - mu, u = y - mu * a.r[v];
                float b1 = 0.f, dd = 0.f;
                if (a.lam_getis > 0.f) b1 += ag * a.Tg[o] + bg * y;
                if (a.lam_moran > 0.f) {
                    const float gh = am * a.Tm[o] + bm * (z * u);
                    b1 += gh * z;
                    dd += gh * u;
                    cms += gh * u + a.r[v] * gh * z;          // column sum of (direct + W^T part) of the Moran gradient
                }
                if (a.lam_geary > 0.f)
                    dd += gam * ((x * a.rc[v] - a.Z[o] - y) / m2 - A * z / (m2 * m2 * (Vf - 1.f)));
                a.B1[o] = b1;
                a.D[o] = dd;
            }
        }
        a.cmpart[(size_t)rb * a.Kp + k] = cms / Vf;
    }
}
