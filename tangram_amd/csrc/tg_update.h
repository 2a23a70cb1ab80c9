// tg_update.h -- K4: softmax backward + Adam (streaming and register-resident row kernels), row sums, regulariser scalars,
// MapperConstrained's filter, softmax statistics merge.  Included by tg_kernels.h.
#pragma once
// ----------------------------------------------------------------------------------------------
// K4: streaming softmax-backward + Adam (mapping_optimizer.py:394-396; torch _single_tensor_adam).
//   One workgroup per cell (row of M): dM = P (dP - r_c) [+ l1 sign(M) + 2 l2 M], Adam, store M, m, v,
//   and the (max, sum exp) of the NEW row for the next forward pass.  Pure HBM stream:
//   reads X, M, m, v (16 B / element), writes M, m, v (12 B / element); algorithmic traffic 24 B / element.
// ----------------------------------------------------------------------------------------------
struct TgUpdateArgs {
    const void* X; float* M; float* am; float* av;    // [C][Vp] (X fp32, or bf16 when X16)
    const float* rshift; const float* rinvz;          // softmax statistics of the CURRENT M
    const float* fgate; const float* dens_w;          // [C] or null
    const float* vcoef;                               // a_v at [2*Vr + v]
    const float* r;                                   // [C] row dots
    float* pair_out;                                  // [2][C] (max, Z) of the new row (cross-GPU exchange)
    float* rowq_out;                                  // [TGP1_N][C] row sums written by tg_adam_rowpass (FULL), else unused
    float* new_shift; float* new_invz; float* new_mul; float* new_scale;   // finalised statistics (single GPU) or null; new_mul = 1/Z and
                                                                           // new_scale = (max + ln Z) log2(e) are the forward's row constants
    int C, V, Vp, Vr, finalize;
    int c_begin;                                      // first cell of this launch (grid = number of cells)
    int c_end;                                        // one past the last cell of this launch
    float lambda_r, lambda_l1, lambda_l2;
    float step_size, bc2_sqrt, beta1, beta2, eps;
    int fin_on;                                       // 1: the LAST workgroup of the grid computes the history scalars instead of a row
    TgFinalizeArgs fin;                               //    (tg_loss_scalars; see tg_dghat_emit<SELF>)
    // spot shards on the peer transport with a step area (round 6): the row pairs (and the history workgroup's parts of the sums over spots)
    // are pushed into every rank's mailbox from the tail of the update kernel; tg_merge_stats_x polls for them
    int xch;                                          // 1: on; 0 everywhere else
    TgPeerLink link;
};

// ---- the arithmetic of one element, shared by both update kernels -------------------------------------------------------------
// Round 5 ("the update on a diet": round 4 counted 89 VALU instructions per element, SQ_INSTS_VALU): the softmax weight P of pass 1 is
// kept for pass 2 instead of a second exponential; Adam's square root and two divisions are tg_sqrt_cr / tg_div_by / tg_div_fr
// (tg_device.h) instead of hipcc's IEEE sequences; whole quads of a row run without per-element predication (only the one quad of a row
// that straddles V takes the masked path); the new row's (max, sum exp) is taken as a per-thread maximum first and ONE pass of
// exponentials against it, instead of an online rescale per quad; the wave reductions are DPP butterflies.
struct TgAdamK { float b1c, beta2, b2c, bc2, ibc2, eps, step; };
TG_DEV TgAdamK tg_adam_k(const TgUpdateArgs& a) {
    TgAdamK k;
    k.b1c = 1.f - a.beta1; k.beta2 = a.beta2; k.b2c = 1.f - a.beta2; k.bc2 = a.bc2_sqrt; k.ibc2 = 1.f / a.bc2_sqrt; k.eps = a.eps; k.step = a.step_size;
    return k;
}
// exp_avg.lerp_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2); denom = sqrt(v)/bc2 + eps; p.addcdiv_(m, denom, -step)
TG_DEV void tg_adam_elem(float gm, float& mo, float& m1, float& m2, const TgAdamK& k) {
    m1 = m1 + (gm - m1) * k.b1c;
    m2 = m2 * k.beta2 + k.b2c * gm * gm;
    const float den = tg_div_by(tg_sqrt_cr(m2), k.bc2, k.ibc2) + k.eps;
    mo = mo - k.step * tg_div_fr(m1, den);
}
// the row-uniform constants of the softmax backward
struct TgRowK { float sh, iz, fg, wc, logiz, lr, l1, l2; };
// dP of one element (mapping_optimizer.py:202 backward + the density term + the entropy term)
template <bool FULL> TG_DEV float tg_dp_elem(float x, float aq, float mo, const TgRowK& r) {
    if constexpr (!FULL) return x + aq * r.wc;               // (the filter gate exists in constrained mode only, which is FULL)
    else {
        float dp = r.fg * (x + aq * r.wc);
        if (r.lr != 0.f) dp -= r.lr * ((mo - r.sh) + r.logiz + 1.f);
        return dp;
    }
}
template <bool FULL> TG_DEV float tg_gm_elem(float p, float dp, float rc, float mo, const TgRowK& r) {
    float gm = p * (dp - rc);
    if constexpr (FULL) {
        if (r.l1 != 0.f) gm += r.l1 * ((mo > 0.f) ? 1.f : ((mo < 0.f) ? -1.f : 0.f));
        if (r.l2 != 0.f) gm += 2.f * r.l2 * mo;
    }
    return gm;
}
// butterfly all-reduce over the wave (fixed order)
TG_DEV float tg_wave_sum(float x) {
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) x += tg_bfly(x, m);
    return x;
}
TG_DEV float tg_wave_max(float x) {
#pragma unroll
    for (int m = 1; m <= 32; m <<= 1) x = tg_fmax(x, tg_bfly(x, m));
    return x;
}
// (max, sum exp) of the new row from the per-thread (max, sum exp against that max): the wave's maximum first, ONE rescale per thread,
// a plain wave sum; then the waves through LDS in wave order.  Thread 0 writes the pair (and, single GPU, the forward's row constants).
template <int NW>
TG_DEV void tg_row_stats_out(float tmax, float tsum, float* red, const TgUpdateArgs& a, int c, int t) {
    const int lane = t & 63, wave = t >> 6;
    const float wmax = tg_wave_max(tmax);
    const float wsum = tg_wave_sum(tsum * tg_exp(tmax - wmax));        // (a thread without elements: 0 * exp(-big) = 0)
    if (lane == 0) { red[wave * 2] = wmax; red[wave * 2 + 1] = wsum; }
    __syncthreads();
    if (wave == 0) {
        // every lane of wave 0 derives the pair (same reads, same order: the same bits in every lane); lane 0 writes it; on a spot shard of
        // the peer transport (a.xch) it also travels now -- lane r < world stores the maximum into rank r's mailbox, lane world + r the
        // sum -- and tg_merge_stats_x polls for it
        float mx = red[0];
        for (int w = 1; w < NW; ++w) mx = tg_fmax(mx, red[w * 2]);
        float z = 0.f;
        for (int w = 0; w < NW; ++w) z += red[w * 2 + 1] * tg_exp(red[w * 2] - mx);
        if (a.xch && lane < 2 * a.link.world) {
            const int r = lane % a.link.world, second = lane / a.link.world;
            tg_link_push_to(a.link, r, a.link.e1 + (second ? a.C : 0) + c, second ? z : mx);
        }
        if (lane == 0) {
            a.pair_out[c] = mx;
            a.pair_out[a.C + c] = z;
            if (a.finalize) {
                const float inz = 1.f / z;
                a.new_shift[c] = mx;
                a.new_invz[c] = inz;
                a.new_mul[c] = inz;                            // (the constrained filter is folded in by tg_merge_stats)
                a.new_scale[c] = (mx + tg_log(z)) * TG_LOG2E;
            }
        }
    }
}

// TEST HOOK (tg_debug_adam_math, tests/test_gpu_parity.py): the three helpers on arrays, so that a test can hold them to IEEE results
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_adam_math_probe(const float* a, const float* b, float bc, float* out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = tg_sqrt_cr(a[i]);
    out[n + i] = tg_div_fr(a[i], b[i]);
    out[2 * n + i] = tg_div_by(a[i], bc, 1.f / bc);
}

// NT = 256 threads per cell; 1 024 for a handful of long rows (clusters mode beyond 16 384 spots: with 18 workgroups the kernel is
// one dependent chain of V / (4 NT) trips per thread -- 81 us at 50 000 spots with 256 threads)
template <bool FULL, bool X16, bool STREAM, int NT = 256>
TG_KERNEL void TG_LAUNCH_BOUNDS(NT) tg_adam_update(TgUpdateArgs a) {
    constexpr int NW = NT / 64;
    TG_LDS_DECL;
    float* red = (float*)tg_lds;          // [NW waves][2]  (history workgroup: [NW][5])
    if (a.fin_on && blockIdx.x == gridDim.x - 1) {
        tg_loss_scalars<false>(a.fin, red);
        if (a.xch && threadIdx.x == 0 && a.fin.part_out) {     // spot shard, peer transport: this rank's parts of the sums over spots ride with
            tg_link_push(a.link, a.link.e1 + 2 * (size_t)a.C, a.fin.part_out[0]);           // the row pairs (tg_row_stats_out)
            tg_link_push(a.link, a.link.e1 + 2 * (size_t)a.C + 1, a.fin.part_out[1]);
        }
        return;
    }
    const int c = a.c_begin + blockIdx.x, t = threadIdx.x;
    TgRowK rk;
    rk.sh = a.rshift[c]; rk.iz = a.rinvz[c];
    rk.fg = a.fgate ? a.fgate[c] : 1.f;
    rk.wc = a.dens_w ? a.dens_w[c] : 1.f;
    rk.lr = a.lambda_r; rk.l1 = a.lambda_l1; rk.l2 = a.lambda_l2;
    rk.logiz = (FULL && a.lambda_r != 0.f) ? tg_log(rk.iz) : 0.f;
    const float rc = a.r[c];
    const TgAdamK ak = tg_adam_k(a);
    const size_t row = (size_t)c * a.Vp;
    // (max, sum exp) of the new row, per thread: every trip rescales once against the trip's maximum (4 elements)
    float lmax = TG_NEG_BIG, lsum = 0.f;
    for (int v = 4 * t; v < a.V; v += 4 * NT) {
        f32x4 xq;
        if constexpr (X16) {
            const u32x2 xp = tg_ld_stream<STREAM>((const u32x2*)((const unsigned short*)a.X + row + v));
            xq = f32x4{tg_bf16_lo_to_f32(xp[0]), tg_bf16_hi_to_f32(xp[0]), tg_bf16_lo_to_f32(xp[1]), tg_bf16_hi_to_f32(xp[1])};
        } else {
            xq = tg_ld_stream<STREAM>((const f32x4*)((const float*)a.X + row + v));
        }
        // (streamed once per iteration: non-temporal accesses keep these 8.4 GB from churning L2 / MALL; measured -9 %)
        f32x4 mq = tg_ld_stream<STREAM>((const f32x4*)(a.M + row + v));
        f32x4 m1 = tg_ld_stream<STREAM>((const f32x4*)(a.am + row + v));
        f32x4 m2 = tg_ld_stream<STREAM>((const f32x4*)(a.av + row + v));
        const f32x4 aq = *(const f32x4*)(a.vcoef + 2 * (size_t)a.Vr + v);
        float qmax = TG_NEG_BIG;
        if (v + 4 <= a.V) {                                    // a whole quad: no per-element predication
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mo = mq[e], e1 = m1[e], e2 = m2[e];
                const float p = tg_exp(mo - rk.sh) * rk.iz;
                const float gm = tg_gm_elem<FULL>(p, tg_dp_elem<FULL>(xq[e], aq[e], mo, rk), rc, mo, rk);
                tg_adam_elem(gm, mo, e1, e2, ak);
                mq[e] = mo; m1[e] = e1; m2[e] = e2;
                qmax = tg_fmax(qmax, mo);
            }
        } else {                                               // the quad that straddles V (at most one per row)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = (v + e) < a.V;
                float mo = mq[e], e1 = m1[e], e2 = m2[e];
                const float p = tg_exp(mo - rk.sh) * rk.iz;
                const float gm = tg_gm_elem<FULL>(p, tg_dp_elem<FULL>(xq[e], aq[e], mo, rk), rc, mo, rk);
                tg_adam_elem(gm, mo, e1, e2, ak);
                if (ok) { mq[e] = mo; m1[e] = e1; m2[e] = e2; qmax = tg_fmax(qmax, mo); }
            }
        }
        tg_st_stream<STREAM>(mq, (f32x4*)(a.M + row + v));
        tg_st_stream<STREAM>(m1, (f32x4*)(a.am + row + v));
        tg_st_stream<STREAM>(m2, (f32x4*)(a.av + row + v));
        const float nmx = tg_fmax(lmax, qmax);
        float qs = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) qs += ((v + e) < a.V) ? tg_exp(mq[e] - nmx) : 0.f;
        lsum = lsum * tg_exp(lmax - nmx) + qs;
        lmax = nmx;
    }
    tg_row_stats_out<NW>(lmax, lsum, red, a, c, t);
}

// K4': the same update for the single-GPU schedule, with the softmax-backward row dot taken in the SAME kernel:
// one workgroup of NT threads per cell holds its whole row of M, X and both moments in registers (NQ float4 per thread
// per array, V <= 4 * NT * NQ; every load of the row is in flight before the first use),
//   pass 1: P, dP -> r_c (block reduction; plus the entropy / L1 / L2 / filter row sums when FULL); P stays in registers,
//   pass 2: dM = P (dP - r_c), Adam, stores, then (max, sum exp) of the new row.
// HBM traffic is that of tg_adam_update; tg_bwd_kernel no longer reads M nor writes row-dot partials.
template <bool FULL, bool X16, int NQ, int NT, bool STREAM>
TG_DEV void tg_adam_rowpass_body(const TgUpdateArgs& a) {
    TG_LDS_DECL;
    constexpr int NW = NT / 64;
    constexpr int NP = FULL ? (int)TGP1_N : 1;
    float* red = (float*)tg_lds;          // [NW waves][NP] (pass 1), then [NW][2] behind it (row statistics): no reuse, one barrier each
    float* red2 = red + NW * NP;
    if (a.fin_on && blockIdx.x == gridDim.x - 1) { tg_loss_scalars<false>(a.fin, red); return; }
    const int c = a.c_begin + blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    TgRowK rk;
    rk.sh = a.rshift[c]; rk.iz = a.rinvz[c];
    rk.fg = a.fgate ? a.fgate[c] : 1.f;
    rk.wc = a.dens_w ? a.dens_w[c] : 1.f;
    rk.lr = a.lambda_r; rk.l1 = a.lambda_l1; rk.l2 = a.lambda_l2;
    rk.logiz = (FULL && a.lambda_r != 0.f) ? tg_log(rk.iz) : 0.f;
    const size_t row = (size_t)c * a.Vp;
    const float* avec = a.vcoef + 2 * (size_t)a.Vr;
    f32x4 mq[NQ];
    typename std::conditional<X16, u32x2, f32x4>::type xr[NQ];
    auto xval = [&](int q) -> f32x4 {
        if constexpr (X16) return f32x4{tg_bf16_lo_to_f32(xr[q][0]), tg_bf16_hi_to_f32(xr[q][0]), tg_bf16_lo_to_f32(xr[q][1]), tg_bf16_hi_to_f32(xr[q][1])};
        else return xr[q];
    };
    // ---- pass 1: loads (all in flight together) and the row sums
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = 4 * (t + NT * q);
        const int vl = v < a.V ? v : 0;
        mq[q] = tg_ld_stream<STREAM>((const f32x4*)(a.M + row + vl));      // streamed once: non-temporal (see tg_adam_update)
        if constexpr (X16) xr[q] = tg_ld_stream<STREAM && !TG_X_TEMPORAL>((const u32x2*)((const unsigned short*)a.X + row + vl));
        else xr[q] = tg_ld_stream<STREAM && !TG_X_TEMPORAL>((const f32x4*)((const float*)a.X + row + vl));
    }
    // the moments travel while pass 1 computes -- except that the variants at the 128-register limit (4 waves per SIMD,
    // NT * NQ = 2560) request the second moment only behind the pass-1 sums, under the block reduction
    constexpr bool LATE_M2 = STREAM && (NT * NQ == 2560);
    f32x4 m1q[NQ], m2q[NQ], pq[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = 4 * (t + NT * q);
        const int vl = v < a.V ? v : 0;
        m1q[q] = tg_ld_stream<STREAM>((const f32x4*)(a.am + row + vl));
        if constexpr (!LATE_M2) m2q[q] = tg_ld_stream<STREAM>((const f32x4*)(a.av + row + vl));
    }
    float acc[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = 4 * (t + NT * q);
        if (v >= a.V) continue;
        const f32x4 aq = *(const f32x4*)(avec + v);
        const f32x4 xq = xval(q);
        auto elem = [&](int e, bool ok) {                      // `ok` is the constant true on the whole-quad path: no selects there
            const float mo = mq[q][e];
            const float p = tg_exp(mo - rk.sh) * rk.iz;
            pq[q][e] = p;
            const float dp = tg_dp_elem<FULL>(xq[e], aq[e], mo, rk);
            if constexpr (FULL) {
                if (rk.lr != 0.f) acc[TGP1_ENT % NP] += ok ? p * ((mo - rk.sh) + rk.logiz) : 0.f;
                acc[TGP1_Q % NP] += ok ? p * xq[e] : 0.f;
                acc[TGP1_PA % NP] += ok ? p * aq[e] : 0.f;
                acc[TGP1_L1 % NP] += ok ? fabsf(mo) : 0.f;
                acc[TGP1_L2 % NP] += ok ? mo * mo : 0.f;
            }
            acc[TGP1_R] += ok ? p * dp : 0.f;
        };
        if (v + 4 <= a.V) {                                    // every quad but the one that straddles V
#pragma unroll
            for (int e = 0; e < 4; ++e) elem(e, true);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) elem(e, (v + e) < a.V);
        }
    }
    if constexpr (LATE_M2) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int v = 4 * (t + NT * q);
            m2q[q] = tg_ld_stream<STREAM>((const f32x4*)(a.av + row + (v < a.V ? v : 0)));
        }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const float x = tg_wave_sum(acc[i]);
        if (lane == 0) red[wave * NP + i] = x;
    }
    __syncthreads();
    float rc = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) rc += red[w * NP + TGP1_R];
    if (FULL && t < NP) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) x += red[w * NP + t];
        a.rowq_out[(size_t)t * a.C + c] = x;
    }
    // ---- pass 2: Adam on the registers held since pass 1
    const TgAdamK ak = tg_adam_k(a);
    float tmax = TG_NEG_BIG;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = 4 * (t + NT * q);
        if (v >= a.V) continue;
        f32x4 m1 = m1q[q], m2 = m2q[q], mo4 = mq[q];
        const f32x4 aq = *(const f32x4*)(avec + v);
        const f32x4 xq = xval(q);
        if (v + 4 <= a.V) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float mo = mo4[e], e1 = m1[e], e2 = m2[e];
                const float gm = tg_gm_elem<FULL>(pq[q][e], tg_dp_elem<FULL>(xq[e], aq[e], mo, rk), rc, mo, rk);
                tg_adam_elem(gm, mo, e1, e2, ak);
                mo4[e] = mo; m1[e] = e1; m2[e] = e2;
                tmax = tg_fmax(tmax, mo);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = (v + e) < a.V;
                float mo = mo4[e], e1 = m1[e], e2 = m2[e];
                const float gm = tg_gm_elem<FULL>(pq[q][e], tg_dp_elem<FULL>(xq[e], aq[e], mo, rk), rc, mo, rk);
                tg_adam_elem(gm, mo, e1, e2, ak);
                if (ok) { mo4[e] = mo; m1[e] = e1; m2[e] = e2; tmax = tg_fmax(tmax, mo); }
            }
        }
        tg_st_stream<STREAM>(mo4, (f32x4*)(a.M + row + v));
        tg_st_stream<STREAM>(m1, (f32x4*)(a.am + row + v));
        tg_st_stream<STREAM>(m2, (f32x4*)(a.av + row + v));
        mq[q] = mo4;
    }
    // ---- (max, sum exp) of the new row: the thread's maximum is known, one exponential per element against it
    float tsum = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int v = 4 * (t + NT * q);
        if (v >= a.V) continue;
        if (v + 4 <= a.V) {
#pragma unroll
            for (int e = 0; e < 4; ++e) tsum += tg_exp(mq[q][e] - tmax);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) tsum += ((v + e) < a.V) ? tg_exp(mq[q][e] - tmax) : 0.f;
        }
    }
    tg_row_stats_out<NW>(tmax, tsum, red2, a, c, t);
}

// ----------------------------------------------------------------------------------------------
// small per-row kernels
// ----------------------------------------------------------------------------------------------
// r_c = sum over spot tiles of the phase-1 partials; also the scalar regulariser sums
struct TgRowsumArgs {
    const float* part; int nvt; int C; int np;
    float* rowq;               // [np][C] summed partials (row 0 = r_c)
    int c_begin, c_end;        // cells handled by this launch
    int xch; TgPeerLink link;  // xch (spot shard, peer transport with a step area): the thread that owns a cell's sum pushes it into every rank's
                               // mailbox and stores the rank-order sum of the world's granules: what an all-reduce after this kernel delivers
};
// 16 cells x 16 groups of spot tiles per workgroup: a cell's partials p = g, g + 16, ... side by side, then the groups in fixed order
// (one thread per cell walking all V / 128 partials took 61 us at 50 000 spots and 18 rows of M)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_rowsum_parts(TgRowsumArgs a) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;                                 // [16][16]
    float* loc = red + 256;                                      // xch: [np][16] this rank's sums of the block's cells
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4;
    // (a grid-stride walk over the blocks of 16 cells: one trip unless the grid was cut short -- ranks sharing a device, tg_polling_grid)
    for (int blk = blockIdx.x; a.c_begin + blk * 16 < a.c_end; blk += (int)gridDim.x) {
        const int c = a.c_begin + blk * 16 + r;
        for (int q = 0; q < a.np; ++q) {
            float s = 0.f;
            if (c < a.c_end)
                for (int p = g; p < a.nvt; p += 16) s += a.part[((size_t)p * a.np + q) * a.C + c];
            red[g * 16 + r] = s;
            __syncthreads();
            if (g == 0 && c < a.c_end) {
                float t = 0.f;
                for (int i = 0; i < 16; ++i) t += red[i * 16 + r];
                if (a.xch) loc[q * 16 + r] = t; else a.rowq[(size_t)q * a.C + c] = t;
            }
            __syncthreads();
        }
        if (a.xch) {                                             // the exchange, every (sum, cell) of the block at once: one round trip
            const int q = threadIdx.x >> 4;
            if (q < a.np && c < a.c_end) {
                const size_t place = a.link.e3 + (size_t)q * a.C + c;
                tg_link_push(a.link, place, loc[q * 16 + r]);
                a.rowq[(size_t)q * a.C + c] = tg_link_sum(a.link, place);
            }
            __syncthreads();
        }
    }
}

// entropy / L1 / L2 scalars (mapping_optimizer.py:224-231) from the per-row sums -> history row
struct TgHistRegArgs { const float* rowq; int C; float* hist; float lambda_r, lambda_l1, lambda_l2; int constrained; };

TG_DEV void tg_hist_regs_body(const TgHistRegArgs& a) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    float e = 0.f, l1 = 0.f, l2 = 0.f;
    constexpr int U = 8;                                   // eight cells per trip, loads first (see tg_filter_body); same summation order
    for (int c0 = threadIdx.x; c0 < a.C; c0 += 1024 * U) {
        float ve[U], v1[U], v2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 1024 * u;
            const bool ok = c < a.C;
            ve[u] = ok ? a.rowq[(size_t)TGP1_ENT * a.C + c] : 0.f;
            v1[u] = ok ? a.rowq[(size_t)TGP1_L1 * a.C + c] : 0.f;
            v2[u] = ok ? a.rowq[(size_t)TGP1_L2 * a.C + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (c0 + 1024 * u >= a.C) continue;
            e += ve[u]; l1 += v1[u]; l2 += v2[u];
        }
    }
    e = tg_block_sum_1024(e, red);
    l1 = tg_block_sum_1024(l1, red);
    l2 = tg_block_sum_1024(l2, red);
    if (threadIdx.x == 0) {
        float total = a.hist[TGH_TOTAL];
        // Mapper reports -sum P log P (:224-225); MapperConstrained reports +sum P log P and subtracts it (:526,:575)
        if (a.lambda_r != 0.f) { a.hist[TGH_ENTROPY] = a.constrained ? e : -e; total += a.lambda_r * (-e); }
        if (a.lambda_l1 != 0.f) { a.hist[TGH_L1] = l1; total += a.lambda_l1 * l1; }
        if (a.lambda_l2 != 0.f) { a.hist[TGH_L2] = l2; total += a.lambda_l2 * l2; }
        a.hist[TGH_TOTAL] = total;
    }
}

// ----------------------------------------------------------------------------------------------
// MapperConstrained filter F (mapping_optimizer.py:490-493, :507, :528-532, :607): one block.
//   init  : f = sigmoid(F), fsum = sum f
//   update: df_c = Q_c + PA_c + lambda_d * dsum / fsum + lambda_count * sign(fsum - target) + lambda_f (1 - 2 f_c)
//           dF = df f (1 - f); Adam(F); then f, fsum of the NEW F; count / f_reg scalars of the OLD f -> history
// ----------------------------------------------------------------------------------------------
struct TgFilterArgs {
    float* F; float* mF; float* vF;      // [C] filter logits and Adam moments
    float* fgate;                        // [C] sigmoid(F)
    float* fsum;                         // [2]: fsum, scratch
    const float* rowq;                   // [TGP1_N][C]
    const float* dsum;                   // [1] sum of the density prior over ALL spots (set-up; all-reduced over spot shards)
    float* hist;
    int C, do_update, has_density;
    float lambda_d, lambda_count, lambda_f_reg, target_count;
    float step_size, bc2_sqrt, beta1, beta2, eps;
};
TG_DEV void tg_filter_body(const TgFilterArgs& a) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const int t = threadIdx.x;
    // One block; a thread owns the cells t, t + 1024, ... in every phase, so the phases fuse per cell (old gate -> f_reg term and
    // gradient -> Adam on F -> new gate) and only the two block sums synchronise.  Round 5: eight cells per trip with every load of
    // the trip requested before the first use (the one-cell-per-trip loops were three chains of ~30 dependent global loads at
    // 30 000 cells: 43 us per step of constrained mode); per-thread summation order unchanged, i.e. the same bits.
    constexpr int U = 8;
    const float fsum = a.do_update ? a.fsum[0] : 0.f;
    const float dsum = (a.do_update && a.has_density) ? a.dsum[0] : 0.f;
    const float cnt = fsum - a.target_count;
    const float sgn = (cnt > 0.f) ? 1.f : ((cnt < 0.f) ? -1.f : 0.f);
    float fr = 0.f, fs = 0.f;
    for (int c0 = t; c0 < a.C; c0 += 1024 * U) {
        float Fv[U], f[U], q[U], pa[U], m1[U], m2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 1024 * u;
            const bool ok = c < a.C;
            Fv[u] = ok ? a.F[c] : 0.f;
            if (a.do_update) {
                f[u] = ok ? a.fgate[c] : 0.f;
                q[u] = ok ? a.rowq[(size_t)TGP1_Q * a.C + c] : 0.f;
                pa[u] = ok ? a.rowq[(size_t)TGP1_PA * a.C + c] : 0.f;
                m1[u] = ok ? a.mF[c] : 0.f;
                m2[u] = ok ? a.vF[c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 1024 * u;
            if (c >= a.C) continue;
            float Fn = Fv[u];
            if (a.do_update) {
                fr += f[u] - f[u] * f[u];
                float df = q[u] + pa[u];
                if (a.has_density) df += a.lambda_d * dsum / fsum;
                df += a.lambda_count * sgn + a.lambda_f_reg * (1.f - 2.f * f[u]);
                const float g = df * f[u] * (1.f - f[u]);
                const float e1 = m1[u] + (g - m1[u]) * (1.f - a.beta1);
                const float e2 = m2[u] * a.beta2 + (1.f - a.beta2) * g * g;
                const float den = sqrtf(e2) / a.bc2_sqrt + a.eps;
                a.mF[c] = e1; a.vF[c] = e2;
                Fn = Fv[u] - a.step_size * (e1 / den);
                a.F[c] = Fn;
            }
            const float fn = 1.f / (1.f + tg_exp(-Fn));
            a.fgate[c] = fn;
            fs += fn;
        }
    }
    if (a.do_update) {
        const float freg = tg_block_sum_1024(fr, red);
        if (t == 0) {
            a.hist[TGH_COUNT] = fabsf(cnt);
            a.hist[TGH_FREG] = freg;
            a.hist[TGH_TOTAL] += a.lambda_count * fabsf(cnt) + a.lambda_f_reg * freg;
        }
    }
    const float fsum_new = tg_block_sum_1024(fs, red);          // (its barriers also order every thread's read of fsum above before this write)
    if (t == 0) a.fsum[0] = fsum_new;
}
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_filter_kernel(TgFilterArgs a) { tg_filter_body(a); }
// batched (tg_batch of MapperConstrained handles): Adam step constants and the history row travel by value
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_filter_kernel_b(const TgFilterArgs* argv, TgStepVar var, float* const* scratch_rows) {
    TgFilterArgs a = argv[blockIdx.z];
    a.step_size = var.step_size; a.bc2_sqrt = var.bc2_sqrt;
    a.hist = (var.hist_row >= 0 && a.hist) ? a.hist + var.hist_row * TGH_NTERMS : scratch_rows[blockIdx.z];
    tg_filter_body(a);
}

// merge (max, sum exp) partials over `nparts` -> rshift = max, rinvz = 1/Z ; optional raw output.
// Spot shards (nparts = ranks, `part` = the all-gathered blocks of `stride` floats: [2][C] pairs + TG_PAIR_TAIL history
// scalars): thread 0 of block 0 also turns this rank's history row into the GLOBAL one -- the terms that are sums over spots
// (voxel score, KL) arrive as per-rank partials in the tail of every block and are added in rank order (deterministic).
#define TG_PAIR_TAIL 64        // floats appended to the [2][C] statistics block of a rank: [0] = vg partial, [1] = KL partial
struct TgMergeArgs {
    const float* part;         // [nparts] blocks of `stride` floats: max at [c], sum exp at [C + c]
    int nparts, C;
    size_t stride;
    float* rshift; float* rinvz;       // final (may be null when only the local pair is wanted)
    float* pair_out;           // [2][C] local (max, Z) for the cross-GPU exchange, or null
    const float* fgate; float* rmul; float* rscale;   // forward row constants: f_c / Z_c and (max + ln Z - ln f_c) * log2(e)
    float* hist; int rank;     // spot shards: history row to complete with the global spot sums (or null), this rank's index
    float lambda_g2, lambda_d; int has_density;
};
// PEER (spot shard on the peer transport, round 6): the ranks' blocks are not in `part` but arrive as granules in this rank's mailbox,
// pushed from the tail of every rank's update kernel (tg_row_stats_out, the history workgroup): the poll of that exchange happens
// here, at the head of the kernel that consumes it.  Same maxima, same sums in the same (rank) order as the gathered form.
template <bool PEER>
TG_DEV void tg_merge_stats_body(const TgMergeArgs& a, const TgPeerLink* link = nullptr) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if constexpr (PEER) {
        if (a.hist && blockIdx.x == 0 && threadIdx.x == 0) {
            float vg, kl;
            tg_link_sum2(*link, link->e1 + 2 * (size_t)a.C, link->e1 + 2 * (size_t)a.C + 1, vg, kl);
            float total = a.hist[TGH_TOTAL];
            if (a.lambda_g2 != 0.f) { a.hist[TGH_VG] = vg; total -= a.lambda_g2 * vg; }
            if (a.has_density) { a.hist[TGH_KL] = kl; total += a.lambda_d * kl; }
            a.hist[TGH_TOTAL] = total;
            // an exchange of this rank has given up waiting for a peer (status word raised): the step's numbers are garbage -- say so in the
            // history row itself, not only in tg_comm_peer_status (a caller that prints or consumes rows during a long run sees NaN at once)
            if (tg_sys_load_u32((const unsigned*)link->box[link->rank]) != 0u)
                for (int i = 0; i < TGH_NTERMS; ++i) a.hist[i] = __builtin_nanf("");
        }
        // (a grid-stride walk: ranks that SHARE a device -- the one-GPU tests -- poll with few workgroups, see tg_polling_grid)
        for (; c < a.C; c += (int)gridDim.x * 256) {
        float pm[TG_PEER_MAX], pz[TG_PEER_MAX];
        float mx = TG_NEG_BIG;
#pragma unroll
        for (int h = 0; h < TG_PEER_MAX / 8; ++h) {
            if (8 * h >= a.nparts) break;
            float vm[8], vz[8];
            tg_link_get8x2(*link, link->e1 + c, link->e1 + a.C + c, 8 * h, vm, vz);
#pragma unroll
            for (int q = 0; q < 8; ++q) { pm[8 * h + q] = (8 * h + q < a.nparts) ? vm[q] : TG_NEG_BIG; pz[8 * h + q] = vz[q]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) mx = tg_fmax(mx, pm[8 * h + q]);
        }
        float z = 0.f;
#pragma unroll
        for (int p = 0; p < TG_PEER_MAX; ++p) if (p < a.nparts) z += pz[p] * tg_exp(pm[p] - mx);
        if (a.pair_out) { a.pair_out[c] = mx; a.pair_out[a.C + c] = z; }
        if (a.rshift) {
            const float iz = 1.f / z;
            a.rshift[c] = mx;
            a.rinvz[c] = iz;
            a.rmul[c] = (a.fgate ? a.fgate[c] : 1.f) * iz;
            a.rscale[c] = (mx + tg_log(z) - (a.fgate ? tg_log(a.fgate[c]) : 0.f)) * TG_LOG2E;
        }
        }
        return;
    }
    if (a.hist && blockIdx.x == 0 && threadIdx.x == 0) {
        const float* tail = a.part + 2 * (size_t)a.C;
        float vg = 0.f, kl = 0.f;
        for (int p = 0; p < a.nparts; ++p) { vg += tail[p * a.stride]; kl += tail[p * a.stride + 1]; }
        float total = a.hist[TGH_TOTAL];                 // so far: every term that is not a sum over spots (tg_loss_scalars)
        if (a.lambda_g2 != 0.f) { a.hist[TGH_VG] = vg; total -= a.lambda_g2 * vg; }
        if (a.has_density) { a.hist[TGH_KL] = kl; total += a.lambda_d * kl; }
        a.hist[TGH_TOTAL] = total;
    }
    if (c >= a.C) return;
    float pm[8], pz[8];                                    // all loads of a cell in flight at once (<= 8 parts per trip)
    float mx = TG_NEG_BIG;
    for (int p0 = 0; p0 < a.nparts; p0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) pm[q] = (p0 + q < a.nparts) ? a.part[(size_t)(p0 + q) * a.stride + c] : TG_NEG_BIG;
#pragma unroll
        for (int q = 0; q < 8; ++q) mx = tg_fmax(mx, pm[q]);
    }
    float z = 0.f;
    for (int p0 = 0; p0 < a.nparts; p0 += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool on = p0 + q < a.nparts;
            pm[q] = on ? a.part[(size_t)(p0 + q) * a.stride + c] : TG_NEG_BIG;
            pz[q] = on ? a.part[(size_t)(p0 + q) * a.stride + a.C + c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) if (p0 + q < a.nparts) z += pz[q] * tg_exp(pm[q] - mx);
    }
    if (a.pair_out) { a.pair_out[c] = mx; a.pair_out[a.C + c] = z; }
    if (a.rshift) {
        const float iz = 1.f / z;
        a.rshift[c] = mx;
        a.rinvz[c] = iz;
        a.rmul[c] = (a.fgate ? a.fgate[c] : 1.f) * iz;
        a.rscale[c] = (mx + tg_log(z) - (a.fgate ? tg_log(a.fgate[c]) : 0.f)) * TG_LOG2E;
    }
}

TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_merge_stats(TgMergeArgs a) { tg_merge_stats_body<false>(a); }
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_merge_stats_b(const TgMergeArgs* argv) { tg_merge_stats_body<false>(argv[blockIdx.z]); }
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_merge_stats_x(TgMergeArgs a, TgPeerLink link) { tg_merge_stats_body<true>(a, &link); }

// forward row constant of the bf16 path WITHOUT the constrained-mode filter: (max + ln Z) * log2(e)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_plain_rscale(const float* rshift, const float* rinvz, int C, float* out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) out[c] = (rshift[c] - tg_log(rinvz[c])) * TG_LOG2E;
}

// one block per row: (max, sum exp) of a row of M (initialisation / fallback path)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_row_stats(const float* M, int C, int V, int Vp, float* part /*[1][2][C]*/) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;
    const int c = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float* row = M + (size_t)c * Vp;
    float mx = TG_NEG_BIG;
    for (int v = t; v < V; v += 256) mx = tg_fmax(mx, row[v]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = tg_fmax(mx, tg_shfl_xor(mx, m));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = tg_fmax(tg_fmax(red[0], red[1]), tg_fmax(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int v = t; v < V; v += 256) s += tg_exp(row[v] - mx);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += tg_shfl_xor(s, m);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (t == 0) { part[c] = mx; part[C + c] = red[0] + red[1] + red[2] + red[3]; }
}

// P_out[c][v] = softmax(M)[c][v]  (mapping_optimizer.py:407), dense pitch V; one block per cell
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_softmax_out(const float* M, const float* rshift, const float* rinvz,
                                                    int C, int V, int Vp, float* out) {
    const int c = blockIdx.x;
    const float sh = rshift[c], iz = rinvz[c];
    for (int v = threadIdx.x; v < V; v += 256) out[(size_t)c * V + v] = tg_exp(M[(size_t)c * Vp + v] - sh) * iz;
}
