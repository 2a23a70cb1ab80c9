// tg_kernels.h -- hand-written gfx950 kernels for one Tangram mapping iteration.
//
// Math (reference tangram/mapping_optimizer.py:189-309, :358-408):
//   P = softmax(M, axis=1)            C x V      (:201)
//   Ghat = P^T [S | 1]                V x Kp     (:202; the extra "ones" column yields sum_c P_cv, :217)
//   gv = mean_k cos(Ghat[:,k], G[:,k]); vg = mean_v cos(Ghat[v,:], G[v,:]); KL(d || colsum/C)   (:205-221)
//   dGhat = d(loss)/dGhat             V x Kp
//   X = S dGhat^T                     C x V      (autograd of :202)
//   dP = X + a_v w_c - lambda_r (log P + 1) ; r_c = sum_v P dP ; dM = P (dP - r) + l1/l2 terms
//   Adam(M, dM)                                   (:373,:396)
//
// Kernel map (one iteration = 6 launches on one GPU without regularisers, no host synchronisation):
//   tg_fwd_kernel        softmax-apply fused into the A-operand path of the MFMA GEMM P^T S (split over cell ranges)
//   tg_ghat_reduce       sum the splits, per-gene / per-spot cosine statistics partials
//   tg_gene_reduce       deterministic second stage of the per-gene statistics
//   tg_dghat_emit        dGhat in matrix-core operand format (<SELF>: derives alpha_k, beta_k, a_v itself)
//   tg_bwd_kernel        MFMA GEMM X = S dGhat^T, X stored as full row segments (+ row-dot partials on a spot shard)
//   tg_adam_rowpass      one workgroup per cell row: row dot, softmax backward, Adam, statistics of the new row,
//                        plus ONE extra workgroup that writes the history row (tg_loss_scalars)
//   also: tg_loss_finalize (history + coefficients as a kernel of its own: spatial terms, spot shards),
//   tg_rowsum_parts + tg_adam_update (two-kernel update: spot shards, rows > 16 384 spots), tg_hist_regs, tg_filter_kernel
//   (MapperConstrained), tg_merge_stats, tg_spmm / tg_ct_* / tg_ac_* (spatial terms), tg_row_entropy / tg_val_finalize.
//
// Data layout in HBM: M, Adam m, Adam v are C x Vp fp32 row-major (Vp = V rounded up to 64);
// S is kept twice in operand format: St [Kp][Cp] (cell index contiguous, for the forward contraction
// over cells) and Sk [Cr][Kp] (gene index contiguous, for the backward contraction over genes).
#pragma once
#include "tg_device.h"

// Loads / stores of the arrays that are streamed once per iteration (M, X, Adam m, v).  STREAM (compile time; a run-time
// `flag ? nontemporal_load : load` makes hipcc issue BOTH loads and select): non-temporal, so that 8.4 GB per iteration do
// not churn L2 / MALL (-9 % on the update kernel at 30k x 10k); problems whose four arrays fit the 256 MB MALL keep ordinary
// accesses and find M still cached in the next forward pass (+5 % there, profiles/r01 run34).
#ifndef TG_X_TEMPORAL
#define TG_X_TEMPORAL 0       // experiment switch (scripts/build_variant.sh -DTG_X_TEMPORAL=1): the backward product X stored and re-read with
#endif                        // ordinary instead of non-temporal accesses, so that the update may find its tail in the Infinity Cache (review r04, item 6)
template <bool STREAM, class T> TG_DEV T tg_ld_stream(const T* p) {
    if constexpr (STREAM) return __builtin_nontemporal_load(p); else return *p;
}
template <bool STREAM, class T> TG_DEV void tg_st_stream(const T& v, T* p) {
    if constexpr (STREAM) __builtin_nontemporal_store(v, p); else *p = v;
}

#ifndef TG_FWD_STAGGER
#define TG_FWD_STAGGER (-1)   // forward kernel A-operand staging schedule: 0 = block after the MFMAs, 1 / 2 = phase-shifted halves
#endif                        // (waves 0-3 / 4-7 early); -1 = per precision (measured, see tg_fwd_kernel)
#define TG_NEG_BIG (-3.0e38f)
#define TG_COS_EPS 1e-8f

// history row layout (floats)
enum { TGH_TOTAL = 0, TGH_MAIN, TGH_VG, TGH_KL, TGH_ENTROPY, TGH_L1, TGH_L2, TGH_NB, TGH_CT, TGH_COUNT, TGH_FREG,
       TGH_GETIS, TGH_MORAN, TGH_GEARY, TGH_NTERMS = 16 };

// what changes from step to step in a batch of mappings (tg_batch; everything else is constant per mapping and lives in argument arrays)
struct TgStepVar { float step_size, bc2_sqrt; long long hist_row; };    // hist_row < 0: no history wanted


#include "tg_peer.h"
#include "tg_gemm.h"
#include "tg_stats.h"
#include "tg_spatial.h"
#include "tg_small.h"
#include "tg_update.h"
#include "tg_setup.h"

// ----------------------------------------------------------------------------------------------
// Kernel entry points of the iteration.  Every kernel of the single-GPU Mapper step exists twice: with its arguments by value
// (one mapping per launch) and as `_b` (BATCHED): the arguments of B independent mappings of one shape sit in an
// array in device memory, blockIdx.z picks the mapping -- B cross-validation folds / seeds advance in ONE launch per kernel
// instead of B (clusters-mode problems, 18 x 250 x 9852, are launch-bound: one fold leaves > 90 % of the chip idle).
// ----------------------------------------------------------------------------------------------
struct TgGeneReduceArgs { const float* genepart; int nrb, Kp; float* genestat; };
// what changes from step to step in a batch (everything else is constant per mapping and lives in the argument arrays)

template <class PR, class GE> TG_KERNEL void TG_LAUNCH_BOUNDS2(GE::NT, 2) tg_fwd_kernel(TgFwdArgs a) { tg_fwd_body<PR, GE>(a); }
template <class PR, class GE> TG_KERNEL void TG_LAUNCH_BOUNDS2(GE::NT, 2) tg_fwd_kernel_b(const TgFwdArgs* argv) {
    const TgFwdArgs a = argv[blockIdx.z];            // (a by-value copy: referencing the block in memory cost the wide geometry 16 spilled registers)
    tg_fwd_body<PR, GE>(a);
}
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ghat_reduce(TgGhatReduceArgs a) { tg_ghat_reduce_body(a); }
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_ghat_reduce_b(const TgGhatReduceArgs* argv) { tg_ghat_reduce_body(argv[blockIdx.z]); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_gene_reduce(const float* genepart, int nrb, int Kp, float* genestat) { tg_gene_reduce_body<64>(genepart, nrb, Kp, genestat); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_gene_reduce_x(const float* genepart, int nrb, int Kp, float* genestat, TgPeerLink link) { tg_gene_reduce_body<64>(genepart, nrb, Kp, genestat, &link); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_gene_reduce_tall_x(const float* genepart, int nrb, int Kp, float* genestat, TgPeerLink link) { tg_gene_reduce_body<16>(genepart, nrb, Kp, genestat, &link); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_gene_reduce_tall(const float* genepart, int nrb, int Kp, float* genestat) { tg_gene_reduce_body<16>(genepart, nrb, Kp, genestat); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_gene_reduce_b(const TgGeneReduceArgs* argv) {
    const TgGeneReduceArgs a = argv[blockIdx.z];
    tg_gene_reduce_body<64>(a.genepart, a.nrb, a.Kp, a.genestat);
}
template <class PR, bool EXTRA, bool SELF> TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_dghat_emit(TgEmitArgs a) { tg_dghat_emit_body<PR, EXTRA, SELF>(a); }
template <class PR> TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_dghat_emit_b(const TgEmitArgs* argv) { tg_dghat_emit_body<PR, false, true>(argv[blockIdx.z]); }
template <class PR, class GE, bool FULL, bool ROWDOT, bool STREAM>
TG_KERNEL void TG_LAUNCH_BOUNDS2(GE::NT, 2) tg_bwd_kernel(TgBwdArgs a) { tg_bwd_body<PR, GE, FULL, ROWDOT, STREAM>(a); }
template <class PR, class GE>
TG_KERNEL void TG_LAUNCH_BOUNDS2(GE::NT, 2) tg_bwd_kernel_b(const TgBwdArgs* argv) { tg_bwd_body<PR, GE, false, false, false>(argv[blockIdx.z]); }
template <bool FULL, bool X16, int NQ, int NT, bool STREAM>
TG_KERNEL void TG_LAUNCH_BOUNDS2(NT, (NT * NQ <= 2560 ? 4 : 2)) tg_adam_rowpass(TgUpdateArgs a) { tg_adam_rowpass_body<FULL, X16, NQ, NT, STREAM>(a); }
// (batched: the argument block of the mapping lives in SGPRs, which leaves the 5-quad variant short at 128 VGPRs: 3 waves per SIMD there)
template <bool FULL, bool X16, int NQ, int NT>
TG_KERNEL void TG_LAUNCH_BOUNDS2(NT, (NT * NQ < 2560 ? 4 : (NT * NQ == 2560 ? 3 : 2))) tg_adam_rowpass_b(const TgUpdateArgs* argv, TgStepVar var) {
    TgUpdateArgs a = argv[blockIdx.z];
    a.step_size = var.step_size; a.bc2_sqrt = var.bc2_sqrt;
    a.fin.hist = (var.hist_row >= 0 && a.fin.hist) ? a.fin.hist + var.hist_row * TGH_NTERMS : a.fin.coef;   // (no history: the row lands in the
    tg_adam_rowpass_body<FULL, X16, NQ, NT, false>(a);                                                     //  unused coefficient scratch)
}
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_hist_regs(TgHistRegArgs a) { tg_hist_regs_body(a); }
TG_KERNEL void TG_LAUNCH_BOUNDS(1024) tg_hist_regs_b(const TgHistRegArgs* argv, TgStepVar var) {
    TgHistRegArgs a = argv[blockIdx.z];
    if (var.hist_row < 0 || !a.hist) return;
    a.hist += var.hist_row * TGH_NTERMS;
    tg_hist_regs_body(a);
}
