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

// ----------------------------------------------------------------------------------------------
// shared GEMM tile machinery.  Output tile TM x TN, 64*WM*WN threads, each wave owns (TM/WM) x (TN/WN)
// = FM x FN MFMA 16x16 fragments.  One LDS stage = {A tile: TM rows, B tile: TN rows} of 128-byte rows
// (8 chunks of 16 B, XOR-swizzled by tg_swz); two stages are double-buffered.
//   small geometry: 128 x 128, 256 threads (2 x 2 waves of 64 x 64)   -- small problems, 2 workgroups / CU
//   large geometry: 256 x 256, 512 threads (2 x 4 waves of 128 x 64)  -- half the staged bytes per flop
// (profiles/r01: the 128^2 kernels are bound by the global->LDS staging rate, ~17 B/clk/CU, not by MFMA)
// ----------------------------------------------------------------------------------------------
template <int TM_, int TN_, int WM_, int WN_>
struct TgGeo {
    static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = WN_;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int FM = TM / (16 * WM), FN = TN / (16 * WN);
    static constexpr int A_CHUNKS = TM * 8, B_CHUNKS = TN * 8, STAGE_CHUNKS = A_CHUNKS + B_CHUNKS;
    static constexpr int STAGE_BYTES = STAGE_CHUNKS * 16, LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int BWD_LDS_BYTES = LDS_BYTES + TN * 16;            // + the per-cell constants of the row-dot epilogue
    static constexpr int LA = A_CHUNKS / NT, LB = B_CHUNKS / NT;       // 16-byte loads per thread per stage
    static_assert(NT >= 2 * TM && NT % (2 * TM) == 0, "forward A staging: (TM/4 spot quads) x 8 chunk slots threads stage, the rest only multiply");
    static_assert(A_CHUNKS % NT == 0 && B_CHUNKS % NT == 0 && FM % 4 == 0, "tile / thread mismatch");
};
typedef TgGeo<128, 128, 2, 2> TgGeoSmall;
typedef TgGeo<256, 256, 2, 4> TgGeoLarge;
// forward only: 128 spots x 512 genes, 8 waves side by side along the genes (the per-wave fragment grid of TgGeoLarge).  The
// softmax staging of an M panel is redone by every gene tile that shares it: 2 instead of 4 times at K = 1000.  Two stages of
// 80 KB = the whole 160 KB of LDS.
typedef TgGeo<128, 512, 1, 8> TgGeoWide;

// One contraction step of the workgroup tile: software-pipelined over (k-chunk group q) x (blocks of GA A-fragments):
// the ds_read_b128 of the NEXT block are issued before the MFMAs of the current one, so that the LDS latency is
// covered by matrix work inside the wave (the compiler then waits with a partial lgkmcnt instead of lgkmcnt(0)).
// `hook(i)` runs once per group, between the LDS reads of group i+1 and the MFMAs of group i: the kernels issue their
// global loads / LDS-DMA for the next step there, a few per group, instead of one burst of 8-16 vector-memory
// instructions per wave right after the barrier (measured: backward -1 % bf16x3 / -4 % bf16, forward -2 % bf16x3).
template <class PR, class GE, int GA_ = 0, class Hook>
TG_DEV void tg_tile_mma(const u32x4* st, int wm, int wn, int lane, f32x4 (&acc)[GE::FM][GE::FN], Hook&& hook) {
    constexpr int GA = GA_ ? GA_ : ((PR::NP == 2) ? 2 : 4);   // A fragments per block (register budget; 1/2/4 measure the same, run 17)
    constexpr int NB = GE::FM / GA;                           // blocks per k-chunk group
    constexpr int NG = PR::KQ * NB;                           // pipeline length (groups per step)
    const int r = lane & 15, g = lane >> 4;
    constexpr int BRC = PR::BRC;                              // chunks per B tile row: 8, or 4 (hi parts only, PrecBF16x2S)
    const u32x4* sa = st + (wm * (GE::TM / GE::WM) + r) * 8;  // this lane's first A row
    const u32x4* sb = st + GE::A_CHUNKS + (wn * (GE::TN / GE::WN) + r) * BRC;
    // rows of successive fragments differ by 16: (row >> 1) & 7 is the same for all of them, (row >> 4) & 1 alternates
    const int swr = tg_swz(wm * (GE::TM / GE::WM) + r, 0);
    const int swb = BRC == 8 ? tg_swz(wn * (GE::TN / GE::WN) + r, 0) : tg_swz4(wn * (GE::TN / GE::WN) + r, 0);
    u32x4 a[2][GA][PR::NP], b[2][GE::FN][PR::NPB];
    auto load_a = [&](int buf, int q, int blk) {
#pragma unroll
        for (int f = 0; f < GA; ++f)
#pragma unroll
            for (int p = 0; p < PR::NP; ++p) a[buf][f][p] = sa[(blk * GA + f) * 128 + ((4 * (q + p) + g) ^ swr ^ ((blk * GA + f) & 1))];
    };
    auto load_b = [&](int buf, int q) {
#pragma unroll
        for (int f = 0; f < GE::FN; ++f)
#pragma unroll
            for (int p = 0; p < PR::NPB; ++p) {
                if constexpr (BRC == 8) b[buf][f][p] = sb[f * 128 + ((4 * (q + p) + g) ^ swb ^ (f & 1))];
                else b[buf][f][p] = sb[f * 64 + (g ^ swb)];
            }
    };
    load_b(0, 0);
    load_a(0, 0, 0);
    if constexpr (PR::NP == 2 && PR::KQ == 1 && GE::FN == 4) {
        // bf16x3: the GA * 2 fragment loads of the next group are issued one per GA * 2-th of this group's MFMAs (6 MFMAs each)
        // instead of all in front of them (backward -1.7 %, profiles/r01 run40)
#pragma unroll
        for (int i = 0; i < NG; ++i) {
#pragma unroll
            for (int sub = 0; sub < 2 * GA; ++sub) {
                const int fi = sub >> 1;
                if (i + 1 < NG) a[(i + 1) & 1][fi][sub & 1] = sa[((i + 1) * GA + fi) * 128 + ((4 * (sub & 1) + g) ^ swr ^ (((i + 1) * GA + fi) & 1))];
                if (sub == 0) hook(i);
                TG_SCHED_FENCE();
#pragma unroll
                for (int fj = 2 * (sub & 1); fj < 2 * (sub & 1) + 2; ++fj)
                    acc[i * GA + fi][fj] = PR::mma(a[i & 1][fi], b[0][fj], acc[i * GA + fi][fj]);
                TG_SCHED_FENCE();
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int q = i / NB, blk = i % NB;
        if (i + 1 < NG) {
            const int qn = (i + 1) / NB, bn = (i + 1) % NB;
            if (qn != q) load_b(qn & 1, qn);
            load_a((i + 1) & 1, qn, bn);
        }
        hook(i);
        TG_SCHED_FENCE();                                     // next block's LDS reads stay ahead of this block's MFMAs
#pragma unroll
        for (int fi = 0; fi < GA; ++fi)
#pragma unroll
            for (int fj = 0; fj < GE::FN; ++fj)
                acc[blk * GA + fi][fj] = PR::mma(a[i & 1][fi], b[q & 1][fj], acc[blk * GA + fi][fj]);
        TG_SCHED_FENCE();
    }
}

template <class PR, class GE, int GA_ = 0>
struct TgMmaShape {
    static constexpr int GA = GA_ ? GA_ : ((PR::NP == 2) ? 2 : 4);
    static constexpr int NG = PR::KQ * (GE::FM / GA);         // groups per step = calls of the hook
};

// ROWS x 128-byte slab (one contraction step) of an operand stored as [row][step][128 B], copied by LDS-DMA
// (buffer_load_dwordx4 ... lds): no VGPR round trip, no ds_write.  The LDS image of one wave instruction is lane-linear
// (64 x 16 B = 8 tile rows), so the XOR swizzle is applied to the per-lane SOURCE offset (logical chunk = physical chunk ^
// swizzle(row)), the read side applies the same involution.
// A TgKtileDma is set up once per tile: a buffer descriptor over the tile's ROWS operand rows and this lane's byte offsets of its
// ROWS * 8 / NT copies; issue(step, ...) then needs no address arithmetic (the step travels in the copy's scalar offset).
// (`part` of `nparts`: the copies i = part, part + nparts, ... only -- for spreading the issue over the MFMA groups.)
// The copies are issued outside hipcc's wait counters (tg_device.h): the caller drains them with tg_dma_drain() in front of the
// barrier that publishes the tile.
#ifndef TG_DMA_MODE
#define TG_DMA_MODE 2         // A/B switch of the copy instruction (scripts/build_variant.sh -DTG_DMA_MODE=n): 2 = buffer descriptor (default),
#endif                        // 1 = global_load_lds from an asm statement, 0 = the builtin, counted by hipcc (round 3's form)
template <int ROWS, int NT, int RC = 8>      // RC: 16-byte chunks per operand row and step (8; 4 = hi parts only, PrecBF16x2S)
struct TgKtileDma {
    static constexpr int N = ROWS * RC / NT;
    TgRsrc rsrc;
    const unsigned char* base0;
    unsigned voff[N];
    TG_DEVM void setup(const unsigned char* base, size_t row0, size_t pitch_bytes, int t) {
        base0 = base + row0 * pitch_bytes;
        rsrc = tg_make_rsrc(base0, (size_t)ROWS * pitch_bytes);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int idx = t + i * NT, row = idx / RC;
            const int logical = RC == 8 ? tg_swz(row, idx & 7) : tg_swz4(row, idx & 3);               // involution: logical = physical ^ s(row)
            voff[i] = (unsigned)((size_t)row * pitch_bytes) + (unsigned)logical * 16u;
        }
    }
    TG_DEVM void issue(int step, u32x4* tile, int wave, int part = 0, int nparts = 1) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i % nparts != part) continue;
            if (TG_DMA_MODE == 2) tg_glds16_buf(rsrc, voff[i], (unsigned)step * (RC * 16u), (unsigned char*)(tile + i * NT + wave * 64));
            else if (TG_DMA_MODE == 1) tg_glds16_uncounted(base0 + voff[i] + (size_t)step * (RC * 16), (unsigned char*)(tile + i * NT + wave * 64));
            else tg_glds16(base0 + voff[i] + (size_t)step * (RC * 16), (unsigned char*)(tile + i * NT + wave * 64));
        }
    }
};

// XCD-aware workgroup -> tile mapping.  MI355X dispatches workgroup b to XCD b % 8 (observed, used for speed
// only: any mapping is correct).  Each XCD owns a contiguous band of the `major` tile axis (n_major / 8 rows, the first
// n_major % 8 bands one more) and walks it in supertiles of up to 8 x 8 tiles, down the major axis first, so the workgroups
// resident on one XCD share 8 + 8 operand panels through that XCD's private 4 MiB L2 instead of re-fetching them over the
// fabric.  The enumeration is DENSE: supertiles are clipped to the band, so the only workgroups without a tile are the last
// n_minor of the XCDs with the shorter bands (round 1 padded every band to whole 8 x 8 supertiles: 41 % of the workgroups of a
// 118 x 5 grid, 37 % of a 40 x 40 one, were launched -- 128 KB of LDS each -- only to exit; profiles/r02/run11_dense_map).
struct TgTileMap { int mode, n_major, n_minor; };     // mode 0: linear (major = b / n_minor)
TG_HD int tg_tilemap_grid(const TgTileMap& m) {
    if (m.mode == 0) return m.n_major * m.n_minor;
    return 8 * ((m.n_major + 7) / 8) * m.n_minor;
}
TG_HD bool tg_tilemap(const TgTileMap& m, int b, int& major, int& minor) {
    if (m.mode == 0) { major = b / m.n_minor; minor = b % m.n_minor; return true; }
    const int x = b & 7, j = b >> 3;
    const int q = m.n_major >> 3, r = m.n_major & 7;
    const int R = q + (x < r ? 1 : 0);                    // rows of this XCD's band
    const int row0 = x * q + (x < r ? x : r);
    if (j >= R * m.n_minor) { major = 0; minor = 0; return false; }
    const int strip = R * 8;                              // tiles of a full 8-column strip of the band
    int sm = j / strip;
    const int nfull = m.n_minor >> 3;
    if (sm > nfull) sm = nfull;
    const int wc = (sm < nfull) ? 8 : (m.n_minor & 7);    // columns of this strip (the last one may be narrower)
    const int rem = j - sm * strip;
    int sM = rem / (8 * wc);
    const int hr = (R - 8 * sM < 8) ? R - 8 * sM : 8;      // rows of this supertile
    const int rem2 = rem - sM * 8 * wc;
    major = row0 + 8 * sM + rem2 % hr;
    minor = 8 * sm + rem2 / hr;
    return true;
}

// ----------------------------------------------------------------------------------------------
// K1: Ghat_partial[split] = P[c-range]^T [S|1][c-range]          (mapping_optimizer.py:201-202,:217)
//   output tile: TM spots x TN genes; contraction over cells in steps of PR::BKE.
//   A operand (P^T) is produced on the fly: CH x 4 micro-blocks of M are loaded as float4 rows,
//   exponentiated with the per-row shift/scale, transposed in registers and written to LDS
//   as 16-byte chunks along the cell axis.  B operand comes from St (cell axis contiguous).
// ----------------------------------------------------------------------------------------------
struct TgFwdArgs {
    const float* M;
    const float* rmax;       // [Cp] per-row max of M (softmax shift); padding = +3e38 (=> P = 0)
    const float* rmul;       // [Cp] per-row factor f_c / Z_c: P_cv f_c = exp2((M_cv - max_c) * log2(e)) * rmul_c   (fp32, bf16x3)
    const float* rlse2;      // [Cp] (max + ln Z - ln f_c) * log2(e): P_cv f_c = exp2(M_cv * log2(e) - rlse2_c), ONE fma + ONE v_exp_f32.
                             //      Only the bf16 path uses this folded form: the argument is ~20 even for the dominant entries and
                             //      its fp32 rounding costs ~7e-7 relative in every P -- invisible next to bf16 operands (2^-9), but
                             //      10x the reference's softmax error on the fp32-parity paths.
    const unsigned char* St; // [Kp][nsteps][128 B]
    float* Gpart;            // [nsplit][Vr][Kp]
    int C, V, Vp, Vr, Kp, Cp;
    int nkt;                 // gene tiles (Kp / TN)
    int nvt, nsplit;         // spot tiles, partial slots per tile in Gpart (>= the segments any tile is cut into)
    int nsteps;              // Cp / BKE
    int units;               // pieces the (spot tile, step) space of ONE gene tile is cut into (tg_fwd_unit_* below); grid: units * nkt
    int band_index, band_step_begin, band_step_end;   // band mode (band_step_end > 0): ONE cell range -> partial `band_index`
};
// Work decomposition of the forward GEMM ("stream-K").  For ONE gene tile kt, the spot tiles vt = 0 .. nvt-1, each `nsteps`
// contraction steps long, form a step space of G = nvt * nsteps steps, cut into `units` equal pieces: piece j owns the global
// steps [j G / units, (j + 1) G / units) -- a tail of one spot tile and a head of the next, i.e. one or two SEGMENTS (more when a
// piece is longer than a tile).  Every gene tile is cut at the SAME places, and workgroup (j, kt) sits next to (j, kt + 1): the
// nkt workgroups that read one range of an M panel run side by side on one XCD and share it through that XCD's L2 (cutting the
// tiles of all gene tiles as ONE step space put them at different steps at any moment: the M panels were re-read from HBM,
// forward +3 % split-bf16, +50 % bf16 on four gene tiles -- profiles/r04/run5_streamk).  With units * nkt = a multiple of the
// CUs every CU gets the same number of steps whatever the tile count (round 3 cut every tile into nsplit equal ranges: 474
// workgroups at cfg2 = 1.85 rounds of 256, 7 % of the chip idle).  Segment i of spot tile vt (i = j - first piece touching vt) is
// written to partial slot i; tg_ghat_reduce sums the tg_fwd_nseg(vt) slots of a tile in slot order: the sum order is a function
// of the shape alone (bit-reproducible) and the same for every gene column.  units = nvt * s reproduces s equal ranges per tile.
TG_HD long long tg_fwd_unit_begin(long long j, long long G, int units) { return j * G / units; }
TG_HD int tg_fwd_unit_of(long long x, long long G, int units) { return (int)(((x + 1) * units - 1) / G); }       // piece owning global step x
TG_HD int tg_fwd_nseg(int vt, int nsteps, long long G, int units) {
    return tg_fwd_unit_of((long long)(vt + 1) * nsteps - 1, G, units) - tg_fwd_unit_of((long long)vt * nsteps, G, units) + 1;
}
// Workgroup b -> (piece j, gene tile kt).  What the workgroups running side by side on one XCD (b % 8) should share through its
// 4 MB L2 besides the M panel: the S^T tile of their contraction steps -- which they only do when they are at the SAME step of
// their tiles at the same time (an XCD's 32 workgroups turn its L2 over every ~2 steps).
//   pieces that do not cross tiles (units = nvt * s): the round-3 map -- an XCD holds spot tiles vt = x, x + 8, ... of ONE range;
//   stream-K pieces: XCD x takes the pieces j = x, x + 8, x + 16, ...  Their start offsets inside a tile, j L mod nsteps with
//   L = nvt nsteps / units, coincide exactly when units divides 8 nvt (cfg2, 128 x 512 forward tiles: nvt = Vr / 128 = 80 spot tiles, 128 pieces, L = 5/8 of a tile): that
//   is the shape of stream-K decompositions tg_choose_units considers.  (Contiguous ranges of pieces per XCD: the pieces of an
//   XCD are then at 16 different offsets and S^T comes out of the MALL instead: forward 1.32 -> 1.29 ms instead of -> 1.23.)
TG_HD int tg_fwd_units_grid(int units, int nkt) { return 8 * ((units + 7) / 8) * nkt; }
TG_HD bool tg_fwd_unit_map(int b, int units, int nkt, int& j, int& kt) {
    const int x = b & 7, idx = b >> 3;
    j = x + 8 * (idx / nkt);
    kt = idx % nkt;
    return j < units;
}
// grid of the forward kernel: the nkt gene tiles that share one M panel (same spot tile, same cell range) sit
// next to each other on ONE XCD; the panels in flight on an XCD belong to the same cell range and share S^T.
TG_HD int tg_fwd_grid(int nvt, int nkt, int nsplit) { return ((nvt * nsplit + 7) / 8) * 8 * nkt; }
TG_HD bool tg_fwd_map(int b, int nvt, int nkt, int nsplit, int& vt, int& kt, int& split) {
    const int bx = b & 7, bj = b >> 3;
    const int unit = (bj / nkt) * 8 + bx;
    kt = bj % nkt;
    split = unit / nvt;
    vt = unit % nvt;
    return unit < nvt * nsplit;
}

// The segments of workgroup b, in order: f(spot tile, gene tile, partial slot, first step, one past the last step).  Shared by the
// kernel and by tg_debug_fwd_cover (host), which replays every workgroup of a grid and checks that each (tile, step) is taken
// exactly once and each tile's partial slots 0 .. nseg - 1 are each written once.
template <class F>
TG_HD void tg_fwd_walk(int b, int nvt, int nkt, int nsteps, int units, F&& f) {
    int j, kt;
    if (units % nvt == 0) {                                    // pieces inside tiles: the round-3 map (range-major over the XCDs)
        int vt, split;
        const int s = units / nvt;
        if (!tg_fwd_map(b, nvt, nkt, s, vt, kt, split)) return;
        j = vt * s + split;
    } else if (!tg_fwd_unit_map(b, units, nkt, j, kt)) return;
    const long long G = (long long)nvt * nsteps;
    const long long g1 = tg_fwd_unit_begin(j + 1, G, units);
    for (long long g = tg_fwd_unit_begin(j, G, units); g < g1;) {
        const int vt = (int)(g / nsteps), s_begin = (int)(g - (long long)vt * nsteps);
        const int len = (g1 - g < nsteps - s_begin) ? (int)(g1 - g) : nsteps - s_begin;
        f(vt, kt, j - tg_fwd_unit_of((long long)vt * nsteps, G, units), s_begin, s_begin + len);
        g += len;
    }
}

template <class PR, class GE>
TG_DEV void tg_fwd_segment(const TgFwdArgs& a, int vt, int kt, int part_slot, int s_begin, int s_end);

template <class PR, class GE>
TG_DEV void tg_fwd_body(const TgFwdArgs& a) {
    if (a.band_step_end > 0) {                                 // band mode: one workgroup per tile, ONE cell range -> partial `band_index`
        int vt, kt, split;
        if (!tg_fwd_map(blockIdx.x, a.nvt, a.nkt, 1, vt, kt, split)) return;
        tg_fwd_segment<PR, GE>(a, vt, kt, a.band_index, a.band_step_begin, a.band_step_end);
        return;
    }
    bool first = true;
    tg_fwd_walk(blockIdx.x, a.nvt, a.nkt, a.nsteps, a.units, [&](int vt, int kt, int part_slot, int s_begin, int s_end) {
        if (!first) __syncthreads();                           // the LDS stages of the previous segment have been read out
        first = false;
        tg_fwd_segment<PR, GE>(a, vt, kt, part_slot, s_begin, s_end);
    });
}

// one segment: the contraction steps [s_begin, s_end) of tile (vt, kt) -> partial slot `slot`
template <class PR, class GE>
TG_DEV void tg_fwd_segment(const TgFwdArgs& a, int vt, int kt, int part_slot, int s_begin, int s_end) {
    TG_LDS_DECL;
    u32x4* lds = (u32x4*)tg_lds;
    const int t = threadIdx.x, lane = t & 63, wave = tg_uniform(t >> 6);
    const int wm = wave / GE::WN, wn = wave % GE::WN;
    const int v0 = vt * GE::TM, k0 = kt * GE::TN;
    const int split = part_slot;

    f32x4 acc[GE::FM][GE::FN];
#pragma unroll
    for (int i = 0; i < GE::FM; ++i)
#pragma unroll
        for (int j = 0; j < GE::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A staging: thread = (spot quad, slot).  One slot = RS cells x 4 spots loaded as RS float4 rows of M.
    //   bf16 / fp32 : slot = one 16-byte chunk (CH cells);
    //   bf16x3      : slot = HALF a k-chunk (4 cells): the thread writes 8 bytes of the hi chunk and 8 bytes of the lo chunk,
    //                 so that every exponential is evaluated exactly once.
    constexpr int RS = (PR::NP == 2) ? PR::CH / 2 : PR::CH;
    const int quad = t % (GE::TM / 4), slot = (t / (GE::TM / 4)) & 7;
    const bool stager = t < 2 * GE::TM;                        // (wide geometry: waves 4-7 only multiply; wave-uniform)
    const int kc = (PR::NP == 2) ? (slot >> 1) : slot;         // k-chunk of the 128-byte step row
    const int half = (PR::NP == 2) ? (slot & 1) : 0;
    const int vcol = v0 + 4 * quad;
    const int vload = (vcol < a.Vp) ? vcol : 0;
    bool vok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vok[i] = (vcol + i) < a.V;
    const bool full_tile = (v0 + GE::TM) <= a.V;               // wave-uniform: interior tiles skip the per-element selects

    f32x4 mreg[RS];
    float sh[RS], mu[RS];
    const size_t bpitch = (size_t)a.nsteps * (PR::BRC * 16);
    constexpr int LBX = GE::TN * PR::BRC / GE::NT;            // copies of the S^T tile per thread and step
    TgKtileDma<GE::TN, GE::NT, PR::BRC> dmaB;                  // S^T tile: TN gene rows of this workgroup, every step of the cell axis
    dmaB.setup(a.St, (size_t)k0, bpitch, t);

    auto load_m = [&](int step, int j) {                       // one float4 row of the M micro-block of `step`
        if (!stager) return;
        const int c = step * PR::BKE + kc * PR::CH + half * RS + j;
        const int cc = c < a.C ? c : a.C - 1;
        mreg[j] = *(const f32x4*)(a.M + (size_t)cc * a.Vp + vload);
    };
    auto load_sh = [&](int step) {
        if (!stager) return;
        const int cb = step * PR::BKE + kc * PR::CH + half * RS;
#pragma unroll
        for (int j = 0; j < RS; ++j) {
            if constexpr (PR::kId == 1) { sh[j] = a.rlse2[cb + j]; mu[j] = 1.f; }
            else { sh[j] = a.rmax[cb + j]; mu[j] = a.rmul[cb + j]; }
        }
    };
    auto load_stage = [&](int step) {
#pragma unroll
        for (int j = 0; j < RS; ++j) load_m(step, j);
        load_sh(step);
    };
    // the same global loads + the LDS-DMA of S^T as NITEM separate issues, spread over the first NSPREAD MFMA groups
    constexpr int GA_F = (PR::NP == 2 ? 1 : 2);                // the M staging registers leave room for small blocks only
    constexpr int NG_F = TgMmaShape<PR, GE, GA_F>::NG;
    constexpr int NSPREAD = (NG_F * 3) / 4 > 0 ? (NG_F * 3) / 4 : 1;
    constexpr int NITEM = LBX + RS + 1;
    // (step_m: the step whose M micro-block is fetched -- one step further ahead for the early half of the waves, see below;
    //  step_b: the step whose S^T tile is copied into `st`; a negative step = nothing to fetch)
    auto issue_next = [&](int step_m, int step_b, u32x4* st, int i) {
#pragma unroll
        for (int k = 0; k < NITEM; ++k) {
            if ((k * NSPREAD) / NITEM != i) continue;
            if (k < RS) { if (step_m >= 0) load_m(step_m, k); }
            else if (k == RS) { if (step_m >= 0) load_sh(step_m); }
            else if (step_b >= 0) dmaB.issue(step_b, st + GE::A_CHUNKS, wave, k - RS - 1, LBX);
        }
    };
    // (MASKED: edge tiles zero the spots >= V; interior tiles skip the 16 selects.  bf16x3: the arithmetic runs on pairs of
    //  cells so that the compiler can use the packed-fp32 VALU ops, v_pk_add_f32 / v_pk_mul_f32: the staging is VALU-bound.)
    auto store_stage_impl = [&](u32x4* st, auto masked) {
        constexpr bool MASKED = decltype(masked)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * quad + i;
            if constexpr (PR::NP == 2) {
                unsigned h[2], l[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const f32x2 m2 = {mreg[2 * q][i], mreg[2 * q + 1][i]};
                    const f32x2 sh2 = {sh[2 * q], sh[2 * q + 1]}, mu2 = {mu[2 * q], mu[2 * q + 1]};
                    const f32x2 tt = (m2 - sh2) * TG_LOG2E;
                    f32x2 x2 = f32x2{tg_exp2(tt[0]), tg_exp2(tt[1])} * mu2;
                    if (MASKED && !vok[i]) x2 = f32x2{0.f, 0.f};
                    h[q] = tg_pack_bf16(x2[0], x2[1]);
                    const f32x2 hf = {tg_bf16_lo_to_f32(h[q]), tg_bf16_hi_to_f32(h[q])};
                    const f32x2 l2 = x2 - hf;
                    l[q] = tg_pack_bf16(l2[0], l2[1]);
                }
                u32x2* hp = (u32x2*)(st + row * 8 + tg_swz(row, kc));
                u32x2* lp = (u32x2*)(st + row * 8 + tg_swz(row, 4 + kc));
                hp[half] = u32x2{h[0], h[1]};
                lp[half] = u32x2{l[0], l[1]};
            } else {
                float x[RS];
#pragma unroll
                for (int j = 0; j < RS; ++j) {
                    const float p = (PR::kId == 1) ? tg_exp2(fmaf(mreg[j][i], TG_LOG2E, -sh[j]))
                                                   : tg_exp2((mreg[j][i] - sh[j]) * TG_LOG2E) * mu[j];
                    x[j] = (full_tile || vok[i]) ? p : 0.f;
                }
                u32x4 hi, lo;
                PR::cvt(x, hi, lo);
                st[row * 8 + tg_swz(row, slot)] = hi;
            }
        }
    };
    auto store_stage = [&](u32x4* st) {
        if (!stager) return;
        if (PR::NP == 2 && full_tile) store_stage_impl(st, std::false_type());     // (the second copy only pays off for bf16x3)
        else store_stage_impl(st, std::true_type());
    };

    // Phase-shifted operand staging.  The softmax staging of the next step (exp2, hi/lo split, transposed ds_write: VALU) is
    // work of the same waves that issue the MFMAs; done by all eight waves at the same point of the step (after their MFMAs,
    // before the barrier) it leaves the matrix pipes idle for its whole duration.  The two waves that share a SIMD therefore
    // do it at OPPOSITE ends of the step: the EARLY half converts the block of step s+1 first thing in step s (its M loads run
    // one step further ahead: issued during step s-1, so they also have a whole step to arrive), the LATE half after its MFMAs
    // as before -- while one wave of a SIMD is in its VALU block the other one feeds the matrix pipe.  Same values, same
    // order of arithmetic: results are bit-identical to the unshifted schedule.
    // Measured at 30k x 1k x 10k (profiles/r02/run2-4): bf16 0.775 -> 0.675 ms with the YOUNGER half early (the older half
    // early: 0.89); bf16x3 1.47 -> 1.58 either way, so the split-bf16 path keeps the unshifted schedule.  Slicing the staging
    // between the MFMA groups of every wave instead (with or without vector-memory traffic in the staging slices) was 1.7x
    // SLOWER (bf16x3 2.45 ms): VALU in the MFMA stream costs far more than its issue slots (profiles/r02/README.md).
    constexpr int STAG = (TG_FWD_STAGGER >= 0) ? TG_FWD_STAGGER : ((PR::kId == 1) ? 2 : 0);
    const bool early = (STAG == 1) ? (wave < GE::NT / 128) : ((STAG == 2) ? (wave >= GE::NT / 128) : false);
    // (every wave early -- all M loads a full step ahead, VALU block first: bf16x3 1.63 ms, bf16 0.77: the M loads do not cost latency)
    if (s_begin < s_end) {
        dmaB.issue(s_begin, lds + GE::A_CHUNKS, wave);
        load_stage(s_begin);
        store_stage(lds);
        if (early && s_begin + 1 < s_end) load_stage(s_begin + 1);
        tg_dma_drain();
        __syncthreads();
        for (int s = s_begin; s < s_end; ++s) {
            u32x4* cur = lds + ((s - s_begin) & 1) * GE::STAGE_CHUNKS;
            u32x4* nxt = lds + ((s - s_begin + 1) & 1) * GE::STAGE_CHUNKS;
            const bool more = (s + 1) < s_end;
            const int step_m = early ? ((s + 2) < s_end ? s + 2 : -1) : (more ? s + 1 : -1);    // M block to fetch during this step
            if (early && more) store_stage(nxt);    // (`nxt` was last read in step s-1: every wave has passed that barrier)
            if (PR::NP == 2) {                      // next step's global loads / LDS-DMA trickle in between the MFMA groups
                                                    // (bf16x3: -2 %; slower for the 8-row micro-blocks of bf16, profiles/r01/run26)
                tg_tile_mma<PR, GE, GA_F>(cur, wm, wn, lane, acc, [&](int i) { issue_next(step_m, more ? s + 1 : -1, nxt, i); });
            } else {
                if (more) dmaB.issue(s + 1, nxt + GE::A_CHUNKS, wave);
                if (step_m >= 0) load_stage(step_m);
                tg_tile_mma<PR, GE, GA_F>(cur, wm, wn, lane, acc, [](int) {});
            }
            if (!early && more) store_stage(nxt);
            tg_dma_drain();
            __syncthreads();
        }
    }

    // epilogue: lane holds 4 consecutive spots (regs) x 1 gene (lane&15) per fragment
    float* out = a.Gpart + (size_t)split * a.Vr * a.Kp;
    const int g = lane >> 4, r15 = lane & 15;
#pragma unroll
    for (int fi = 0; fi < GE::FM; ++fi)
#pragma unroll
        for (int fj = 0; fj < GE::FN; ++fj) {
            const int v = v0 + wm * (GE::TM / GE::WM) + fi * 16 + 4 * g;
            const int k = k0 + wn * (GE::TN / GE::WN) + fj * 16 + r15;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(v + r) * a.Kp + k] = acc[fi][fj][r];
        }
}

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
template <int KX>
TG_DEV void tg_gene_reduce_body(const float* genepart, int nrb, int Kp, float* genestat /*[2][Kp]*/) {
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
    const int vbeg = blockIdx.x * TG_RB;
    const size_t pitch = (size_t)(a.Kp / PR::BKE) * 128;
    const float* coef = a.coef;
    if constexpr (SELF) {
        for (int k = threadIdx.x; k < a.Kp; k += 256) {
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
            if (v < a.Vr) { a.fin.vcoef[v] = va; a.fin.vcoef[a.Vr + v] = vb; a.fin.vcoef[2 * a.Vr + v] = av; }   // a_v: read by the backward / update kernels
        }
        if (threadIdx.x < 64 && a.fin.spotpart) {        // the spots' loss terms summed per block: the history workgroup adds the blocks up
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { c_blk += tg_shfl_xor(c_blk, m); kl_blk += tg_shfl_xor(kl_blk, m); }
            if (threadIdx.x == 0) { a.fin.spotpart[2 * blockIdx.x] = c_blk; a.fin.spotpart[2 * blockIdx.x + 1] = kl_blk; }
        }
        __syncthreads();
        coef = cf;
    }
    for (int idx = threadIdx.x; idx < nch * TG_RB; idx += 256) {
        const int i = idx / nch, ch = idx % nch;
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

// ----------------------------------------------------------------------------------------------
// K3: X^T tile = dGhat[v-tile] . S[c-tile]^T  (contraction over genes), fused epilogues.
//   epilogue: X[c][v] stored (fp32) for the update kernel, and
//             r_part[vt][c] = sum_{v in tile} P_cv dP_cv                 (softmax backward row dot)
//             (+ row partials of the entropy / L1 / L2 scalars and of the filter gradient when FULL)
//   Softmax backward needs the complete row dot r_c before any element of the row can be updated, so the
//   update runs as a second, purely streaming kernel (tg_adam_update) on the stored X.
//   Fragment ownership: lane holds 4 consecutive spots (one float4 of M) for cell c = lane&15.
// ----------------------------------------------------------------------------------------------
struct TgBwdArgs {
    const unsigned char* dG;      // A operand [Vr][nsteps][128 B]
    const unsigned char* Sk;      // B operand [Cr][nsteps][128 B]
    const float* M;                                // logits, pitch Vp
    void* X;                                       // [C][Vp] backward GEMM result S dGhat^T (fp32, or bf16 when PR::X16) for tg_adam_update
    const float* rshift; const float* rinvz;       // [Cp] softmax shift and 1/Z of the CURRENT M
    const float* fgate;                            // [C] filter f_c (constrained) or null
    const float* vcoef;                            // a_v at [2*Vr + v]
    const float* dens_w;                           // [C] w_c (d_source) or null (=1)
    float* part;                                   // [nvt][NP1][C] row-dot partials
    int C, V, Vp, Vr, Kp, nsteps;
    TgTileMap map;                                 // major/minor = (cell tile, spot tile) or swapped
    int map_major_is_cells;
    int ct_offset;                                 // first cell tile of this launch (cell-band pipelining)
    float lambda_r, lambda_l1, lambda_l2;
};
enum { TGP1_R = 0, TGP1_ENT, TGP1_L1, TGP1_L2, TGP1_Q, TGP1_PA, TGP1_N };

template <class PR, class GE, bool FULL, bool ROWDOT, bool STREAM>
TG_DEV void tg_bwd_body(const TgBwdArgs& a) {
    TG_LDS_DECL;
    u32x4* lds = (u32x4*)tg_lds;
    const int t = threadIdx.x, lane = t & 63, wave = tg_uniform(t >> 6);
    const int wm = wave / GE::WN, wn = wave % GE::WN;
    int t_major, t_minor;
    if (!tg_tilemap(a.map, blockIdx.x, t_major, t_minor)) return;
    const int vt = a.map_major_is_cells ? t_minor : t_major, ct = a.ct_offset + (a.map_major_is_cells ? t_major : t_minor);
    const int v0 = vt * GE::TM, c0 = ct * GE::TN;
    const int nsteps = a.nsteps;

    f32x4 acc[GE::FM][GE::FN];
#pragma unroll
    for (int i = 0; i < GE::FM; ++i)
#pragma unroll
        for (int j = 0; j < GE::FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    {
        const size_t pitch = (size_t)nsteps * 128;
        constexpr int LBX = GE::TN * PR::BRC / GE::NT;        // copies of the S tile per thread and step
        TgKtileDma<GE::TM, GE::NT> dmaA;
        TgKtileDma<GE::TN, GE::NT, PR::BRC> dmaB;
        dmaA.setup(a.dG, (size_t)v0, pitch, t);
        dmaB.setup(a.Sk, (size_t)c0, (size_t)nsteps * (PR::BRC * 16), t);
        dmaA.issue(0, lds, wave);
        dmaB.issue(0, lds + GE::A_CHUNKS, wave);
        tg_dma_drain();
        __syncthreads();
        // The last step is peeled off the loop: inside the loop the DMA issue is unconditional, so a step is ONE basic block
        // (a `more` test per MFMA group made four, and hipcc then waits lgkmcnt(0) at the head of every block).
        for (int s = 0; s + 1 < nsteps; ++s) {
            u32x4* cur = lds + (s & 1) * GE::STAGE_CHUNKS;
            u32x4* nxt = lds + ((s + 1) & 1) * GE::STAGE_CHUNKS;
            tg_tile_mma<PR, GE>(cur, wm, wn, lane, acc, [&](int i) {     // DMA of the next step lands while the matrix cores run,
                constexpr int NG_B = TgMmaShape<PR, GE>::NG, NSP = (NG_B * 3) / 4 > 0 ? (NG_B * 3) / 4 : 1, NIT = GE::LA + LBX;
#pragma unroll
                for (int k = 0; k < NIT; ++k) {                          // issued a few copies per MFMA group (see tg_tile_mma)
                    if ((k * NSP) / NIT != i) continue;
                    if (k < GE::LA) dmaA.issue(s + 1, nxt, wave, k, GE::LA);
                    else dmaB.issue(s + 1, nxt + GE::A_CHUNKS, wave, k - GE::LA, LBX);
                }
            });
            tg_dma_drain();
            __syncthreads();                        // publishes the next tile and releases `cur` for the step after next
        }
        tg_tile_mma<PR, GE>(lds + ((nsteps - 1) & 1) * GE::STAGE_CHUNKS, wm, wn, lane, acc, [](int) {});
        __syncthreads();
    }

    // ---------------- epilogue ----------------
    // The MFMA result layout gives a lane 4 consecutive spots of ONE cell (16 cells per wave instruction), i.e. 16 separate
    // 64-byte pieces per global access: ~13 us per tile for the X store alone, not overlapped with anything (one workgroup
    // per CU).  The tile is therefore transposed through the (now idle) LDS in NPASS passes of CPP cells and handled as
    // full rows: RC lanes own the TM spots of one cell, X leaves (and M arrives) as 1 KB row segments.
    //   ROWDOT == false (single GPU): only X leaves the kernel; the row dots are taken by tg_adam_rowpass.
    //   ROWDOT == true  (spot shard, or rows too long for tg_adam_rowpass): also r_part[vt][c] = sum_{v in tile} P dP
    //                    (+ the entropy / L1 / L2 / filter row sums when FULL), reduced over the RC lanes of the row.  The M
    //                    segments of a whole pass are requested before the staging barrier (NIT loads in flight per lane);
    //                    the per-cell constants of the tile wait in the 4 KB of LDS behind the staging area.
    constexpr int RC = GE::TM / 4;                                        // 16-byte columns of a staged row (one cell, TM spots)
    constexpr int CPP = (GE::LDS_BYTES / (GE::TM * 4) < GE::TN) ? GE::LDS_BYTES / (GE::TM * 4) : GE::TN;   // cells per pass
    constexpr int NPASS = GE::TN / CPP, FPP = GE::FN / NPASS;             // passes, cell fragments per wave and pass
    constexpr int NIT = (CPP * RC) / GE::NT;                              // row segments per lane and pass
    static_assert(FPP * NPASS == GE::FN && CPP == GE::WN * FPP * 16 && (CPP * RC) % GE::NT == 0 && RC >= 16 && RC <= 64 && GE::NT % RC == 0,
                  "epilogue staging geometry");
    constexpr int NP = FULL ? (int)TGP1_N : 1;
    f32x4* stg = (f32x4*)tg_lds;
    f32x4* rowc = (f32x4*)(tg_lds + GE::LDS_BYTES);                       // ROWDOT: [TN] (shift, 1/Z, f, w) of the tile's cells
    const int g = lane >> 4, r15 = lane & 15;
    const int j = t % RC, v = v0 + 4 * j;                                 // this lane's 4 spots: the same in every row it visits
    f32x4 aq = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ROWDOT) {
        if (t < GE::TN) {
            const int c = c0 + t, cc = c < a.C ? c : a.C - 1;
            rowc[t] = f32x4{a.rshift[cc], a.rinvz[cc], a.fgate ? a.fgate[cc] : 1.f, a.dens_w ? a.dens_w[cc] : 1.f};
        }
        aq = *(const f32x4*)(a.vcoef + 2 * (size_t)a.Vr + (v < a.Vr ? v : 0));
    }
    auto cell_of = [&](int pass, int row) {                               // staged row -> cell index of the tile (0 .. TN-1)
        const int wn_r = row / (FPP * 16), rem = row % (FPP * 16);
        return wn_r * (GE::TN / GE::WN) + (pass * FPP + rem / 16) * 16 + (rem & 15);
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass) __syncthreads();                                        // the previous pass has been read out
#pragma unroll
        for (int fjl = 0; fjl < FPP; ++fjl) {
            const int cell_l = (wn * FPP + fjl) * 16 + r15;
#pragma unroll
            for (int fi = 0; fi < GE::FM; ++fi) {
                const int col = (wm * (GE::TM / GE::WM) + fi * 16) / 4 + g;
                stg[cell_l * RC + (col ^ r15)] = acc[fi][pass * FPP + fjl];   // XOR swizzle: the 16 cells of a lane group hit 16 different columns
            }
        }
        f32x4 mqs[ROWDOT ? NIT : 1];
        if constexpr (ROWDOT) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int c = c0 + cell_of(pass, (it * GE::NT + t) / RC);
                mqs[it] = *(const f32x4*)(a.M + (size_t)(c < a.C ? c : a.C - 1) * a.Vp + (v < a.Vp ? v : 0));
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int row = (it * GE::NT + t) / RC;
            const int cl = cell_of(pass, row), c = c0 + cl;
            const bool ok = c < a.C && v < a.Vp;
            const f32x4 x = stg[row * RC + (j ^ (row & 15))];
            if (ok) {
                if constexpr (PR::X16)
                    tg_st_stream<STREAM && !TG_X_TEMPORAL>(u32x2{tg_pack_bf16(x[0], x[1]), tg_pack_bf16(x[2], x[3])}, (u32x2*)((unsigned short*)a.X + (size_t)c * a.Vp + v));
                else
                    tg_st_stream<STREAM && !TG_X_TEMPORAL>(x, (f32x4*)((float*)a.X + (size_t)c * a.Vp + v));
            }
            if constexpr (ROWDOT) {
                float pacc[NP];
#pragma unroll
                for (int q = 0; q < NP; ++q) pacc[q] = 0.f;
                if (ok) {
                    const f32x4 mq = mqs[it], rcst = rowc[cl];
                    const float sh = rcst[0], iz = rcst[1], fg = rcst[2], wc = rcst[3];
                    const float logiz = (FULL && a.lambda_r != 0.f) ? tg_log(iz) : 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if ((v + e) >= a.V) continue;
                        const float p = tg_exp(mq[e] - sh) * iz;
                        float dp = fg * (x[e] + aq[e] * wc);
                        if (FULL) {
                            if (a.lambda_r != 0.f) {
                                const float lp = (mq[e] - sh) + logiz;        // log P, no underflow
                                dp -= a.lambda_r * (lp + 1.f);
                                pacc[TGP1_ENT % NP] += p * lp;
                            }
                            pacc[TGP1_Q % NP] += p * x[e];
                            pacc[TGP1_PA % NP] += p * aq[e];
                            pacc[TGP1_L1 % NP] += fabsf(mq[e]);
                            pacc[TGP1_L2 % NP] += mq[e] * mq[e];
                        }
                        pacc[TGP1_R] += p * dp;
                    }
                }
#pragma unroll
                for (int q = 0; q < NP; ++q) {                            // the RC lanes of this row segment sit in one wave
                    float sm = pacc[q];
#pragma unroll
                    for (int m = RC / 2; m >= 1; m >>= 1) sm += tg_shfl_xor(sm, m);
                    if (j == 0 && c < a.C) a.part[((size_t)vt * NP + q) * a.C + c] = sm;
                }
            }
        }
    }
}

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
    if (t == 0) {
        float mx = red[0];
        for (int w = 1; w < NW; ++w) mx = tg_fmax(mx, red[w * 2]);
        float z = 0.f;
        for (int w = 0; w < NW; ++w) z += red[w * 2 + 1] * tg_exp(red[w * 2] - mx);
        a.pair_out[c] = mx;
        a.pair_out[a.C + c] = z;
        if (a.finalize) {
            const float inz = 1.f / z;
            a.new_shift[c] = mx;
            a.new_invz[c] = inz;
            a.new_mul[c] = inz;                                // (the constrained filter is folded in by tg_merge_stats)
            a.new_scale[c] = (mx + tg_log(z)) * TG_LOG2E;
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
    if (a.fin_on && blockIdx.x == gridDim.x - 1) { tg_loss_scalars<false>(a.fin, red); return; }
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
// Peer-memory exchange: the third tg_comm transport (tg_capi.hip: tg_comm_peer_create / _connect).
//   The three per-step exchanges of a spot shard are 8 KB - 1 MB vectors whose cost on a collective library is its fixed latency
//   (a ring all-reduce on 8 ranks is 14 dependent hops).  Here an exchange is ONE kernel per rank and ONE hop: every rank owns a
//   MAILBOX in its own HBM that every peer has mapped (hipIpc between processes of a node: xGMI stores; plain pointers between
//   shards inside one process).  The mailbox holds, per generation slot (2) and source rank, the vector as 8-byte GRANULES
//   {float value, sequence number of the exchange}, each written by ONE naturally aligned write-through store (system-scope relaxed
//   atomic: sc0 sc1) -- value and tag arrive together or not at all, so there is no flag, no fence and no barrier
//   (MI355X_MICROARCH.md, "handoff-1to1": tagged granules cost half of payload + flag).  A thread
//     1. reads its elements of this rank's vector and stores their granules into slot [seq & 1][this rank] of EVERY rank's mailbox
//        (its own included), peers visited from rank + 1 on;
//     2. polls the granules of the same elements from every rank in its OWN mailbox until their tag is this exchange's sequence
//        number (bounded: a peer that never arrives costs TG_PEER_TIMEOUT_MS, raises the error word, and the kernel ends);
//     3. all-reduce: adds the world values in RANK ORDER (every rank adds the same floats in the same order: bit-identical on every
//        rank, and equal to the callback transport's rank-order sum); all-gather: copies them out.
//   Elements are independent: no workgroup or grid barrier.  Two generation slots suffice: a rank leaves exchange g only after every
//   peer has pushed g, and a peer pushes g + 1 only after its own kernel of g has finished reading, so nobody writes generation g + 2
//   into a slot somebody still reads generation g from.  Sequence numbers only grow (the mailbox starts zeroed; the first is 1).
// ----------------------------------------------------------------------------------------------
#define TG_PEER_MAX 16
#define TG_PEER_CHUNK 2048              // floats per workgroup
#define TG_PEER_HDR 256                 // bytes in front of the granules: [0] error word (1: a poll timed out)
#ifdef TG_SIM
#include <chrono>
#include <sched.h>
TG_DEV void tg_sys_store_u64(unsigned long long* p, unsigned long long v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
TG_DEV unsigned long long tg_sys_load_u64(const unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
TG_DEV void tg_sys_store_u32(unsigned* p, unsigned v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }
TG_DEV unsigned tg_sys_load_u32(const unsigned* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
TG_DEV unsigned long long tg_wall_ticks() {      // 100 MHz like wall_clock64()
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10;
}
TG_DEV void tg_poll_pause() { sched_yield(); }
#else
TG_DEV void tg_sys_store_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned long long tg_sys_load_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV void tg_sys_store_u32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned tg_sys_load_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
TG_DEV unsigned long long tg_wall_ticks() { return wall_clock64(); }
TG_DEV void tg_poll_pause() { __builtin_amdgcn_s_sleep(2); }
#endif
struct TgPeerArgs {
    unsigned char* box[TG_PEER_MAX];    // every rank's mailbox as mapped by THIS rank; box[rank] is its own
    int world, rank;
    unsigned long long cap;             // granules of one (slot, rank) region
    int slot; unsigned seq;
    const float* send; float* recv;     // all-reduce: in place (send == recv)
    unsigned long long n;               // floats (per rank)
    int gather; unsigned long long ld;  // gather: recv[r * ld + i]
    unsigned long long timeout_ticks;   // bound of a poll in 10-ns ticks
};
TG_HD size_t tg_peer_box_bytes(int world, size_t cap) { return TG_PEER_HDR + (size_t)2 * world * cap * 8; }
// the value of granule *g once its tag is `seq` (0.f after a time-out, with the error word raised)
TG_DEV float tg_peer_take(const unsigned long long* g, unsigned seq, unsigned long long timeout, unsigned* err) {
    unsigned long long x = tg_sys_load_u64(g);
    if ((unsigned)(x >> 32) != seq) {
        const unsigned long long t0 = tg_wall_ticks();
        do {
            tg_poll_pause();
            x = tg_sys_load_u64(g);
            if ((unsigned)(x >> 32) == seq) break;
            if (tg_wall_ticks() - t0 > timeout) { tg_sys_store_u32(err, 1u); return 0.f; }
        } while (true);
    }
    return __builtin_bit_cast(float, (unsigned)x);
}
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_peer_exchange(TgPeerArgs a) {
    constexpr int NJ = TG_PEER_CHUNK / 256;
    const int t = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * TG_PEER_CHUNK;
    const size_t region = ((size_t)a.slot * a.world + a.rank) * a.cap;         // my region in anybody's mailbox
    const unsigned long long tag = (unsigned long long)a.seq << 32;
    // 1. push
    float mine[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const size_t i = lo + t + 256 * (size_t)j; mine[j] = i < a.n ? a.send[i] : 0.f; }
    for (int p = 0; p < a.world; ++p) {
        unsigned long long* dst = (unsigned long long*)(a.box[(a.rank + 1 + p) % a.world] + TG_PEER_HDR) + region;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t i = lo + t + 256 * (size_t)j;
            if (i < a.n) tg_sys_store_u64(dst + i, tag | (unsigned long long)__builtin_bit_cast(unsigned, mine[j]));
        }
    }
    // 2. + 3. take every rank's granules of my elements out of MY mailbox.  Once a poll has timed out (error word raised) a peer is
    // gone: later exchanges do not wait again -- the run ends quickly with garbage and tg_comm_peer_status says why.
    const unsigned long long* in = (const unsigned long long*)(a.box[a.rank] + TG_PEER_HDR) + (size_t)a.slot * a.world * a.cap;
    unsigned* err = (unsigned*)a.box[a.rank];
    if (tg_sys_load_u32(err) != 0u) a.timeout_ticks = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const size_t i = lo + t + 256 * (size_t)j;
        if (i >= a.n) continue;
        if (a.gather) {
            for (int r = 0; r < a.world; ++r) a.recv[(size_t)r * a.ld + i] = tg_peer_take(in + (size_t)r * a.cap + i, a.seq, a.timeout_ticks, err);
        } else {
            float s = tg_peer_take(in + i, a.seq, a.timeout_ticks, err);
            for (int r = 1; r < a.world; ++r) s += tg_peer_take(in + (size_t)r * a.cap + i, a.seq, a.timeout_ticks, err);
            a.recv[i] = s;
        }
    }
}

// ----------------------------------------------------------------------------------------------
// small per-row kernels
// ----------------------------------------------------------------------------------------------
// r_c = sum over spot tiles of the phase-1 partials; also the scalar regulariser sums
struct TgRowsumArgs {
    const float* part; int nvt; int C; int np;
    float* rowq;               // [np][C] summed partials (row 0 = r_c)
    int c_begin, c_end;        // cells handled by this launch
};
// 16 cells x 16 groups of spot tiles per workgroup: a cell's partials p = g, g + 16, ... side by side, then the groups in fixed order
// (one thread per cell walking all V / 128 partials took 61 us at 50 000 spots and 18 rows of M)
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_rowsum_parts(TgRowsumArgs a) {
    TG_LDS_DECL;
    float* red = (float*)tg_lds;                                 // [16][16]
    const int r = threadIdx.x & 15, g = threadIdx.x >> 4, c = a.c_begin + blockIdx.x * 16 + r;
    for (int q = 0; q < a.np; ++q) {
        float s = 0.f;
        if (c < a.c_end)
            for (int p = g; p < a.nvt; p += 16) s += a.part[((size_t)p * a.np + q) * a.C + c];
        red[g * 16 + r] = s;
        __syncthreads();
        if (g == 0 && c < a.c_end) {
            float t = 0.f;
            for (int i = 0; i < 16; ++i) t += red[i * 16 + r];
            a.rowq[(size_t)q * a.C + c] = t;
        }
        __syncthreads();
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
TG_DEV void tg_merge_stats_body(const TgMergeArgs& a) {
    const int c = blockIdx.x * 256 + threadIdx.x;
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

TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_merge_stats(TgMergeArgs a) { tg_merge_stats_body(a); }
TG_KERNEL void TG_LAUNCH_BOUNDS(256) tg_merge_stats_b(const TgMergeArgs* argv) { tg_merge_stats_body(argv[blockIdx.z]); }

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
