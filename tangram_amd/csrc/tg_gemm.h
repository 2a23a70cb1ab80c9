// tg_gemm.h -- the two MFMA GEMMs of the iteration: shared tile machinery, K1 forward (softmax fused into the A-operand path,
// stream-K), K3 backward (X = S dGhat^T).  Included by tg_kernels.h (after tg_device.h and the common definitions there).
#pragma once
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
