// tg_capi.hip -- host side of libtangram_hip.so: the C ABI of include/tangram_hip.h.
// Enqueues the kernels of tg_kernels.h on the caller's stream; allocates nothing on the device.
#include "../../include/tangram_hip.h"
#include "tg_kernels.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <new>
#include <string>
#include <vector>

#ifdef TG_SIM
typedef void* tg_stream_t;
#define TG_LAUNCH(kern, gx, gy, block, lds, stream, ...) \
    hipsim::launch(hipsim::uint3s{(unsigned)(gx), (unsigned)(gy), 1u}, hipsim::uint3s{(unsigned)(block), 1u, 1u}, [&] { kern(__VA_ARGS__); })
static int tg_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, tg_stream_t) {
    for (size_t r = 0; r < height; ++r) memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
    return 0;
}
static int tg_memset(void* dst, int v, size_t n, tg_stream_t) { memset(dst, v, n); return 0; }
static int tg_memcpy_h2d(void* dst, const void* src, size_t n, tg_stream_t) { memcpy(dst, src, n); return 0; }
static int tg_memcpy(void* dst, const void* src, size_t n, tg_stream_t) { memcpy(dst, src, n); return 0; }
#define TG_LAUNCH3(kern, gx, gy, gz, block, lds, stream, ...) \
    hipsim::launch(hipsim::uint3s{(unsigned)(gx), (unsigned)(gy), (unsigned)(gz)}, hipsim::uint3s{(unsigned)(block), 1u, 1u}, [&] { kern(__VA_ARGS__); })
static int tg_launch_error(const char** name) { *name = nullptr; return 0; }
static bool tg_launch_failed() { return false; }
static const char* tg_hip_errstr(int) { return "emulator"; }
typedef int tg_event_t;
static int tg_stream_create(tg_stream_t* s) { *s = nullptr; return 0; }
static void tg_stream_destroy(tg_stream_t) {}
static int tg_event_create(tg_event_t* e) { *e = 0; return 0; }
static void tg_event_destroy(tg_event_t) {}
static void tg_event_record(tg_event_t, tg_stream_t) {}           // the emulator runs launches synchronously in program order
static void tg_stream_wait(tg_stream_t, tg_event_t) {}
static void tg_event_host_wait(tg_event_t) {}
static void* tg_host_pinned_alloc(size_t n) { return malloc(n); }
static void tg_host_pinned_free(void* p) { free(p); }
static bool tg_stream_capturing(tg_stream_t) { return false; }
#else
typedef hipStream_t tg_stream_t;
// Every launch is checked where it is issued: the FIRST failing kernel of a call is remembered by name (thread-local) and
// reported by tg_launch_status() -- a failed launch in the middle of a step is no longer an anonymous error at its end.
static thread_local int g_launch_rc = 0;
static thread_local const char* g_launch_name = nullptr;
#define TG_LAUNCH(kern, gx, gy, block, lds, stream, ...)                                                                          \
    do {                                                                                                                          \
        hipLaunchKernelGGL(kern, dim3((unsigned)(gx), (unsigned)(gy), 1), dim3((unsigned)(block), 1, 1), (size_t)(lds), stream,   \
                           __VA_ARGS__);                                                                                          \
        const int _le = (int)hipGetLastError();                                                                                   \
        if (_le != 0 && g_launch_rc == 0) { g_launch_rc = _le; g_launch_name = #kern; }                                           \
    } while (0)
#define TG_LAUNCH3(kern, gx, gy, gz, block, lds, stream, ...)                                                                      \
    do {                                                                                                                          \
        hipLaunchKernelGGL(kern, dim3((unsigned)(gx), (unsigned)(gy), (unsigned)(gz)), dim3((unsigned)(block), 1, 1), (size_t)(lds), \
                           stream, __VA_ARGS__);                                                                                  \
        const int _le = (int)hipGetLastError();                                                                                   \
        if (_le != 0 && g_launch_rc == 0) { g_launch_rc = _le; g_launch_name = #kern; }                                           \
    } while (0)
static int tg_launch_error(const char** name) {
    int e = g_launch_rc;
    *name = g_launch_name;
    if (e == 0) { e = (int)hipGetLastError(); *name = "(asynchronous error of an earlier call)"; }
    g_launch_rc = 0; g_launch_name = nullptr;
    return e;
}
static bool tg_launch_failed() { return g_launch_rc != 0; }
static const char* tg_hip_errstr(int e) { return hipGetErrorString((hipError_t)e); }
static int tg_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, tg_stream_t s) {
    return (int)hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, s);
}
static int tg_memset(void* dst, int v, size_t n, tg_stream_t s) { return (int)hipMemsetAsync(dst, v, n, s); }
static int tg_memcpy_h2d(void* dst, const void* src, size_t n, tg_stream_t s) {      // pageable source: staged before the call returns
    return (int)hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s);
}
static int tg_memcpy(void* dst, const void* src, size_t n, tg_stream_t s) {
    return (int)hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s);
}
typedef hipEvent_t tg_event_t;
static int tg_stream_create(tg_stream_t* s) { return (int)hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
static void tg_stream_destroy(tg_stream_t s) { if (s) (void)hipStreamDestroy(s); }
static int tg_event_create(tg_event_t* e) { return (int)hipEventCreateWithFlags(e, hipEventDisableTiming); }
static void tg_event_destroy(tg_event_t e) { if (e) (void)hipEventDestroy(e); }
static void tg_event_record(tg_event_t e, tg_stream_t s) { (void)hipEventRecord(e, s); }
static void tg_stream_wait(tg_stream_t s, tg_event_t e) { (void)hipStreamWaitEvent(s, e, 0); }
static void tg_event_host_wait(tg_event_t e) { (void)hipEventSynchronize(e); }
// page-locked HOST memory (the library still allocates no device memory): the source of a truly asynchronous, capturable copy
static void* tg_host_pinned_alloc(size_t n) { void* p = nullptr; return hipHostMalloc(&p, n, hipHostMallocDefault) == hipSuccess ? p : nullptr; }
static void tg_host_pinned_free(void* p) { if (p) (void)hipHostFree(p); }
static bool tg_stream_capturing(tg_stream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(s, &st) == hipSuccess && st == hipStreamCaptureStatusActive;
}
#endif

// ----------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int tg_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
extern "C" const char* tg_last_error(void) { return g_err.c_str(); }
// status of the launches issued since the last call: TG_OK, or TG_ERR_HIP naming the first kernel whose launch failed
static int tg_launch_status() {
    const char* name = nullptr;
    const int e = tg_launch_error(&name);
    if (e == 0) return TG_OK;
    return tg_fail(TG_ERR_HIP, "launch of %s failed with HIP error %d (%s)", name ? name : "?", e, tg_hip_errstr(e));
}
#define TG_LAUNCH_CK() do { int _rc = tg_launch_status(); if (_rc) return _rc; } while (0)
extern "C" int tg_abi_version(void) { return TG_ABI_VERSION; }
// internal precision code of a handle whose S is bf16-exact (PrecBF16x2S, tg_device.h): never accepted from a caller's tg_config
enum { TG_PREC_BF16X2S = 3 };

static inline size_t rup(size_t x, size_t m) { return (x + m - 1) / m * m; }

#define TG_MAX_BANDS 16
static constexpr int TG_ROWPASS_MAX_V = 16384;        // tg_adam_rowpass: 512 threads x 8 float4 per array
#ifndef TG_FWD_WIDE
#define TG_FWD_WIDE 1                                 // forward GEMM on 128 x 512 tiles where the 256^2 geometry would be used and Kp % 512 == 0
#endif
struct TgLayout {
    int C, K, V, Vtot, Kp, Vp, Vr, Cp, Cr, nvt, nct, nkt, nrb, nsplit, ESZ, BKE, prec, full, T;
    size_t o_Sk, o_St, o_StP, o_dG, o_Gp, o_Ghat, o_Gpart, o_genepart, o_genestat, o_gnorm2, o_voxstat, o_vnorm2,
        o_d, o_coef, o_vcoef, o_rshift, o_rinvz, o_rscale, o_rmul, o_fgate, o_densw, o_part, o_rowq, o_rowpair, o_scal, o_fsum, o_X,
        o_gfrac, o_rowent, o_extra, o_WG, o_Y, o_nbpart, o_nbstat, o_wgn2, o_nbcoef, o_ctmask, o_ctpart, o_csr[6][3],
        o_acY, o_acZ, o_acTg, o_acTm, o_acrefp, o_acr, o_acrc, o_acpart, o_acstat, o_acstat2, o_accoef, o_acB1, o_acD,
        o_accmpart, o_accm, o_actnorm, total;
    int T_ct, Tp, has_nb, has_ct, has_ac, bands, nranks;
    // spatial terms: the spots their kernels run over.  One GPU: Vs = V.  Spot shard: Vs = ALL spots -- every rank gathers Ghat (and,
    // once, G) and evaluates the spatial terms redundantly on the whole graph; shards are then blocks of Vmaxl = ceil(Vtot / ranks)
    // spots (only the last one shorter), so that the gathered blocks ARE the global matrix.  Vsr = rows allocated (>= Vs).
    int Vs, Vsr, Vmaxl, sp_shard, nrb_s;
    size_t o_GhatFull, o_Gfull;
    int smallc;                                       // C <= 32 (clusters mode): the iteration runs on tg_sc_forward / tg_sc_backward
    size_t o_Sa, o_Sx, o_spotpart;
    int fwd_wide;                                     // forward GEMM on 128 x 512 tiles (TgGeoWide)
    int fwd_units;                                    // workgroups the forward (tile, step) space is cut into (tg_kernels.h, tg_fwd_unit_*)
    int bwd_T;                                        // tile edge of the backward GEMM (T, or 128 under the 256 layout: tg_make_layout)
    size_t o_gathered, pair_stride;
    size_t o_peertab;                                 // spot shards: every rank's mailbox as mapped here (TgPeerLink::box), 128 bytes
    size_t step_e2, step_e3, step_e1, step_floats;    // spot shards: regions of the peer transport's step area (granules), tg_peer.h
    size_t s_M, s_m1, s_m2, s_F, s_total;
};

// Pieces per gene tile of the forward GEMM (tg_kernels.h: the (spot tile, step) space of one gene tile cut into `units` equal
// pieces, the same for every gene tile; workgroups = units * nkt): minimise a small cost model (microseconds, measured on
// MI355X at cfg2: profiles/r01) of
//     rounds of workgroups x (steps per piece x time per step + fixed cost per segment) + the write and re-read of the partial tiles
// (tg_ghat_reduce streams them at ~1.5 TB/s) over the candidates: the tiles cut into s = 1, 2, ... equal ranges (no piece crosses
// a tile), and r = 1, 2, 3 full rounds of the chip's workgroup slots (every CU the same number of steps whatever the tile count:
// cfg2's split-bf16 forward runs on 128 x 512 tiles: nvt = Vr / 128 = 80 spot tiles x nkt = 2 gene tiles of 938 steps -- three
// equal ranges each were 480 workgroups = 1.875 rounds of 256; 128 pieces x 2 gene tiles are exactly one round, and 128 divides
// 8 nvt = 640, which is what admits the stream-K candidate below.  On 256^2 tiles (nvt = 40, nkt = 4: plain bf16, fp32) the
// candidate is 64 pieces x 4, 64 | 320.)
static int tg_choose_units(int nvt, int nkt, int nsteps, int slots, int precision, int tile_edge, size_t tile_bytes) {
    double t_step = precision == TG_PREC_F32 ? 7.7 : 2.4;       // 256^2 tile, one contraction step (bf16: 64 elements)
    if (tile_edge != 256) t_step *= 0.5;                        // a quarter of the work on half a CU
    const long long G = (long long)nvt * nsteps;
    int best = nvt;
    double best_cost = 1e30;
    auto consider = [&](long long units, bool aligned, double partial_rate = 1.5e6) {
        if (units < 1 || units > G || G / units < (units > nvt ? 8 : 1)) return;        // keep >= 8 contraction steps per piece
        const long long wgs = units * nkt, rounds = (wgs + slots - 1) / slots, segs = (aligned ? units : units + nvt) * nkt;
        const double steps = (double)G / (double)units;
        const double cost = (double)rounds * (steps * t_step + 8.0 * (double)segs / (double)wgs) + (double)segs * (double)tile_bytes * 2.0 / partial_rate;
        if (cost < best_cost * 0.995) { best_cost = cost; best = (int)units; }
    };
    for (int s = 1; s <= 32; ++s) consider((long long)nvt * s, true);
    // stream-K pieces: whole rounds of the chip, and a piece length whose start offsets repeat every 8 pieces (units | 8 nvt), so
    // that the pieces of one XCD walk their tiles in step and share S^T through its L2 (tg_fwd_unit_map)
    bool any_streamk = false;
    for (int r = 1; r <= 3; ++r) {
        const long long u = (long long)slots * r / nkt;
        if (((long long)slots * r) % nkt == 0 && u > 0 && (8LL * nvt) % u == 0 && u % nvt != 0) { consider(u, false); any_streamk = true; }
    }
    // Round 6: shapes without such a piece length (a spot shard: 10 / 20 / 40 spot tiles) fell back to equal ranges per tile -- 12 x 10 x 2 =
    // 240 workgroups on a 1/8 shard of cfg2, 6 % of the chip idle.  ONE exact round of stream-K pieces without the XCD alignment measures
    // faster there: forward 190 -> 177 us at 1/8, 344 -> 321 at 1/4, 658 -> 624 at 1/2 of cfg2, fp32 651 -> 614, 30 000 x 1 000 x 3 300
    // 498 -> 415 (two or three rounds lose) -- but SLOWER with ONE gene tile (26 431 x 249 x 9 852: 349 -> 377 us) and with short pieces
    // (8 000 x 500 x 5 000 in bf16, 19 steps per piece: 60 -> 68 us): profiles/r06/run5_shard_forward.  Hence: only where the equal-ranges
    // choice is itself a single, partly filled round, several gene tiles share an M panel, and a piece is at least 64 steps long; its extra
    // partial segments priced at what tg_ghat_reduce really streams them at (3.7 - 4.8 TB/s since round 3, not the 1.5 this model was
    // first fitted with).
    const long long u1 = slots % nkt == 0 ? (long long)slots / nkt : 0;
    if (!any_streamk && nkt >= 2 && u1 > 0 && (long long)best * nkt <= slots && u1 % nvt != 0 && G / u1 >= 64) consider(u1, false, 4.0e6);
    return best;
}
// partial slots per tile that `units` pieces need: the most segments any spot tile is cut into
static int tg_fwd_slots(int nvt, int nsteps, int units) {
    const long long G = (long long)nvt * nsteps;
    int mx = 1;
    for (int vt = 0; vt < nvt; ++vt) { const int n = tg_fwd_nseg(vt, nsteps, G, units); if (n > mx) mx = n; }
    return mx;
}

// Does this configuration train on the clusters-mode kernels (tg_sc_forward / tg_sc_backward)?  ONE predicate for the precision
// override at the top of tg_make_layout and for L->smallc: at most TG_SC_MAXC rows of M, one GPU, no spatial terms, rows within the
// one-kernel update, tile_size not pinned.  (At most 32 cells are ONE cell tile, so the cell-band pipeline -- bands <= cell tiles --
// never applies to such a problem: no separate bands condition.)
static bool tg_is_clusters_problem(const tg_config* c) {
    const bool spatial = c->lambda_neighborhood_g1 > 0.f || c->lambda_ct_islands > 0.f || c->lambda_getis_ord > 0.f || c->lambda_moran > 0.f ||
                         c->lambda_geary > 0.f;
    const int vtot = c->n_spots_total > 0 ? c->n_spots_total : c->n_spots;
    return c->n_cells >= 1 && c->n_cells <= TG_SC_MAXC && !spatial && c->n_ranks == 0 && vtot == c->n_spots && c->n_spots <= TG_ROWPASS_MAX_V &&
           c->tile_size == 0;
}

static int tg_make_layout(const tg_config* cfg_in, TgLayout* L) {
    if (!cfg_in) return tg_fail(TG_ERR_INVALID, "null config");
    // Clusters-mode problems (at most TG_SC_MAXC rows of M, one GPU, no spatial terms, tile_size not pinned) TRAIN on the exact-fp32
    // tg_sc_* kernels whatever gemm precision is configured.  Everything else such a handle does -- tg_mapper_validate,
    // tg_mapper_project, tg_mapper_project_genes: GEMMs with at most 32 contraction rows, i.e. free -- then runs in exact fp32 too,
    // so that validation losses and projections come from the same numerical path as the training history (round 3 ran them at
    // the configured precision).  The EFFECTIVE precision is L->prec; tg_mapper_create stores it back into the handle's config.
    tg_config cfg_eff = *cfg_in;
    if (tg_is_clusters_problem(cfg_in) && cfg_in->precision >= 0 && cfg_in->precision <= 2) cfg_eff.precision = TG_PREC_F32;
    const tg_config* cfg = &cfg_eff;
    if (cfg->abi_version != TG_ABI_VERSION) return tg_fail(TG_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, TG_ABI_VERSION);
    if (cfg->n_cells < 1 || cfg->n_genes < 1 || cfg->n_spots < 1) return tg_fail(TG_ERR_INVALID, "empty problem: C=%d K=%d V=%d", cfg->n_cells, cfg->n_genes, cfg->n_spots);
    if (cfg->precision < 0 || cfg->precision > 2) return tg_fail(TG_ERR_INVALID, "unknown precision %d", cfg->precision);
    if (cfg->mode != TG_MODE_MAPPER && cfg->mode != TG_MODE_CONSTRAINED) return tg_fail(TG_ERR_INVALID, "unknown mode %d", cfg->mode);
    if (cfg->lambda_g1 == 0.f) return tg_fail(TG_ERR_INVALID, "lambda_g1 cannot be 0.");   // mapping_utils.py:206-207
    if (cfg->has_d_source && !cfg->has_density) return tg_fail(TG_ERR_INVALID, "d_source requires d");
    if (cfg->n_ranks < 0 || cfg->n_ranks > 4096) return tg_fail(TG_ERR_INVALID, "n_ranks out of range");
    memset(L, 0, sizeof *L);
    L->C = cfg->n_cells; L->K = cfg->n_genes; L->V = cfg->n_spots;
    L->Vtot = cfg->n_spots_total > 0 ? cfg->n_spots_total : cfg->n_spots;
    L->prec = cfg->precision;
    L->ESZ = cfg->precision == TG_PREC_BF16 ? 2 : 4;      // operand bytes per contraction element (bf16x3: hi + lo)
    L->BKE = cfg->precision == TG_PREC_BF16 ? 64 : 32;     // contraction elements per 128-byte step
    if (cfg->tile_size != 0 && cfg->tile_size != 128 && cfg->tile_size != 256) return tg_fail(TG_ERR_INVALID, "tile_size must be 0, 128 or 256");
    // large geometry (256 x 256 tiles, one 512-thread workgroup per CU) once every tile axis is long enough to fill the chip
    // (measured, profiles/r02/run10_tiles: 30k x 1k x 500 and x 1000 are 8 - 11 % faster on 256 tiles, 20k x 1k x 324 is 14 % faster on
    //  128: there the spots pad to 512 instead of 384)
    const bool pad_ok = rup((size_t)L->V, 256) * 100 <= rup((size_t)L->V, 128) * 115;
    // The tile copies (TgKtileDma, tg_kernels.h) address an operand tile through ONE buffer descriptor: byte count, per-lane offsets
    // and the step offset are 32-bit, and out-of-range buffer loads are clamped to zero instead of faulting.  The longest operand rows
    // are those of the forward's S^T image (one gene row = every cell: 4 bytes per cell in fp32 / split bf16, 2 in bf16): a tile of
    // `rows` gene rows must stay below 4 GiB.  The geometry choice keeps it there (128 x 512 tiles up to ~2.1 M cells, 256^2 up to
    // ~4.2 M, 128^2 up to ~8.4 M, twice that in bf16); beyond, create refuses instead of computing a wrong Ghat.
    const size_t st_row_bytes = (rup((size_t)L->C, 64) / (size_t)L->BKE) * 128;
    auto dma_fits = [&](int rows) { return (size_t)rows * st_row_bytes < ((size_t)1 << 32); };
    L->T = cfg->tile_size ? cfg->tile_size : ((L->C >= 4096 && L->V >= 448 && pad_ok && dma_fits(256)) ? 256 : 128);
    if (!dma_fits(L->T))
        return tg_fail(TG_ERR_UNSUPPORTED, "n_cells = %d: a %d-row tile of the S^T operand image (%zu bytes per gene row) exceeds the 32-bit tile offsets of the copy engine path%s",
                       L->C, L->T, st_row_bytes, cfg->tile_size == 256 ? " (tile_size 128 or 0 would fit)" : "");
    L->has_nb = cfg->lambda_neighborhood_g1 > 0.f;
    L->has_ct = cfg->lambda_ct_islands > 0.f;
    L->has_ac = cfg->lambda_getis_ord > 0.f || cfg->lambda_moran > 0.f || cfg->lambda_geary > 0.f;
    if (L->has_ac && cfg->nnz_s < 1) return tg_fail(TG_ERR_INVALID, "the spatial autocorrelation terms need the spatial_weights graph");
    if ((L->has_nb || L->has_ct || L->has_ac) && cfg->mode != TG_MODE_MAPPER) return tg_fail(TG_ERR_INVALID, "spatial terms exist only in Mapper (mapping_utils.py:366-375 ignores them in constrained mode)");
    const bool spatial = L->has_nb || L->has_ct || L->has_ac;
    L->sp_shard = (spatial && (cfg->n_ranks >= 1 || L->Vtot != L->V)) ? 1 : 0;
    if (L->sp_shard) {
        if (cfg->n_ranks < 1) return tg_fail(TG_ERR_INVALID, "a spot shard with spatial terms needs n_ranks");
        L->Vmaxl = (L->Vtot + cfg->n_ranks - 1) / cfg->n_ranks;
        if (cfg->spot_offset < 0 || cfg->spot_offset % L->Vmaxl != 0 || cfg->spot_offset >= L->Vtot ||
            L->V != ((L->Vtot - cfg->spot_offset < L->Vmaxl) ? L->Vtot - cfg->spot_offset : L->Vmaxl))
            return tg_fail(TG_ERR_INVALID, "spatial terms on spot shards: shard r must hold the spots [r * ceil(V / ranks), ...) (got offset %d, %d of %d spots, %d ranks)",
                           cfg->spot_offset, L->V, L->Vtot, cfg->n_ranks);
    }
    if (L->has_ct && cfg->n_cell_types < 1) return tg_fail(TG_ERR_INVALID, "lambda_ct_islands > 0 needs n_cell_types >= 1");
    if (L->has_nb && cfg->nnz_w < 1) return tg_fail(TG_ERR_INVALID, "lambda_neighborhood_g1 > 0 needs the voxel_weights graph");
    if (L->has_ct && cfg->nnz_n < 1) return tg_fail(TG_ERR_INVALID, "lambda_ct_islands > 0 needs the neighborhood_filter graph");
    L->T_ct = L->has_ct ? cfg->n_cell_types : 0;
    L->Tp = (int)rup((size_t)(L->T_ct > 0 ? L->T_ct : 1), 4);
    L->Kp = (int)rup((size_t)L->K + 1 + L->T_ct, L->T);
    L->Vp = (int)rup(L->V, 64);
    L->Vr = (int)rup(L->V, L->T);
    L->Cp = (int)rup(L->C, 64);
    L->Cr = (int)rup(L->C, L->T);
    L->nvt = L->Vr / L->T; L->nct = L->Cr / L->T; L->nkt = L->Kp / L->T;
    if ((size_t)L->T * ((size_t)L->Kp / (size_t)L->BKE) * 128 >= ((size_t)1 << 32))      // backward operand tiles: rows of Kp genes
        return tg_fail(TG_ERR_UNSUPPORTED, "n_genes = %d: an operand tile exceeds the 32-bit tile offsets of the copy engine path", L->K);
    L->fwd_wide = (TG_FWD_WIDE && L->T == 256 && L->Kp % 512 == 0 && cfg->precision == TG_PREC_BF16X3 && dma_fits(512)) ? 1 : 0;   // (measured: plain bf16 is faster on 256^2, profiles/r02/run8_wide)
    // Tile edge of the backward GEMM.  Under the 256 layout both 256^2 (one workgroup per CU) and 128^2 (two) are legal; the choice
    // is a FIXED function of the shape (round 2 timed both on the first step: a hidden host synchronisation inside tg_mapper_step
    // and a box-dependent kernel choice).  With the dense XCD-banded tile map 256^2 wins or ties every X-only shape measured
    // (profiles/r02/run10_tiles + run11_dense_map).  The row-dot epilogue (spot shards, rows beyond TG_ROWPASS_MAX_V) takes 128^2
    // tiles when the 256^2 grid is under 4 rounds of workgroups with a last round under 40 % full (1/8 shard of cfg2: 590 tiles =
    // 2.3 rounds, 242 -> 222 us; at 4.6 and 9.2 rounds 256^2 wins).  cfg->bwd_tile pins it.
    if (cfg->bwd_tile != 0 && cfg->bwd_tile != 128 && cfg->bwd_tile != 256) return tg_fail(TG_ERR_INVALID, "bwd_tile must be 0, 128 or 256");
    L->bwd_T = L->T;
    if (L->T == 256) {
        const bool rowdot = (cfg->n_ranks >= 1) || (L->Vtot != L->V) || (L->V > TG_ROWPASS_MAX_V);
        const double rounds = (double)L->nct * (double)L->nvt / 256.0, frac = rounds - (double)(long)rounds;
        if (cfg->bwd_tile) L->bwd_T = cfg->bwd_tile;
        else if (rowdot && rounds < 4.0 && frac > 0.0 && frac < 0.4) L->bwd_T = 128;
        // (Round 6 measured a SPLIT pass -- the cell tiles of the whole rounds on 256^2 tiles, the ragged band as a second launch on 128^2:
        //  slower on every shard shape, 220 -> 277 / 395 -> 464 / 763 -> 795 us at 1/8, 1/4, 1/2 of cfg2: a round of 512 small tiles costs
        //  0.7 of a 256^2 round however full it is, and the second kernel starts behind the first one's drain.  profiles/r06/run2.)
    }
    L->nrb = (L->Vr + TG_RB - 1) / TG_RB;
    L->Vs = L->sp_shard ? L->Vtot : L->V;
    L->Vsr = L->sp_shard ? (int)rup((size_t)cfg->n_ranks * L->Vmaxl, TG_RB) : L->Vr;
    L->nrb_s = (L->Vsr + TG_RB - 1) / TG_RB;
    L->full = (cfg->mode == TG_MODE_CONSTRAINED) || cfg->lambda_r != 0.f || cfg->lambda_l1 != 0.f || cfg->lambda_l2 != 0.f;
    const int nsteps = L->Cp / L->BKE;
    const int slots = 256 * (L->T == 256 ? 1 : 2);
    {   // forward work decomposition (the forward kernel may run on 128 x 512 tiles: same tile count x area)
        const int f_nvt = L->fwd_wide ? L->Vr / 128 : L->nvt, f_nkt = L->fwd_wide ? L->Kp / 512 : L->nkt;
        const int splits = cfg->fwd_splits > nsteps ? nsteps : cfg->fwd_splits;
        const long long G = (long long)f_nvt * nsteps;
        L->fwd_units = splits > 0 ? f_nvt * splits
                     : splits < 0 ? (int)(-(long long)splits > G ? G : -(long long)splits)          // (tests / tuning: that many pieces per gene tile)
                                  : tg_choose_units(f_nvt, f_nkt, nsteps, slots, cfg->precision, L->T, (size_t)(L->fwd_wide ? 128 * 512 : L->T * L->T) * 4);
        L->nsplit = tg_fwd_slots(f_nvt, nsteps, L->fwd_units);
    }
    // cell-band software pipeline (backward GEMM | streaming Adam | next forward GEMM on three streams): single-GPU Mapper only.
    // Opt-in (pipeline_bands >= 2): measured SLOWER than the sequential schedule on MI355X (profiles/r01/run14): the 256^2 GEMM
    // workgroups fill the VGPR file of their CU, so the streaming kernel cannot co-reside and only takes CUs away from the GEMMs.
    if (cfg->pipeline_bands < 0 || cfg->pipeline_bands > TG_MAX_BANDS) return tg_fail(TG_ERR_INVALID, "pipeline_bands must be in [0, %d]", TG_MAX_BANDS);
    L->bands = 1;
    if (cfg->mode == TG_MODE_MAPPER && L->Vtot == L->V) {
        if (cfg->pipeline_bands > 1) L->bands = cfg->pipeline_bands;
        if (L->bands > L->nct) L->bands = L->nct;
    }
    if (L->bands > 1) {                                   // one forward partial per cell band; a whole forward pass (first step, validate,
        L->nsplit = L->bands;                             // project) cuts every tile into as many equal ranges
        L->fwd_units = (L->fwd_wide ? L->Vr / 128 : L->nvt) * L->bands;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += rup(bytes, 256); return o; };
    L->o_Sk = take((size_t)L->Cr * L->Kp * L->ESZ);
    L->o_St = take((size_t)L->Kp * L->Cp * L->ESZ);
    L->o_StP = take((size_t)L->Kp * L->Cp * L->ESZ);      // operand image of a block of genes handed to tg_mapper_project_genes
    L->o_dG = take((size_t)L->Vr * L->Kp * L->ESZ);
    const size_t vrows = (L->sp_shard && L->Vmaxl > L->Vr) ? (size_t)L->Vmaxl : (size_t)L->Vr;   // (a gathered block is Vmaxl rows)
    L->o_Gp = take(vrows * L->Kp * 4);
    L->o_Ghat = take(vrows * L->Kp * 4);
    L->o_Gpart = take((size_t)L->nsplit * L->Vr * L->Kp * 4);
    L->o_genepart = take((size_t)L->nrb * 2 * L->Kp * 4);
    L->o_genestat = take((size_t)2 * L->Kp * 4);
    L->o_gnorm2 = take((size_t)(L->Kp + 64) * 4);         // [Kp] |G_k|^2, then [Kp] = sum of the density prior (set-up exchange vector)
    L->o_voxstat = take((size_t)((L->Kp + TG_GH_COLS - 1) / TG_GH_COLS) * 2 * L->Vr * 4);
    L->o_vnorm2 = take((size_t)L->Vr * 4);
    L->o_d = take((size_t)L->Vr * 4);
    L->o_coef = take((size_t)2 * L->Kp * 4);
    L->o_vcoef = take((size_t)3 * L->Vr * 4);
    L->o_rshift = take((size_t)L->Cp * 4);
    L->o_rinvz = take((size_t)L->Cp * 4);
    L->o_rscale = take((size_t)L->Cp * 4);
    L->o_rmul = take((size_t)L->Cp * 4);
    L->o_fgate = take((size_t)L->Cp * 4);
    L->o_densw = take((size_t)L->Cp * 4);
    const size_t np1 = L->full ? TGP1_N : 1;
    L->o_part = take((size_t)(L->Vr / 128) * np1 * L->C * 4);      // (one partial per spot tile of the backward GEMM, 128 or 256 wide)
    L->o_rowq = take((size_t)TGP1_N * L->C * 4);
    L->pair_stride = (size_t)2 * L->C + TG_PAIR_TAIL;     // (max, sum exp) pairs + the per-rank history partials
    L->o_rowpair = take(L->pair_stride * 4);
    L->nranks = cfg->n_ranks > 1 ? cfg->n_ranks : 1;
    if (cfg->n_ranks >= 1 || L->Vtot != L->V) {
        L->o_gathered = take((size_t)L->nranks * L->pair_stride * 4);   // (n_ranks = 1: a 1-rank communicator)
        L->o_peertab = take(256);
        L->step_e2 = 0;
        L->step_e3 = rup((size_t)2 * L->Kp, 64);
        L->step_e1 = L->step_e3 + rup((size_t)TGP1_N * L->C, 64);
        L->step_floats = L->step_e1 + rup(L->pair_stride, 64);
    }
    L->o_scal = take(64 * 4);
    L->o_fsum = take(64 * 4);
    L->o_X = take((size_t)L->C * L->Vp * 4);
    L->o_gfrac = take((size_t)L->Kp * 4);
    L->o_rowent = take((size_t)L->Cp * 4);
    // small-C path (tg_kernels.h): single GPU, rows within the one-kernel update, no spatial terms (their extra gradient rides dGhat)
    L->smallc = (tg_is_clusters_problem(cfg) && L->bands == 1) ? 1 : 0;           // (bands == 1 always holds here: one cell tile)
    L->o_spotpart = take((size_t)L->nrb * 2 * 4);               // per-block sums of the spots' loss terms (32- or 64-spot blocks)
    if (L->smallc) {
        const size_t cm = (size_t)tg_sc_cm(L->C);
        L->o_Sa = take(cm * L->Kp * 4);                               // operand images of S (tg_prep_ssmall)
        L->o_Sx = take((size_t)16 * ((cm + 15) / 16) * L->Kp * 4);
    }
    if (L->sp_shard) {
        L->o_GhatFull = take((size_t)L->Vsr * L->Kp * 4);
        L->o_Gfull = take((size_t)L->Vsr * L->Kp * 4);
    }
    if (L->has_nb || L->has_ct || L->has_ac) L->o_extra = take((size_t)L->Vsr * L->Kp * 4);
    if (L->has_nb) {
        L->o_WG = take((size_t)L->Vsr * L->Kp * 4);
        L->o_Y = take((size_t)L->Vsr * L->Kp * 4);
        L->o_nbpart = take((size_t)L->nrb_s * 2 * L->Kp * 4);
        L->o_nbstat = take((size_t)2 * L->Kp * 4);
        L->o_wgn2 = take((size_t)2 * L->Kp * 4);
        L->o_nbcoef = take((size_t)2 * L->Kp * 4);
    }
    if (L->has_ct) {
        L->o_ctmask = take((size_t)L->Vsr * L->Tp * 4);
        L->o_ctpart = take((size_t)L->Vsr * 4);
    }
    if (L->has_ac) {
        const size_t vk = (size_t)L->Vsr * L->Kp * 4, kp = (size_t)L->Kp * 4;
        L->o_acY = take(vk); L->o_acZ = take(vk); L->o_acTg = take(vk); L->o_acTm = take(vk); L->o_acB1 = take(vk); L->o_acD = take(vk);
        L->o_acrefp = take(kp); L->o_acr = take((size_t)L->Vsr * 4); L->o_acrc = take((size_t)L->Vsr * 4);
        L->o_acpart = take((size_t)L->nrb_s * TGAC_NSTAT * kp); L->o_acstat = take(TGAC_NSTAT * kp); L->o_acstat2 = take(3 * kp);
        L->o_accoef = take(TGAC_NCOEF * kp); L->o_accmpart = take((size_t)L->nrb_s * kp); L->o_accm = take(kp); L->o_actnorm = take(4 * kp);
    }
    for (int gph = 0; gph < 6; ++gph) {           // 0: W, 1: W^T, 2: N, 3: N^T, 4: Ws, 5: Ws^T
        const bool on = gph < 2 ? L->has_nb : (gph < 4 ? L->has_ct : L->has_ac);
        const size_t nnz = gph < 2 ? (size_t)cfg->nnz_w : (gph < 4 ? (size_t)cfg->nnz_n : (size_t)cfg->nnz_s);
        if (!on) continue;
        L->o_csr[gph][0] = take((size_t)(L->Vs + 1) * 4);
        L->o_csr[gph][1] = take(nnz * 4);
        L->o_csr[gph][2] = take(nnz * 4);
    }
    L->total = off;
    size_t so = 0;
    auto stake = [&](size_t bytes) { size_t o = so; so += rup(bytes, 256); return o; };
    L->s_M = stake((size_t)L->C * L->Vp * 4);
    L->s_m1 = stake((size_t)L->C * L->Vp * 4);
    L->s_m2 = stake((size_t)L->C * L->Vp * 4);
    L->s_F = stake((size_t)3 * L->Cp * 4);
    L->s_total = so;
    return TG_OK;
}

extern "C" int tg_query_sizes(const tg_config* cfg, tg_sizes* out) {
    TgLayout L;
    int rc = tg_make_layout(cfg, &L);
    if (rc) return rc;
    if (!out) return tg_fail(TG_ERR_INVALID, "null out");
    out->state_bytes = L.s_total;
    out->workspace_bytes = L.total;
    out->m_pitch = L.Vp;
    out->history_terms = TG_H_NTERMS;
    out->peer_step_floats = L.step_floats;
    return TG_OK;
}

struct tg_mapper {
    tg_config cfg;
    TgLayout L;
    unsigned char* ws;
    unsigned char* st;
    tg_stream_t stream;
    int64_t step;
    bool ready;
    tg_stream_t s_adam, s_fwd;                       // library-owned streams of the cell-band pipeline
    tg_event_t e_bwd[TG_MAX_BANDS], e_adam[TG_MAX_BANDS], e_fwd;
    // history scalars deferred from tg_launch_loss to one extra workgroup of the next update kernel (tg_dghat_emit<SELF>)
    bool stream_once;                                // the per-iteration arrays exceed the MALL: non-temporal accesses (tg_ld_stream)
    bool fin_pending;
    TgFinalizeArgs fin_args;
    tg_comm* comm;                                   // spot-sharded run: the communicator (borrowed), else null
    bool fused;                                      // ... whose exchanges happen inside the kernels (peer transport with a step area)
    // profiling
    bool prof;
    std::vector<std::string> prof_names;
#ifndef TG_SIM
    std::vector<hipEvent_t> prof_events;
#endif
    float* fp(size_t off) const { return (float*)(ws + off); }
};

#define TG_CK(expr)                                                                          \
    do {                                                                                     \
        int _e = (int)(expr);                                                                \
        if (_e != 0) return tg_fail(TG_ERR_HIP, "%s failed with HIP error %d (%s:%d)", #expr, _e, __FILE__, __LINE__); \
    } while (0)

static void tg_prof_mark(tg_mapper* m, const char* name) {
    if (!m->prof) return;
#ifndef TG_SIM
    hipEvent_t e;
    hipEventCreate(&e);
    hipEventRecord(e, m->stream);
    m->prof_events.push_back(e);
#endif
    m->prof_names.push_back(name);
}

template <class PR, class GE>
static int tg_lds_attr() {
#ifndef TG_SIM
    const int bytes = GE::LDS_BYTES;
    TG_CK(hipFuncSetAttribute((const void*)tg_fwd_kernel<PR, GE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    TG_CK(hipFuncSetAttribute((const void*)tg_fwd_kernel_b<PR, GE>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if constexpr (GE::TM == 256 && PR::NP == 2) {
        TG_CK(hipFuncSetAttribute((const void*)tg_fwd_kernel<PR, TgGeoWide>, hipFuncAttributeMaxDynamicSharedMemorySize, TgGeoWide::LDS_BYTES));
        TG_CK(hipFuncSetAttribute((const void*)tg_fwd_kernel_b<PR, TgGeoWide>, hipFuncAttributeMaxDynamicSharedMemorySize, TgGeoWide::LDS_BYTES));
    }
    TG_CK(hipFuncSetAttribute((const void*)tg_bwd_kernel_b<PR, GE>, hipFuncAttributeMaxDynamicSharedMemorySize, GE::BWD_LDS_BYTES));
#define TG_BWD_ATTR(F, R, S) TG_CK(hipFuncSetAttribute((const void*)tg_bwd_kernel<PR, GE, F, R, S>, hipFuncAttributeMaxDynamicSharedMemorySize, GE::BWD_LDS_BYTES))
    TG_BWD_ATTR(false, true, true); TG_BWD_ATTR(true, true, true); TG_BWD_ATTR(false, false, true); TG_BWD_ATTR(false, false, false);
#undef TG_BWD_ATTR
#endif
    return TG_OK;
}

// small-C path: one switch over the compile-time cluster bound (C rounded up to a multiple of 4) and the per-spot-sums flag
#define TG_SC_DISPATCH(C, VOX, STMT)                                                                               \
    do {                                                                                                           \
        switch (tg_sc_cm(C)) {                                                                                     \
        case 4: if (VOX) { STMT(4, true); } else { STMT(4, false); } break;                                        \
        case 8: if (VOX) { STMT(8, true); } else { STMT(8, false); } break;                                        \
        case 12: if (VOX) { STMT(12, true); } else { STMT(12, false); } break;                                     \
        case 16: if (VOX) { STMT(16, true); } else { STMT(16, false); } break;                                     \
        case 20: if (VOX) { STMT(20, true); } else { STMT(20, false); } break;                                     \
        case 24: if (VOX) { STMT(24, true); } else { STMT(24, false); } break;                                     \
        case 28: if (VOX) { STMT(28, true); } else { STMT(28, false); } break;                                     \
        default: if (VOX) { STMT(32, true); } else { STMT(32, false); } break;                                     \
        }                                                                                                          \
    } while (0)
// ---- set-up ------------------------------------------------------------------------------------
static TgPrepSArgs tg_prep_s_args(tg_mapper* m, const tg_inputs* in) {
    const TgLayout& L = m->L;
    TgPrepSArgs a;
    a.S = in->S_dev; a.C = L.C; a.K = L.K; a.ldS = L.K;
    a.aug = m->cfg.has_d_source ? in->d_source_dev : nullptr;
    a.ct = L.has_ct ? in->ct_encode_dev : nullptr; a.T = L.T_ct;
    a.Sk = m->ws + L.o_Sk; a.Cr = L.Cr; a.Kp = L.Kp;
    a.St = m->ws + L.o_St; a.Cp = L.Cp;
    return a;
}
// Is S (with its augmentation columns) exactly representable in bf16?  One elementwise pass and ONE 4-byte read-back: the only
// point where tg_mapper_create waits for the stream (only with tg_config.s_exact_mode = 1).  *exact = 1 / 0.
static int tg_s_is_exact(tg_mapper* m, const tg_inputs* in, int* exact) {
    const TgLayout& L = m->L;
    int* flag = (int*)(m->fp(L.o_fsum) + 48);              // (64 floats of scratch, zeroed with the workspace; [0] is the filter sum)
    const size_t n = (size_t)L.C * (L.K + 1 + L.T_ct);
    const int grid = (int)(n / 1024 + 1 < 4096 ? n / 1024 + 1 : 4096);
    TG_LAUNCH(tg_s_exact_check, grid, 1, 256, 0, m->stream, tg_prep_s_args(m, in), flag);
    TG_LAUNCH_CK();
    int h = 1;
#ifdef TG_SIM
    h = *flag; *flag = 0;
#else
    TG_CK(hipMemcpyAsync(&h, flag, sizeof h, hipMemcpyDeviceToHost, m->stream));
    TG_CK(hipStreamSynchronize(m->stream));
    TG_CK(hipMemsetAsync(flag, 0, sizeof h, m->stream));
#endif
    *exact = (h == 0) ? 1 : 0;
    return TG_OK;
}

template <class PR>
static int tg_setup_operands(tg_mapper* m, const tg_inputs* in) {
    const TgLayout& L = m->L;
    const TgPrepSArgs a = tg_prep_s_args(m, in);
    const size_t n1 = (size_t)L.Cr * (L.Kp / PR::CH), n2 = (size_t)L.Kp * (L.Cp / PR::CH);
    TG_LAUNCH((tg_prep_sk<PR>), (n1 + 255) / 256, 1, 256, 0, m->stream, a);
    TG_LAUNCH((tg_prep_st<PR>), (n2 + 255) / 256, 1, 256, 0, m->stream, a);
    TG_LAUNCH_CK();
    if (L.smallc) {
        const int cm = tg_sc_cm(L.C);
        TG_LAUNCH(tg_prep_ssmall, (16 * ((cm + 15) / 16) * L.Kp + 255) / 256, 1, 256, 0, m->stream, in->S_dev, (long long)L.K, a.aug, L.C, cm, L.K, L.Kp, m->fp(L.o_Sa),
                  m->fp(L.o_Sx));
        TG_LAUNCH_CK();
    }
    if (L.T == 256) { const int rc = tg_lds_attr<PR, TgGeoSmall>(); if (rc != TG_OK) return rc; }     // (the backward GEMM may run on 128^2 tiles)
    return L.T == 256 ? tg_lds_attr<PR, TgGeoLarge>() : tg_lds_attr<PR, TgGeoSmall>();
}

static int tg_softmax_stats_from_scratch(tg_mapper* m) {
    const TgLayout& L = m->L;
    float* M = (float*)(m->st + L.s_M);
    TG_LAUNCH(tg_row_stats, L.C, 1, 256, 64, m->stream, (const float*)M, L.C, L.V, L.Vp, m->fp(L.o_rowpair));
    TG_LAUNCH_CK();
    return TG_OK;
}

static TgMergeArgs tg_merge_args(tg_mapper* m, const float* parts, int nparts, bool finalize, bool want_pair, float* global_hist_row, int rank) {
    const TgLayout& L = m->L;
    TgMergeArgs a;
    a.part = parts; a.nparts = nparts; a.C = L.C; a.stride = L.pair_stride;
    a.rshift = finalize ? m->fp(L.o_rshift) : nullptr;
    a.rinvz = m->fp(L.o_rinvz);
    a.rmul = m->fp(L.o_rmul); a.rscale = m->fp(L.o_rscale);
    a.pair_out = want_pair ? m->fp(L.o_rowpair) : nullptr;
    a.fgate = (m->cfg.mode == TG_MODE_CONSTRAINED) ? m->fp(L.o_fgate) : nullptr;
    a.hist = global_hist_row; a.rank = rank;
    a.lambda_g2 = m->cfg.lambda_g2; a.lambda_d = m->cfg.lambda_d; a.has_density = m->cfg.has_density;
    return a;
}
static int tg_merge(tg_mapper* m, const float* parts, int nparts, bool finalize, bool want_pair, float* global_hist_row = nullptr,
                    int rank = 0) {
    const TgLayout& L = m->L;
    const TgMergeArgs a = tg_merge_args(m, parts, nparts, finalize, want_pair, global_hist_row, rank);
    TG_LAUNCH(tg_merge_stats, (L.C + 255) / 256, 1, 256, 0, m->stream, a);
    tg_prof_mark(m, "tg_merge_stats");
    TG_LAUNCH_CK();
    return TG_OK;
}

static TgCsr tg_csr(const tg_mapper* m, int gph) {
    const TgLayout& L = m->L;
    TgCsr c;
    c.indptr = (const int*)(m->ws + L.o_csr[gph][0]);
    c.indices = (const int*)(m->ws + L.o_csr[gph][1]);
    c.data = (const float*)(m->ws + L.o_csr[gph][2]);
    return c;
}

// the Ghat / G the spatial terms are evaluated on: this handle's own, or (spot shard) the gathered global matrices
static float* tg_sp_ghat(tg_mapper* m) { return m->L.sp_shard ? m->fp(m->L.o_GhatFull) : m->fp(m->L.o_Ghat); }
static float* tg_sp_g(tg_mapper* m) { return m->L.sp_shard ? m->fp(m->L.o_Gfull) : m->fp(m->L.o_Gp); }

// spatial terms, set-up: library-owned copies of the CSR graphs, W G and |W G_k|^2 (constant: :236 recomputes it every iteration)
static int tg_setup_spatial(tg_mapper* m, const tg_inputs* in) {
    const TgLayout& L = m->L;
    const void* src[6][3] = {{in->w_indptr, in->w_indices, in->w_data}, {in->wt_indptr, in->wt_indices, in->wt_data},
                             {in->n_indptr, in->n_indices, in->n_data}, {in->nt_indptr, in->nt_indices, in->nt_data},
                             {in->s_indptr, in->s_indices, in->s_data}, {in->st_indptr, in->st_indices, in->st_data}};
    for (int gph = 0; gph < 6; ++gph) {
        const bool on = gph < 2 ? L.has_nb : (gph < 4 ? L.has_ct : L.has_ac);
        if (!on) continue;
        const size_t nnz = gph < 2 ? (size_t)m->cfg.nnz_w : (gph < 4 ? (size_t)m->cfg.nnz_n : (size_t)m->cfg.nnz_s);
        for (int j = 0; j < 3; ++j)
            if (!src[gph][j]) return tg_fail(TG_ERR_INVALID, "a spatial term is enabled but its CSR graph (or the transpose) is NULL");
        TG_CK(tg_memcpy(m->ws + L.o_csr[gph][0], src[gph][0], (size_t)(L.Vs + 1) * 4, m->stream));
        TG_CK(tg_memcpy(m->ws + L.o_csr[gph][1], src[gph][1], nnz * 4, m->stream));
        TG_CK(tg_memcpy(m->ws + L.o_csr[gph][2], src[gph][2], nnz * 4, m->stream));
    }
    if (L.has_ct && !in->ct_encode_dev) return tg_fail(TG_ERR_INVALID, "lambda_ct_islands > 0 needs ct_encode");
    return TG_OK;
}

// ... and what is derived from G over ALL the spots the spatial terms see: W G and |W G_k|^2.  One GPU: at create.  Spot shard: at
// tg_mapper_attach_comm, once the blocks of G have been gathered.
static int tg_setup_spatial_derived(tg_mapper* m) {
    const TgLayout& L = m->L;
    if (L.has_nb) {
        TgSpmmArgs a = {};
        a.W = tg_csr(m, 0); a.A = tg_sp_g(m); a.B = nullptr; a.ca = nullptr; a.cb = nullptr;
        a.Y = m->fp(L.o_WG); a.V = L.Vs; a.Kp = L.Kp; a.k_begin = 0; a.k_end = L.K;
        TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, a);
        const int nrb = (L.Vs + TG_RB - 1) / TG_RB;
        TG_LAUNCH(tg_colstats, nrb, 1, 256, 0, m->stream, (const float*)m->fp(L.o_WG), (const float*)m->fp(L.o_WG), L.Vs, L.Kp, m->fp(L.o_nbpart));
        TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_nbpart), nrb, L.Kp, m->fp(L.o_wgn2));
        TG_LAUNCH_CK();
    }
    return TG_OK;
}

// spatial terms, per iteration, part 1 (before tg_loss_finalize): W Ghat and its per-gene statistics; ct-islands mask
static int tg_launch_spatial_stats(tg_mapper* m) {
    const TgLayout& L = m->L;
    if (L.has_nb) {
        TgSpmmArgs a = {};
        a.W = tg_csr(m, 0); a.A = tg_sp_ghat(m); a.B = nullptr; a.ca = nullptr; a.cb = nullptr;
        a.Y = m->fp(L.o_Y); a.V = L.Vs; a.Kp = L.Kp; a.k_begin = 0; a.k_end = L.K;
        TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, a);
        const int nrb = (L.Vs + TG_RB - 1) / TG_RB;
        TG_LAUNCH(tg_colstats, nrb, 1, 256, 0, m->stream, (const float*)m->fp(L.o_Y), (const float*)m->fp(L.o_WG), L.Vs, L.Kp, m->fp(L.o_nbpart));
        TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_nbpart), nrb, L.Kp, m->fp(L.o_nbstat));
        tg_prof_mark(m, "tg_spatial_nb_stats");
    }
    if (L.has_ct) {
        TgCtArgs c;
        c.N = tg_csr(m, 2); c.Ghat = tg_sp_ghat(m); c.mask = m->fp(L.o_ctmask); c.ctpart = m->fp(L.o_ctpart);
        c.extra = m->fp(L.o_extra); c.V = L.Vs; c.Kp = L.Kp; c.K = L.K; c.T = L.T_ct; c.Tp = L.Tp; c.lambda_ct = m->cfg.lambda_ct_islands;
        TG_LAUNCH(tg_ct_mask, L.Vs, 1, 64, 0, m->stream, c);
        c.N = tg_csr(m, 3);
        TG_LAUNCH(tg_ct_grad, L.Vs, 1, 64, 0, m->stream, c);
        tg_prof_mark(m, "tg_spatial_ct");
    }
    return TG_OK;
}

// ---- spatial autocorrelation terms (Getis-Ord / Moran / Geary) --------------------------------------------------
static TgAcArgs tg_ac_args(tg_mapper* m, const float* X, bool setup, float* hist_row) {
    const TgLayout& L = m->L;
    TgAcArgs a = {};
    a.X = X; a.Y = m->fp(L.o_acY); a.Z = m->fp(L.o_acZ); a.r = m->fp(L.o_acr); a.rc = m->fp(L.o_acrc);
    a.Tg = m->fp(L.o_acTg); a.Tm = m->fp(L.o_acTm); a.refp = m->fp(L.o_acrefp);
    a.part = m->fp(L.o_acpart); a.stat = m->fp(L.o_acstat); a.stat2 = m->fp(L.o_acstat2); a.coef = m->fp(L.o_accoef);
    a.B1 = m->fp(L.o_acB1); a.D = m->fp(L.o_acD); a.cmpart = m->fp(L.o_accmpart); a.cm = m->fp(L.o_accm);
    a.hist = hist_row ? hist_row : m->fp(L.o_scal);
    a.V = L.Vs; a.Vr = L.Vsr; a.Kp = L.Kp; a.K = L.K; a.setup = setup ? 1 : 0;
    a.lam_getis = m->cfg.lambda_getis_ord; a.lam_moran = m->cfg.lambda_moran; a.lam_geary = m->cfg.lambda_geary;
    return a;
}

// Y = Ws X (+ local Geary sums into D), first and second stage statistics
static void tg_ac_indicators(tg_mapper* m, TgAcArgs& a) {
    const TgLayout& L = m->L;
    const int nrb = (L.Vs + TG_RB - 1) / TG_RB, kb = (L.Kp + 255) / 256;
    TgSpmmArgs sp = {};
    sp.W = tg_csr(m, 4); sp.A = a.X; sp.Y = m->fp(L.o_acY); sp.E = m->fp(L.o_acD); sp.V = L.Vs; sp.Kp = L.Kp; sp.k_begin = 0; sp.k_end = L.K;
    TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, sp);
    TG_LAUNCH(tg_ac_stats1, nrb, 1, 256, 0, m->stream, a);
    TG_LAUNCH(tg_stat_reduce, kb, 1, 256, 0, m->stream, (const float*)a.part, nrb, (int)TGAC_NSTAT, L.Kp, a.stat);
    TG_LAUNCH(tg_ac_stats2, nrb, 1, 256, 0, m->stream, a);
    TG_LAUNCH(tg_stat_reduce, kb, 1, 256, 0, m->stream, (const float*)a.part, nrb, 3, L.Kp, a.stat2);
}

static int tg_setup_autocorr(tg_mapper* m) {
    const TgLayout& L = m->L;
    const int nrb = (L.Vs + TG_RB - 1) / TG_RB;
    TG_LAUNCH(tg_csr_rowsum, (L.Vs + 255) / 256, 1, 256, 0, m->stream, tg_csr(m, 4), L.Vs, m->fp(L.o_acr), 0);
    TG_LAUNCH(tg_csr_rowsum, (L.Vs + 255) / 256, 1, 256, 0, m->stream, tg_csr(m, 4), L.Vs, m->fp(L.o_acrc), 0);
    TG_LAUNCH(tg_csr_rowsum, (L.Vs + 255) / 256, 1, 256, 0, m->stream, tg_csr(m, 5), L.Vs, m->fp(L.o_acrc), 1);
    TgAcArgs a = tg_ac_args(m, tg_sp_g(m), true, nullptr);       // indicators of G: the references (:144)
    tg_ac_indicators(m, a);
    TG_LAUNCH(tg_ac_refs, nrb, 1, 256, 0, m->stream, a);
    // |Tg_k|^2 and |Tm_k|^2 (rows 0 and 2 of tnorm)
    TG_LAUNCH(tg_colstats, nrb, 1, 256, 0, m->stream, (const float*)a.Tg, (const float*)a.Tg, L.Vs, L.Kp, a.part);
    TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)a.part, nrb, L.Kp, m->fp(L.o_actnorm));
    TG_LAUNCH(tg_colstats, nrb, 1, 256, 0, m->stream, (const float*)a.Tm, (const float*)a.Tm, L.Vs, L.Kp, a.part);
    TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)a.part, nrb, L.Kp, m->fp(L.o_actnorm) + 2 * (size_t)L.Kp);
    TG_LAUNCH_CK();
    return TG_OK;
}

// per iteration, after tg_loss_finalize (which starts the history row)
static int tg_launch_autocorr(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    const int nrb = (L.Vs + TG_RB - 1) / TG_RB, kb = (L.Kp + 255) / 256;
    TgAcArgs a = tg_ac_args(m, tg_sp_ghat(m), false, hist_row);
    tg_ac_indicators(m, a);
    if (a.lam_geary > 0.f) {
        TgSpmmArgs sz = {};
        sz.W = tg_csr(m, 5); sz.A = a.X; sz.Y = m->fp(L.o_acZ); sz.V = L.Vs; sz.Kp = L.Kp; sz.k_begin = 0; sz.k_end = L.K;
        TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, sz);
    }
    TgAcFinArgs f; f.a = a; f.tnorm = m->fp(L.o_actnorm);
    TG_LAUNCH(tg_ac_finalize, 1, 1, 1024, 64, m->stream, f);
    TG_LAUNCH(tg_ac_grad, nrb, 1, 256, 0, m->stream, a);
    TG_LAUNCH(tg_stat_reduce, kb, 1, 256, 0, m->stream, (const float*)a.cmpart, nrb, 1, L.Kp, a.cm);
    TgSpmmArgs sg = {};       // extra[:, :K] (+)= Ws^T B1 + D - cm
    sg.W = tg_csr(m, 5); sg.A = a.B1; sg.Y = m->fp(L.o_extra); sg.V = L.Vs; sg.Kp = L.Kp; sg.k_begin = 0; sg.k_end = L.K;
    sg.accumulate = L.has_nb ? 1 : 0; sg.addD = a.D; sg.addc = a.cm;
    TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, sg);
    tg_prof_mark(m, "tg_spatial_autocorr");
    return TG_OK;
}

// part 2 (after tg_loss_finalize): extra[:, :K] = W^T (nbcoef0 * WG + nbcoef1 * W Ghat)
static int tg_launch_spatial_grad(tg_mapper* m) {
    const TgLayout& L = m->L;
    if (L.has_nb) {
        TgSpmmArgs a = {};
        a.W = tg_csr(m, 1); a.A = m->fp(L.o_WG); a.B = m->fp(L.o_Y); a.ca = m->fp(L.o_nbcoef); a.cb = m->fp(L.o_nbcoef) + L.Kp;
        a.Y = m->fp(L.o_extra); a.V = L.Vs; a.Kp = L.Kp; a.k_begin = 0; a.k_end = L.K;
        TG_LAUNCH(tg_spmm, L.Vs, 1, 256, 0, m->stream, a);
        tg_prof_mark(m, "tg_spatial_nb_grad");
    }
    return TG_OK;
}

static TgFilterArgs tg_filter_args(tg_mapper* m, bool update, float lr, float* hist_row) {
    const TgLayout& L = m->L;
    TgFilterArgs a;
    float* F = (float*)(m->st + L.s_F);
    a.F = F; a.mF = F + L.Cp; a.vF = F + 2 * (size_t)L.Cp;
    a.fgate = m->fp(L.o_fgate); a.fsum = m->fp(L.o_fsum); a.rowq = m->fp(L.o_rowq);
    a.dsum = m->fp(L.o_gnorm2) + L.Kp; a.hist = hist_row ? hist_row : m->fp(L.o_scal);
    a.C = L.C; a.do_update = update ? 1 : 0; a.has_density = m->cfg.has_density;
    a.lambda_d = m->cfg.lambda_d; a.lambda_count = m->cfg.lambda_count; a.lambda_f_reg = m->cfg.lambda_f_reg;
    a.target_count = m->cfg.target_count;
    const double t = (double)(m->step + 1);
    a.step_size = (float)((double)lr / (1.0 - pow((double)m->cfg.beta1, t)));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)m->cfg.beta2, t));
    a.beta1 = m->cfg.beta1; a.beta2 = m->cfg.beta2; a.eps = m->cfg.eps;
    return a;
}
static int tg_launch_filter(tg_mapper* m, bool update, float lr, float* hist_row) {
    const TgFilterArgs a = tg_filter_args(m, update, lr, hist_row);
    TG_LAUNCH(tg_filter_kernel, 1, 1, 1024, 64, m->stream, a);
    tg_prof_mark(m, "tg_filter_kernel");
    return TG_OK;
}

extern "C" int tg_mapper_create(const tg_config* cfg, const tg_inputs* in, void* state_dev, void* workspace_dev,
                                void* hip_stream, tg_mapper** out) {
    TgLayout L;
    int rc = tg_make_layout(cfg, &L);
    if (rc) return rc;
    if (!in || !state_dev || !workspace_dev || !out) return tg_fail(TG_ERR_INVALID, "null argument");
    if (!in->S_dev || !in->G_dev || !in->M0_dev) return tg_fail(TG_ERR_INVALID, "S, G and M0 are required");
    if (cfg->has_density && !in->d_dev) return tg_fail(TG_ERR_INVALID, "has_density set but d is NULL");
    if (cfg->has_d_source && !in->d_source_dev) return tg_fail(TG_ERR_INVALID, "has_d_source set but d_source is NULL");
    if (cfg->mode == TG_MODE_CONSTRAINED && !in->F0_dev) return tg_fail(TG_ERR_INVALID, "constrained mode needs F0");
    tg_mapper* m = new (std::nothrow) tg_mapper();
    if (!m) return tg_fail(TG_ERR_INVALID, "out of host memory");
    m->cfg = *cfg; m->L = L;
    m->cfg.precision = L.prec;                            // the EFFECTIVE precision (tg_make_layout: fp32 for clusters-mode handles)
    cfg = &m->cfg;
    m->ws = (unsigned char*)workspace_dev; m->st = (unsigned char*)state_dev;
    m->stream = (tg_stream_t)hip_stream;
    m->step = 0; m->ready = false; m->prof = false; m->fin_pending = false; m->comm = nullptr; m->fused = false;
    // M, Adam m, v (fp32) and X (fp32 or bf16) of this handle against the 256 MB MALL, with room left for the GEMM operands
    m->stream_once = (size_t)L.C * L.Vp * (12 + (cfg->precision == TG_PREC_BF16 ? 2 : 4)) > ((size_t)192 << 20);
    m->s_adam = nullptr; m->s_fwd = nullptr;
    if (L.bands > 1) {
        int e = tg_stream_create(&m->s_adam) | tg_stream_create(&m->s_fwd) | tg_event_create(&m->e_fwd);
        for (int b = 0; b < L.bands; ++b) e |= tg_event_create(&m->e_bwd[b]) | tg_event_create(&m->e_adam[b]);
        if (e) { delete m; return tg_fail(TG_ERR_HIP, "could not create the pipeline streams / events"); }
    }
    auto bail = [&](int code) { delete m; return code; };

    if (tg_memset(m->ws, 0, L.total, m->stream)) return bail(tg_fail(TG_ERR_HIP, "memset(workspace) failed"));
    if (tg_memset(m->st, 0, L.s_total, m->stream)) return bail(tg_fail(TG_ERR_HIP, "memset(state) failed"));
    // M0 [C][V] dense -> M [C][Vp]   (mapping_optimizer.py:155-157)
    if (tg_memcpy2d(m->st + L.s_M, (size_t)L.Vp * 4, in->M0_dev, (size_t)L.V * 4, (size_t)L.V * 4, L.C, m->stream))
        return bail(tg_fail(TG_ERR_HIP, "copy of M0 failed"));
    if (cfg->has_density && tg_memcpy(m->ws + L.o_d, in->d_dev, (size_t)L.V * 4, m->stream))
        return bail(tg_fail(TG_ERR_HIP, "copy of d failed"));
    if (cfg->has_d_source && tg_memcpy(m->ws + L.o_densw, in->d_source_dev, (size_t)L.C * 4, m->stream))
        return bail(tg_fail(TG_ERR_HIP, "copy of d_source failed"));

    if (cfg->precision == TG_PREC_BF16X3 && m->cfg.s_exact_mode == 1) {      // bf16-exact S: two products per element instead of three
        int exact = 0;
        if ((rc = tg_s_is_exact(m, in, &exact))) return bail(rc);
        if (exact) { m->cfg.precision = TG_PREC_BF16X2S; m->L.prec = TG_PREC_BF16X2S; }
    }
    switch (cfg->precision) {
        case TG_PREC_F32: rc = tg_setup_operands<PrecF32>(m, in); break;
        case TG_PREC_BF16: rc = tg_setup_operands<PrecBF16>(m, in); break;
        case TG_PREC_BF16X2S: rc = tg_setup_operands<PrecBF16x2S>(m, in);
                              if (!rc) rc = L.T == 256 ? tg_lds_attr<PrecBF16x3, TgGeoLarge>() : tg_lds_attr<PrecBF16x3, TgGeoSmall>();   // (tg_mapper_project_genes)
                              break;
        default: rc = tg_setup_operands<PrecBF16x3>(m, in); break;
    }
    if (rc) return bail(rc);
    // padded G, |G_v|^2, |G_k|^2 partials (reuse genepart as scratch)
    // (genepart [nrb][2][Kp] doubles as scratch: first half |G|^2 partials, second half non-zero counts)
    TG_LAUNCH(tg_prep_g, L.nrb, 1, 256, 4 * TG_RB * 4, m->stream, in->G_dev, L.V, L.K, L.Vr, L.Kp, m->fp(L.o_Gp),
              m->fp(L.o_vnorm2), m->fp(L.o_genepart), m->fp(L.o_genepart) + (size_t)L.nrb * L.Kp);
    TG_LAUNCH(tg_colsum_parts, (L.Kp + 63) / 64, 1, 1024, 16 * 64 * 4, m->stream, (const float*)m->fp(L.o_genepart), L.nrb, L.Kp,
              m->fp(L.o_gnorm2), 1.f);
    TG_LAUNCH(tg_colsum_parts, (L.Kp + 63) / 64, 1, 1024, 16 * 64 * 4, m->stream,
              (const float*)(m->fp(L.o_genepart) + (size_t)L.nrb * L.Kp), L.nrb, L.Kp, m->fp(L.o_gfrac), 1.f / (float)L.V);
    if (cfg->has_density) TG_LAUNCH(tg_vec_sum, 1, 1, 1024, 64, m->stream, (const float*)m->fp(L.o_d), L.V, m->fp(L.o_gnorm2) + L.Kp);
    if ((L.has_nb || L.has_ct || L.has_ac) && (rc = tg_setup_spatial(m, in))) return bail(rc);
    if (!L.sp_shard) {                      // (a spot shard derives these at tg_mapper_attach_comm, from the gathered G)
        if ((L.has_nb || L.has_ct || L.has_ac) && (rc = tg_setup_spatial_derived(m))) return bail(rc);
        if (L.has_ac && (rc = tg_setup_autocorr(m))) return bail(rc);
    }
    // padding of the softmax statistics: shift = +3e38, scale = 0  => exp(M - shift) * scale == 0
    TG_LAUNCH(tg_fill, (L.Cp + 255) / 256, 1, 256, 0, m->stream, m->fp(L.o_rshift), (size_t)L.Cp, 3.0e38f);
    TG_LAUNCH(tg_fill, (L.Cp + 255) / 256, 1, 256, 0, m->stream, m->fp(L.o_rscale), (size_t)L.Cp, 3.0e38f);
    if ((rc = tg_launch_status())) return bail(rc);
    if (cfg->mode == TG_MODE_CONSTRAINED) {
        if (tg_memcpy(m->st + L.s_F, in->F0_dev, (size_t)L.C * 4, m->stream)) return bail(tg_fail(TG_ERR_HIP, "copy of F0 failed"));
        if ((rc = tg_launch_filter(m, false, 0.f, nullptr))) return bail(rc);
    }
    rc = tg_softmax_stats_from_scratch(m);
    if (!rc) rc = tg_merge(m, m->fp(L.o_rowpair), 1, /*finalize=*/true, /*want_pair=*/false);
    if (rc) return bail(rc);
    m->ready = true;
    *out = m;
    return TG_OK;
}

extern "C" void tg_mapper_destroy(tg_mapper* m) {
    if (!m) return;
    if (m->L.bands > 1) {
        tg_stream_destroy(m->s_adam); tg_stream_destroy(m->s_fwd); tg_event_destroy(m->e_fwd);
        for (int b = 0; b < m->L.bands; ++b) { tg_event_destroy(m->e_bwd[b]); tg_event_destroy(m->e_adam[b]); }
    }
    delete m;
}

// ---- one iteration ------------------------------------------------------------------------------
// cell rows / contraction steps of band b
static void tg_band_range(const TgLayout& L, int b, int* ct0, int* ct1, int* c0, int* c1) {
    *ct0 = (int)((long long)L.nct * b / L.bands);
    *ct1 = (int)((long long)L.nct * (b + 1) / L.bands);
    *c0 = *ct0 * L.T;
    *c1 = (*ct1 * L.T < L.C) ? *ct1 * L.T : L.C;
}

template <class PR>
static TgFwdArgs tg_fwd_args(tg_mapper* m, int band, const unsigned char* St_alt, bool unfiltered, int* grid_out) {
    const TgLayout& L = m->L;
    TgFwdArgs a;
    a.M = (const float*)(m->st + L.s_M);
    a.rmax = m->fp(L.o_rshift);
    a.rmul = unfiltered ? (const float*)m->fp(L.o_rinvz) : (const float*)m->fp(L.o_rmul);     // f_c / Z_c (padding rows: 0)
    a.rlse2 = unfiltered ? (const float*)m->fp(L.o_rowent) : (const float*)m->fp(L.o_rscale);
    a.St = St_alt ? St_alt : m->ws + L.o_St;
    a.Gpart = m->fp(L.o_Gpart);
    a.C = L.C; a.V = L.V; a.Vp = L.Vp; a.Vr = L.Vr; a.Kp = L.Kp; a.Cp = L.Cp;
    const int nvt = L.fwd_wide ? L.Vr / 128 : L.nvt, nkt = L.fwd_wide ? L.Kp / 512 : L.nkt;
    a.nkt = nkt; a.nvt = nvt; a.nsplit = L.nsplit; a.nsteps = L.Cp / PR::BKE;
    a.units = L.fwd_units;
    a.band_index = 0; a.band_step_begin = 0; a.band_step_end = 0;
    int grid = (L.fwd_units % nvt == 0) ? tg_fwd_grid(nvt, nkt, L.fwd_units / nvt) : tg_fwd_units_grid(L.fwd_units, nkt);
    if (band >= 0) {
        int ct0, ct1, c0, c1;
        tg_band_range(L, band, &ct0, &ct1, &c0, &c1);
        a.band_index = band;
        a.band_step_begin = c0 / PR::BKE;
        a.band_step_end = (band == L.bands - 1) ? a.nsteps : (ct1 * L.T) / PR::BKE;
        if (a.band_step_end > a.nsteps) a.band_step_end = a.nsteps;
        grid = tg_fwd_grid(nvt, nkt, 1);
    }
    *grid_out = grid;
    return a;
}

template <class PR>
static int tg_launch_forward(tg_mapper* m, tg_stream_t stream = nullptr, int band = -1,
                             const unsigned char* St_alt = nullptr, bool unfiltered = false) {
    const TgLayout& L = m->L;
    if (band < 0) stream = m->stream;
    int grid;
    const TgFwdArgs a = tg_fwd_args<PR>(m, band, St_alt, unfiltered, &grid);
    if (L.fwd_wide) {
        if constexpr (PR::NP == 2) TG_LAUNCH((tg_fwd_kernel<PR, TgGeoWide>), grid, 1, TgGeoWide::NT, TgGeoWide::LDS_BYTES, stream, a);
    } else if (L.T == 256) TG_LAUNCH((tg_fwd_kernel<PR, TgGeoLarge>), grid, 1, TgGeoLarge::NT, TgGeoLarge::LDS_BYTES, stream, a);
    else TG_LAUNCH((tg_fwd_kernel<PR, TgGeoSmall>), grid, 1, TgGeoSmall::NT, TgGeoSmall::LDS_BYTES, stream, a);
    if (band < 0) tg_prof_mark(m, "tg_fwd_kernel");
    return TG_OK;
}

static TgGhatReduceArgs tg_ghat_args(tg_mapper* m, bool force_vox) {
    const TgLayout& L = m->L;
    TgGhatReduceArgs a;
    a.Gpart = m->fp(L.o_Gpart); a.nsplit = L.nsplit; a.G = m->fp(L.o_Gp); a.Ghat = m->fp(L.o_Ghat);
    a.units = L.fwd_units; a.f_tm = L.fwd_wide ? 128 : L.T; a.f_nsteps = L.Cp / L.BKE;
    a.genepart = m->fp(L.o_genepart); a.voxstat = m->fp(L.o_voxstat);
    a.V = L.V; a.Vr = L.Vr; a.Kp = L.Kp; a.K = L.K; a.want_vox = force_vox || (m->cfg.lambda_g2 != 0.f);
    return a;
}

struct TgPeerLink;
static int tg_launch_ghat_stats(tg_mapper* m, bool force_vox = false, const TgPeerLink* link = nullptr) {
    const TgLayout& L = m->L;
    const TgGhatReduceArgs a = tg_ghat_args(m, force_vox);
    const int nrb = (L.V + TG_RB - 1) / TG_RB;
    TG_LAUNCH(tg_ghat_reduce, nrb, (L.Kp + TG_GH_COLS - 1) / TG_GH_COLS, 256, 4 * 64 * 2 * 16, m->stream, a);
    tg_prof_mark(m, "tg_ghat_reduce");
    if (link) {                   // spot shard, peer transport: the exchange of the statistics happens inside this kernel
        if (nrb > 512) TG_LAUNCH(tg_gene_reduce_tall_x, (L.Kp + 15) / 16, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_genepart), nrb, L.Kp, m->fp(L.o_genestat), *link);
        else TG_LAUNCH(tg_gene_reduce_x, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_genepart), nrb, L.Kp, m->fp(L.o_genestat), *link);
    } else if (nrb > 512) {       // many row blocks, e.g. clusters mode on 50 000 spots: 16 genes x 64 groups per workgroup
        TG_LAUNCH(tg_gene_reduce_tall, (L.Kp + 15) / 16, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_genepart), nrb, L.Kp,
                  m->fp(L.o_genestat));
    } else {
        TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_genepart), nrb, L.Kp,
                  m->fp(L.o_genestat));
    }
    tg_prof_mark(m, "tg_gene_reduce");
    return TG_OK;
}

// arguments of the loss / gradient-coefficient stage: the statistics -> coefficient map (f) and the dGhat emitter (e)
static bool tg_emit_self_ok(const tg_mapper* m);
static void tg_loss_args(tg_mapper* m, float* hist_row, TgFinalizeArgs& f, TgEmitArgs& e) {
    const TgLayout& L = m->L;
    f.genestat = m->fp(L.o_genestat); f.gnorm2 = m->fp(L.o_gnorm2); f.Ghat = m->fp(L.o_Ghat);
    f.voxstat = m->fp(L.o_voxstat); f.nky = (L.Kp + TG_GH_COLS - 1) / TG_GH_COLS; f.vnorm2 = m->fp(L.o_vnorm2); f.d = m->fp(L.o_d);
    f.coef = m->fp(L.o_coef); f.vcoef = m->fp(L.o_vcoef);
    f.hist = hist_row ? hist_row : m->fp(L.o_scal);
    f.lambda_g1 = m->cfg.lambda_g1; f.lambda_g2 = m->cfg.lambda_g2; f.lambda_d = m->cfg.lambda_d;
    f.rho_scale = m->cfg.has_d_source ? 1.f : 1.f / (float)L.C;
    f.fsum_dev = (m->cfg.mode == TG_MODE_CONSTRAINED) ? m->fp(L.o_fsum) : nullptr;
    f.K = L.K; f.Kp = L.Kp; f.V = L.V; f.Vr = L.Vr; f.V_total = L.Vtot; f.has_density = m->cfg.has_density;
    f.nbstat = L.has_nb ? m->fp(L.o_nbstat) : nullptr; f.wgnorm2 = L.has_nb ? m->fp(L.o_wgn2) : nullptr;
    f.nbcoef = L.has_nb ? m->fp(L.o_nbcoef) : nullptr;
    f.ctpart = L.has_ct ? m->fp(L.o_ctpart) : nullptr; f.n_ctpart = L.Vs; f.V_sp = L.Vs;
    f.lambda_nb = m->cfg.lambda_neighborhood_g1; f.lambda_ct = m->cfg.lambda_ct_islands; f.T = L.T_ct;
    f.part_out = m->comm ? m->fp(L.o_rowpair) + 2 * (size_t)L.C : nullptr;       // spot shard: this rank's parts of the spot sums
    f.spotpart = nullptr; f.n_spotpart = 0;
    if (tg_emit_self_ok(m)) { f.spotpart = m->fp(L.o_spotpart); f.n_spotpart = (L.V + TG_RB - 1) / TG_RB; }     // (tg_dghat_emit<SELF> leaves them)
    e.Ghat = m->fp(L.o_Ghat); e.G = m->fp(L.o_Gp); e.coef = m->fp(L.o_coef); e.vcoef = m->fp(L.o_vcoef);
    e.dG = m->ws + L.o_dG;
    // (spot shard: the extra gradient is evaluated for ALL spots; this handle's rows start at its spot offset)
    e.extra = (L.has_nb || L.has_ct || L.has_ac) ? m->fp(L.o_extra) + (L.sp_shard ? (size_t)m->cfg.spot_offset * L.Kp : 0) : nullptr;
    e.V = L.V; e.Vr = L.Vr; e.Kp = L.Kp; e.K = L.K; e.n_aug = 1 + L.T_ct;
    e.fin = f;
}
static bool tg_emit_self_ok(const tg_mapper* m) {      // the emit kernel can derive its coefficients itself (no spatial terms, LDS fits)
    const TgLayout& L = m->L;
    return !(L.has_nb || L.has_ct || L.has_ac) && L.bands == 1 && (L.Vtot == L.V || m->comm) && (size_t)(2 * L.Kp + 2 * TG_RB) * 4 <= 48 * 1024;
}

template <class PR>
static int tg_launch_loss(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    TgFinalizeArgs f;
    TgEmitArgs e;
    tg_loss_args(m, hist_row, f, e);
    // Without spatial terms every gradient coefficient is a local function of the reduced statistics: the emit kernel
    // derives them itself and the scalars of the history row are left to one extra workgroup of the update kernel.
    const bool self = tg_emit_self_ok(m);
    if (self) {
        m->fin_args = f; m->fin_pending = true;
        const int nrb = (L.V + TG_RB - 1) / TG_RB;
        int ncol = (512 + nrb - 1) / nrb;                // (column blocks: enough workgroups for two per CU on thin shapes; 1 from 512 spot blocks on)
        if (ncol > 8) ncol = 8;
        if (ncol > L.Kp / 128) ncol = L.Kp / 128 > 0 ? L.Kp / 128 : 1;
        TG_LAUNCH((tg_dghat_emit<PR, false, true>), nrb, ncol, 256, (2 * L.Kp + 2 * TG_RB) * 4, m->stream, e);
        tg_prof_mark(m, "tg_dghat_emit");
        return TG_OK;
    }
    m->fin_pending = false;
    int rcs = tg_launch_spatial_stats(m);
    if (rcs) return rcs;
    TG_LAUNCH(tg_loss_finalize, 1, 1, 1024, 16 * 5 * 4, m->stream, f);
    tg_prof_mark(m, "tg_loss_finalize");
    if ((rcs = tg_launch_spatial_grad(m))) return rcs;
    if (L.has_ac && (rcs = tg_launch_autocorr(m, hist_row))) return rcs;
    if (e.extra) TG_LAUNCH((tg_dghat_emit<PR, true, false>), (L.V + TG_RB - 1) / TG_RB, 1, 256, 0, m->stream, e);
    else TG_LAUNCH((tg_dghat_emit<PR, false, false>), (L.V + TG_RB - 1) / TG_RB, 1, 256, 0, m->stream, e);
    tg_prof_mark(m, "tg_dghat_emit");
    return TG_OK;
}

// backward GEMM (X, row-dot partials) over the cell tiles [ct0, ct1) on `stream`; `x_only`: no row dots (they are
// taken by tg_adam_rowpass)
template <class PR>
static TgBwdArgs tg_bwd_args(tg_mapper* m, int ct0, int ct1, int* grid_out, int tile = 0) {      // (ct0, ct1 in tiles of `tile` cells; 0 = L.T)
    const TgLayout& L = m->L;
    TgBwdArgs a;
    a.dG = m->ws + L.o_dG;
    a.Sk = m->ws + L.o_Sk;
    a.M = (const float*)(m->st + L.s_M); a.X = (void*)m->fp(L.o_X);
    a.rshift = m->fp(L.o_rshift); a.rinvz = m->fp(L.o_rinvz);
    a.fgate = (m->cfg.mode == TG_MODE_CONSTRAINED) ? m->fp(L.o_fgate) : nullptr;
    a.vcoef = m->fp(L.o_vcoef);
    a.dens_w = m->cfg.has_d_source ? m->fp(L.o_densw) : nullptr;
    a.part = m->fp(L.o_part);
    a.C = L.C; a.V = L.V; a.Vp = L.Vp; a.Vr = L.Vr; a.Kp = L.Kp; a.nsteps = L.Kp / PR::BKE;
    const int nct = ct1 - ct0, nvt = tile ? L.Vr / tile : L.nvt;
    a.ct_offset = ct0;
    // XCD bands along the longer tile axis when it is long enough to feed 8 XCDs, otherwise a plain linear order
    if (nct >= 16 && nct >= nvt) { a.map = TgTileMap{1, nct, nvt}; a.map_major_is_cells = 1; }
    else if (nvt >= 16) { a.map = TgTileMap{1, nvt, nct}; a.map_major_is_cells = 0; }
    else { a.map = TgTileMap{0, nvt, nct}; a.map_major_is_cells = 0; }
    a.lambda_r = m->cfg.lambda_r; a.lambda_l1 = m->cfg.lambda_l1; a.lambda_l2 = m->cfg.lambda_l2;
    *grid_out = tg_tilemap_grid(a.map);
    return a;
}

template <class PR>
static void tg_launch_bwd(tg_mapper* m, tg_stream_t stream, int ct0, int ct1, bool x_only = false, int tile = 0) {
    const TgLayout& L = m->L;
    int grid;
    // (the cached-access variant exists for the single-GPU X-only epilogue only: the row-dot variants serve spot shards and
    //  very long rows, i.e. big problems, and every GEMM instantiation costs seconds of compile time)
    // ct0, ct1 count tiles of L.T cells; under the 256 layout the kernel may run on 128^2 tiles (L.bwd_T, see tg_make_layout)
    if (tile == 0) tile = L.bwd_T;
    const int f = L.T / tile;
    const TgBwdArgs a = tg_bwd_args<PR>(m, f * ct0, f * ct1, &grid, tile);
#define TG_BWD_GO(GE, F, R, S) TG_LAUNCH((tg_bwd_kernel<PR, GE, F, R, S>), grid, 1, GE::NT, GE::BWD_LDS_BYTES, stream, a)
    if (tile == 256) {
        if (x_only) { if (m->stream_once) TG_BWD_GO(TgGeoLarge, false, false, true); else TG_BWD_GO(TgGeoLarge, false, false, false); }
        else if (L.full) TG_BWD_GO(TgGeoLarge, true, true, true);
        else TG_BWD_GO(TgGeoLarge, false, true, true);
    } else {
        if (x_only) { if (m->stream_once) TG_BWD_GO(TgGeoSmall, false, false, true); else TG_BWD_GO(TgGeoSmall, false, false, false); }
        else if (L.full) TG_BWD_GO(TgGeoSmall, true, true, true);
        else TG_BWD_GO(TgGeoSmall, false, true, true);
    }
#undef TG_BWD_GO
}

static int tg_polling_grid(tg_mapper* m, const void* fn, int nt, int lds, int want);      // (spot-sharded section below)
static bool tg_grid_strided(const tg_mapper* m);
static void tg_launch_rowsum(tg_mapper* m, tg_stream_t stream, int c0, int c1, const TgPeerLink* link = nullptr) {
    const TgLayout& L = m->L;
    TgRowsumArgs r;
    r.part = m->fp(L.o_part); r.nvt = L.Vr / L.bwd_T; r.C = L.C; r.rowq = m->fp(L.o_rowq); r.np = L.full ? TGP1_N : 1;
    r.c_begin = c0; r.c_end = c1;
    r.xch = 0; r.link.world = 0;
    int grid = (c1 - c0 + 15) / 16;
    if (link) { r.xch = 1; r.link = *link; if (tg_grid_strided(m)) grid = tg_polling_grid(m, (const void*)tg_rowsum_parts, 256, 2048, grid); }
    TG_LAUNCH(tg_rowsum_parts, grid, 1, 256, 2048, stream, r);
}

static void tg_launch_hist_regs(tg_mapper* m, tg_stream_t stream, float* hist_row) {
    const TgLayout& L = m->L;
    TgHistRegArgs h;
    h.rowq = m->fp(L.o_rowq); h.C = L.C; h.hist = hist_row ? hist_row : m->fp(L.o_scal);
    h.lambda_r = m->cfg.lambda_r; h.lambda_l1 = m->cfg.lambda_l1; h.lambda_l2 = m->cfg.lambda_l2;
    h.constrained = (m->cfg.mode == TG_MODE_CONSTRAINED);
    TG_LAUNCH(tg_hist_regs, 1, 1, 1024, 64, stream, h);
}

// arguments shared by the update kernels and the row-dot pass, for the cells [c0, c1)
static TgUpdateArgs tg_update_args(tg_mapper* m, float lr, bool finalize, int c0, int c1) {
    const TgLayout& L = m->L;
    TgUpdateArgs u;
    u.X = m->fp(L.o_X);
    u.M = (float*)(m->st + L.s_M); u.am = (float*)(m->st + L.s_m1); u.av = (float*)(m->st + L.s_m2);
    u.rshift = m->fp(L.o_rshift); u.rinvz = m->fp(L.o_rinvz);
    u.fgate = (m->cfg.mode == TG_MODE_CONSTRAINED) ? m->fp(L.o_fgate) : nullptr;
    u.dens_w = m->cfg.has_d_source ? m->fp(L.o_densw) : nullptr;
    u.vcoef = m->fp(L.o_vcoef); u.r = m->fp(L.o_rowq);
    u.pair_out = m->fp(L.o_rowpair); u.rowq_out = m->fp(L.o_rowq);
    u.new_shift = m->fp(L.o_rshift); u.new_invz = m->fp(L.o_rinvz); u.new_mul = m->fp(L.o_rmul); u.new_scale = m->fp(L.o_rscale);
    u.C = L.C; u.V = L.V; u.Vp = L.Vp; u.Vr = L.Vr; u.finalize = finalize ? 1 : 0; u.c_begin = c0; u.c_end = c1;
    u.lambda_r = m->cfg.lambda_r; u.lambda_l1 = m->cfg.lambda_l1; u.lambda_l2 = m->cfg.lambda_l2;
    const double t = (double)(m->step + 1);
    u.step_size = (float)((double)lr / (1.0 - pow((double)m->cfg.beta1, t)));
    u.bc2_sqrt = (float)sqrt(1.0 - pow((double)m->cfg.beta2, t));
    u.beta1 = m->cfg.beta1; u.beta2 = m->cfg.beta2; u.eps = m->cfg.eps;
    u.fin_on = 0;
    u.xch = 0; u.link.world = 0;
    return u;
}

// backward GEMM with the row-dot epilogue + the sum of its per-spot-tile partials (rows too long for tg_adam_rowpass; spot shards
// take the same two kernels in tg_one_step_sharded).  A separate row-dot pass over the stored X (one wave per row) was measured
// instead of the epilogue: 202 + 60 us against 247 + 7 us on a 1/8 spot shard of cfg2, i.e. no better (profiles/r02/run3).
template <class PR>
static int tg_launch_rowdots(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    tg_launch_bwd<PR>(m, m->stream, 0, L.nct);
    tg_prof_mark(m, "tg_bwd_kernel");
    tg_launch_rowsum(m, m->stream, 0, L.C);
    tg_prof_mark(m, "tg_rowsum_parts");
    if (L.full && !m->fin_pending) {          // (deferred history row: tg_one_step launches this after the update kernel)
        tg_launch_hist_regs(m, m->stream, hist_row);
        tg_prof_mark(m, "tg_hist_regs");
    }
    return TG_OK;
}

// streaming softmax-backward + Adam over the cells [c0, c1); `finalize`: write the next softmax statistics directly
// (single GPU, no filter)
template <bool FULL, bool X16, bool STREAM>
static void tg_launch_rowpass(const TgUpdateArgs& u, int rows, int V, tg_stream_t stream) {
#define TG_RP(NQ, NT, S) TG_LAUNCH((tg_adam_rowpass<FULL, X16, NQ, NT, S>), rows, 1, NT, 256, stream, u)
    if (V <= 4096) {                                  // 256 threads, up to 4 quads each
        const int nq = (V + 1023) / 1024;
        if (nq <= 1) TG_RP(1, 256, STREAM); else if (nq <= 2) TG_RP(2, 256, STREAM); else TG_RP(4, 256, STREAM);
    } else {                                          // 512 threads, up to 8 quads each (V <= TG_ROWPASS_MAX_V); streaming accesses
        const int nq = (V + 2047) / 2048;             // only: with rows this long the cached variant would serve a few hundred cells
        if (nq <= 3) TG_RP(3, 512, true); else if (nq <= 4) TG_RP(4, 512, true); else if (nq <= 5) TG_RP(5, 512, true);
        else if (nq <= 6) TG_RP(6, 512, true); else TG_RP(8, 512, true);
    }
#undef TG_RP
}

static int tg_launch_update(tg_mapper* m, float lr, bool finalize, tg_stream_t stream = nullptr, int c0 = 0, int c1 = -1,
                            bool rowpass = false, const TgPeerLink* link = nullptr) {
    const TgLayout& L = m->L;
    const bool whole = c1 < 0;
    if (whole) { stream = m->stream; c0 = 0; c1 = L.C; }
    TgUpdateArgs u = tg_update_args(m, lr, finalize, c0, c1);
    const bool x16 = (m->cfg.precision == TG_PREC_BF16) && !L.smallc;     // PrecBF16::X16 (the small-C kernels store X in fp32)
    u.fin_on = 0;
    if (link) { u.xch = 1; u.link = *link; }                              // (fused sharded step: the row pairs are pushed from the kernel's tail)
    int extra_wg = 0;
    if (m->fin_pending && whole) { u.fin = m->fin_args; u.fin_on = 1; extra_wg = 1; m->fin_pending = false; }
    if (rowpass) {
#define TG_RPGO(F, X) do { if (m->stream_once) tg_launch_rowpass<F, X, true>(u, c1 - c0 + extra_wg, L.V, stream); \
                           else tg_launch_rowpass<F, X, false>(u, c1 - c0 + extra_wg, L.V, stream); } while (0)
        if (L.full) { if (x16) TG_RPGO(true, true); else TG_RPGO(true, false); }
        else { if (x16) TG_RPGO(false, true); else TG_RPGO(false, false); }
#undef TG_RPGO
        if (whole) tg_prof_mark(m, "tg_adam_rowpass");
        return TG_OK;
    }
    const bool few = (c1 - c0) <= 64 && L.V > 4096;      // a handful of long rows: 1 024 threads per cell
#define TG_AUGO(F, X) do { if (few) TG_LAUNCH((tg_adam_update<F, X, true, 1024>), c1 - c0 + extra_wg, 1, 1024, 512, stream, u); \
                           else TG_LAUNCH((tg_adam_update<F, X, true>), c1 - c0 + extra_wg, 1, 256, 128, stream, u); } while (0)
    if (L.full) { if (x16) TG_AUGO(true, true); else TG_AUGO(true, false); }
    else { if (x16) TG_AUGO(false, true); else TG_AUGO(false, false); }
#undef TG_AUGO
    if (whole) tg_prof_mark(m, "tg_adam_update");
    return TG_OK;
}

// ---- small-C path (clusters mode) ---------------------------------------------------------------------------------
static TgSmallArgs tg_small_args(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    TgFinalizeArgs f;
    TgEmitArgs e;
    tg_loss_args(m, hist_row, f, e);
    f.nky = (L.Kp + TG_SC_KC - 1) / TG_SC_KC;     // tg_sc_forward: one block of per-spot statistics per gene chunk
    f.spotpart = m->fp(L.o_spotpart); f.n_spotpart = (L.V + TG_SC_SB - 1) / TG_SC_SB;
    TgSmallArgs a;
    a.M = (const float*)(m->st + L.s_M); a.rmax = m->fp(L.o_rshift); a.rmul = m->fp(L.o_rmul);
    a.Sa = m->fp(L.o_Sa); a.Sx = m->fp(L.o_Sx); a.G = m->fp(L.o_Gp); a.Ghat = m->fp(L.o_Ghat);
    a.genepart = m->fp(L.o_genepart); a.voxstat = m->fp(L.o_voxstat); a.X = m->fp(L.o_X);
    a.C = L.C; a.CM = tg_sc_cm(L.C); a.V = L.V; a.Vp = L.Vp; a.Vr = L.Vr; a.Kp = L.Kp; a.K = L.K; a.want_vox = (m->cfg.lambda_g2 != 0.f);
    a.fin = f;
    return a;
}
static int tg_small_nblk(const TgLayout& L) { return (L.V + TG_SC_SB - 1) / TG_SC_SB; }        // blocks of 64 spots = rows of genepart

// forward + statistics, backward of one iteration on the small-C kernels; the history scalars are left to the extra workgroup of the
// update kernel like on the GEMM path (fin_pending)
static int tg_launch_small(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    const TgSmallArgs a = tg_small_args(m, hist_row);
    const int nblk = tg_small_nblk(L), nch = (L.Kp + TG_SC_KC - 1) / TG_SC_KC;
#define TG_SC_FWD(CM, VX) TG_LAUNCH((tg_sc_forward<CM, VX>), nblk, nch, TG_SC_KC, tg_sc_lds_fwd(), m->stream, a)
    TG_SC_DISPATCH(L.C, a.want_vox, TG_SC_FWD);
#undef TG_SC_FWD
    tg_prof_mark(m, "tg_sc_forward");
    TG_LAUNCH(tg_gene_reduce, (L.Kp + 63) / 64, 1, 1024, TG_GR_GROUPS * 64 * 2 * 4, m->stream, (const float*)m->fp(L.o_genepart), nblk, L.Kp,
              m->fp(L.o_genestat));
    tg_prof_mark(m, "tg_gene_reduce");
    m->fin_args = a.fin; m->fin_pending = true;
#define TG_SC_BWD(CM, VX) TG_LAUNCH((tg_sc_backward<CM>), nblk, 1, TG_SC_KC, tg_sc_lds_bwd(), m->stream, a)
    TG_SC_DISPATCH(L.C, false, TG_SC_BWD);
#undef TG_SC_BWD
    tg_prof_mark(m, "tg_sc_backward");
    return TG_OK;
}

template <class PR>
static int tg_one_step(tg_mapper* m, float lr, float* hist_row) {
    int rc;
    if (m->L.smallc) {
        const bool constrained = (m->cfg.mode == TG_MODE_CONSTRAINED);
        if ((rc = tg_launch_small(m, hist_row))) return rc;
        if (tg_launch_failed()) return tg_launch_status();
        if ((rc = tg_launch_update(m, lr, !constrained, nullptr, 0, -1, true))) return rc;
        if (m->L.full) {
            tg_launch_hist_regs(m, m->stream, hist_row);
            tg_prof_mark(m, "tg_hist_regs");
        }
        if (constrained) {
            if ((rc = tg_launch_filter(m, true, lr, hist_row))) return rc;
            if ((rc = tg_merge(m, m->fp(m->L.o_rowpair), 1, true, false))) return rc;
        }
        m->step += 1;
        TG_LAUNCH_CK();
        return TG_OK;
    }
    if ((rc = tg_launch_forward<PR>(m))) return rc;
    if ((rc = tg_launch_ghat_stats(m))) return rc;
    if ((rc = tg_launch_loss<PR>(m, hist_row))) return rc;
    const bool constrained = (m->cfg.mode == TG_MODE_CONSTRAINED);
    if (m->L.V <= TG_ROWPASS_MAX_V) {
        // a row of M and X fits the registers of one workgroup: the backward GEMM only stores X, the row dots are fused
        // into the update (tg_adam_rowpass), which also leaves the regulariser row sums for tg_hist_regs / the filter
        tg_launch_bwd<PR>(m, m->stream, 0, m->L.nct, true);
        tg_prof_mark(m, "tg_bwd_kernel");
        if (tg_launch_failed()) return tg_launch_status();       // stop at the first failed launch, named
        if ((rc = tg_launch_update(m, lr, !constrained, nullptr, 0, -1, true))) return rc;
        if (m->L.full) {
            tg_launch_hist_regs(m, m->stream, hist_row);
            tg_prof_mark(m, "tg_hist_regs");
        }
    } else {
        const bool deferred = m->fin_pending;
        if ((rc = tg_launch_rowdots<PR>(m, hist_row))) return rc;
        if ((rc = tg_launch_update(m, lr, !constrained))) return rc;
        if (m->L.full && deferred) {
            tg_launch_hist_regs(m, m->stream, hist_row);
            tg_prof_mark(m, "tg_hist_regs");
        }
    }
    if (constrained) {      // Adam on F, then fold the NEW filter into the forward row scale
        if ((rc = tg_launch_filter(m, true, lr, hist_row))) return rc;
        if ((rc = tg_merge(m, m->fp(m->L.o_rowpair), 1, true, false))) return rc;
    }
    m->step += 1;
    TG_LAUNCH_CK();
    return TG_OK;
}

// One iteration as a software pipeline over bands of cells on three streams:
//   caller's stream : [forward of this step unless pre-launched] -> loss kernels -> backward GEMM(band 0), (band 1), ...
//   s_adam          : for each band, as soon as its backward is done: row sums -> streaming softmax-backward + Adam
//   s_fwd           : for each band, as soon as its rows are updated: forward GEMM partial of the NEXT step
// The HBM-bound update of band b runs beside the matrix-core-bound GEMMs of the neighbouring bands.  Every
// dependency is a HIP event; nothing is reordered that the sequential schedule orders (same arithmetic, same results).
template <class PR>
static int tg_one_step_pipelined(tg_mapper* m, float lr, float* hist_row, bool fwd_prelaunched, bool prelaunch_next) {
    const TgLayout& L = m->L;
    int rc;
    if (fwd_prelaunched) tg_stream_wait(m->stream, m->e_fwd);
    else if ((rc = tg_launch_forward<PR>(m))) return rc;
    if ((rc = tg_launch_ghat_stats(m))) return rc;
    if ((rc = tg_launch_loss<PR>(m, hist_row))) return rc;
    for (int b = 0; b < L.bands; ++b) {
        int ct0, ct1, c0, c1;
        tg_band_range(L, b, &ct0, &ct1, &c0, &c1);
        tg_launch_bwd<PR>(m, m->stream, ct0, ct1);
        tg_event_record(m->e_bwd[b], m->stream);
        tg_stream_wait(m->s_adam, m->e_bwd[b]);
        tg_launch_rowsum(m, m->s_adam, c0, c1);
        if ((rc = tg_launch_update(m, lr, true, m->s_adam, c0, c1))) return rc;
        tg_event_record(m->e_adam[b], m->s_adam);
        if (prelaunch_next) {
            tg_stream_wait(m->s_fwd, m->e_adam[b]);
            if ((rc = tg_launch_forward<PR>(m, m->s_fwd, b))) return rc;
        }
    }
    if (L.full) tg_launch_hist_regs(m, m->s_adam, hist_row);          // needs the row sums of every band
    tg_event_record(m->e_adam[L.bands - 1], m->s_adam);
    if (prelaunch_next) tg_event_record(m->e_fwd, m->s_fwd);
    tg_stream_wait(m->stream, m->e_adam[L.bands - 1]);               // join: the caller's stream sees the updated state
    m->step += 1;
    TG_LAUNCH_CK();
    return TG_OK;
}

// ---- batched independent mappings ------------------------------------------------------------------------------
// B handles of ONE shape and configuration (cross-validation folds, seeds: utils.py:576-600, mapping_parameter_tuning.py:109-131)
// advance together: one launch per kernel with blockIdx.z = mapping, the per-mapping kernel arguments in device arrays.
#define TG_BATCH_MAX_GROUPS 4
// groups a batch of n handles is stepped in (measured at 18 x 250 x 9 852, us per iteration of the whole batch, one group -> this rule:
// n = 2: 41.8 -> 37.9, 3: 45.3 -> 44.3, 4: 58.8 -> 50.7, 6: 70.1 -> 61.5, 8: 84.5 -> 69.4, 16: 148.6 -> 125; eight groups lose)
static int tg_batch_groups(int n) { return n <= 3 ? n : std::min((int)TG_BATCH_MAX_GROUPS, (n + 1) / 2); }
struct tg_batch {
    std::vector<tg_mapper*> h;
    unsigned char* dev;                              // caller-provided scratch: the argument arrays
    size_t o_fwd, o_ghat, o_gene, o_emit, o_bwd, o_upd, o_hreg, o_filt, o_merge, o_scr, o_small, total;
    std::vector<float*> hist;                        // history base pointers the argument arrays currently hold
    bool args_valid;
    // The argument arrays are assembled in page-locked host memory the batch owns and go to the device as ONE asynchronous copy:
    // tg_batch_step never synchronises the stream (round 3 built them in vectors on the stack and had to wait for the copies
    // before returning).  `e_upload` marks the last copy out of `stage`; the host waits for it only before REWRITING the staging
    // area, i.e. when a later call passes other history pointers while that copy is still queued.
    unsigned char* stage;
    tg_event_t e_upload;
    bool upload_pending;
    // A batch of two or more mappings is stepped as 2 - 4 groups, group 0 on the handles' stream and the others on streams the batch
    // owns (forked from / joined to the handles' stream inside every tg_batch_step call): the workgroups of one group's forward
    // kernel fill the gaps of another group's backward kernel (profiles/r03/run12_streams: + 18 - 22 % at 16 - 32 folds).
    int n_groups;
    tg_stream_t sub[TG_BATCH_MAX_GROUPS - 1];
    tg_event_t e_fork, e_join[TG_BATCH_MAX_GROUPS - 1];
};
static size_t tg_batch_layout(int n, tg_batch* b) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += rup(bytes, 256); return o; };
    const size_t o_fwd = take(n * sizeof(TgFwdArgs)), o_ghat = take(n * sizeof(TgGhatReduceArgs)), o_gene = take(n * sizeof(TgGeneReduceArgs)),
                 o_emit = take(n * sizeof(TgEmitArgs)), o_bwd = take(n * sizeof(TgBwdArgs)), o_upd = take(n * sizeof(TgUpdateArgs)),
                 o_hreg = take(n * sizeof(TgHistRegArgs)), o_filt = take(n * sizeof(TgFilterArgs)), o_merge = take(n * sizeof(TgMergeArgs)),
                 o_scr = take(n * sizeof(float*)), o_small = take(n * sizeof(TgSmallArgs));
    if (b) { b->o_small = o_small; b->o_fwd = o_fwd; b->o_ghat = o_ghat; b->o_gene = o_gene; b->o_emit = o_emit; b->o_bwd = o_bwd; b->o_upd = o_upd; b->o_hreg = o_hreg;
             b->o_filt = o_filt; b->o_merge = o_merge; b->o_scr = o_scr; b->total = off; }
    return off;
}
extern "C" size_t tg_batch_query_bytes(int n_mappers) { return n_mappers > 0 ? tg_batch_layout(n_mappers, nullptr) : 0; }

extern "C" int tg_batch_create(tg_mapper* const* mappers, int n, void* scratch_dev, tg_batch** out) {
    if (!mappers || n < 1 || !scratch_dev || !out) return tg_fail(TG_ERR_INVALID, "null argument or empty batch");
    const tg_mapper* m0 = mappers[0];
    for (int i = 0; i < n; ++i) {
        const tg_mapper* m = mappers[i];
        if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper %d of the batch is not ready", i);
        const TgLayout &A = m->L, &B = m0->L;
        if (A.C != B.C || A.K != B.K || A.V != B.V || A.prec != B.prec || A.T != B.T || A.nsplit != B.nsplit || A.fwd_units != B.fwd_units || A.full != B.full)
            return tg_fail(TG_ERR_INVALID, "mapper %d differs in shape / precision / terms from mapper 0 (a batch is B mappings of ONE shape)", i);
        if (m->stream != m0->stream) return tg_fail(TG_ERR_INVALID, "all mappers of a batch must be created on the same stream");
        if (m->step != m0->step) return tg_fail(TG_ERR_INVALID, "all mappers of a batch must be at the same step");
        if (m->cfg.mode != m0->cfg.mode) return tg_fail(TG_ERR_INVALID, "a batch holds handles of ONE class (Mapper or MapperConstrained)");
        if (m->comm || A.Vtot != A.V || A.bands > 1 || !tg_emit_self_ok(m) || A.V > TG_ROWPASS_MAX_V)
            return tg_fail(TG_ERR_UNSUPPORTED, "mapper %d uses spatial terms, spot shards, the band pipeline or rows longer than %d spots", i, TG_ROWPASS_MAX_V);
        if (m->cfg.beta1 != m0->cfg.beta1 || m->cfg.beta2 != m0->cfg.beta2) return tg_fail(TG_ERR_INVALID, "Adam betas differ inside the batch");
        for (int j = 0; j < i; ++j) if (mappers[j] == m) return tg_fail(TG_ERR_INVALID, "mapper %d appears twice in the batch", i);
    }
    tg_batch* b = new (std::nothrow) tg_batch();
    if (!b) return tg_fail(TG_ERR_INVALID, "out of host memory");
    b->h.assign(mappers, mappers + n);
    b->dev = (unsigned char*)scratch_dev;
    tg_batch_layout(n, b);
    b->args_valid = false;
    b->upload_pending = false;
    b->e_upload = tg_event_t();
    b->stage = (unsigned char*)tg_host_pinned_alloc(b->total);
    if (!b->stage || tg_event_create(&b->e_upload) != 0) { tg_batch_destroy(b); return tg_fail(TG_ERR_HIP, "could not allocate the batch's page-locked argument staging (%zu bytes)", b->total); }
    b->n_groups = tg_batch_groups(n);
    for (int g = 0; g + 1 < TG_BATCH_MAX_GROUPS; ++g) { b->sub[g] = nullptr; b->e_join[g] = tg_event_t(); }
    b->e_fork = tg_event_t();
    if (b->n_groups > 1) {
        bool ok = tg_event_create(&b->e_fork) == 0;
        for (int g = 0; ok && g + 1 < b->n_groups; ++g) ok = tg_stream_create(&b->sub[g]) == 0 && tg_event_create(&b->e_join[g]) == 0;
        if (!ok) { tg_batch_destroy(b); return tg_fail(TG_ERR_HIP, "could not create the streams of the batch's groups"); }
    }
    *out = b;
    return TG_OK;
}
extern "C" void tg_batch_destroy(tg_batch* b) {
    if (!b) return;
    if (b->n_groups > 1) {
        for (int g = 0; g + 1 < b->n_groups; ++g) { tg_stream_destroy(b->sub[g]); tg_event_destroy(b->e_join[g]); }
        tg_event_destroy(b->e_fork);
    }
    if (b->upload_pending) tg_event_host_wait(b->e_upload);
    tg_event_destroy(b->e_upload);
    tg_host_pinned_free(b->stage);
    delete b;
}

template <class PR>
static int tg_batch_upload(tg_batch* b, float* const* hist) {
    const int n = (int)b->h.size();
    tg_stream_t s = b->h[0]->stream;
    if (tg_stream_capturing(s))
        return tg_fail(TG_ERR_STATE, "tg_batch_step under stream capture must not change the batch's argument arrays: step the batch once "
                                     "with the same history pointers before capturing");
    if (b->upload_pending) { tg_event_host_wait(b->e_upload); b->upload_pending = false; }       // the staging area is about to be rewritten
    memset(b->stage, 0, b->total);
    TgFwdArgs* fw = (TgFwdArgs*)(b->stage + b->o_fwd); TgGhatReduceArgs* gh = (TgGhatReduceArgs*)(b->stage + b->o_ghat);
    TgGeneReduceArgs* gr = (TgGeneReduceArgs*)(b->stage + b->o_gene); TgEmitArgs* em = (TgEmitArgs*)(b->stage + b->o_emit);
    TgBwdArgs* bw = (TgBwdArgs*)(b->stage + b->o_bwd); TgUpdateArgs* up = (TgUpdateArgs*)(b->stage + b->o_upd);
    TgHistRegArgs* hr = (TgHistRegArgs*)(b->stage + b->o_hreg); TgFilterArgs* fl = (TgFilterArgs*)(b->stage + b->o_filt);
    TgMergeArgs* mg = (TgMergeArgs*)(b->stage + b->o_merge); float** scr = (float**)(b->stage + b->o_scr);
    TgSmallArgs* sm = (TgSmallArgs*)(b->stage + b->o_small);
    const bool constrained = (b->h[0]->cfg.mode == TG_MODE_CONSTRAINED);
    for (int i = 0; i < n; ++i) {
        tg_mapper* m = b->h[i];
        const TgLayout& L = m->L;
        int grid;
        fw[i] = tg_fwd_args<PR>(m, -1, nullptr, false, &grid);
        gh[i] = tg_ghat_args(m, false);
        gr[i] = TgGeneReduceArgs{m->fp(L.o_genepart), (L.V + TG_RB - 1) / TG_RB, L.Kp, m->fp(L.o_genestat)};
        TgFinalizeArgs f;
        tg_loss_args(m, nullptr, f, em[i]);
        f.hist = hist ? hist[i] : nullptr;                      // BASE of the mapping's history (row offset: TgStepVar)
        em[i].fin = f;
        bw[i] = tg_bwd_args<PR>(m, 0, L.nct, &grid);
        if (L.smallc) {                                             // clusters mode: tg_sc_forward / tg_sc_backward take the place of the GEMMs
            sm[i] = tg_small_args(m, nullptr);
            sm[i].fin.hist = f.hist;
            f = sm[i].fin;                                          // (one block of per-spot statistics: nky = 1)
            gr[i].nrb = tg_small_nblk(L);
        }
        up[i] = tg_update_args(m, 0.f, !constrained, 0, L.C);       // (MapperConstrained: tg_merge_stats folds the NEW filter in afterwards)
        up[i].fin = f; up[i].fin_on = 1;
        hr[i] = TgHistRegArgs{m->fp(L.o_rowq), L.C, hist ? hist[i] : nullptr, m->cfg.lambda_r, m->cfg.lambda_l1, m->cfg.lambda_l2, constrained ? 1 : 0};
        if (constrained) {
            fl[i] = tg_filter_args(m, true, 0.f, nullptr);
            fl[i].hist = hist ? hist[i] : nullptr;               // BASE of the history (row offset: TgStepVar), like the update kernel's
            mg[i] = tg_merge_args(m, m->fp(L.o_rowpair), 1, true, false, nullptr, 0);
        }
        scr[i] = m->fp(L.o_scal);
    }
    TG_CK(tg_memcpy_h2d(b->dev, b->stage, b->total, s));          // one copy, asynchronous (page-locked source that outlives the call)
    tg_event_record(b->e_upload, s);
    b->upload_pending = true;
    b->hist.assign(n, nullptr);
    for (int i = 0; i < n; ++i) b->hist[i] = hist ? hist[i] : nullptr;
    b->args_valid = true;
    return TG_OK;
}

template <bool FULL, bool X16>
static void tg_launch_rowpass_b(const TgUpdateArgs* argv, TgStepVar var, int rows, int V, int nb, tg_stream_t stream) {
#define TG_RPB(NQ, NT) TG_LAUNCH3((tg_adam_rowpass_b<FULL, X16, NQ, NT>), rows, 1, nb, NT, 256, stream, argv, var)
    if (V <= 4096) {
        const int nq = (V + 1023) / 1024;
        if (nq <= 1) TG_RPB(1, 256); else if (nq <= 2) TG_RPB(2, 256); else TG_RPB(4, 256);
    } else {
        const int nq = (V + 2047) / 2048;
        if (nq <= 3) TG_RPB(3, 512); else if (nq <= 4) TG_RPB(4, 512); else if (nq <= 5) TG_RPB(5, 512);
        else if (nq <= 6) TG_RPB(6, 512); else TG_RPB(8, 512);
    }
#undef TG_RPB
}

template <class PR>
static int tg_batch_step_impl(tg_batch* b, int n_steps, float lr, float* const* hist, int first_row) {
    const int n = (int)b->h.size();
    bool same = b->args_valid;
    for (int i = 0; same && i < n; ++i) same = (b->hist[i] == (hist ? hist[i] : nullptr));
    if (!same) { int rc = tg_batch_upload<PR>(b, hist); if (rc) return rc; }
    tg_mapper* m0 = b->h[0];
    const TgLayout& L = m0->L;
    tg_stream_t s = m0->stream;
    const int nrb = (L.V + TG_RB - 1) / TG_RB;
    const bool x16 = (m0->cfg.precision == TG_PREC_BF16) && !L.smallc;
    int gf, gb;
    (void)tg_fwd_args<PR>(m0, -1, nullptr, false, &gf);
    (void)tg_bwd_args<PR>(m0, 0, L.nct, &gb);
    const TgFwdArgs* a_fwd0 = (const TgFwdArgs*)(b->dev + b->o_fwd);
    const TgGhatReduceArgs* a_gh0 = (const TgGhatReduceArgs*)(b->dev + b->o_ghat);
    const TgGeneReduceArgs* a_gr0 = (const TgGeneReduceArgs*)(b->dev + b->o_gene);
    const TgEmitArgs* a_em0 = (const TgEmitArgs*)(b->dev + b->o_emit);
    const TgBwdArgs* a_bw0 = (const TgBwdArgs*)(b->dev + b->o_bwd);
    const TgUpdateArgs* a_up0 = (const TgUpdateArgs*)(b->dev + b->o_upd);
    const TgHistRegArgs* a_hr0 = (const TgHistRegArgs*)(b->dev + b->o_hreg);
    const TgSmallArgs* a_sm0 = (const TgSmallArgs*)(b->dev + b->o_small);
    const int nblk = tg_small_nblk(L), sc_nch = (L.Kp + TG_SC_KC - 1) / TG_SC_KC;
    const bool want_vox = (m0->cfg.lambda_g2 != 0.f);
    const int NG = b->n_groups, n_all = n;
    if (NG > 1) {                                    // fork: the groups' streams start behind everything already on the handles' stream
        tg_event_record(b->e_fork, m0->stream);
        for (int g = 0; g + 1 < NG; ++g) tg_stream_wait(b->sub[g], b->e_fork);
    }
    for (int it = 0; it < n_steps; ++it) {
      for (int grp = 0; grp < NG; ++grp) {
        const int z0 = (int)((long long)grp * n_all / NG), n = (int)((long long)(grp + 1) * n_all / NG) - z0;     // this group's mappings
        tg_stream_t s = grp == 0 ? m0->stream : b->sub[grp - 1];
        const TgFwdArgs* a_fwd = a_fwd0 + z0; const TgGhatReduceArgs* a_gh = a_gh0 + z0; const TgGeneReduceArgs* a_gr = a_gr0 + z0;
        const TgEmitArgs* a_em = a_em0 + z0; const TgBwdArgs* a_bw = a_bw0 + z0; const TgUpdateArgs* a_up = a_up0 + z0;
        const TgHistRegArgs* a_hr = a_hr0 + z0; const TgSmallArgs* a_sm = a_sm0 + z0;
        if (L.smallc) {
#define TG_SC_FWD_B(CM, VX) TG_LAUNCH3((tg_sc_forward_b<CM, VX>), nblk, sc_nch, n, TG_SC_KC, tg_sc_lds_fwd(), s, a_sm)
            TG_SC_DISPATCH(L.C, want_vox, TG_SC_FWD_B);
#undef TG_SC_FWD_B
            TG_LAUNCH3(tg_gene_reduce_b, (L.Kp + 63) / 64, 1, n, 1024, TG_GR_GROUPS * 64 * 2 * 4, s, a_gr);
#define TG_SC_BWD_B(CM, VX) TG_LAUNCH3((tg_sc_backward_b<CM>), nblk, 1, n, TG_SC_KC, tg_sc_lds_bwd(), s, a_sm)
            TG_SC_DISPATCH(L.C, false, TG_SC_BWD_B);
#undef TG_SC_BWD_B
        } else {
        if (L.fwd_wide) {
            if constexpr (PR::NP == 2) TG_LAUNCH3((tg_fwd_kernel_b<PR, TgGeoWide>), gf, 1, n, TgGeoWide::NT, TgGeoWide::LDS_BYTES, s, a_fwd);
        } else if (L.T == 256) TG_LAUNCH3((tg_fwd_kernel_b<PR, TgGeoLarge>), gf, 1, n, TgGeoLarge::NT, TgGeoLarge::LDS_BYTES, s, a_fwd);
        else TG_LAUNCH3((tg_fwd_kernel_b<PR, TgGeoSmall>), gf, 1, n, TgGeoSmall::NT, TgGeoSmall::LDS_BYTES, s, a_fwd);
        TG_LAUNCH3(tg_ghat_reduce_b, nrb, (L.Kp + TG_GH_COLS - 1) / TG_GH_COLS, n, 256, 4 * 64 * 2 * 16, s, a_gh);
        TG_LAUNCH3(tg_gene_reduce_b, (L.Kp + 63) / 64, 1, n, 1024, TG_GR_GROUPS * 64 * 2 * 4, s, a_gr);
        TG_LAUNCH3((tg_dghat_emit_b<PR>), nrb, 1, n, 256, (2 * L.Kp + 2 * TG_RB) * 4, s, a_em);
        if (L.T == 256) TG_LAUNCH3((tg_bwd_kernel_b<PR, TgGeoLarge>), gb, 1, n, TgGeoLarge::NT, TgGeoLarge::BWD_LDS_BYTES, s, a_bw);
        else TG_LAUNCH3((tg_bwd_kernel_b<PR, TgGeoSmall>), gb, 1, n, TgGeoSmall::NT, TgGeoSmall::BWD_LDS_BYTES, s, a_bw);
        }
        const double t = (double)(m0->step + 1);
        TgStepVar var;
        var.step_size = (float)((double)lr / (1.0 - pow((double)m0->cfg.beta1, t)));
        var.bc2_sqrt = (float)sqrt(1.0 - pow((double)m0->cfg.beta2, t));
        var.hist_row = hist ? (long long)(first_row + it) : -1;
        if (L.full) { if (x16) tg_launch_rowpass_b<true, true>(a_up, var, L.C + 1, L.V, n, s); else tg_launch_rowpass_b<true, false>(a_up, var, L.C + 1, L.V, n, s); }
        else { if (x16) tg_launch_rowpass_b<false, true>(a_up, var, L.C + 1, L.V, n, s); else tg_launch_rowpass_b<false, false>(a_up, var, L.C + 1, L.V, n, s); }
        if (L.full) TG_LAUNCH3(tg_hist_regs_b, 1, 1, n, 1024, 64, s, a_hr, var);
        if (m0->cfg.mode == TG_MODE_CONSTRAINED) {      // Adam on the filters, then the new filters folded into the forward row constants
            TG_LAUNCH3(tg_filter_kernel_b, 1, 1, n, 1024, 64, s, (const TgFilterArgs*)(b->dev + b->o_filt) + z0, var, (float* const*)(b->dev + b->o_scr) + z0);
            TG_LAUNCH3(tg_merge_stats_b, (L.C + 255) / 256, 1, n, 256, 0, s, (const TgMergeArgs*)(b->dev + b->o_merge) + z0);
        }
      }
        for (int i = 0; i < n_all; ++i) b->h[i]->step += 1;
        if (tg_launch_failed()) break;
    }
    if (NG > 1)                                      // join: the handles' stream continues behind every group (also after a failed launch)
        for (int g = 0; g + 1 < NG; ++g) { tg_event_record(b->e_join[g], b->sub[g]); tg_stream_wait(m0->stream, b->e_join[g]); }
    if (tg_launch_failed()) return tg_launch_status();
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_batch_step(tg_batch* b, int n_steps, float lr, float* const* history_dev, int first_row) {
    if (!b || b->h.empty()) return tg_fail(TG_ERR_STATE, "empty batch");
    if (n_steps < 0 || first_row < 0) return tg_fail(TG_ERR_INVALID, "n_steps < 0 or first_row < 0");
    for (size_t i = 0; i < b->h.size(); ++i) {
        if (b->h[i]->step != b->h[0]->step) return tg_fail(TG_ERR_STATE, "the mappers of the batch are at different steps (one was stepped on its own)");
        b->h[i]->fin_pending = false;
    }
    switch (b->h[0]->cfg.precision) {
        case TG_PREC_F32: return tg_batch_step_impl<PrecF32>(b, n_steps, lr, history_dev, first_row);
        case TG_PREC_BF16: return tg_batch_step_impl<PrecBF16>(b, n_steps, lr, history_dev, first_row);
        case TG_PREC_BF16X2S: return tg_batch_step_impl<PrecBF16x2S>(b, n_steps, lr, history_dev, first_row);
        default: return tg_batch_step_impl<PrecBF16x3>(b, n_steps, lr, history_dev, first_row);
    }
}

// ---- spot-sharded multi-GPU path --------------------------------------------------------------------------------------
#ifndef TG_SIM
#include <dlfcn.h>
#endif
struct tg_comm {
    int world, rank;
    tg_all_reduce_sum_fn ar; tg_all_gather_fn ag; void* ctx;       // callback transport
    // RCCL transport (librccl.so bound at run time: the library itself has no link-time dependency on it)
    void* lib; void* nccl;
    int (*p_allreduce)(const void*, void*, size_t, int, int, void*, void*);
    int (*p_allgather)(const void*, void*, size_t, int, void*, void*);
    int (*p_destroy)(void*);
    const char* (*p_errstr)(int);
    // peer-memory transport (tg_peer_exchange, tg_kernels.h): this rank's mailbox, every rank's mailbox as mapped here
    unsigned char* box; unsigned char* peer[TG_PEER_MAX];
    size_t cap, box_bytes; unsigned seq; int peer_mode, connected;                   // peer_mode 0: off; 1: hipIpc handles; 2: raw pointers
    size_t step_cap; unsigned step_seq; int colocated;                               // step area (tg_comm_peer_create_stepped): granules per (slot, rank), steps so far
    unsigned long long timeout_ticks;
    char shm_name[64];                                                               // (emulated build: the mailbox is a POSIX shm object)
};
struct tg_nccl_id { char internal[128]; };     // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
enum { TG_NCCL_FLOAT = 7, TG_NCCL_SUM = 0 };   // ncclFloat32, ncclSum (rccl.h)

extern "C" int tg_comm_create_callbacks(int world, int rank, tg_all_reduce_sum_fn ar, tg_all_gather_fn ag, void* ctx, tg_comm** out) {
    if (!out || !ar || !ag || world < 1 || rank < 0 || rank >= world) return tg_fail(TG_ERR_INVALID, "bad communicator arguments");
    tg_comm* c = new (std::nothrow) tg_comm();
    if (!c) return tg_fail(TG_ERR_INVALID, "out of host memory");
    c->world = world; c->rank = rank; c->ar = ar; c->ag = ag; c->ctx = ctx; c->lib = nullptr; c->nccl = nullptr;
    *out = c;
    return TG_OK;
}

// ---- peer-memory transport ---------------------------------------------------------------------------------------------------
#ifdef TG_SIM
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#endif
static unsigned long long tg_peer_timeout_ticks() {
    const char* e = getenv("TG_PEER_TIMEOUT_MS");
    const double ms = (e && *e) ? atof(e) : 20000.0;
    return (unsigned long long)((ms > 1.0 ? ms : 1.0) * 1e5);          // 10-ns ticks
}
extern "C" int tg_comm_peer_create_stepped(int world, int rank, size_t capacity_floats, size_t step_floats, int colocated, int same_process,
                                    void* handle64_out, tg_comm** out) {
    if (!out || !handle64_out || world < 1 || world > TG_PEER_MAX || rank < 0 || rank >= world || capacity_floats < 1 || colocated < 1)
        return tg_fail(TG_ERR_INVALID, "bad peer communicator arguments (1 <= world <= %d, colocated >= 1)", TG_PEER_MAX);
    tg_comm* c = new (std::nothrow) tg_comm();
    if (!c) return tg_fail(TG_ERR_INVALID, "out of host memory");
    memset(c, 0, sizeof *c);
    c->world = world; c->rank = rank;
    c->cap = rup(capacity_floats, TG_PEER_CHUNK);
    c->step_cap = rup(step_floats, 64); c->colocated = colocated;
    c->box_bytes = tg_peer_box_bytes(world, c->cap, c->step_cap);
    c->peer_mode = same_process ? 2 : 1;
    c->timeout_ticks = tg_peer_timeout_ticks();
    memset(handle64_out, 0, 64);
#ifdef TG_SIM
    if (same_process) {
        c->box = (unsigned char*)calloc(1, c->box_bytes);
        if (!c->box) { delete c; return tg_fail(TG_ERR_INVALID, "out of host memory"); }
    } else {
        static int counter = 0;
        snprintf(c->shm_name, sizeof c->shm_name, "/tg_peer_%d_%d", (int)getpid(), counter++);
        const int fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->box_bytes) != 0) { if (fd >= 0) close(fd); delete c; return tg_fail(TG_ERR_HIP, "shm_open / ftruncate failed for the emulated mailbox"); }
        c->box = (unsigned char*)mmap(nullptr, c->box_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (c->box == (unsigned char*)MAP_FAILED) { shm_unlink(c->shm_name); delete c; return tg_fail(TG_ERR_HIP, "mmap of the emulated mailbox failed"); }
        memset(c->box, 0, c->box_bytes);
        memcpy(handle64_out, c->shm_name, strlen(c->shm_name) + 1);
    }
#else
    // fine-grained device memory: peers write it and this GPU reads it while kernels are running (coarse-grained allocations are
    // only coherent at kernel boundaries).  The ONE device allocation of the library; it belongs to the communicator, not to a handle.
    void* p = nullptr;
    hipError_t e = hipExtMallocWithFlags(&p, c->box_bytes, hipDeviceMallocFinegrained);      // (no coarse-grained fallback: it would be incoherent)
    if (e != hipSuccess) { (void)hipGetLastError(); delete c; return tg_fail(TG_ERR_HIP, "cannot allocate the %zu-byte fine-grained mailbox (%s)", c->box_bytes, hipGetErrorString(e)); }
    c->box = (unsigned char*)p;
    if ((e = hipMemset(p, 0, c->box_bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
        (void)hipFree(p); delete c; return tg_fail(TG_ERR_HIP, "clearing the mailbox failed (%s)", hipGetErrorString(e));
    }
    if (!same_process) {
        hipIpcMemHandle_t h;
        static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle larger than the 64 bytes the ABI reserves");
        if ((e = hipIpcGetMemHandle(&h, p)) != hipSuccess) { (void)hipFree(p); delete c; return tg_fail(TG_ERR_HIP, "hipIpcGetMemHandle failed (%s)", hipGetErrorString(e)); }
        memcpy(handle64_out, &h, sizeof h);
    }
#endif
    if (same_process) memcpy(handle64_out, &c->box, sizeof c->box);
    *out = c;
    return TG_OK;
}
extern "C" int tg_comm_peer_create(int world, int rank, size_t capacity_floats, int same_process, void* handle64_out, tg_comm** out) {
    return tg_comm_peer_create_stepped(world, rank, capacity_floats, 0, 1, same_process, handle64_out, out);
}
// handles: world x 64 bytes, rank r's from tg_comm_peer_create (gathered over any out-of-band channel).  Collective in the sense that
// every rank must have created its mailbox before anybody connects.
extern "C" int tg_comm_peer_connect(tg_comm* c, const void* handles) {
    if (!c || !handles || !c->peer_mode) return tg_fail(TG_ERR_INVALID, "not a peer communicator");
    if (c->connected) return tg_fail(TG_ERR_STATE, "peer communicator already connected");
    for (int r = 0; r < c->world; ++r) {
        const unsigned char* h = (const unsigned char*)handles + 64 * (size_t)r;
        if (r == c->rank) { c->peer[r] = c->box; continue; }
        if (c->peer_mode == 2) { memcpy(&c->peer[r], h, sizeof c->peer[r]); continue; }
#ifdef TG_SIM
        char name[64]; memcpy(name, h, 64); name[63] = 0;
        const int fd = shm_open(name, O_RDWR, 0600);
        if (fd < 0) return tg_fail(TG_ERR_HIP, "cannot open the emulated mailbox %s of rank %d", name, r);
        void* p = mmap(nullptr, c->box_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) return tg_fail(TG_ERR_HIP, "mmap of rank %d's emulated mailbox failed", r);
        c->peer[r] = (unsigned char*)p;
#else
        hipIpcMemHandle_t ih; memcpy(&ih, h, sizeof ih);
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int q = 0; q < r; ++q) if (q != c->rank && c->peer[q]) { (void)hipIpcCloseMemHandle(c->peer[q]); c->peer[q] = nullptr; }   // (what was opened so far)
            return tg_fail(TG_ERR_HIP, "hipIpcOpenMemHandle for rank %d's mailbox failed (%s)", r, hipGetErrorString(e));
        }
        c->peer[r] = (unsigned char*)p;
#endif
    }
    c->connected = 1;
    return TG_OK;
}
// 0: every poll of every exchange so far met its peers; 1: a poll timed out (results since then are garbage).  Synchronises the device.
extern "C" int tg_comm_peer_status(tg_comm* c, int* timed_out) {
    if (!c || !timed_out || !c->peer_mode) return tg_fail(TG_ERR_INVALID, "not a peer communicator");
    unsigned w = 0;
#ifdef TG_SIM
    w = *(volatile unsigned*)c->box;
#else
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(&w, c->box, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return tg_fail(TG_ERR_HIP, "reading the mailbox status failed (%s)", hipGetErrorString(e));
#endif
    *timed_out = w ? 1 : 0;
    return TG_OK;
}

#ifndef TG_SIM
static void* tg_rccl_open(const char* path) {
    void* h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    return h;
}
#endif

extern "C" int tg_comm_rccl_unique_id(const char* librccl_path, void* id128_out) {
#ifdef TG_SIM
    (void)librccl_path; (void)id128_out;
    return tg_fail(TG_ERR_UNSUPPORTED, "RCCL is not available in the emulated build");
#else
    if (!id128_out) return tg_fail(TG_ERR_INVALID, "null id buffer");
    void* h = tg_rccl_open(librccl_path);
    if (!h) return tg_fail(TG_ERR_HIP, "cannot load librccl.so (%s)", dlerror());
    auto get_id = (int (*)(tg_nccl_id*))dlsym(h, "ncclGetUniqueId");
    if (!get_id) return tg_fail(TG_ERR_HIP, "librccl.so has no ncclGetUniqueId");
    const int e = get_id((tg_nccl_id*)id128_out);
    if (e) return tg_fail(TG_ERR_HIP, "ncclGetUniqueId failed with %d", e);
    return TG_OK;
#endif
}

extern "C" int tg_comm_create_rccl(const char* librccl_path, const void* id128, int world, int rank, tg_comm** out) {
#ifdef TG_SIM
    (void)librccl_path; (void)id128; (void)world; (void)rank; (void)out;
    return tg_fail(TG_ERR_UNSUPPORTED, "RCCL is not available in the emulated build");
#else
    if (!out || !id128 || world < 1 || rank < 0 || rank >= world) return tg_fail(TG_ERR_INVALID, "bad communicator arguments");
    void* h = tg_rccl_open(librccl_path);
    if (!h) return tg_fail(TG_ERR_HIP, "cannot load librccl.so (%s)", dlerror());
    tg_comm* c = new (std::nothrow) tg_comm();
    if (!c) return tg_fail(TG_ERR_INVALID, "out of host memory");
    c->world = world; c->rank = rank; c->ar = nullptr; c->ag = nullptr; c->ctx = nullptr; c->lib = h; c->nccl = nullptr;
    auto init = (int (*)(void**, int, tg_nccl_id, int))dlsym(h, "ncclCommInitRank");
    c->p_allreduce = (int (*)(const void*, void*, size_t, int, int, void*, void*))dlsym(h, "ncclAllReduce");
    c->p_allgather = (int (*)(const void*, void*, size_t, int, void*, void*))dlsym(h, "ncclAllGather");
    c->p_destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    c->p_errstr = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!init || !c->p_allreduce || !c->p_allgather || !c->p_destroy) { delete c; return tg_fail(TG_ERR_HIP, "librccl.so lacks a required entry point"); }
    tg_nccl_id id;
    memcpy(&id, id128, sizeof id);
    const int e = init(&c->nccl, world, id, rank);
    if (e) { const char* msg = c->p_errstr ? c->p_errstr(e) : "?"; delete c; return tg_fail(TG_ERR_HIP, "ncclCommInitRank failed with %d (%s)", e, msg); }
    *out = c;
    return TG_OK;
#endif
}

extern "C" void tg_comm_destroy(tg_comm* c) {
    if (!c) return;
#ifndef TG_SIM
    if (c->nccl && c->p_destroy) (void)c->p_destroy(c->nccl);
    if (c->peer_mode) {
        if (c->peer_mode == 1 && c->connected)
            for (int r = 0; r < c->world; ++r) if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
        if (c->box) (void)hipFree(c->box);
    }
#else
    if (c->peer_mode == 2) free(c->box);
    else if (c->peer_mode == 1) {
        if (c->connected) for (int r = 0; r < c->world; ++r) if (r != c->rank && c->peer[r]) munmap(c->peer[r], c->box_bytes);
        if (c->box) { munmap(c->box, c->box_bytes); shm_unlink(c->shm_name); }
    }
#endif
    delete c;
}

// one exchange of the peer transport: pieces of at most `cap` floats, one kernel each (tg_peer_exchange)
static int tg_peer_exchange_go(tg_comm* c, tg_stream_t stream, const float* send, float* recv, size_t n, int gather) {
    if (!c->connected) return tg_fail(TG_ERR_STATE, "peer communicator not connected (tg_comm_peer_connect)");
    if (tg_stream_capturing(stream))          // the sequence number of an exchange is a launch argument: a replay would find its granules already there
        return tg_fail(TG_ERR_UNSUPPORTED, "an exchange over the peer transport cannot be captured into a HIP graph");
    for (size_t off = 0; off < n; off += c->cap) {
        const size_t piece = (n - off < c->cap) ? n - off : c->cap;
        TgPeerArgs a;
        for (int r = 0; r < TG_PEER_MAX; ++r) a.box[r] = r < c->world ? c->peer[r] : nullptr;
        a.world = c->world; a.rank = c->rank; a.cap = c->cap;
        c->seq += 1;
        a.seq = c->seq; a.slot = (int)(c->seq & 1u);
        a.send = send + off; a.recv = recv + off; a.n = piece; a.gather = gather; a.ld = n; a.timeout_ticks = c->timeout_ticks;
        TG_LAUNCH(tg_peer_exchange, (piece + TG_PEER_CHUNK - 1) / TG_PEER_CHUNK, 1, 256, 0, stream, a);
    }
    return tg_launch_failed() ? tg_launch_status() : TG_OK;
}

// The communicator's two collectives as entry points of their own (what tg_mapper_step issues between its kernels): in-place
// all-reduce(sum) of n floats, all-gather of n floats per rank into recv[rank * n ...], enqueued on `hip_stream`.
extern "C" int tg_comm_all_reduce_sum(tg_comm* c, float* buf_dev, size_t n, void* hip_stream) {
    if (!c || !buf_dev) return tg_fail(TG_ERR_INVALID, "null communicator or buffer");
    int e = 0;
    if (c->peer_mode) { const int rc = tg_peer_exchange_go(c, (tg_stream_t)hip_stream, buf_dev, buf_dev, n, 0); if (rc) return rc; }
    else if (c->nccl) e = c->p_allreduce(buf_dev, buf_dev, n, TG_NCCL_FLOAT, TG_NCCL_SUM, c->nccl, hip_stream);
    else e = c->ar(c->ctx, buf_dev, n, hip_stream);
    if (e) return tg_fail(TG_ERR_HIP, "all-reduce of %zu floats failed with %d", n, e);
    return TG_OK;
}
extern "C" int tg_comm_all_gather(tg_comm* c, const float* send_dev, float* recv_dev, size_t n_per_rank, void* hip_stream) {
    if (!c || !send_dev || !recv_dev) return tg_fail(TG_ERR_INVALID, "null communicator or buffer");
    int e = 0;
    if (c->peer_mode) { const int rc = tg_peer_exchange_go(c, (tg_stream_t)hip_stream, send_dev, recv_dev, n_per_rank, 1); if (rc) return rc; }
    else if (c->nccl) e = c->p_allgather(send_dev, recv_dev, n_per_rank, TG_NCCL_FLOAT, c->nccl, hip_stream);
    else e = c->ag(c->ctx, send_dev, recv_dev, n_per_rank, hip_stream);
    if (e) return tg_fail(TG_ERR_HIP, "all-gather of %zu floats per rank failed with %d", n_per_rank, e);
    return TG_OK;
}
extern "C" int tg_comm_peer_set_timeout_ms(tg_comm* c, double ms) {
    if (!c || !c->peer_mode) return tg_fail(TG_ERR_INVALID, "not a peer communicator");
    c->timeout_ticks = (unsigned long long)((ms > 1.0 ? ms : 1.0) * 1e5);
    return TG_OK;
}

static int tg_exchange_all_reduce(tg_mapper* m, float* buf, size_t n) {
    const int rc = tg_comm_all_reduce_sum(m->comm, buf, n, (void*)m->stream);
    if (rc) return rc;
    tg_prof_mark(m, "exchange_all_reduce");
    return TG_OK;
}
static int tg_exchange_all_gather(tg_mapper* m, const float* send, float* recv, size_t n) {
    const int rc = tg_comm_all_gather(m->comm, send, recv, n, (void*)m->stream);
    if (rc) return rc;
    tg_prof_mark(m, "exchange_all_gather");
    return TG_OK;
}

// gather every rank's (max, sum exp) block, merge; `hist_row`: also turn this rank's history row into the global one
static int tg_exchange_row_stats(tg_mapper* m, float* hist_row) {
    const TgLayout& L = m->L;
    int rc = tg_exchange_all_gather(m, m->fp(L.o_rowpair), m->fp(L.o_gathered), L.pair_stride);
    if (rc) return rc;
    return tg_merge(m, m->fp(L.o_gathered), m->comm->world, /*finalize=*/true, /*want_pair=*/false, hist_row, m->comm->rank);
}

// ---- spot shards: kernels that exchange with the other ranks themselves (peer transport with a step area, round 6) ----------------
// Grid of a kernel whose workgroups WAIT for pushes of other ranks.  Deployment (one rank per device): callers launch their natural grid
// (one workgroup per row: the dispatcher starts them in order).  Ranks that SHARE a device (the one-GPU tests; tg_comm_peer_create_stepped's
// `colocated`): a waiting kernel must leave the device to the kernels of the ranks it waits for -- half of the CUs divided by the ranks,
// one workgroup each, all co-resident, walking their rows with a grid stride.
static int tg_polling_grid(tg_mapper* m, const void* fn, int nt, int lds, int want) {
    int g;
#ifdef TG_SIM
    (void)fn; (void)nt; (void)lds;
    g = 3;                                                   // (a few rows per workgroup: the grid-stride walk gets exercised)
#else
    static thread_local std::map<const void*, int> per_cu_of;
    static thread_local int cus = 0;
    if (!cus) { int dev = 0; (void)hipGetDevice(&dev); if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 64; }
    int& per_cu = per_cu_of[fn];
    if (!per_cu) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, nt, (size_t)lds) != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 1; } }
    const int col = m->comm ? m->comm->colocated : 1;
    g = col > 1 ? cus / (2 * col) : cus * per_cu;
#endif
    if (g < 1) g = 1;
    return g < want ? g : want;
}

static bool tg_grid_strided(const tg_mapper* m) {
#ifdef TG_SIM
    (void)m; return true;                                    // (the emulator always walks: the loop is what needs testing there)
#else
    return m->comm->colocated > 1;
#endif
}

static TgPeerLink tg_step_link(tg_mapper* m) {
    const tg_comm* c = m->comm; const TgLayout& L = m->L;
    TgPeerLink k;
    k.box = (unsigned char* const*)(m->ws + L.o_peertab);               // (the table tg_mapper_attach_comm left there)
    k.world = c->world; k.rank = c->rank;
    k.base = (unsigned long long)2 * c->world * c->cap; k.cap = c->step_cap; k.timeout_ticks = c->timeout_ticks;
    k.seq = c->step_seq; k.slot = (int)(c->step_seq & 1u);
    k.e2 = L.step_e2; k.e3 = L.step_e3; k.e1 = L.step_e1;
    return k;
}

// the statistics of the new rows on a fused step: every rank's pairs arrive as granules (pushed from the tails of the update kernels);
// tg_merge_stats_x polls for them at its head
static int tg_merge_fused(tg_mapper* m, float* hist_row, const TgPeerLink& link) {
    const TgLayout& L = m->L;
    const TgMergeArgs a = tg_merge_args(m, nullptr, m->comm->world, /*finalize=*/true, /*want_pair=*/false, hist_row, m->comm->rank);
    const int want = (L.C + 255) / 256;
    const int grid = tg_grid_strided(m) ? tg_polling_grid(m, (const void*)tg_merge_stats_x, 256, 0, want) : want;
    TG_LAUNCH(tg_merge_stats_x, grid, 1, 256, 0, m->stream, a, link);
    tg_prof_mark(m, "tg_merge_stats");
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_mapper_attach_comm(tg_mapper* m, tg_comm* comm) {
    if (!m || !m->ready || !comm) return tg_fail(TG_ERR_STATE, "mapper not ready or null communicator");
    const TgLayout& L = m->L;
    if (m->step != 0) return tg_fail(TG_ERR_STATE, "attach the communicator before the first step");
    if (comm->world > L.nranks) return tg_fail(TG_ERR_INVALID, "communicator of %d ranks but the handle was sized for n_ranks = %d", comm->world, L.nranks);
    if (!L.o_gathered) return tg_fail(TG_ERR_INVALID, "the handle was not created as a spot shard (n_ranks / n_spots_total)");
    if (L.bands > 1) return tg_fail(TG_ERR_UNSUPPORTED, "the cell-band pipeline is a single-GPU schedule");
    if (L.sp_shard && (comm->world != L.nranks || m->cfg.spot_offset != comm->rank * L.Vmaxl))
        return tg_fail(TG_ERR_INVALID, "spatial terms on spot shards: rank %d of %d must hold the spots from %d on (handle: %d ranks, offset %d)",
                       comm->rank, comm->world, comm->rank * L.Vmaxl, L.nranks, m->cfg.spot_offset);
    m->comm = comm;
    // Round 6: on the peer transport with a step area the three exchanges of a step happen inside its kernels (tg_one_step_sharded).
    // TG_PEER_FUSED=0 (environment) keeps the exchange kernels of round 5 (A / B measurements).
    {
        const char* e = getenv("TG_PEER_FUSED");
        m->fused = comm->peer_mode && comm->step_cap >= L.step_floats && L.step_floats > 0 && !L.sp_shard && !(e && *e == '0');
    }
    if (comm->peer_mode) {                                       // every rank's mailbox as mapped here: the kernels' table (TgPeerLink::box)
        static_assert(TG_PEER_MAX * sizeof(void*) <= 256, "the mailbox table has a 256-byte block");
        TG_CK(tg_memcpy_h2d(m->ws + L.o_peertab, comm->peer, sizeof(void*) * (size_t)comm->world, m->stream));
    }
    if (L.sp_shard) {
        // the spatial terms see the whole spot graph: gather the blocks of G once (the references W G, |W G_k|^2 and the autocorrelation
        // indicators of G are constants of the run), like Ghat every iteration
        int rcs = tg_exchange_all_gather(m, m->fp(L.o_Gp), m->fp(L.o_Gfull), (size_t)L.Vmaxl * L.Kp);
        if (rcs) return rcs;
        if ((rcs = tg_setup_spatial_derived(m))) return rcs;
        if (L.has_ac && (rcs = tg_setup_autocorr(m))) return rcs;
    }
    // set-up exchange: |G_k|^2 over all spots and the total of the density prior, then the softmax statistics of the initial logits
    int rc = tg_exchange_all_reduce(m, m->fp(L.o_gnorm2), (size_t)L.Kp + 1);
    if (rc) return rc;
    if ((rc = tg_softmax_stats_from_scratch(m))) return rc;
    return tg_exchange_row_stats(m, nullptr);
}

// One iteration on a spot shard: the kernels of tg_one_step on this rank's spots, with the three exchanges issued between them
// on the same stream (RCCL: no host code, no second stream, no event between a kernel and the collective that consumes its output).
template <class PR>
static int tg_one_step_sharded(tg_mapper* m, float lr, float* hist_row) {
    const TgLayout& L = m->L;
    int rc;
    // fused (peer transport with a step area, round 6): no exchange launches -- every exchange happens inside the small kernel that produces
    // or consumes its vector: E2 inside tg_gene_reduce, E3 inside tg_rowsum_parts, E1 pushed from the tail of the update kernel and polled at
    // the head of the merge: 8 launches per step instead of 11, every sum in the same order as on the other transports (bit-identical).
    // (Measured and NOT adopted, profiles/r06: E3 per row inside the update kernel -- at the head of the streaming kernel 164 + 9 + 7 -> 216 us,
    //  inside a register-resident row kernel that also drops the GEMM's row-dot epilogue 220 + 164 -> 198 + 376 us: a round trip through the
    //  fine-grained mailbox costs a row ~4 us, which no residency a 1 250-spot row allows can hide.)
    const bool fused = m->fused;
    TgPeerLink link;
    link.world = 0;
    if (fused) {
        if (tg_stream_capturing(m->stream))              // (the sequence number of a step is a launch argument)
            return tg_fail(TG_ERR_UNSUPPORTED, "a sharded step over the peer transport cannot be captured into a HIP graph");
        m->comm->step_seq += 1;
        link = tg_step_link(m);
    }
    if ((rc = tg_launch_forward<PR>(m))) return rc;
    if ((rc = tg_launch_ghat_stats(m, false, fused ? &link : nullptr))) return rc;
    if (!fused && (rc = tg_exchange_all_reduce(m, m->fp(L.o_genestat), (size_t)2 * L.Kp))) return rc;   // E2: per-gene cosine statistics
    if (L.sp_shard && (rc = tg_exchange_all_gather(m, m->fp(L.o_Ghat), m->fp(L.o_GhatFull), (size_t)L.Vmaxl * L.Kp))) return rc;   // spatial terms: all of Ghat
    if ((rc = tg_launch_loss<PR>(m, hist_row))) return rc;                                    // (coefficients; dGhat operand image)
    tg_launch_bwd<PR>(m, m->stream, 0, L.nct);                                                // X + row-dot partials of this rank's spots
    tg_prof_mark(m, "tg_bwd_kernel");
    tg_launch_rowsum(m, m->stream, 0, L.C, fused ? &link : nullptr);                          // fused: E3 inside
    tg_prof_mark(m, "tg_rowsum_parts");
    if (tg_launch_failed()) return tg_launch_status();
    if (!fused && (rc = tg_exchange_all_reduce(m, m->fp(L.o_rowq), (size_t)(L.full ? TGP1_N : 1) * L.C))) return rc;   // E3: row dots (+ regulariser row sums)
    if ((rc = tg_launch_update(m, lr, false, nullptr, 0, -1, false, fused ? &link : nullptr))) return rc;   // Adam; local (max, sum exp) [fused: pushed];
                                                                                                            // deferred history row by its extra workgroup
    if (L.full) {                                               // the row sums are global now: every rank adds the same scalars
        tg_launch_hist_regs(m, m->stream, hist_row);
        tg_prof_mark(m, "tg_hist_regs");
    }
    if (m->cfg.mode == TG_MODE_CONSTRAINED && (rc = tg_launch_filter(m, true, lr, hist_row))) return rc;   // replicated F: same result on every rank
    m->step += 1;
    if (fused) return tg_merge_fused(m, hist_row, link);        // E1 polled at the head of the merge
    return tg_exchange_row_stats(m, hist_row);                  // E1: statistics of the new rows (+ globalise the history row)
}

static int tg_dispatch_step(tg_mapper* m, float lr, float* hist_row, bool prelaunched = false, bool prelaunch_next = false) {
    if (m->comm) {
        switch (m->cfg.precision) {
            case TG_PREC_F32: return tg_one_step_sharded<PrecF32>(m, lr, hist_row);
            case TG_PREC_BF16: return tg_one_step_sharded<PrecBF16>(m, lr, hist_row);
            case TG_PREC_BF16X2S: return tg_one_step_sharded<PrecBF16x2S>(m, lr, hist_row);
            default: return tg_one_step_sharded<PrecBF16x3>(m, lr, hist_row);
        }
    }
    if (m->L.bands > 1 && !m->prof) {
        switch (m->cfg.precision) {
            case TG_PREC_F32: return tg_one_step_pipelined<PrecF32>(m, lr, hist_row, prelaunched, prelaunch_next);
            case TG_PREC_BF16: return tg_one_step_pipelined<PrecBF16>(m, lr, hist_row, prelaunched, prelaunch_next);
            case TG_PREC_BF16X2S: return tg_one_step_pipelined<PrecBF16x2S>(m, lr, hist_row, prelaunched, prelaunch_next);
            default: return tg_one_step_pipelined<PrecBF16x3>(m, lr, hist_row, prelaunched, prelaunch_next);
        }
    }
    switch (m->cfg.precision) {
        case TG_PREC_F32: return tg_one_step<PrecF32>(m, lr, hist_row);
        case TG_PREC_BF16: return tg_one_step<PrecBF16>(m, lr, hist_row);
        case TG_PREC_BF16X2S: return tg_one_step<PrecBF16x2S>(m, lr, hist_row);
        default: return tg_one_step<PrecBF16x3>(m, lr, hist_row);
    }
}

extern "C" int tg_mapper_step(tg_mapper* m, int n_steps, float lr, float* history_dev, int first_row) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (n_steps < 0) return tg_fail(TG_ERR_INVALID, "n_steps < 0");
    if (!m->comm && m->L.Vtot != m->L.V) return tg_fail(TG_ERR_STATE, "a spot shard needs tg_mapper_attach_comm before it can step");
    const bool pipelined = m->L.bands > 1 && !m->prof;
    bool prelaunched = false;
    for (int i = 0; i < n_steps; ++i) {
        float* row = history_dev ? history_dev + (size_t)(first_row + i) * TG_H_NTERMS : nullptr;
        const bool next = pipelined && (i + 1 < n_steps);        // the last step of a call leaves nothing in flight
        int rc = tg_dispatch_step(m, lr, row, prelaunched, next);
        if (rc) return rc;
        prelaunched = next;
    }
    return TG_OK;
}

// ---- results ------------------------------------------------------------------------------------
extern "C" int tg_mapper_result(tg_mapper* m, float* P_out_dev, float* F_out_dev) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (!P_out_dev) return tg_fail(TG_ERR_INVALID, "P_out is NULL");
    const TgLayout& L = m->L;
    TG_LAUNCH(tg_softmax_out, L.C, 1, 256, 0, m->stream, (const float*)(m->st + L.s_M),
              (const float*)m->fp(L.o_rshift), (const float*)m->fp(L.o_rinvz), L.C, L.V, L.Vp, P_out_dev);
    if (F_out_dev) {
        if (m->cfg.mode != TG_MODE_CONSTRAINED) return tg_fail(TG_ERR_INVALID, "F_out requested from an unconstrained mapper");
        TG_CK(tg_memcpy(F_out_dev, m->ws + L.o_fgate, (size_t)L.C * 4, m->stream));
    }
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_mapper_project(tg_mapper* m, float* Ghat_out_dev) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (!Ghat_out_dev) return tg_fail(TG_ERR_INVALID, "Ghat_out is NULL");
    const TgLayout& L = m->L;
    int rc;
    switch (m->cfg.precision) {
        case TG_PREC_F32: rc = tg_launch_forward<PrecF32>(m); break;
        case TG_PREC_BF16: rc = tg_launch_forward<PrecBF16>(m); break;
        case TG_PREC_BF16X2S: rc = tg_launch_forward<PrecBF16x2S>(m); break;
        default: rc = tg_launch_forward<PrecBF16x3>(m); break;
    }
    if (rc) return rc;
    if ((rc = tg_launch_ghat_stats(m))) return rc;
    TG_CK(tg_memcpy2d(Ghat_out_dev, (size_t)L.K * 4, m->ws + L.o_Ghat, (size_t)L.Kp * 4, (size_t)L.K * 4, L.V, m->stream));
    return TG_OK;
}

template <class PR>
static int tg_project_block(tg_mapper* m, const float* S_blk, long long ld_s, int kc, bool unfiltered) {
    const TgLayout& L = m->L;
    TgPrepSArgs a;
    a.S = S_blk; a.C = L.C; a.K = kc; a.ldS = ld_s; a.aug = nullptr; a.ct = nullptr; a.T = 0;
    a.Sk = nullptr; a.Cr = L.Cr; a.Kp = L.Kp;
    a.St = m->ws + L.o_StP; a.Cp = L.Cp;
    const size_t n2 = (size_t)L.Kp * (L.Cp / PR::CH);
    TG_LAUNCH((tg_prep_st<PR>), (n2 + 255) / 256, 1, 256, 0, m->stream, a);
    return tg_launch_forward<PR>(m, nullptr, -1, m->ws + L.o_StP, unfiltered);
}

extern "C" int tg_mapper_project_genes(tg_mapper* m, const float* S_dev, int64_t ld_s, int32_t n_genes, float* out_dev,
                                       int64_t ld_out, int32_t unfiltered) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (!S_dev || !out_dev) return tg_fail(TG_ERR_INVALID, "S or out is NULL");
    if (n_genes < 1 || ld_s < n_genes || ld_out < n_genes) return tg_fail(TG_ERR_INVALID, "n_genes < 1 or a row pitch smaller than n_genes");
    const TgLayout& L = m->L;
    // adata_map.X is softmax(M) without the filter (mapping_optimizer.py:637): row constants 1/Z and (max + ln Z) log2(e)
    const bool plain = unfiltered && m->cfg.mode == TG_MODE_CONSTRAINED;
    if (plain)
        TG_LAUNCH(tg_plain_rscale, (L.C + 255) / 256, 1, 256, 0, m->stream, (const float*)m->fp(L.o_rshift),
                  (const float*)m->fp(L.o_rinvz), L.C, m->fp(L.o_rowent));
    for (int k0 = 0; k0 < n_genes; k0 += L.K) {
        const int kc = (n_genes - k0 < L.K) ? n_genes - k0 : L.K;
        int rc;
        switch (m->cfg.precision) {
            case TG_PREC_F32: rc = tg_project_block<PrecF32>(m, S_dev + k0, ld_s, kc, plain); break;
            case TG_PREC_BF16: rc = tg_project_block<PrecBF16>(m, S_dev + k0, ld_s, kc, plain); break;
            default: rc = tg_project_block<PrecBF16x3>(m, S_dev + k0, ld_s, kc, plain); break;
        }
        if (rc) return rc;
        if ((rc = tg_launch_ghat_stats(m))) return rc;
        TG_CK(tg_memcpy2d(out_dev + k0, (size_t)ld_out * 4, m->ws + L.o_Ghat, (size_t)L.Kp * 4, (size_t)kc * 4, L.V, m->stream));
    }
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_csr_columns_to_dense(const int64_t* indptr_dev, const int32_t* indices_dev, const float* data_dev, int64_t n_rows,
                                       int32_t col0, int32_t n_cols, float* out_dev, int64_t ld_out, void* hip_stream) {
    if (!indptr_dev || !indices_dev || !data_dev || !out_dev) return tg_fail(TG_ERR_INVALID, "null argument");
    if (n_rows < 1 || n_cols < 1 || col0 < 0 || ld_out < n_cols) return tg_fail(TG_ERR_INVALID, "bad block: rows %lld, columns %d at %d, pitch %lld",
                                                                                 (long long)n_rows, n_cols, col0, (long long)ld_out);
    TG_LAUNCH(tg_csr_cols_to_dense, n_rows, 1, 256, 0, (tg_stream_t)hip_stream, (const long long*)indptr_dev, (const int*)indices_dev, data_dev,
              col0, n_cols, out_dev, (long long)ld_out);
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_csr_gather_columns(const int64_t* indptr_dev, const int32_t* indices_dev, const float* data_dev, int64_t n_rows,
                                     const int32_t* colmap_dev, int32_t n_out_cols, float* out_dev, int64_t ld_out, void* hip_stream) {
    if (!indptr_dev || !indices_dev || !data_dev || !colmap_dev || !out_dev) return tg_fail(TG_ERR_INVALID, "null argument");
    if (n_rows < 1 || n_out_cols < 1 || ld_out < n_out_cols) return tg_fail(TG_ERR_INVALID, "bad gather: rows %lld, columns %d, pitch %lld",
                                                                            (long long)n_rows, n_out_cols, (long long)ld_out);
    TG_LAUNCH(tg_csr_gather_cols, n_rows, 1, 256, 0, (tg_stream_t)hip_stream, (const long long*)indptr_dev, (const int*)indices_dev, data_dev,
              (const int*)colmap_dev, n_out_cols, out_dev, (long long)ld_out);
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_row_sums(const float* X_dev, int64_t ld, int32_t n_cols, const int64_t* indptr_dev, const float* data_dev, int64_t n_rows,
                           float* out_dev, int32_t normalize, void* hip_stream) {
    if (!out_dev || (!X_dev && !(indptr_dev && data_dev))) return tg_fail(TG_ERR_INVALID, "null argument");
    if (n_rows < 1 || (X_dev && (n_cols < 1 || ld < n_cols))) return tg_fail(TG_ERR_INVALID, "bad matrix shape");
    TG_LAUNCH(tg_row_sums, (n_rows + 3) / 4, 1, 256, 4 * 64 * 8, (tg_stream_t)hip_stream, X_dev, (long long)ld, n_cols,
              X_dev ? (const long long*)nullptr : (const long long*)indptr_dev, data_dev, (long long)n_rows, out_dev);
    if (normalize) TG_LAUNCH(tg_normalize_total, 1, 1, 1024, 1024 * 8, (tg_stream_t)hip_stream, out_dev, (long long)n_rows);
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_init_logits_normal(float* out_dev, int64_t n_rows, int64_t n_cols, int64_t ld, uint64_t seed, int64_t col0,
                                     int64_t n_cols_total, void* hip_stream) {
    if (!out_dev) return tg_fail(TG_ERR_INVALID, "null argument");
    if (n_rows < 1 || n_cols < 1 || ld < n_cols || col0 < 0 || n_cols_total < col0 + n_cols)
        return tg_fail(TG_ERR_INVALID, "bad block: %lld x %lld (ld %lld) at column %lld of %lld", (long long)n_rows, (long long)n_cols, (long long)ld,
                       (long long)col0, (long long)n_cols_total);
    const long long quads = n_rows * ((n_cols + 3) / 4);
    const int grid = (int)(quads / 256 + 1 < 16384 ? quads / 256 + 1 : 16384);
    TG_LAUNCH(tg_init_normal, grid, 1, 256, 0, (tg_stream_t)hip_stream, out_dev, (long long)n_rows, (long long)n_cols, (long long)ld,
              (unsigned long long)seed, (long long)col0, (long long)n_cols_total);
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_cluster_aggregate(const float* X_dev, int64_t ld, int32_t n_cols, const int32_t* member_indptr_dev,
                                    const int32_t* member_rows_dev, int32_t n_clusters, int32_t mean, float* out_dev, int64_t ld_out,
                                    void* hip_stream) {
    if (!X_dev || !member_indptr_dev || !member_rows_dev || !out_dev) return tg_fail(TG_ERR_INVALID, "null argument");
    if (n_clusters < 1 || n_cols < 1 || ld < n_cols || ld_out < n_cols) return tg_fail(TG_ERR_INVALID, "bad aggregation shape");
    TG_LAUNCH(tg_cluster_sums, n_clusters, (n_cols + 255) / 256, 256, 0, (tg_stream_t)hip_stream, X_dev, (long long)ld, n_cols,
              (const int*)member_indptr_dev, (const int*)member_rows_dev, mean, out_dev, (long long)ld_out);
    TG_LAUNCH_CK();
    return TG_OK;
}

// The precision a handle really computes in: tg_precision, or 3 = split bf16 with two products per element (S found bf16-exact at
// tg_mapper_create, tg_config.s_exact_mode).  Clusters-mode handles report TG_PREC_F32 (tg_make_layout).
extern "C" int tg_mapper_effective_precision(const tg_mapper* m) { return m ? m->cfg.precision : -1; }

extern "C" int tg_mapper_validate(tg_mapper* m, float* out4_dev) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (!out4_dev) return tg_fail(TG_ERR_INVALID, "out is NULL");
    const TgLayout& L = m->L;
    if (L.Vtot != L.V && !m->comm) return tg_fail(TG_ERR_STATE, "a spot shard needs tg_mapper_attach_comm before it can validate");
    if (m->cfg.mode != TG_MODE_MAPPER) return tg_fail(TG_ERR_INVALID, "MapperConstrained has no validation loss (mapping_optimizer.py:589)");
    int rc;
    switch (m->cfg.precision) {
        case TG_PREC_F32: rc = tg_launch_forward<PrecF32>(m); break;
        case TG_PREC_BF16: rc = tg_launch_forward<PrecBF16>(m); break;
        case TG_PREC_BF16X2S: rc = tg_launch_forward<PrecBF16x2S>(m); break;
        default: rc = tg_launch_forward<PrecBF16x3>(m); break;
    }
    if (rc) return rc;
    if ((rc = tg_launch_ghat_stats(m, true))) return rc;
    if (m->comm && (rc = tg_exchange_all_reduce(m, m->fp(L.o_genestat), (size_t)2 * L.Kp))) return rc;      // per-gene sums over all spots
    TG_LAUNCH(tg_row_entropy, L.C, 1, 256, 64, m->stream, (const float*)(m->st + L.s_M), (const float*)m->fp(L.o_rshift),
              (const float*)m->fp(L.o_rinvz), L.V, L.Vp, m->fp(L.o_rowent));
    TgValArgs a;
    a.genestat = m->fp(L.o_genestat); a.gnorm2 = m->fp(L.o_gnorm2); a.gfrac = m->fp(L.o_gfrac);
    a.voxstat = m->fp(L.o_voxstat); a.nky = (L.Kp + TG_GH_COLS - 1) / TG_GH_COLS; a.vnorm2 = m->fp(L.o_vnorm2); a.rowent = m->fp(L.o_rowent);
    a.out = out4_dev; a.K = L.K; a.Kp = L.Kp; a.V = L.V; a.Vr = L.Vr; a.C = L.C;
    a.V_total = L.Vtot; a.partial = 0; a.part = nullptr; a.gfrac_scale = (float)((double)L.V / (double)L.Vtot);
    if (m->comm) {
        // the sums over spots, per rank -> all-reduce -> every rank finishes with the same numbers.  Scratch: the gene coefficient
        // buffer [2][Kp] (rewritten by the loss kernels of the next step); Kp >= 128, so 64 + Kp floats fit.
        a.part = m->fp(L.o_coef); a.partial = 1;
        TG_LAUNCH(tg_val_finalize, 1, 1, 1024, 64, m->stream, a);
        if ((rc = tg_exchange_all_reduce(m, a.part, (size_t)64 + L.Kp))) return rc;
        a.partial = 0;
    }
    TG_LAUNCH(tg_val_finalize, 1, 1, 1024, 64, m->stream, a);
    TG_LAUNCH_CK();
    return TG_OK;
}

extern "C" int tg_mapper_state(tg_mapper* m, float** M_dev, float** m1_dev, float** m2_dev, int32_t* pitch, int64_t* step) {
    if (!m) return tg_fail(TG_ERR_INVALID, "null mapper");
    if (M_dev) *M_dev = (float*)(m->st + m->L.s_M);
    if (m1_dev) *m1_dev = (float*)(m->st + m->L.s_m1);
    if (m2_dev) *m2_dev = (float*)(m->st + m->L.s_m2);
    if (pitch) *pitch = m->L.Vp;
    if (step) *step = m->step;
    return TG_OK;
}

extern "C" int tg_mapper_filter_state(tg_mapper* m, float** F_rows_dev, int32_t* pitch) {
    if (!m) return tg_fail(TG_ERR_INVALID, "null mapper");
    if (m->cfg.mode != TG_MODE_CONSTRAINED) return tg_fail(TG_ERR_STATE, "only MapperConstrained has a filter");
    if (F_rows_dev) *F_rows_dev = (float*)(m->st + m->L.s_F);
    if (pitch) *pitch = m->L.Cp;
    return TG_OK;
}

extern "C" int tg_mapper_set_step(tg_mapper* m, int64_t step) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
    if (step < 0) return tg_fail(TG_ERR_INVALID, "step < 0");
    m->step = step;
    int rc = TG_OK;
    if (m->cfg.mode == TG_MODE_CONSTRAINED && (rc = tg_launch_filter(m, false, 0.f, nullptr))) return rc;
    rc = tg_softmax_stats_from_scratch(m);
    if (rc) return rc;
    if (m->comm) return tg_exchange_row_stats(m, nullptr);          // collective: every rank restores its shard and calls this
    return tg_merge(m, m->fp(m->L.o_rowpair), 1, true, false);
}

extern "C" int tg_mapper_profile(tg_mapper* m, int enable) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
#ifndef TG_SIM
    for (auto e : m->prof_events) hipEventDestroy(e);
    m->prof_events.clear();
#endif
    m->prof_names.clear();
    m->prof = enable != 0;
    if (m->prof) tg_prof_mark(m, "start");
    return TG_OK;
}

extern "C" int tg_mapper_profile_read(tg_mapper* m, char* names_out, size_t names_cap, float* total_ms_out,
                                      int* count_out, int n_max, int* n_out) {
    if (!m || !m->ready) return tg_fail(TG_ERR_STATE, "mapper not ready");
#ifndef TG_SIM
    TG_CK(hipStreamSynchronize(m->stream));
#endif
    std::vector<std::string> uniq;
    std::vector<double> tot;
    std::vector<int> cnt;
    for (size_t i = 1; i < m->prof_names.size(); ++i) {
        float ms = 0.f;
#ifndef TG_SIM
        hipEventElapsedTime(&ms, m->prof_events[i - 1], m->prof_events[i]);
#endif
        size_t j = 0;
        for (; j < uniq.size(); ++j) if (uniq[j] == m->prof_names[i]) break;
        if (j == uniq.size()) { uniq.push_back(m->prof_names[i]); tot.push_back(0.0); cnt.push_back(0); }
        tot[j] += ms;
        cnt[j] += 1;
    }
    std::string names;
    for (size_t j = 0; j < uniq.size(); ++j) {
        if (j) names += ";";
        names += uniq[j];
        if ((int)j < n_max) {
            if (total_ms_out) total_ms_out[j] = (float)tot[j];
            if (count_out) count_out[j] = cnt[j];
        }
    }
    if (names_out && names_cap) { strncpy(names_out, names.c_str(), names_cap - 1); names_out[names_cap - 1] = 0; }
    if (n_out) *n_out = (int)uniq.size();
    return tg_mapper_profile(m, 0);
}

// ---- test hooks (not part of the ABI): expose the workgroup -> tile maps so that their bijectivity can be checked on the host
extern "C" int tg_debug_tilemap(int mode, int n_major, int n_minor, int b, int* major, int* minor) {
    TgTileMap m{mode, n_major, n_minor};
    if (b < 0) return tg_tilemap_grid(m);
    return tg_tilemap(m, b, *major, *minor) ? 1 : 0;
}
extern "C" int tg_debug_fwd_map(int nvt, int nkt, int nsplit, int b, int* vt, int* kt, int* split) {
    if (b < 0) return tg_fwd_grid(nvt, nkt, nsplit);
    return tg_fwd_map(b, nvt, nkt, nsplit, *vt, *kt, *split) ? 1 : 0;
}
// the forward kernel's work decomposition for a configuration: out[0..3] = pieces per gene tile, gene tiles, spot tiles, contraction steps
extern "C" int tg_debug_fwd_decomposition(const tg_config* cfg, int* out) {
    TgLayout L;
    const int rc = tg_make_layout(cfg, &L);
    if (rc != TG_OK) return rc;
    out[0] = L.fwd_units; out[1] = L.fwd_wide ? L.Kp / 512 : L.nkt; out[2] = L.fwd_wide ? L.Vr / 128 : L.nvt; out[3] = L.Cp / L.BKE;
    return TG_OK;
}
// Host replay of the forward launch of a configuration (no device needed): every workgroup of the grid walks its segments through the
// kernel's own tg_fwd_walk; checked: each (spot tile, gene tile, step) is taken exactly once, a tile's partial slots are 0 .. nseg - 1
// (what tg_ghat_reduce sums), each written once, nseg <= the slots the layout reserves.  out[0..5] = grid, workgroups with work,
// fewest / most steps of a working workgroup, most segments of a workgroup, partial slots reserved.
extern "C" int tg_debug_fwd_cover(const tg_config* cfg, long long* out) {
    TgLayout L;
    const int rc = tg_make_layout(cfg, &L);
    if (rc != TG_OK) return rc;
    if (L.smallc || L.bands > 1) return tg_fail(TG_ERR_INVALID, "tg_debug_fwd_cover: this configuration does not launch the decomposed forward GEMM");
    const int nvt = L.fwd_wide ? L.Vr / 128 : L.nvt, nkt = L.fwd_wide ? L.Kp / 512 : L.nkt, nsteps = L.Cp / L.BKE, units = L.fwd_units;
    const int grid = (units % nvt == 0) ? tg_fwd_grid(nvt, nkt, units / nvt) : tg_fwd_units_grid(units, nkt);
    const long long G = (long long)nvt * nsteps;
    std::vector<unsigned char> taken((size_t)nvt * nkt * nsteps, 0), slot((size_t)nvt * nkt * L.nsplit, 0);
    long long working = 0, lo = -1, hi = 0, most_seg = 0;
    const char* bad = nullptr;
    for (int b = 0; b < grid; ++b) {
        long long steps = 0, segs = 0;
        tg_fwd_walk(b, nvt, nkt, nsteps, units, [&](int vt, int kt, int part_slot, int s_begin, int s_end) {
            if (vt < 0 || vt >= nvt || kt < 0 || kt >= nkt || s_begin < 0 || s_end > nsteps || s_begin >= s_end) { bad = "segment out of range"; return; }
            if (part_slot < 0 || part_slot >= L.nsplit || part_slot >= tg_fwd_nseg(vt, nsteps, G, units)) { bad = "partial slot out of range"; return; }
            if (slot[((size_t)vt * nkt + kt) * L.nsplit + part_slot]++) bad = "partial slot written twice";
            for (int st = s_begin; st < s_end; ++st) if (taken[((size_t)vt * nkt + kt) * nsteps + st]++) bad = "step taken twice";
            steps += s_end - s_begin; ++segs;
        });
        if (segs) { ++working; if (lo < 0 || steps < lo) lo = steps; if (steps > hi) hi = steps; if (segs > most_seg) most_seg = segs; }
    }
    for (size_t i = 0; i < taken.size() && !bad; ++i) if (taken[i] != 1) bad = "step not taken";
    for (int vt = 0; vt < nvt && !bad; ++vt)
        for (int kt = 0; kt < nkt; ++kt)
            for (int i = 0; i < L.nsplit; ++i)
                if ((slot[((size_t)vt * nkt + kt) * L.nsplit + i] != 0) != (i < tg_fwd_nseg(vt, nsteps, G, units))) bad = "partial slots are not 0 .. nseg - 1";
    if (bad) return tg_fail(TG_ERR_INVALID, "tg_debug_fwd_cover: %s (nvt %d nkt %d nsteps %d units %d)", bad, nvt, nkt, nsteps, units);
    out[0] = grid; out[1] = working; out[2] = lo; out[3] = hi; out[4] = most_seg; out[5] = L.nsplit;
    return TG_OK;
}
// TEST HOOK: Adam's square root and divisions as the update kernels evaluate them (tg_device.h: tg_sqrt_cr, tg_div_fr, tg_div_by):
// out[0..n) = sqrt(a), out[n..2n) = a / b, out[2n..3n) = a / bc (device arrays; enqueued on hip_stream)
extern "C" int tg_debug_adam_math(const float* a_dev, const float* b_dev, float bc, float* out_dev, long long n, void* hip_stream) {
    if (!a_dev || !b_dev || !out_dev || n < 1) return tg_fail(TG_ERR_INVALID, "null argument");
    TG_LAUNCH(tg_adam_math_probe, (n + 255) / 256, 1, 256, 0, (tg_stream_t)hip_stream, a_dev, b_dev, bc, out_dev, n);
    TG_LAUNCH_CK();
    return TG_OK;
}
// the launch geometry tg_make_layout derives from a configuration: out[0..7] = tile edge, cell tiles, spot tiles, gene tiles,
// forward splits, forward on 128 x 512 tiles (0/1), cell bands, clusters-mode kernels (0/1)
extern "C" int tg_debug_layout(const tg_config* cfg, int* out) {
    TgLayout L;
    const int rc = tg_make_layout(cfg, &L);
    if (rc != TG_OK) return rc;
    out[0] = L.T; out[1] = L.nct; out[2] = L.nvt; out[3] = L.nkt; out[4] = L.nsplit; out[5] = L.fwd_wide; out[6] = L.bands; out[7] = L.smallc;
    return TG_OK;
}
