/*
 * tangram_hip.h -- C ABI of the MI355X-native Tangram mapping optimizer (libtangram_hip.so).
 *
 * This is the drop-in boundary for ONE path of broadinstitute/Tangram: the training loop of
 * `tangram.mapping_optimizer.Mapper` / `MapperConstrained`
 * (reference: tangram/mapping_optimizer.py:14-408 and :411-639), reached through
 * `tg.map_cells_to_space()` (tangram/mapping_utils.py:141-428, operator seam at :355-363 and
 * :383-389).  The reference has no FFI of its own (it is pure Python on top of PyTorch); each entry
 * point below names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a
 * Tangram maintainer would add.
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is a device (HBM) pointer owned by the caller;
 *   - the library allocates NO device memory: the caller provides `state` (logits + Adam moments)
 *     and `workspace` buffers whose sizes come from tg_query_sizes();
 *   - all work is enqueued on the caller's hipStream_t (passed as void*); nothing synchronises
 *     the stream except tg_mapper_profile_read();
 *   - return value 0 = success, negative = tg_status; tg_last_error() gives the message of the last
 *     failure on the calling thread.  The library never aborts.
 *   - a handle is not thread-safe; different handles may be used from different threads.
 */
#ifndef TANGRAM_HIP_H
#define TANGRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_ABI_VERSION 6

typedef enum tg_status {
    TG_OK = 0,
    TG_ERR_INVALID = -1,     /* bad argument (maps to ValueError in the Python mirror)            */
    TG_ERR_HIP = -2,         /* a HIP runtime call failed                                         */
    TG_ERR_STATE = -3,       /* call sequence violated (e.g. step before setup)                   */
    TG_ERR_UNSUPPORTED = -4  /* feature of the reference not available in this build              */
} tg_status;

typedef enum tg_mode {
    TG_MODE_MAPPER = 0,      /* Mapper            (mapping_optimizer.py:14)  modes 'cells'/'clusters' */
    TG_MODE_CONSTRAINED = 1  /* MapperConstrained (mapping_optimizer.py:411) mode 'constrained'       */
} tg_mode;

typedef enum tg_precision {
    TG_PREC_F32 = 0,     /* exact fp32 matrix-core products (v_mfma_f32_16x16x4_f32)                 */
    TG_PREC_BF16 = 1,    /* bf16 operands, fp32 accumulate                                           */
    TG_PREC_BF16X3 = 2   /* split-bf16 (hi+lo) operands, 3 products, fp32 accumulate: fp32-parity    */
} tg_precision;

/* Hyper-parameters: Mapper.__init__ (mapping_optimizer.py:19-45) / MapperConstrained.__init__ (:417-432) */
typedef struct tg_config {
    int32_t abi_version;     /* TG_ABI_VERSION */
    int32_t mode;            /* tg_mode */
    int32_t precision;       /* tg_precision of the two GEMMs (problems with n_cells <= 32 on one GPU without spatial terms run on
                                exact-fp32 matrix-core kernels that recompute P^T S instead of storing it, whatever this says;
                                tile_size != 0 pins the GEMM kernels there too) */
    int32_t n_cells;         /* C: rows of S and M */
    int32_t n_genes;         /* K: training genes */
    int32_t n_spots;         /* V: spots held by THIS handle (a shard when spots are partitioned) */
    int32_t n_spots_total;   /* V over all shards (= n_spots on one GPU) */
    int32_t has_density;     /* d given (target_density_enabled, :114) */
    int32_t has_d_source;    /* d_source given (:118) */
    int32_t fwd_splits;      /* work decomposition of the forward GEMM.  0 = choose automatically; s > 0 = every output tile's cell range cut
                                into s equal ranges; u < 0 = the (spot tile, contraction step) space of every gene tile cut into -u equal pieces,
                                whatever the tile boundaries ("stream-K": what the automatic choice uses to fill whole rounds of the chip) */
    int32_t tile_size;       /* 0 = choose automatically; 128 or 256 = GEMM output tile edge (tuning / tests)        */
    int32_t pipeline_bands;  /* 0/1 = sequential schedule (default); 2..16 = cell bands of the opt-in 3-stream pipeline */
    float lambda_g1, lambda_d, lambda_g2, lambda_r, lambda_l1, lambda_l2;
    float lambda_count, lambda_f_reg, target_count;     /* constrained mode (:426-428, :480-483) */
    float lambda_neighborhood_g1;                       /* spatially weighted gene term (:33, :234-239); needs W, W^T */
    float lambda_ct_islands;                            /* cell-type islands (:40, :242-248); needs N, N^T, ct_encode   */
    int32_t n_cell_types;                               /* T: columns of ct_encode */
    int32_t nnz_w, nnz_n;                               /* non-zeros of the two spot graphs */
    float lambda_getis_ord, lambda_moran, lambda_geary; /* spatial autocorrelation terms (:35-37, :159-187, :251-263); need Ws, Ws^T */
    int32_t nnz_s;                                      /* non-zeros of spatial_weights */
    float beta1, beta2, eps;                            /* torch.optim.Adam defaults 0.9, 0.999, 1e-8 (:373) */
    int32_t n_ranks;         /* 0 = the handle holds every spot and steps alone; >= 1 = it is one spot shard (one per GPU) and will be
                                stepped through a communicator of that many ranks.  Sizes the gather buffer of the per-cell softmax
                                statistics ([n_ranks][2 C + 64] floats of workspace).                                               */
    int32_t bwd_tile;        /* 0 = the library's fixed rule; 128 or 256 = tile edge of the backward GEMM under the 256 layout (tuning /
                                tests).  The stored product X is bit-identical for both; the row-dot partials of a spot shard depend on
                                the tile width, which is why the rule is a function of the shape alone (never of a timing).            */
    int32_t spot_offset;     /* spot shard: index of this shard's first spot among the n_spots_total spots (0 on one GPU)            */
    int32_t s_exact_mode;    /* TG_PREC_BF16X3 only.  0 (default): the general three-product split-bf16 path.  1 (opt-in): tg_mapper_create
                                checks once whether every element of S (and of the augmentation columns that ride with it: d_source, the
                                cell-type encoding) is exactly representable in bf16 -- raw counts below 256 are -- and, if so, runs both
                                GEMMs with TWO matrix-core products per element instead of three: the lo part of S is identically zero,
                                the product a_hi * S_lo adds exact zeros and is skipped; the results are those of the three-product path
                                (bit for bit, up to the sign of an exact zero).  If S is not exact the general path runs.  The check
                                reads one flag back: create then synchronises the stream once.  (ABI version 4.)                   */
} tg_config;

typedef struct tg_sizes {
    size_t state_bytes;      /* logits M + Adam m + Adam v (+ filter F and its moments), fp32 */
    size_t workspace_bytes;  /* operand images of S, G copy, partial sums, coefficient vectors */
    int32_t m_pitch;         /* row pitch of M in floats (n_spots rounded up to 64) */
    int32_t history_terms;   /* floats per history row */
    size_t peer_step_floats; /* spot shards (n_ranks >= 1), ABI version 6: granules per (slot, rank) of the STEP AREA a peer communicator needs for
                                this handle to run its three per-step exchanges inside its kernels (tg_comm_peer_create_stepped); 0 = not applicable */
} tg_sizes;

/* Inputs of the constructor (mapping_optimizer.py:83-157): caller-owned device arrays, fp32 row-major. */
typedef struct tg_inputs {
    const float* S_dev;         /* [C][K]  single-cell matrix                     (:83)  */
    const float* G_dev;         /* [V][K]  spatial matrix (this shard's rows)     (:84)  */
    const float* d_dev;         /* [V]     density prior or NULL                  (:116) */
    const float* d_source_dev;  /* [C]     source density or NULL                 (:120) */
    const float* M0_dev;        /* [C][V]  initial logits, dense pitch V          (:150-157) */
    const float* F0_dev;        /* [C]     initial filter logits (constrained)    (:490) */
    const float* ct_encode_dev; /* [C][T]  one-hot cell types                     (:134-136) or NULL */
    /* spot graphs as CSR (int32 indptr [V+1], int32 indices [nnz], float data [nnz]) instead of the reference's dense
     * V x V matrices (spatial_weights.py:5-29): W = voxel_weights (:125-127) and its transpose,
     * N = neighborhood_filter (:130-132) and its transpose.  NULL when the term is off. */
    const int32_t* w_indptr;  const int32_t* w_indices;  const float* w_data;
    const int32_t* wt_indptr; const int32_t* wt_indices; const float* wt_data;
    const int32_t* n_indptr;  const int32_t* n_indices;  const float* n_data;
    const int32_t* nt_indptr; const int32_t* nt_indices; const float* nt_data;
    /* Ws = spatial_weights (:139-141) and its transpose, for the Getis-Ord / Moran / Geary terms */
    const int32_t* s_indptr;  const int32_t* s_indices;  const float* s_data;
    const int32_t* st_indptr; const int32_t* st_indices; const float* st_data;
} tg_inputs;

typedef struct tg_mapper tg_mapper;

/* indices into a history row (floats); unused terms are NaN like the reference's filtered terms (:301) */
enum { TG_H_TOTAL = 0, TG_H_MAIN = 1, TG_H_VG = 2, TG_H_KL = 3, TG_H_ENTROPY = 4, TG_H_L1 = 5, TG_H_L2 = 6,
       TG_H_NB = 7, TG_H_CT = 8, TG_H_COUNT = 9, TG_H_FREG = 10, TG_H_GETIS = 11, TG_H_MORAN = 12, TG_H_GEARY = 13,
       TG_H_NTERMS = 16 };

/* ---- communicator of the spot-sharded multi-GPU path (the reference has no distributed code) ----------
 * One process per GPU; rank g owns the spots V_g: M[:, V_g] + Adam moments, G[V_g], d[V_g]; S is replicated.  Per iteration exactly
 * three small vectors cross GPUs, all issued by the library itself on the handle's stream, between its own kernels:
 *   per-gene cosine statistics [2][Kp]            all-reduce(sum)   after the forward GEMM
 *   per-cell softmax-backward row dots [1|6][C]   all-reduce(sum)   after the backward GEMM
 *   per-cell (max, sum exp) of the new logits     all-gather        after the Adam update   (+ 2 history scalars per rank)
 * A tg_comm is RCCL (librccl.so is dlopen'ed: collectives run on the handle's stream, no host code between the phases of a
 * step), a pair of callbacks (any other transport: torch.distributed/gloo in the CPU tests, in-process shards in the GPU tests), or
 * peer memory (below: one-hop exchange kernels over mailboxes the ranks map from each other). */
typedef struct tg_comm tg_comm;
typedef int (*tg_all_reduce_sum_fn)(void* ctx, float* buf_dev, size_t n_floats, void* hip_stream);          /* in place; 0 = ok */
typedef int (*tg_all_gather_fn)(void* ctx, const float* send_dev, float* recv_dev, size_t n_floats_per_rank, void* hip_stream);
int tg_comm_create_callbacks(int world, int rank, tg_all_reduce_sum_fn all_reduce_sum, tg_all_gather_fn all_gather, void* ctx,
                             tg_comm** out);
/* RCCL: rank 0 calls tg_comm_rccl_unique_id and hands the 128 bytes to every rank (any out-of-band channel, e.g. a
 * torch.distributed broadcast); every rank then calls tg_comm_create_rccl (collective).  librccl_path: the librccl.so to bind
 * (the one PyTorch-ROCm ships is the natural choice); NULL = "librccl.so" by the loader's search path.                        */
int tg_comm_rccl_unique_id(const char* librccl_path, void* id128_out);
int tg_comm_create_rccl(const char* librccl_path, const void* id128, int world, int rank, tg_comm** out);
/* Peer memory (ABI version 5): the third transport.  Every rank owns a MAILBOX in its own HBM which every peer maps (hipIpc between the
 * processes of one node: xGMI stores; plain pointers between shards that live in one process); an exchange is ONE kernel per rank on
 * the handle's stream and ONE hop: store this rank's vector into every mailbox as 8-byte {value, sequence number} granules (one
 * write-through store each: no flag, no fence), poll the peers' granules in the own mailbox, sum in RANK ORDER (bit-identical on
 * every rank) or copy out.  No collective library, no host code, no extra stream; its latency is a kernel launch plus one xGMI
 * write, not a ring of 2 (N - 1) hops.  tg_comm_peer_create allocates the mailbox (fine-
 * grained device memory: the one allocation this library makes, owned by the communicator) and returns a 64-byte handle; gather the
 * handles of all ranks over any out-of-band channel (torch.distributed) and pass them, in rank order, to tg_comm_peer_connect.
 * capacity_floats: the longest vector moved in one piece (longer ones go in pieces); same_process != 0: the handle is a raw device
 * pointer (ranks living in one process; every rank needs a stream -- and a hardware queue -- of its own: a rank's kernel waits for
 * its peers' kernels).  One handle per peer communicator; sharded steps through it cannot be captured into a HIP graph (the sequence number of
 * an exchange is a launch argument).  A poll that does not meet its peers within TG_PEER_TIMEOUT_MS (environment, default 20 000)
 * gives up and sets a flag that tg_comm_peer_status reports: a lost peer costs a bounded wait, never a hang. */
int tg_comm_peer_create(int world, int rank, size_t capacity_floats, int same_process, void* handle64_out, tg_comm** out);
/* ABI version 6: the same with a STEP AREA behind the generic one.  step_floats = tg_sizes.peer_step_floats of the handle that will be
 * attached (0: none, = tg_comm_peer_create).  With it tg_mapper_step no longer launches an exchange kernel between its kernels: the
 * per-gene statistics are pushed and polled inside tg_gene_reduce; the update kernel sums the backward GEMM's row-dot partials of its
 * row itself, pushes them, polls the peers' and goes on (its first loads already in flight); the row pairs are pushed from the update
 * kernel's tail and polled at the head of tg_merge_stats: 7 launches per step instead of 11, every sum in the same order as on the other
 * transports (bit-identical results).  Runs with spatial terms keep the exchange kernels.  colocated: how many ranks of this communicator
 * run on THIS rank's device (1 in deployment; the one-GPU tests pass the world size: kernels that wait for their peers then leave room
 * for the peers' kernels). */
int tg_comm_peer_create_stepped(int world, int rank, size_t capacity_floats, size_t step_floats, int colocated, int same_process,
                         void* handle64_out, tg_comm** out);
int tg_comm_peer_connect(tg_comm* c, const void* handles_world_x_64);
int tg_comm_peer_status(tg_comm* c, int* timed_out);
int tg_comm_peer_set_timeout_ms(tg_comm* c, double ms);      /* bound of the polls of the exchanges issued from now on */
/* The two collectives a communicator of ANY transport runs between the kernels of a sharded step, as entry points of their own
 * (enqueued on hip_stream; a caller can verify a transport on its topology before it trusts it with a run -- tangram_amd/sharded.py
 * does so for the peer transport): in-place all-reduce(sum) of n floats; all-gather of n_per_rank floats into recv[rank * n_per_rank]. */
int tg_comm_all_reduce_sum(tg_comm* c, float* buf_dev, size_t n, void* hip_stream);
int tg_comm_all_gather(tg_comm* c, const float* send_dev, float* recv_dev, size_t n_per_rank, void* hip_stream);
void tg_comm_destroy(tg_comm* c);

int tg_abi_version(void);
const char* tg_last_error(void);

/* Sizes of the caller-provided buffers for a configuration. */
int tg_query_sizes(const tg_config* cfg, tg_sizes* out);

/* Replaces Mapper.__init__/MapperConstrained.__init__ (mapping_optimizer.py:19-157, :417-493):
 * builds operand images of S and G, copies M0 (and F0) into `state`, zeroes the Adam moments.
 * `S_dev`, `G_dev`, `d_dev`, `d_source_dev` are only read during this call.                       */
int tg_mapper_create(const tg_config* cfg, const tg_inputs* in, void* state_dev, void* workspace_dev,
                     void* hip_stream, tg_mapper** out);
void tg_mapper_destroy(tg_mapper* m);

/* Replaces the body of Mapper.train / MapperConstrained.train (mapping_optimizer.py:382-396, :621-634):
 * runs n_steps iterations (loss, backward, Adam) with learning rate lr; writes one history row per step
 * into history_dev[(first_row + i) * TG_H_NTERMS ...] (device memory, may be NULL).
 * Asynchronous on the handle's stream: nothing in it synchronises, times or queries the device, so a call can be captured into a
 * HIP graph from the first step on (the kernel selection is a fixed function of the configuration, tg_config.bwd_tile).  What a
 * captured call bakes into the graph: the step indices it covers (Adam's bias corrections 1 - beta^t are launch arguments
 * computed on the host) and the history rows it writes -- a replay re-runs exactly those steps, bit for bit, from whatever state
 * the handle's buffers hold (tests/test_gpu_graph_capture.py replays 50 captured steps from a restored state).  */
int tg_mapper_step(tg_mapper* m, int n_steps, float lr, float* history_dev, int first_row);

/* Spot-sharded multi-GPU run: attach a communicator to a handle created with n_spots < n_spots_total (collective; performs the
 * set-up exchanges: |G_k|^2 and sum(d) all-reduced, softmax statistics of the initial logits gathered).  Afterwards
 * tg_mapper_step runs the sharded schedule: same kernels on this rank's spots + the three exchanges above per iteration; the
 * history rows it writes are GLOBAL (identical on every rank).  The communicator must outlive the handle.                      */
int tg_mapper_attach_comm(tg_mapper* m, tg_comm* comm);

/* Replaces `softmax(self.M, dim=1).cpu().numpy()` (mapping_optimizer.py:407; :637-638 constrained):
 * P_out_dev [C][V] dense, F_out_dev [C] (sigmoid(F)) or NULL.                                       */
int tg_mapper_result(tg_mapper* m, float* P_out_dev, float* F_out_dev);

/* Replaces `adata_map.X.T @ S` (mapping_utils.py:402): Ghat_out_dev [V][K] = softmax(M)^T S (times f). */
int tg_mapper_project(tg_mapper* m, float* Ghat_out_dev);

/* Replaces `adata_map.X.T @ adata_sc.X` of project_genes (utils.py:366-368) and of the train-score epilogue over any
 * gene set (mapping_utils.py:402-410) without moving the C x V mapping to the host: out_dev[V][ld_out] (first n_genes
 * columns) = softmax(M)^T S_dev, S_dev [C][ld_s] holding n_genes expression columns (all genes of adata_sc, in blocks
 * of cfg.n_genes internally).  unfiltered != 0: in constrained mode use softmax(M) alone, as adata_map.X holds it
 * (mapping_optimizer.py:637); 0: times the filter f, like the training forward.  GEMM precision = cfg.precision.  */
int tg_mapper_project_genes(tg_mapper* m, const float* S_dev, int64_t ld_s, int32_t n_genes, float* out_dev,
                            int64_t ld_out, int32_t unfiltered);

/* Replaces `adata_sc.X.toarray()` on the host (utils.py:364-365; mapping_utils.py:259-266): the dense block of gene columns
 * [col0, col0 + n_cols) of a cells x genes CSR matrix (int64 indptr [n_rows + 1], int32 indices, float data, all on the device),
 * written to out_dev[n_rows][ld_out].  No handle: enqueued on `hip_stream`.  Feeds tg_mapper_project_genes block by block.  */
int tg_csr_columns_to_dense(const int64_t* indptr_dev, const int32_t* indices_dev, const float* data_dev, int64_t n_rows,
                            int32_t col0, int32_t n_cols, float* out_dev, int64_t ld_out, void* hip_stream);

/* ---- batched independent mappings ------------------------------------------------------------------------------
 * The reference trains independent mappings one after the other: one per held-out gene in `cross_val` (utils.py:576-600; 249 in
 * the tutorial), three seeds per trial in the tuner (mapping_parameter_tuning.py:109-131).  A tg_batch advances B Mapper handles
 * of ONE shape / configuration together: every kernel of the iteration is launched once with blockIdx.z = mapping (its arguments
 * come from arrays in `scratch_dev`, tg_batch_query_bytes(B) bytes of device memory).  The handles stay usable on their own
 * (result, project, state); they must share the stream they were created on and stay at the same step.  Results are the bits of
 * stepping each handle alone.  Handles of ONE class (all Mapper or all MapperConstrained), without spatial terms, rows <= 16 384 spots.
 * history_dev: host array of B device pointers (one history buffer per mapping) or NULL.
 * A batch steps its handles as 2 - 4 groups side by side: group 0 on the handles' stream, the others on
 * streams the batch creates (hipStreamNonBlocking; destroyed by tg_batch_destroy).  Every tg_batch_step forks them from the
 * handles' stream when it starts and joins them before it returns, so to the caller everything still happens on that stream.
 * tg_batch_step never synchronises: the per-mapping argument arrays are assembled in page-locked host memory the batch owns and
 * reach `scratch_dev` as one asynchronous copy, re-issued only when `history_dev` holds other pointers than in the previous call.
 * A call that does not change them is kernel launches + the fork / join events only and can be captured into a HIP graph (a call
 * that would have to re-upload under capture is refused with TG_ERR_STATE: step the batch once with the same history pointers
 * first).  As for tg_mapper_step, the Adam bias corrections and the history rows of the captured steps are launch arguments: a
 * replay repeats exactly the captured step indices (tests/test_gpu_graph_capture.py). */
typedef struct tg_batch tg_batch;
size_t tg_batch_query_bytes(int n_mappers);
int tg_batch_create(tg_mapper* const* mappers, int n_mappers, void* scratch_dev, tg_batch** out);
int tg_batch_step(tg_batch* b, int n_steps, float lr, float* const* history_dev, int first_row);
void tg_batch_destroy(tg_batch* b);

/* ---- host pre-processing on the device (no handle; enqueued on `hip_stream`) -------------------------------------------------
 * Replaces `adata[:, training_genes].X.toarray()` on the host (mapping_utils.py:259-275): the selected gene columns of a cells x genes
 * (or spots x genes) CSR matrix written straight into the dense S / G the mapper consumes.  colmap_dev[j] = destination column of
 * source column j, or -1; values are copied (bit-identical to the host gather).                                                  */
int tg_csr_gather_columns(const int64_t* indptr_dev, const int32_t* indices_dev, const float* data_dev, int64_t n_rows,
                          const int32_t* colmap_dev, int32_t n_out_cols, float* out_dev, int64_t ld_out, void* hip_stream);
/* Replaces `np.array(adata_sp.X.sum(axis=1)).squeeze()` and, with normalize != 0, `/ np.sum(rna_count_per_spot)` (pp_adatas,
 * mapping_utils.py:88-89): row sums of a dense (X_dev, ld, n_cols) or CSR (X_dev = NULL; indptr_dev, data_dev) matrix, accumulated
 * in double, rounded once; normalize: divided by their total (also double) -> the rna_count_based density prior.               */
int tg_row_sums(const float* X_dev, int64_t ld, int32_t n_cols, const int64_t* indptr_dev, const float* data_dev, int64_t n_rows,
                float* out_dev, int32_t normalize, void* hip_stream);
/* Initial logits generated on the device: out[r][c] (row pitch ld floats) = standard normal of (seed, r * n_cols_total + col0 + c),
 * r < n_rows, c < n_cols.  Opt-in replacement of `np.random.normal(0, 1, (n_cells, n_spots))` (mapping_optimizer.py:147-157) for
 * problems whose cells x spots plane must never exist on the host (BASELINE config 4: 40 GB of logits; `Mapper(..., init="device")`).
 * Counter-based (SplitMix64 finaliser + Box-Muller per element): a spot shard passes its first global spot as `col0` and
 * generates exactly the columns it owns -- any partition of the spots yields the same logits.  Not NumPy's stream: parity runs
 * use the reference's generator.  (ABI version 4.) */
int tg_init_logits_normal(float* out_dev, int64_t n_rows, int64_t n_cols, int64_t ld, uint64_t seed, int64_t col0,
                          int64_t n_cols_total, void* hip_stream);

/* Replaces the per-cluster `adata[mask].X.sum(axis=0)` / `.mean(axis=0)` loop of adata_to_cluster_expression (mapping_utils.py:126-132):
 * out_dev[c][k] = sum (mean != 0: mean) of X_dev[r][k] over the member rows r of cluster c (member_rows_dev[member_indptr_dev[c] ..
 * member_indptr_dev[c+1])), accumulated in double in member order.                                                              */
int tg_cluster_aggregate(const float* X_dev, int64_t ld, int32_t n_cols, const int32_t* member_indptr_dev, const int32_t* member_rows_dev,
                         int32_t n_clusters, int32_t mean, float* out_dev, int64_t ld_out, void* hip_stream);

/* Replaces Mapper._val_loss_fn (mapping_optimizer.py:311-356), evaluated with the CURRENT logits:
 * out4_dev = { gene score + voxel score, gene score, sparsity-weighted gene score, normalised map entropy }.
 * On a spot shard (attached communicator) the call is collective: the per-gene sums, the spot-cosine and entropy sums and the
 * non-zero fractions are all-reduced, every rank receives the same four numbers over ALL spots.                 */
int tg_mapper_validate(tg_mapper* m, float* out4_dev);

/* Checkpoint access (the reference's adata_map resume is a stub, mapping_optimizer.py:151-153):
 * raw pointers to M / Adam m / Adam v inside `state` and the step counter.                          */
int tg_mapper_state(tg_mapper* m, float** M_dev, float** m1_dev, float** m2_dev, int32_t* pitch, int64_t* step);
int tg_mapper_set_step(tg_mapper* m, int64_t step);      /* after restoring state: recomputes the softmax statistics */
/* The precision the handle computes in: a tg_precision value, or 3 = split bf16 with two matrix-core products per element (S was
 * found bf16-exact at create, tg_config.s_exact_mode); TG_PREC_F32 for clusters-mode handles (see tg_config.precision). */
int tg_mapper_effective_precision(const tg_mapper* m);
/* Constrained mode: the filter logits F and their two Adam moments, three rows of `pitch` floats (row 0 = F, mapping_optimizer.py:
 * 486-493; rows 1, 2 = exp_avg, exp_avg_sq of torch.optim.Adam).  A checkpoint = these + tg_mapper_state; restore into a fresh
 * handle, then tg_mapper_set_step.  The resumed run equals the uninterrupted one up to the rounding of the softmax normaliser
 * (set_step rebuilds the per-row statistics that the update kernel otherwise carries from step to step).              */
int tg_mapper_filter_state(tg_mapper* m, float** F_rows_dev, int32_t* pitch);

/* Timing hooks for bench.py: while enabled, a HIP event is recorded on the handle's stream after every
 * kernel launch of tg_mapper_step (no synchronisation).  tg_mapper_profile_read synchronises the stream once,
 * aggregates the event intervals per kernel name (';'-separated list, total milliseconds, launch counts)
 * and disables recording.                                                                              */
int tg_mapper_profile(tg_mapper* m, int enable);
int tg_mapper_profile_read(tg_mapper* m, char* names_out, size_t names_cap, float* total_ms_out, int* count_out,
                           int n_max, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* TANGRAM_HIP_H */
