/*
 * lgcn_hip.h -- C ABI of the MI355X-native LightGCN/BPR training hot path.
 *
 * One shared library (liblgcn_hip.so, built from
 * graph-and-sequential-recommendation-systems_amd/csrc by hipcc for gfx950).
 * Plain C types only: device buffers are raw pointers (e.g. torch
 * `tensor.data_ptr()`), streams are `hipStream_t` passed as void*.  Every
 * device entry point is asynchronous on the given stream, never allocates,
 * never synchronises, and is therefore hipGraph-capturable.  Return value:
 * 0 = ok, non-zero = error (text via lgcn_last_error()).
 *
 * Each entry point names the reference interface it replaces; paths are
 * relative to LightGCN_work/code in saamiya225/Graph-and-sequential-recommendation-systems.
 * The reference's only FFI on this path is the pybind11 module `sampling`
 * (sources/sampling.cpp:95-106); everything else replaces PyTorch library
 * calls made from model.py / utils.py.  INTEGRATION.md shows the bindings.
 */
#ifndef LGCN_HIP_H
#define LGCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LGCN_ABI_VERSION 11
#define LGCN_MAX_LAYERS 8

/* storage type of propagated activations (accumulation is always fp32) */
enum { LGCN_F32 = 0, LGCN_BF16 = 1, LGCN_FP8 = 2 };
/* LGCN_FP8: a table of n rows is n*d bytes of OCP E4M3 values FOLLOWED BY n fp32 row scales (value = scale * fp8), one pointer
 * for both; scales are powers of two with max|row| / scale in [64, 128).  d must be 64, 128 or 256.  The bytes of a row are
 * CHUNK-INTERLEAVED (an opaque layout: lgcn_to_fp8 writes it, the kernels read it): with L = d/16, byte l*16 + 4j + e holds
 * column (j*L + l)*4 + e, so that a lane gathering 16 bytes owns four float4 chunks L chunks apart and the fp32 side of every
 * epilogue is coalesced.  Accumulation is fp32 as
 * with every storage type.  lgcn_table_bytes: bytes of one [n_rows, d] table of a storage type (fp8: rows + scales, padded
 * to 256); lgcn_to_fp8: quantise an fp32 table (device pointers).                                                     */
int64_t lgcn_table_bytes(int64_t n_rows, int32_t d, int32_t dtype);
int lgcn_to_fp8(const float *src, void *dst, int64_t n_rows, int32_t d, void *stream);

int lgcn_abi_version(void);
const char *lgcn_last_error(void);
/* 1 if a HIP device is usable from this process, else 0 (never throws) */
int lgcn_device_available(void);

/* ------------------------------------------------------------------------ */
/* Host: BPR triplet sampler -- replaces the pybind11 module `sampling`      */
/* ------------------------------------------------------------------------ */
/* sampling.seed(seed)            sources/sampling.cpp:88-91,101  (srand)     */
void lgcn_sampling_seed(unsigned int seed);
/* sampling.randint(end)          sources/sampling.cpp:22-25,100  (rand()%end) */
int lgcn_sampling_randint(int end);
/* sampling.sample_negative(user_num,item_num,train_num,allPos,neg_num)
 *                                sources/sampling.cpp:27-56,102-103
 * allPos is passed zero-copy as the CSR (indptr[user_num+1], indices) of
 * UserItemNet (dataloader.py:133-136,178-180; columns sorted ascending).
 * S_out: int32 [user_num*(train_num/user_num), 2+neg_num], row-major.
 * Bit-exact glibc rand() stream.  Returns 2 (no crash) if a user has no
 * positives, where the reference dies with SIGFPE on rand()%0. */
int lgcn_sample_negative(int user_num, int item_num, int64_t train_num,
                         const int64_t *indptr, const int32_t *indices,
                         int neg_num, int32_t *S_out);
/* sampling.sample_negative_ByUser(users,item_num,allPos,neg_num)
 *                                sources/sampling.cpp:58-86,104-105          */
int lgcn_sample_negative_by_user(const int32_t *users, int n_users_listed, int item_num,
                                 const int64_t *indptr, const int32_t *indices,
                                 int neg_num, int32_t *S_out);

/* The same sampler on the GPU (neg_num = 1): identical rows from the identical rand() stream, written
 * to DEVICE memory (d_S int32 [user_num*(train_num/user_num), 3]); the host generator is then moved
 * past the draws the device consumed (jump-ahead), so host and device calls can be mixed freely.
 * h_indptr: host copy of the CSR row pointers (degree checks); d_indptr/d_indices: device CSR;
 * workspace: device bytes from lgcn_sample_negative_device_workspace().  Synchronises `stream`. */
int64_t lgcn_sample_negative_device_workspace(int user_num, int64_t train_num);
int lgcn_sample_negative_device(int user_num, int item_num, int64_t train_num,
                                const int64_t *h_indptr, const int64_t *d_indptr, const int32_t *d_indices,
                                int32_t *d_S, void *workspace, int64_t workspace_bytes, void *stream);
/* TEST HOOK: the device sampler expands 2 draws per triplet + T / divisor + fixed more (default 50, 65536) and
 * returns 5 -- host generator untouched -- when an epoch's rejections need more.  A test makes the margin small to
 * drive the segmented path into the end of the stream; values <= 0 restore the defaults.  Affects the workspace size. */
void lgcn_sampler_test_margin(int64_t divisor, int64_t fixed);

/* numpy legacy global RandomState stream (MT19937), used by the reference for
 * the fallback sampler and the epoch shuffle.                                 */
/* utils.set_seed -> np.random.seed(seed)                    utils.py:114-120 */
void lgcn_np_seed(uint32_t seed);
/* utils.UniformSample_original_python                       utils.py:84-110
 * S_out: int64 [train_num,3]; returns the number of rows written (users with no
 * positives are skipped), or -1 on error.                                      */
int64_t lgcn_sample_python(int n_users, int m_items, int64_t train_num,
                           const int64_t *indptr, const int32_t *indices, int64_t *S_out);
/* utils.shuffle: idx = arange(n); np.random.shuffle(idx)    utils.py:148-149 */
int lgcn_np_shuffle_perm(int64_t n, int64_t *perm_out);

/* ------------------------------------------------------------------------ */
/* Host: graph builder -- replaces Loader.getSparseGraph dataloader.py:218-234 */
/* ------------------------------------------------------------------------ */
/* COO interactions (dataloader.py:133-136) -> UserItemNet CSR with sorted,
 * de-duplicated columns (values = multiplicity).  Call with indices==NULL to get
 * the de-duplicated nnz in *nnz_out first.                                     */
int lgcn_build_user_item_csr(int n_users, int m_items, int64_t n_inter,
                             const int64_t *train_user, const int64_t *train_item,
                             int64_t *indptr, int32_t *indices, float *vals, int64_t *nnz_out);
/* fp32 row sums of A=[[0,R],[R^T,0]]                        dataloader.py:230 */
int lgcn_adj_rowsum(int n_users, int m_items, const int64_t *r_indptr, const int32_t *r_indices,
                    const float *r_vals, float *rowsum);
/* A_hat = D^-1/2 A D^-1/2 as CSR (int32 indptr[N+1], indices[2E], fp32 data[2E]),
 * value = fl32(fl32(d_inv[i]*a_ij)*d_inv[j]); d_inv is supplied by the caller
 * (the host mirror computes it with numpy.power exactly as dataloader.py:231). */
int lgcn_build_norm_adj(int n_users, int m_items, const int64_t *r_indptr, const int32_t *r_indices,
                        const float *r_vals, const float *d_inv,
                        int32_t *indptr, int32_t *indices, float *data);

/* ------------------------------------------------------------------------ */
/* Device: graph object + single kernels                                      */
/* ------------------------------------------------------------------------ */
/* The device-resident A_hat (what Loader.getSparseGraph returns, dataloader.py:203-246,
 * here as CSR: int32 indptr[n_rows+1], int32 indices[nnz] sorted per row, fp32 vals[nnz]).
 * The arrays are borrowed (must outlive the graph).  Creation is synchronous: it reads
 * indptr back once to plan the splitting of long rows and allocates that plan's scratch
 * (d_max = largest embedding dim that will be used with this graph).  Column indices are
 * range-checked here (rc 3), so no later launch can gather out of bounds.  Launches on one
 * graph share its scratch: the library orders them (a launch on another stream than the
 * previous one first waits for it).
 * row_order (DEVICE int32[n_order], may be NULL = natural order of all rows) is an optional
 * PROCESSING order of the rows -- a permutation chosen for L2 locality (n_order = n_rows), or
 * a SUBSET of the rows without repetition (n_order < n_rows: launches then compute only those
 * rows and leave the others of Y untouched -- a rank's share of a row-sharded job);
 * xcd_start (HOST int64[9], may be NULL) cuts that order into the 8 slices the 8 XCDs work on
 * (NULL: 8 slices of equal work).  Neither changes the memory layout or any result bit.   */
typedef struct lgcn_graph lgcn_graph;   /* opaque */
int lgcn_graph_create(const int32_t *indptr, const int32_t *indices, const float *vals,
                      int64_t n_rows, int64_t nnz, int32_t d_max, const int32_t *row_order,
                      int64_t n_order, const int64_t *xcd_start, lgcn_graph **out);
void lgcn_graph_destroy(lgcn_graph *g);

/* Y = A_hat X  -- replaces torch.sparse.mm(g, x)                model.py:217
 * (and its autograd backward A^T g: A_hat is symmetric).  X,Y: [n_rows,d]
 * row-major, fp32 or bf16 (x_dtype / y_dtype); d in {32,64,128,256}.           */
int lgcn_spmm_csr(const lgcn_graph *g, const void *X, int x_dtype, void *Y, int y_dtype, int d,
                  void *stream);

/* out[N,d] fp32 = mean(X_0, A X_0, ..., A^K X_0) -- replaces LightGCN.computer()
 * model.py:201-231 (cat + K sparse.mm + stack + mean).  work: (K-1)*N*d elements
 * of act_dtype (may be NULL for K == 1).                                       */
int lgcn_propagate_mean(const lgcn_graph *g, const float *E0, int K, int d, int act_dtype,
                        void *work, float *out, void *stream);

/* The same permutation from the same stream, produced ON THE DEVICE (d_perm: device int64[n]): one wave twists MT19937 in LDS
 * and aligns the draws to the Fisher-Yates steps as it generates them (64 per pass), then the swaps are resolved as sorted chains +
 * pointer doubling (csrc/lgcn_shuffle.hip).  The host generator is set to where the device stopped, so host and device calls mix
 * freely.  workspace: device bytes from lgcn_np_shuffle_perm_device_workspace(n).  Returns 3 for n >= 2^31 - 16 (the caller then
 * uses lgcn_np_shuffle_perm).  Synchronises `stream`. */
int64_t lgcn_np_shuffle_perm_device_workspace(int64_t n);
int lgcn_np_shuffle_perm_device(int64_t n, int64_t *d_perm, void *workspace, int64_t workspace_bytes, void *stream);
/* users/pos/neg[T] = S[perm[t], 0..2] -- the device side of utils.shuffle
 * (utils.py:150) applied to the sampler output (main.py:217-220).            */
int lgcn_apply_perm(const int32_t *S, int s_cols, const int64_t *perm, int64_t T,
                    int32_t *users, int32_t *pos, int32_t *neg, void *stream);

/* ------------------------------------------------------------------------ */
/* Device: fused training step -- replaces BPRLoss.stageOne   utils.py:53-64  */
/*   = LightGCN.bpr_loss (model.py:162-183) + loss.backward() (autograd of     */
/*     model.py:201-231) + torch.optim.Adam.step (utils.py:51,62)              */
/* ------------------------------------------------------------------------ */
typedef struct lgcn_ctx lgcn_ctx;   /* opaque */

typedef struct {
    const lgcn_graph *graph;    /* A_hat, N = n_users + m_items rows */
    int32_t n_users;
    int32_t d;                  /* latent_dim_rec: 32/64/128/256 */
    int32_t K;                  /* lightGCN_n_layers: 1..LGCN_MAX_LAYERS */
    int32_t act_dtype;          /* LGCN_F32 | LGCN_BF16 | LGCN_FP8: storage of layer activations (forward X_k and backward h_k) */
    /* parameters + Adam state, fp32 [N,d]: rows [0,n_users) = embedding_user.weight,
     * rows [n_users,N) = embedding_item.weight (model.py:57-60) */
    float *E0;
    float *adam_m;
    float *adam_v;
    /* workspace, caller-allocated, zero-initialised before the first step */
    void *act;                  /* max(1, dense_last ? K : K-1) tables of lgcn_table_bytes(N, d, act_dtype) bytes each */
    int64_t *G64;               /* [N,d] fixed-point (2^50) accumulator of the sparse-row gradient */
    uint32_t *bitmap;           /* [2*ceil(N/32)] rows of G64 that are non-zero (two, used alternately) */
    float *terms;               /* [2*max_batch] per-triplet loss / reg terms */
    float *contrib;             /* [3*max_batch*d + 2*max_batch] (data-parallel exchange buffer) or NULL */
    int32_t *err;               /* [1] device error flag */
    int32_t max_batch;
    /* hyper-parameters (utils.py:47-51, torch.optim.Adam defaults) */
    float decay;                /* config['decay'] */
    double lr, beta1, beta2, eps;
    int32_t xcd_remap;          /* 1: contiguous row ranges per XCD */
    int32_t dense_last;         /* 1: propagate the LAST layer densely too and read the batch rows from it (needs K
                                   activation buffers); 0: compute it only on the 3B batch rows (K-1 buffers).  Pays
                                   when the batch rows together hold more non-zeros than the graph (hub-heavy data) */
    /* hub plan of the batch-row kernel (dense_last = 0): rows with more than hub_nnz non-zeros get their last-layer row
     * from a whole-chip SpMM over just those rows, once per step, instead of from the one workgroup of each triplet
     * that names them.  0 = library default (131072), < 0 = no hub plan.  hub_chunk: non-zeros per chunk of those
     * rows (0 = default 2048).  Neither changes the algorithm, only who sums a hub row (fp32 summation order). */
    int32_t hub_nnz;
    int32_t hub_chunk;
    /* ---- the fork's optional branches (all NULL / 0: the default model).  With either one on, the step propagates every
     * layer densely (dense_last must be 1), forms the layer mean T, and runs the loss on ONE final table.
     * Item-item smoothing, model.py:99-109,228-229: items = T_items + alpha * (I2I @ T_items).  i2i / i2i_t: graphs over the
     * m_items item rows ([m_items, m_items]) holding alpha * I2I and its transpose (the caller scales the values).      */
    const lgcn_graph *i2i;
    const lgcn_graph *i2i_t;
    /* Popularity gate, model.py:66-96,139-157,176-181.  gate_params: ONE fp32 buffer holding the eight tensors of pop_mlp
     * and gate_mlp back to back in torch's named_parameters order and torch.nn.Linear layouts:
     *   pop_mlp.0.weight [Hp,1] | pop_mlp.0.bias [Hp] | pop_mlp.2.weight [d,Hp] | pop_mlp.2.bias [d] |
     *   gate_mlp.0.weight [Hg,2d] | gate_mlp.0.bias [Hg] | gate_mlp.2.weight [1,Hg] | gate_mlp.2.bias [1]
     * (P = 2 Hp + d Hp + d + 2 d Hg + 2 Hg + 1 floats); gate_adam_m / gate_adam_v / gate_grad: [P] each (Adam state of those
     * parameters, zero-initialised; the last reduced gradient, for inspection).  Hp, Hg <= 64, d <= 128.
     * terms must then hold 3 * max_batch floats (third block: the gates' entropy per triplet).                         */
    const float *item_pop;      /* [m_items] item_pop_scalar, or NULL: no gate */
    float *gate_params;
    float *gate_adam_m;
    float *gate_adam_v;
    float *gate_grad;
    int32_t pop_hidden;         /* Hp */
    int32_t gate_hidden;        /* Hg */
    float gate_entropy_coeff;
    float pop_gate_temp;
    /* Which rows the L2 term of the loss is taken on.  0 (default) = the reference: the PROPAGATED rows of the batch
     * (model.py:173: u.norm(2), pos_e.norm(2), neg_e.norm(2) of getEmbedding's outputs).  1 = UPSTREAM LightGCN, the code the
     * reference forked and whose recorded 1000-epoch run / README table it keeps (code/runs/07-10-17h52m32s--lgn,
     * LightGCN_work/README.md:88-95): the embedding tables' OWN rows of the batch (userEmb0 / posEmb0 / negEmb0), whose
     * gradient decay/B * count(row) * E0[row] does not pass through the propagation -- the Adam epilogue adds it from a
     * per-row slot count.  Not available together with the optional branches.                                        */
    int32_t reg_ego;
} lgcn_train_config;

/* Besides the caller's workspace the context owns device allocations made here with hipMalloc and released by
 * lgcn_ctx_destroy: N*d*4 bytes (fp32 copy of the step's sparse gradient rows) and, with act_dtype = LGCN_BF16 and
 * K >= 2, N*d*2 bytes (bf16 copy of E0: the input of layer 1 in that mode).  Returns 4 if they cannot be had. */
int lgcn_ctx_create(const lgcn_train_config *cfg, lgcn_ctx **out);
void lgcn_ctx_destroy(lgcn_ctx *ctx);
/* optimizer step counter (torch Adam state['step']) for checkpoint/resume */
int64_t lgcn_ctx_get_step(const lgcn_ctx *ctx);
/* number of rows in the context's hub plan (0: none was built) */
int64_t lgcn_ctx_hub_rows(const lgcn_ctx *ctx);
void lgcn_ctx_set_step(lgcn_ctx *ctx, int64_t step);
void lgcn_ctx_set_lr(lgcn_ctx *ctx, double lr);
/* Data parallel, rows mode: 1 = part 1 of a step also adds this rank's OWN gradient rows into its G64 (besides writing
 * them to the exchange block) and part 2 scatters only the other ranks' blocks -- what lgcn_train_epoch_dp does itself;
 * 0 (default) = part 2 scatters every block (one context may then play several ranks, as the emulation tests do). */
int lgcn_ctx_set_dp_local(lgcn_ctx *ctx, int on);

/* One full stageOne on a batch of B triplets (device int32 ids).
 * loss_out[0..2] (device) = {bpr + decay*reg, bpr, reg}.  No host sync.        */
int lgcn_train_step(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos, const int32_t *neg,
                    int32_t B, float *loss_out, void *stream);

/* The same step for ids as the reference passes them -- torch.long (int64) device tensors (main.py:217-225): one launch narrows the
 * three arrays into ids_scratch (device int32 [3*B], caller-owned) and the step follows on the same stream; an id outside int32
 * is flagged like any out-of-range id.                                                                                       */
int lgcn_train_step_i64(lgcn_ctx *ctx, const int64_t *users, const int64_t *pos, const int64_t *neg,
                        int32_t B, int32_t *ids_scratch, float *loss_out, void *stream);

/* A whole epoch: the loop of main.py:223-225 over ceil(T/B) consecutive batches
 * of the (already shuffled) device arrays.  loss_out: [3*ceil(T/B)].           */
int lgcn_train_epoch(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos, const int32_t *neg,
                     int64_t T, int32_t B, float *loss_out, void *stream);

/* Data-parallel split of the same step (replicated tables, batch sharded):
 *   part 1  forward propagation + per-triplet loss terms and gradient rows for this
 *           rank's shard [rank*S, min((rank+1)*S, B_global)), S = ceil(B_global/world),
 *           written to cfg.contrib as [3*S*d gradient rows | S loss | S reg terms];
 *   (caller: RCCL all-gather of that block over the ranks into `gathered`)
 *   part 2  order-independent reduction of all ranks' rows, backward propagation
 *           and Adam -- bitwise identical on every rank and to lgcn_train_step.  */
/* Floats per rank of the exchange block for a global batch of B_global triplets over `world` ranks (S = ceil(B_global/world)):
 * 3*S*d + 2*S for the default model; with the popularity gate 3*S*d + 3*S (third term block: the gates' entropy), padded to
 * an even count, + 2*P floats holding P int64 = the rank's fixed-point sums of the MLP parameter gradients.  cfg.contrib must
 * hold that many floats for the largest batch, `gathered` world times as many.  The optional branches are supported in the
 * gradient-row exchange (LGCN_DP_ROWS) only.                                                                      */
int64_t lgcn_dp_block_floats(const lgcn_ctx *ctx, int32_t B_global, int32_t world);
int lgcn_train_step_dp_part1(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos,
                             const int32_t *neg, int32_t B_global, int32_t world, int32_t rank,
                             void *stream);
/* The literal north_star form (dense gradient all-reduce), kept as an alternative:
 *   dense part 1  forward + this rank's shard accumulated into ITS OWN G64 / bitmap / loss terms
 *                 (1/B and decay/B of the global batch);
 *   (caller: RCCL all-reduce SUM of G64 [N*d int64] and of terms [2*B_global fp32]; the row bitmap
 *    is flagged for the whole global batch on every rank and needs no collective)
 *   part 2 with gathered == NULL: backward + Adam from the reduced G64.
 * Integer sums commute, so this is again bitwise equal to lgcn_train_step -- but it moves
 * N*d*8 bytes per step (Gowalla 36 MB) instead of 1.6 MB.
 * With the popularity gate on, terms carries a third block (the gates' entropy: all-reduce 3*B_global
 * floats) and the rank's fixed-point sums of the MLP parameter gradients sit in the buffer
 * lgcn_ctx_gate_total names (int64[count]): all-reduce SUM it too before part 2.
 * (the reference trains on one device: model.py:139-157,176-181 define what is summed)      */
int lgcn_train_step_dp_dense_part1(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos,
                                   const int32_t *neg, int32_t B_global, int32_t world, int32_t rank,
                                   void *stream);
int lgcn_ctx_gate_total(const lgcn_ctx *ctx, void **buf, int32_t *count);   /* count 0 without the gate */
int lgcn_train_step_dp_part2(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos,
                             const int32_t *neg, int32_t B_global, int32_t world,
                             const float *gathered, float *loss_out, void *stream);

/* Column-sharded data parallelism (LGCN_DP_COLS): rank r holds columns [r*d/W, (r+1)*d/W) of the tables -- its context is an
 * ordinary context of width d/W (32/64/128/256) over the same graph, cfg.contrib holding 3*max_batch*(d/W) floats.  Every rank
 * sees the WHOLE batch.  part 1: forward + the batch's slot rows + the PARTIAL scores / reg terms over this rank's columns
 * (*partials = device float[3*B]: pos score | neg score | reg term); the caller all-reduces (SUM) those 3*B floats over the
 * ranks IN PLACE; part 2: loss terms, gradient rows of this rank's columns, backward, Adam.  One 3*B-float collective per
 * step and per-rank SpMM work that falls with W.  Equal to lgcn_train_step up to the summation order of the dot products
 * (W partial sums).  lgcn_train_epoch_dp(reduce = LGCN_DP_COLS) runs the loop with the all-reduce on the communicator.     */
int lgcn_train_step_cols_part1(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos, const int32_t *neg,
                               int32_t B, float **partials, void *stream);
int lgcn_train_step_cols_part2(lgcn_ctx *ctx, const int32_t *users, const int32_t *pos, const int32_t *neg,
                               int32_t B, float *loss_out, void *stream);

/* ------------------------------------------------------------------------ */
/* Device: fused full-ranking evaluation -- replaces the body of Procedure.Test  */
/* ------------------------------------------------------------------------ */
/* For every listed user: scores against ALL items (model.getUsersRating, model.py:114-123:
 * U_b . I^T on the propagated table E[N,d] -- matrix cores, fp32 accumulate, fp32-accurate), train positives
 * set to -(1<<10) (Procedure.py:177-181; CSR with int64 indptr[n_users+1] and ascending int32
 * indices = dataset.allPos), top-K (Procedure.py:183) -- in one kernel, the [users, m_items]
 * score matrix is never materialised.  topk_items [n_eval,K] int32, best first (ties: lower
 * item id first; when the sweep is split over several workgroups per user block -- catalogues of >= 4096 items -- the
 * order of two items whose scores tie EXACTLY at the K-th place is unspecified, as torch.topk's is);
 * topk_scores [n_eval,K] or NULL.  1 <= K <= 64 (Procedure.py:183 takes k = max(topks)).          */
int lgcn_eval_topk(const float *E, int32_t n_users, int32_t m_items, int32_t d,
                   const int32_t *users, int32_t n_eval,
                   const int64_t *train_indptr, const int32_t *train_indices,
                   int32_t K, int32_t *topk_items, float *topk_scores, void *stream);
/* The same with the train-positive masks precomputed ONCE per dataset instead of walked from the CSR inside the sweep:
 * masks = lgcn_eval_mask_words(m_items, n_eval) uint32 words filled by lgcn_eval_build_masks (word [t * stride + slot],
 * stride = n_eval rounded up to 128: bit i = item 32 t + i is a train positive of users[slot]).  The sweep then holds
 * no global load besides the item tiles and one coalesced mask load per wave and tile.  Same results.          */
int64_t lgcn_eval_mask_words(int32_t m_items, int32_t n_eval);
int lgcn_eval_build_masks(const int32_t *users, int32_t n_eval, const int64_t *train_indptr, const int32_t *train_indices,
                          int32_t m_items, uint32_t *masks, void *stream);
int lgcn_eval_topk_masked(const float *E, int32_t n_users, int32_t m_items, int32_t d,
                          const int32_t *users, int32_t n_eval,
                          const int64_t *train_indptr, const int32_t *train_indices,
                          int32_t K, int32_t *topk_items, float *topk_scores, const uint32_t *masks, void *stream);
/* The same, with every score produced by the fp32 matrix instructions (v_mfma_f32_32x32x2_f32).  lgcn_eval_topk itself
 * computes the fp32 product on the bf16 matrix cores where it can (d <= 64, at most 65 K items per part of the catalogue: three parts for
 * K <= 20, two for K <= 64; d = 128 with K <= 20): both operands split EXACTLY into three bf16 values each, six of the nine product planes accumulated in
 * fp32 -- the planes left out are below the rounding of an fp32 dot product, so both entry points are fp32-accurate and
 * may differ only where two scores tie within that rounding.  This one is the slower cross-check.          */
int lgcn_eval_topk_fp32(const float *E, int32_t n_users, int32_t m_items, int32_t d,
                        const int32_t *users, int32_t n_eval,
                        const int64_t *train_indptr, const int32_t *train_indices,
                        int32_t K, int32_t *topk_items, float *topk_scores, void *stream);
/* Per-user precision / recall / NDCG at the cut-offs ks[n_ks] (Procedure.test_one_batch,
 * Procedure.py:89-121; utils.RecallPrecision_ATk / NDCGatK_r / getLabel, utils.py:173-217) from the
 * ranked ids and the users' test lists (CSR over the n_eval slots, ids ASCENDING per slot), in
 * float64; per_user [n_eval, 3*n_ks] = precision | recall | ndcg, sums [3*n_ks] = their sums over
 * the users in a fixed order (the reference averages them, Procedure.py:191-192).        */
int lgcn_eval_metrics(const int32_t *topk_items, int32_t n_eval, int32_t K,
                      const int64_t *test_indptr, const int32_t *test_items_sorted,
                      const int32_t *ks, int32_t n_ks, double *per_user, double *sums, void *stream);

/* ------------------------------------------------------------------------ */
/* Data parallel over RCCL (no counterpart in the reference: SURVEY 2, north_star) */
/* ------------------------------------------------------------------------ */
/* RCCL is resolved at run time (dlopen of the librccl already mapped into the process, else
 * $LGCN_RCCL_PATH, else the system one); the library itself does not link it.  One communicator
 * per process / GPU.  Rank 0 creates the 128-byte id (ncclGetUniqueId) and the caller hands it to
 * every rank through any channel it has (torch.distributed store, MPI, a file).             */
#define LGCN_DP_ID_BYTES 128
enum { LGCN_DP_ROWS = 0, LGCN_DP_DENSE = 1, LGCN_DP_ROW_SHARDED = 2, LGCN_DP_COLS = 3 };
typedef struct lgcn_dp lgcn_dp;   /* opaque */
int lgcn_dp_available(void);                         /* 1 if RCCL could be resolved */
int lgcn_dp_unique_id(void *id128);
int lgcn_dp_init(const void *id128, int world, int rank, lgcn_dp **out);   /* on the current HIP device */
/* TEST HOOK, no RCCL involved: `world` communicators for `world` THREADS OF THIS PROCESS on the current device
 * (out[world]; destroy each with lgcn_dp_destroy).  Their collectives meet on a host barrier and move the blocks
 * with hipMemcpyAsync on the calling rank's stream, so lgcn_train_epoch_dp's world > 1 control flow can run -- and
 * be compared bit for bit with the single-GPU epoch -- on a one-GPU machine: every thread calls
 * lgcn_train_epoch_dp with its own context, tables, stream and communicator.  Synchronous by construction. */
int lgcn_dp_init_loopback(int world, lgcn_dp **out);
void lgcn_dp_destroy(lgcn_dp *dp);
/* In-place SUM all-reduce of n device floats over the communicator, on `stream` (no host sync).  bench.py counts the ranks
 * RCCL itself sees with it (an all-reduce of 1.0 per rank), independent of torch.distributed's WORLD_SIZE. */
int lgcn_dp_allreduce_sum_f32(lgcn_dp *dp, float *buf, int64_t n, void *stream);
int lgcn_dp_world(const lgcn_dp *dp);
int lgcn_dp_rank(const lgcn_dp *dp);
/* A whole data-parallel epoch in one host call: the loop of main.py:223-225 over ceil(T/B_global)
 * global batches (arrays identical on every rank); per batch: part 1, the collective on `stream`
 * (reduce = LGCN_DP_ROWS: ncclAllGather of cfg.contrib blocks into `gathered`
 * [world * (3*S*d + 2*S)] floats, S = ceil(B_global/world); LGCN_DP_DENSE: ncclAllReduce of G64 and
 * of the loss terms, `gathered` unused), part 2.  No host synchronisation.  loss_out: [3*steps].
 * LGCN_DP_ROW_SHARDED: the row-sharded step below, every exchange one grouped in-place
 * ncclBroadcast of the owners' row ranges; row_ranges = HOST int64[world][4] = {user rows lo, hi,
 * item rows lo, hi} (row ids of the [N,d] tables) owned by each rank, else NULL.          */
int lgcn_train_epoch_dp(lgcn_ctx *ctx, lgcn_dp *dp, const int32_t *users, const int32_t *pos,
                        const int32_t *neg, int64_t T, int32_t B_global, int32_t reduce,
                        const int64_t *row_ranges, float *gathered, float *loss_out, void *stream);

/* Row-sharded propagation (SURVEY 8e "beyond the contract"; analogue of the reference's A_split row
 * folds, dataloader.py:192-201).  Tables stay replicated in layout; the graph handed to the context was
 * created with row_order = ONLY the rows this rank owns (n_order < n_rows), so every SpMM phase
 * computes 1/world of a layer and the owners' rows are exchanged before the next phase:
 *   FWD k = 1..K-1   X_k[owned] = (A X_{k-1})[owned]                       -> exchange lgcn_rs_buffer(FWD,k)
 *   BPR              this rank's batch shard -> cfg.contrib (as dp part 1)  -> all-gather of the blocks
 *   SCATTER          all ranks' gradient rows -> G64 + row flags (every rank, identical)
 *   BWD k = K..1     h_{k-1}[owned] = Gs + (A h_k)[owned]; k = 1: Adam on the owned rows
 *                                                                          -> exchange lgcn_rs_buffer(BWD,k)
 *   FINISH           zero the batch rows of G64 and their flags, reduce the loss
 * Every row is computed by the same code in the same order wherever it runs, so the result is
 * bitwise identical to lgcn_train_step.  Adam state (m, v) of a row lives on its owner only.
 * An exchanged buffer of dtype LGCN_FP8 is an fp8 table (lgcn_table_bytes): a caller running the
 * phases itself moves the owners' rows (d bytes each) AND their row scales (fp32, behind the rows).   */
enum { LGCN_RS_FWD = 0, LGCN_RS_BPR = 1, LGCN_RS_SCATTER = 2, LGCN_RS_BWD = 3, LGCN_RS_FINISH = 4 };
int lgcn_rs_phase(lgcn_ctx *ctx, int32_t phase, int32_t k, const int32_t *users, const int32_t *pos,
                  const int32_t *neg, int32_t B_global, int32_t world, int32_t rank,
                  const float *gathered, float *loss_out, void *stream);
int lgcn_rs_buffer(const lgcn_ctx *ctx, int32_t phase, int32_t k, void **buf, int32_t *dtype);

/* reads and clears the device error flag (synchronises the stream): 0 = none,
 * 1 = id out of range in users/pos/neg.                                        */
int lgcn_ctx_check(lgcn_ctx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* LGCN_HIP_H */
