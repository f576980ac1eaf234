/*
 * pls_hip.h -- C-ABI of the MI355X (gfx950) PLS fit / predict hot path.
 *
 * This is the drop-in boundary for the body of PLS::Model::plsr and the products
 * behind Model::scores / coefficients / fitted_values of tjhladish/PLS:
 *
 *   reference interface                                   replaced by
 *   ----------------------------------------------------  --------------------------
 *   void Model::plsr(const Mat2D&, const Mat2D&, METHOD)   pls_hip_fit
 *     include/PLS/pls.h:199, src/pls.cpp:390-437
 *   const Mat2Dc Model::coefficients(size_t)               pls_hip_coefficients
 *     include/PLS/pls.h:214, src/pls.cpp:444-447
 *   const Mat2D  Model::fitted_values(const Mat2D&,size_t) pls_hip_xb  (X_new * B)
 *     include/PLS/pls.h:218, src/pls.cpp:449-451
 *   const Mat2Dc Model::scores(const Mat2D&, size_t)       pls_hip_xb  (X_new * R[:, :c])
 *     include/PLS/pls.h:203, src/pls.cpp:439-442
 *
 * The reference has no FFI of its own (it is a C++ library on Eigen); the binding a
 * maintainer adds is the body of Model::plsr in src/pls.cpp -- shown in INTEGRATION.md and
 * shipped in pls_amd/host/pls.cpp.
 *
 * Conventions (the reference's, include/PLS/pls.h:22-27): every matrix is COLUMN-MAJOR
 * with an explicit leading dimension in elements (ld >= rows); dimensions are 64-bit.
 * X is N x K (samples x predictors), Y is N x M, A components.  W,P,R are K x A, Q is
 * M x A, B is K x M, T is N x A.  W,P,Q,R,B and every vector are always fp64; X, Y and T
 * use the storage dtype of the call (fp64, or fp32 storage with fp64 accumulation).
 * All results are real: the reference's std::complex containers (include/PLS/pls.h:26-27)
 * always hold zero imaginary parts and are rebuilt at the C++ boundary.
 *
 * A handle is not thread-safe: one caller at a time per handle (the reference's Model has no shared
 * state either; use one handle per host thread, handles are independent).
 *
 * No torch types, no C++ types, no exceptions cross this boundary; every entry point
 * returns a pls_hip_status and never calls exit().  There is NO CPU fallback: without a
 * gfx950 device every compute entry point fails with PLS_HIP_ERR_DEVICE.
 */
#ifndef PLS_HIP_H
#define PLS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLS_HIP_ABI_VERSION 1

#if defined(PLS_HIP_BUILDING)
#define PLS_HIP_API __attribute__((visibility("default")))
#else
#define PLS_HIP_API
#endif

typedef struct pls_hip_context *pls_hip_handle;

typedef enum {
    PLS_HIP_OK = 0,
    PLS_HIP_ERR_INVALID = 1,     /* bad shape / pointer / enum (the reference only assert()s: src/pls.cpp:345-347) */
    PLS_HIP_ERR_DEVICE = 2,      /* HIP runtime error or no usable gfx950 device */
    PLS_HIP_ERR_ALLOC = 3,       /* device allocation failed */
    PLS_HIP_ERR_UNSUPPORTED = 4, /* valid request this build does not implement */
    PLS_HIP_ERR_REDUCER = 5      /* the injected all-reduce returned non-zero */
} pls_hip_status;

/* PLS::METHOD, include/PLS/pls.h:131 */
typedef enum { PLS_HIP_KERNEL_TYPE1 = 0, PLS_HIP_KERNEL_TYPE2 = 1 } pls_hip_method;

typedef enum { PLS_HIP_F64 = 0, PLS_HIP_F32 = 1 } pls_hip_dtype;

/* where the caller's X/Y/outputs live */
typedef enum { PLS_HIP_MEM_HOST = 0, PLS_HIP_MEM_DEVICE = 1 } pls_hip_mem;

/* How the score/loading passes treat X (identical W,P,Q,R,T,B up to rounding):
 *  KERNEL : the reference's operation sequence -- X is read-only, only the K x M matrix
 *           XY is deflated (src/pls.cpp:419-429).
 *  NIPALS : the north-star sequence -- t = X_a w, p = X_a^T t, then the rank-1 deflation
 *           X_{a+1} = X_a - t p^T on a library-owned working copy.
 *  GRAM   : XX = X^T X once on the matrix cores (v_mfma_f64_16x16x4_f64 SYRK), the component loop on
 *           K x K data as KERNEL_TYPE2 does (tt = r^T XX r, p = XX r / tt, src/pls.cpp:422-425), then
 *           T = X R in one pass.  2 + A*0 passes over X; pays off when A exceeds ~K/50.
 *  AUTO   : KERNEL or GRAM, whichever a bandwidth / matrix-core cost model predicts to be faster for
 *           the shape of the call (GRAM only for K <= 2048, single rank). */
typedef enum {
    PLS_HIP_ALGO_KERNEL = 0,
    PLS_HIP_ALGO_NIPALS = 1,
    PLS_HIP_ALGO_GRAM = 2,
    PLS_HIP_ALGO_AUTO = 3
} pls_hip_algo;

typedef enum {
    PLS_HIP_OPT_ALGO = 1,       /* pls_hip_algo; default PLS_HIP_ALGO_KERNEL */
    PLS_HIP_OPT_FUSE = 2,       /* 0: one kernel per product (Xv, X^T t, deflate); 1 (default): row-tile-resident fused pass when the shape allows */
    PLS_HIP_OPT_PROFILE = 3,    /* HIP events on the launch stream: 1 = around the streaming kernels over X, 2 = around every kernel */
    PLS_HIP_OPT_POWER_ITERS = 4, /* squarings of the S^T S power iteration (m > 1); default 48 */
    PLS_HIP_OPT_FUSED_GRID = 5,  /* workgroups of the fused pass; 0 (default) = 2 per CU */
    PLS_HIP_OPT_WORK_LAYOUT = 6, /* NIPALS work buffer (the deflated copy of X) of a fused fit: 1 (default) row-tile-major, 0 column-major */
    PLS_HIP_OPT_DEFER = 7,       /* NIPALS plan, K <= 512: write the deflated matrix back every D-th component only (1..4); the
                                    D - 1 pending rank-1 updates are re-applied in registers.  1 (default) = explicit deflation */
    PLS_HIP_OPT_GRAPH = 8        /* 1: a device-memory pls_hip_fit that repeats an earlier call (same pointers, shapes, options) is
                                    captured into a hipGraph on its second occurrence and replayed as ONE graph launch from the
                                    third on.  Single rank, profiling off, a stream of its own (not the default stream).  0 (default): every
                                    call enqueues its kernels */
} pls_hip_option;

/*
 * In-place sum over ranks of `count` fp64 values at DEVICE address `buf`, ordered on
 * `stream` (a hipStream_t).  Must leave bit-identical results on every rank.
 * EVERY message is SLICED: `count` is a multiple of PLS_HIP_REDUCE_SLICES, the buffer is
 * PLS_HIP_REDUCE_SLICES slices of count / 8 values, and the library's consumers add the
 * slices of a value in index order.  A reducer may therefore either sum element by element
 * (RCCL, torch.distributed) or leave the total of the 8 slices in slice 0 and zeros in
 * slices 1..7 (the library's device-side exchanges); the library never calls it with an
 * unsliced count.
 * Calls per KERNEL_TYPE1 fit: once with count = 8*K*M (the X^T Y partial), once per
 * component with count = 8*(K+1) (the packed [X^T t, t^T t]), and -- on every fit of a
 * handle with nranks > 1 -- once more with count = 8*8 after the component loop: the
 * replica-divergence guard (checksums of W, P, Q, R, B; the environment switch
 * PLS_HIP_REPLICA_GUARD=0, which must be set identically on EVERY rank, removes it).
 * KERNEL_TYPE2 / GRAM add one message of 8*K*K (X^T X); pls_hip_colwise_z_scores sends two
 * of 8*K, pls_hip_sse_by_components one of 8*A*M per range of component counts.
 * Return 0 on success.
 */
#define PLS_HIP_REDUCE_SLICES 8
typedef int (*pls_hip_allreduce_fn)(void *user, void *buf, int64_t count, void *stream);

/* Per-family device time of every launch since the previous pls_hip_get_timing call (recorded
 * only while PLS_HIP_OPT_PROFILE=1; ms from hipEventElapsedTime on the launch stream), launch
 * counts and algorithmic bytes.  pls_hip_get_timing synchronises the stream and resets. */
enum {
    PLS_HIP_FAM_XTY = 0,     /* X^T Y  and  X^T t   (column reductions)           */
    PLS_HIP_FAM_XB = 1,      /* t = X v, X B        (row products)                */
    PLS_HIP_FAM_DEFLATE = 2, /* X -= t p^T                                        */
    PLS_HIP_FAM_FUSED = 3,   /* tile-resident fused pass ([deflate +] score + loading) */
    PLS_HIP_FAM_SMALL = 4,   /* partial reduction + per-component K-sized bookkeeping */
    PLS_HIP_FAM_COUNT = 5
};
typedef struct {
    double fit_ms;                       /* summed over fits: first launch to last     */
    int64_t fits;                        /* number of pls_hip_fit calls covered        */
    double fam_ms[PLS_HIP_FAM_COUNT];    /* summed over launches of the family        */
    int64_t fam_launches[PLS_HIP_FAM_COUNT];
    int64_t fam_bytes[PLS_HIP_FAM_COUNT]; /* ALGORITHMIC bytes summed over those launches */
} pls_hip_timing;

/* ---- lifetime ------------------------------------------------------------------- */

PLS_HIP_API int pls_hip_abi_version(void);

/* device: HIP ordinal.  stream: the hipStream_t every kernel, copy and event of this handle is
 * issued on; NULL = the device's default (null) stream.  Fails with PLS_HIP_ERR_DEVICE if the
 * device is not gfx950. */
PLS_HIP_API int pls_hip_create(pls_hip_handle *out, int device, void *stream);
PLS_HIP_API int pls_hip_destroy(pls_hip_handle h);
PLS_HIP_API int pls_hip_set_stream(pls_hip_handle h, void *stream);
PLS_HIP_API int pls_hip_set_option(pls_hip_handle h, int option, int64_t value);
PLS_HIP_API int pls_hip_get_option(pls_hip_handle h, int option, int64_t *value);
/* Row-sharded fit over `nranks` processes (one per GPU): every rank passes its own row
 * block and the same K, M, A; fn sums the small partial products.  fn == NULL: single rank. */
PLS_HIP_API int pls_hip_set_reducer(pls_hip_handle h, pls_hip_allreduce_fn fn, void *user, int rank, int nranks);
/* Optional caller-owned DEVICE staging buffer for the reducer
 * (>= PLS_HIP_REDUCE_SLICES * max(K*M, K+1) fp64), so a
 * host runtime can hand its collective a buffer it allocated itself. */
PLS_HIP_API int pls_hip_set_reduce_buffer(pls_hip_handle h, void *buf, int64_t count);
PLS_HIP_API int pls_hip_synchronize(pls_hip_handle h);
PLS_HIP_API const char *pls_hip_last_error(pls_hip_handle h);
PLS_HIP_API int pls_hip_get_timing(pls_hip_handle h, pls_hip_timing *out);

/* ---- the hot path --------------------------------------------------------------- */

/*
 * Fit A components: the body of Model::plsr (src/pls.cpp:390-437).
 * X, Y are never written.  PLS_HIP_KERNEL_TYPE2 (XX = X^T X once, no pass over X per component,
 * src/pls.cpp:398,422-425) does not compute T: T may be NULL and is left untouched.  B (K x M, ld K)
 * may be NULL; otherwise it receives coefficients(A) = R Q^T (src/pls.cpp:444-447).
 * mem == DEVICE: all pointers are device pointers, the call only enqueues work on the
 * stream (pls_hip_synchronize or the caller's own stream sync completes it).
 * mem == HOST: pointers are host memory; the call copies in (pinned double-buffered staging pipeline), fits, copies
 * out and returns with the results in place.  Under PLS_HIP_ALGO_AUTO / _GRAM and for PLS_HIP_KERNEL_TYPE2 (single
 * rank) X^T X and X^T Y are accumulated on the matrix cores while X crosses PCIe and the component loop starts from them.
 * Shapes follow the reference's asserts (src/pls.cpp:345-347): 1 <= A <= K, N >= 1
 * (N may be 0 on a rank of a sharded fit), 1 <= M <= 1024 (beyond 32 responses the M-sized work of the component
 * update runs from global memory: correct, about a millisecond per component slower).  Single-response problems that fit
 * one workgroup's registers (N <= 1024, K <= 26 * floor(16 / ceil(N/64))) run as ONE launch under the KERNEL / AUTO plans.
 * A > rank(X) yields inf/NaN in the surplus columns, as in the reference (:427-428).
 */
PLS_HIP_API int pls_hip_fit(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                int64_t N, int64_t K, int64_t M, int64_t A, int method, int dtype, int mem,
                double *W, double *P, double *Q, double *R, void *T, int64_t ldt, double *B);

/* B(K x M) = R[:, :c] Q[:, :c]^T.  Model::coefficients, src/pls.cpp:444-447. */
PLS_HIP_API int pls_hip_coefficients(pls_hip_handle h, const double *R, const double *Q, int64_t K,
                         int64_t M, int64_t A, int64_t c, int mem, double *B);

/* out(N x C) = X(N x K) * Bm(K x C); Bm fp64, X/out in `dtype`.
 * Model::fitted_values (Bm = B, src/pls.cpp:449-451), Model::scores (Bm = R[:, :c], :439-442). */
PLS_HIP_API int pls_hip_xb(pls_hip_handle h, const void *X, int64_t ldx, int64_t N, int64_t K,
               const double *Bm, int64_t ldb, int64_t C, int dtype, int mem, void *out,
               int64_t ldo);

/* ---- single steps of the path on DEVICE pointers (parity tests, bench, profiling) -- */

/* XY(K x M, ld K, fp64) = X^T Y   (src/pls.cpp:396; with Y = t, M = 1: src/pls.cpp:421) */
PLS_HIP_API int pls_hip_xty(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                int64_t N, int64_t K, int64_t M, int dtype, double *XY);
/* dst = src - t p^T  (N x K; dst may equal src).  The north-star rank-1 deflation. */
PLS_HIP_API int pls_hip_deflate(pls_hip_handle h, const void *src, int64_t lds, void *dst, int64_t ldd,
                    int64_t N, int64_t K, const void *t, const double *p, int dtype);

/* ---- callers either side of the path, on DEVICE pointers ---- */

/* Column z-scores as the reference's main applies them before the fit (src/pls.cpp:69-111,
 * src/main.cpp:24-25): mean[k], sd[k] = sqrt(SST/(n_total-1)), Z = (X - mean)/sd (Z may be NULL to get
 * the statistics only, or equal to X for an in-place transform).  A constant column yields NaN, as
 * upstream (:103).  n_total = rows over all ranks of a sharded matrix (= N on one GPU). */
PLS_HIP_API int pls_hip_colwise_z_scores(pls_hip_handle h, const void *X, int64_t ldx, int64_t N,
                                         int64_t n_total, int64_t K, int dtype, void *Z, int64_t ldz,
                                         double *mean, double *sd);
/* SSE(M x A, ld M)[m, c-1] = sum_i (Y[i,m] - (S[:, :c] Q[:, :c]^T)[i,m])^2 for c = 1..A in one sweep
 * over the scores S = X R (N x A): Model::SSE for every component count (src/pls.cpp:457-459) without the
 * A separate X*B passes of print_explained_variance (:551-562).  Any A (long component lists are swept in ranges of
 * 1024/M component counts); M <= 1024. */
PLS_HIP_API int pls_hip_sse_by_components(pls_hip_handle h, const void *S, int64_t lds, const void *Y,
                                          int64_t ldy, int64_t N, int64_t A, int64_t M, const double *Q,
                                          int dtype, double *SSE);

/* The same for a model (R: K x A, Q: M x A, ld = rows) and data (X, Y) in host or device memory:
 * S = X R (one pass over X, kept on the device) followed by the sweep above. */
PLS_HIP_API int pls_hip_model_sse(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                                  int64_t N, int64_t K, int64_t M, int64_t A, const double *R,
                                  const double *Q, int dtype, int mem, double *SSE);

/*
 * Cross-validation folds in one batched launch (Model::cv_LOO / cv_LSO,
 * src/pls.cpp:469-549).  Fold f refits A components on every row EXCEPT test_idx[f*test_size .. +test_size)
 * and records the residuals of those test rows:
 *     E[m*(nobs*A) + (f*test_size + i) + c*nobs] = Y[row, m] - x_row^T B_{c+1}^{(f)},   nobs = num_folds*test_size
 * i.e. M matrices of nobs x A, the layout of PLS::Residual::errors().  Leave-one-out = test_size 1,
 * num_folds N, test_idx = 0..N-1.  test_idx is HOST memory (distinct indices within a fold); X, Y, E follow
 * `mem`.  All folds share XX = X^T X and XY = X^T Y formed once; a fold works on
 * XX - X_test^T X_test applied on the fly (the KERNEL_TYPE2 recurrence, src/pls.cpp:422-425): no per-fold
 * pass over X.  Small single-response data (N <= 1024, K <= 26 * floor(16 / ceil(N/64)), M = 1) run as one single-launch fit per
 * fold on the masked X instead (no X^T X at all).  Shapes that launch declines (M > 32, A > 4096, K > 16384, workspaces that do not fit) run as one
 * device refit per fold instead -- same results, num_folds fits.  Single rank.  The call returns after the work
 * has completed.
 */
PLS_HIP_API int pls_hip_cv_folds(pls_hip_handle h, const void *X, int64_t ldx, const void *Y, int64_t ldy,
                                 int64_t N, int64_t K, int64_t M, int64_t A, const int64_t *test_idx,
                                 int64_t test_size, int64_t num_folds, int dtype, int mem, double *E);

/* ---- synthetic inputs, generated on the device (DESIGN.md "Synthetic inputs") ------ */

/* rows [row0, row0+nrows) of the global matrix -> X (nrows x K, ld ldx) / Y (nrows x M) */
PLS_HIP_API int pls_hip_synth_x(pls_hip_handle h, void *X, int64_t ldx, int64_t row0, int64_t nrows,
                    int64_t K, uint64_t seed, int dtype);
PLS_HIP_API int pls_hip_synth_y(pls_hip_handle h, void *Y, int64_t ldy, int64_t row0, int64_t nrows,
                    int64_t M, uint64_t seed, int dtype);

/* ---- one process per GPU without RCCL: the device-side exchange across processes ---------------------------------
 *
 * The reducer of a row-sharded fit (pls_hip_set_reducer) as a direct exchange between the ranks' GPUs: every rank owns an
 * inbox in fine-grained device memory, exported with hipIpcGetMemHandle; a collective is two small launches per rank --
 * the rank WRITES its partial sums into every peer's inbox over xGMI and raises a sequence flag there, then spins on its
 * own flags and adds the inboxes in rank order (exchange_kernels.hpp; the same kernels the in-process group uses).  These
 * messages are a few KB and latency-bound, which is what this form is for; RCCL (include/pls_hip_rccl.h) or any other
 * pls_hip_allreduce_fn remain alternatives.  Set-up, on every rank:
 *     pls_hip_xchg_create(h, rank, nranks, mine)         -> `mine`: PLS_HIP_XCHG_HANDLE_BYTES to publish
 *     (all-gather the blobs in rank order with whatever the application has: MPI, torch.distributed, a file)
 *     pls_hip_xchg_connect(h, all)                        opens the peers' inboxes, installs the reducer
 *     pls_hip_xchg_selftest(h)                            COLLECTIVE: one round on known values with a 5 s limit;
 *                                                         PLS_HIP_ERR_REDUCER if this system cannot do it -- then every
 *                                                         rank should pls_hip_xchg_destroy and take another reducer
 * A rank that waits longer than 30 s (PLS_HIP_XCHG_TIMEOUT_S) for a peer -- a rank failed or fell out of step -- gives up:
 * pls_hip_synchronize returns PLS_HIP_ERR_REDUCER from then on.  2 <= nranks <= 16.
 */
#define PLS_HIP_XCHG_HANDLE_BYTES 160
PLS_HIP_API int pls_hip_xchg_create(pls_hip_handle h, int rank, int nranks, void *mine);
PLS_HIP_API int pls_hip_xchg_connect(pls_hip_handle h, const void *all);
PLS_HIP_API int pls_hip_xchg_selftest(pls_hip_handle h);
PLS_HIP_API int pls_hip_xchg_destroy(pls_hip_handle h);

/* ---- one process, several GPUs: a group of handles behind one call ------------------
 *
 * The reference's Model is one object driven by one host thread (include/PLS/pls.h:187-199, src/pls.cpp:340-353).
 * A group keeps that shape while the rows of X and Y are spread over the GPUs of the node: it owns one handle and
 * one stream per member and runs every call with one host thread per member inside the library.  The members'
 * partial products are summed by a reducer the group installs itself -- a fixed-order all-reduce with no RCCL and,
 * when every member has a GPU of its own, nothing on the host at all: each member WRITES its partial sums into an
 * inbox on every peer over xGMI and raises a sequence flag there, and the consumer's kernel spins on its flags and adds
 * the inboxes in rank order ("device-side exchange": two small launches per member and collective, a one-hop direct
 * write as SURVEY.md 8(e) recommends for these latency-bound messages; a member that waits longer than 30 s -- a peer
 * failed or fell out of step -- makes the call return PLS_HIP_ERR_REDUCER).  The exchange is tried once on known
 * values when the group is created.  Where it is not available (members that share a GPU -- see below -- or no peer
 * writes into fine-grained memory), and for messages beyond 512 KB, the members instead READ each other's buffers
 * behind two host-thread barriers per collective ("host-synchronised exchange").  Either way all members end up with
 * identical bits.  PLS_HIP_GROUP_EXCHANGE = device | host overrides the choice.
 * `devices[r]` is the HIP ordinal of member r; an ordinal may repeat ("virtual shards" sharing one GPU: the way the
 * sharded path is exercised on a one-GPU machine).  A group of one member is a plain single-GPU fit.
 *
 * Matrices live on the device(s) between calls (288 GB of HBM per GPU: the training data of a Model stays resident
 * instead of being copied on the host as src/pls.cpp:344 does): pls_hip_group_upload spreads a HOST matrix over the
 * members by rows -- member r owns a contiguous block, the same partition for every matrix of the same row
 * count -- through a double-buffered pinned staging pipeline (host threads repack a tile while the DMA engine
 * transfers the previous one).
 */
typedef struct pls_hip_group_s *pls_hip_group;
typedef struct pls_hip_matrix_s *pls_hip_matrix;

PLS_HIP_API int pls_hip_group_create(pls_hip_group *out, int n, const int *devices);
PLS_HIP_API int pls_hip_group_destroy(pls_hip_group g);
PLS_HIP_API int pls_hip_group_size(pls_hip_group g);
/* 1 = the device-side exchange carries the group's collectives, 0 = the host-synchronised one (or one member) */
PLS_HIP_API int pls_hip_group_exchange(pls_hip_group g);
/* member r's handle (options, timing); it stays owned by the group */
PLS_HIP_API int pls_hip_group_handle(pls_hip_group g, int rank, pls_hip_handle *out);
PLS_HIP_API int pls_hip_group_set_option(pls_hip_group g, int option, int64_t value);
PLS_HIP_API const char *pls_hip_group_last_error(pls_hip_group g);

/* N x K host matrix (column-major, ld) -> resident, row-sharded.  Returns when the data has left `host`. */
PLS_HIP_API int pls_hip_group_upload(pls_hip_group g, const void *host, int64_t ld, int64_t N, int64_t K,
                                     int dtype, pls_hip_matrix *out);
/* X (N x K) and Y (N x M) of one data set together.  While the rows of X stream in over PCIe, every member
 * accumulates X^T X and X^T Y of its rows block by block on the matrix cores (a block's SYRK takes a fraction of
 * its transfer time), and the pair keeps them: a later pls_hip_group_fit(X, Y) under PLS_HIP_ALGO_AUTO or _GRAM (and
 * KERNEL_TYPE2) starts its component loop at once -- no pass over X before it, one (T = X R) after it -- and
 * pls_hip_group_cv_folds skips its own X^T X.  Layouts the matrix-core kernel declines (K > 4096, unaligned)
 * simply upload; the fit then forms the products itself. */
PLS_HIP_API int pls_hip_group_upload_xy(pls_hip_group g, const void *hostX, int64_t ldx, const void *hostY,
                                        int64_t ldy, int64_t N, int64_t K, int64_t M, int dtype,
                                        pls_hip_matrix *X, pls_hip_matrix *Y);
/* uninitialised resident N x K matrix (e.g. the scores T of a fit) */
PLS_HIP_API int pls_hip_group_alloc(pls_hip_group g, int64_t N, int64_t K, int dtype, pls_hip_matrix *out);
/* columns [col0, col0 + ncols) of a resident matrix -> host (N x ncols, ld) */
PLS_HIP_API int pls_hip_group_download(pls_hip_group g, pls_hip_matrix m, int64_t col0, int64_t ncols, void *host,
                                       int64_t ld);
PLS_HIP_API int pls_hip_group_free(pls_hip_group g, pls_hip_matrix m);
PLS_HIP_API int pls_hip_matrix_shape(pls_hip_matrix m, int64_t *N, int64_t *K, int *dtype);
/* member r's block of a resident matrix: device pointer, leading dimension, first row, row count */
PLS_HIP_API int pls_hip_matrix_block(pls_hip_matrix m, int rank, void **data, int64_t *ld, int64_t *row0,
                                     int64_t *nrows);

/* pls_hip_fit on resident X (N x K), Y (N x M); T: resident N x A (scores stay on the devices; NULL for
 * PLS_HIP_KERNEL_TYPE2).  W, P, R (K x A), Q (M x A), B (K x M, may be NULL) are HOST memory, ld = rows.  Every
 * member derives the same W, P, Q, R, B bit for bit; the call checks that before it returns
 * (PLS_HIP_ERR_REDUCER otherwise). */
PLS_HIP_API int pls_hip_group_fit(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A, int method,
                                  double *W, double *P, double *Q, double *R, pls_hip_matrix T, double *B);
/* out (resident N x C) = X * Bm, Bm HOST K x C fp64 (Model::scores / fitted_values, src/pls.cpp:439-451) */
PLS_HIP_API int pls_hip_group_xb(pls_hip_group g, pls_hip_matrix X, const double *Bm, int64_t ldb, int64_t C,
                                 pls_hip_matrix out);
/* pls_hip_model_sse on resident data; R, Q, SSE (M x A) HOST */
PLS_HIP_API int pls_hip_group_model_sse(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A,
                                        const double *R, const double *Q, double *SSE);
/* pls_hip_cv_folds on resident data, E (M x nobs x A) HOST.  Groups of one member only (the fold kernel works on
 * K-sized data of ONE device); PLS_HIP_ERR_UNSUPPORTED otherwise. */
PLS_HIP_API int pls_hip_group_cv_folds(pls_hip_group g, pls_hip_matrix X, pls_hip_matrix Y, int64_t A,
                                       const int64_t *test_idx, int64_t test_size, int64_t num_folds, double *E);

#ifdef __cplusplus
}
#endif
#endif /* PLS_HIP_H */
