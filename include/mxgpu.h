/* mxgpu.h — C-ABI of libmxgpu.so, the MI355X (gfx950) backend for MatrixExtra's
 * CSR hot path.
 *
 * Two layers, both plain C (pointers + sizes, no torch / Rcpp / R types):
 *
 *  (1) mx_*   "export level": one entry point per Rcpp-exported routine of the
 *             reference's hot path (src/RcppExports.cpp CallEntries[] :2233-2242,
 *             :2290-2291, :2297-2298, :2333-2334, :2341-2343).  Arguments are
 *             HOST pointers with exactly the meaning of the reference's
 *             IntegerVector / NumericVector / NumericMatrix views; outputs are
 *             caller-allocated (fixed-size results) or obtained through a
 *             begin/finish pair (variable-size results, so that the caller —
 *             the R .Call shim — can allocate R vectors of the right length
 *             and have the D2H copy land directly in them).  Synchronous at
 *             return, as the reference is (SURVEY §8b "Threading").
 *             Large host buffers are registered with the HIP runtime for the
 *             duration of a call (direct DMA).  R calls from one thread; a host
 *             that calls from several threads at once must not hand the SAME
 *             host buffer to two concurrent calls (one call's unregistration can
 *             pull the range from under the other's copy: the HIP runtime aborts
 *             with "Memobj map does not have ptr").  The sharded path
 *             (mx_set_devices) does share its inputs between its own worker
 *             threads and therefore registers them once, in the calling thread.
 *
 *  (2) mxd_*  "device level": the same operations on DEVICE pointers, enqueued
 *             on a caller-supplied hipStream_t (passed as void*), no
 *             allocation, no synchronisation except where a size must come
 *             back to the host.  This is what bench.py, the multi-GPU path
 *             and layer (1) drive.
 *
 *             Graph capture (tests/test_gpu_graph_capture.py captures one
 *             graph and replays it on fresh inputs): CAPTURABLE after one
 *             uncaptured call on the same stream (the library's grow-only
 *             per-thread scratch then exists): mxd_spmm_plan_run /
 *             _run_rows, mxd_spmm_csr_dense, mxd_spmm_csr_dense_ex2 with
 *             MX_SPMM_ROWWAVE / _SLAB and — with nnz passed — _ROWSPLIT /
 *             _TILE, mxd_spmv_csr_dvec(_ex), mxd_spmv_plan_run, every *_fill,
 *             mxd_values_elemwise, mxd_csr_by_*, mxd_spmv_csr_svec,
 *             mxd_csr_cbind / _rbind_append, mxd_stream_copy,
 *             mxd_exclusive_scan_i32.  NOT capturable (they wait for a value
 *             on the host): every *_count, the *_fused forms, *_plan_create*,
 *             mxd_csr_rows_sorted, mxd_csr_sort_rows, mxd_check_is_seq,
 *             mxd_csr_check_valid, MX_SPMM_AUTO / _PLANNED through
 *             mxd_spmm_csr_dense_ex* (the plan is sized on the host) and
 *             MX_SPMM_ROWSPLIT / _TILE with nnz = -1 (indptr[m] is read back).
 *
 * Every function returns 0 on success, non-zero on failure; the message for the
 * calling thread is read with mx_last_error() (the .Call shim turns it into
 * Rf_error, mirroring BEGIN_RCPP/END_RCPP at src/RcppExports.cpp:16,24).
 *
 * Index type is int32 (R INTSXP) throughout, as in the reference.
 * R's NA_INTEGER / NA_LOGICAL = INT_MIN; NA_REAL = NaN with low word 1954.
 */
#ifndef MXGPU_H
#define MXGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MXGPU_ABI_VERSION 1

/* ---- status / device management ------------------------------------------ */
const char *mx_last_error(void);
int  mx_abi_version(void);
int  mx_device_count(int *count);
int  mx_set_device(int device);
int  mx_device_name(char *buf, size_t buflen);
/* Several GPUs of the node behind the SAME exports: after mx_set_devices(list, n) with n > 1 the large
 * CSR x dense products (mx_tcrossprod_csr_dense_*, mx_matmul_dense_csc_*, mx_tcrossprod_dense_csr_*) cut the rows of the
 * sparse operand into one contiguous range per listed device (mx_partition_rows' balance: 12 bytes up per entry, one result
 * row down per row), run one host thread and three queues per device, and let every device download its rows straight
 * into the caller's result (the result lives on the host: no collective is needed; a device may be listed more than once —
 * its shards then share the GPU).  n = 1: that device becomes the calling thread's current device; n = 0: the current
 * device, unsharded (default).  Everything else stays on the current device. */
int  mx_set_devices(const int *devices, int n);
/* the row cuts that sharding uses: cuts[0 .. nparts], part k = rows [cuts[k], cuts[k+1]) — host arithmetic only */
int  mx_partition_rows(const int32_t *indptr, int nrows, int nparts, int dense_cols, int dense_bytes, int *cuts);
/* SpMM row-sharded over the GPUs of one node WITH the RCCL all-gather of C (csrc/sharded.hip) — BASELINE.json north_star:
 * "Shard SpMM by row-blocks across the 8 GPUs of one node with an RCCL all-gather of C over xGMI".  The exports above return
 * an R matrix in host memory and so need no collective (mx_set_devices); this is the form for a device-resident consumer
 * that keeps multiplying one matrix (the loop of vignettes/Introducing_MatrixExtra.Rmd:452-470 around
 * tcrossprod_csr_dense, src/matmul.cpp:316-375): the CSR is cut ONCE into one row block per listed device (distinct
 * devices; balanced by entries + rows — a block's cost on its device —, or equal_rows = 1: ceil(m / ndev) rows each), every block stays on its
 * device with what AUTO keeps per matrix (profile, sortedness, plan), each product runs on all devices at once — one host
 * thread per device — and ONE in-place ncclAllGather of equal slots (ncclCommInitAll communicators, one process) leaves the
 * full row-major C on EVERY device.  RCCL is dlopen()ed at first use (librccl.so.1; MXGPU_RCCL_LIB overrides).
 * Gathered buffer on a device: ndev slots of slot_rows x n elements, row-major, ld = n; block r = rows [cuts[r], cuts[r+1])
 * of C at the head of slot r (rows of a slot past its block are never written); with equal_rows slot r starts at row
 * r * slot_rows = cuts[r]: a contiguous m x n matrix.  Two buffers alternate: a result stays valid until the product after
 * the next one.  One device listed several times (a one-GPU box): its shards share the device's buffer, nothing is exchanged.
 *   _run      B on the host (K x n row-major = the column-major n x K matrix Y of tcrossprod_csr_dense), uploaded to every
 *             device; returns when every device holds C; C_host (NULL = none): a copy from the first device, row-major
 *             m x n (ldc >= n) or column-major (ldc >= m, transposed on the device);
 *   _run_dev  B_dev[k] = the operand already on shard k's device; flags bit 0: return once everything is queued — the
 *             all-gather of this product then runs under the next product (mx_spmm_sharded_sync waits for all devices);
 *   _result   the last product's gathered buffer on shard k's device;  _layout: the cuts and slot size for a device count
 *             (host arithmetic only). */
typedef struct mx_spmm_sharded mx_spmm_sharded;
int  mx_spmm_sharded_layout(const int32_t *indptr, int m, int ndev, int dense_cols, int dense_bytes, int equal_rows,
                            int *cuts /* ndev + 1 */, int *slot_rows);
int  mx_spmm_sharded_create(const int *devices, int ndev, int m, int K, const int32_t *indptr, const int32_t *indices,
                            const double *values, int equal_rows, mx_spmm_sharded **out);
int  mx_spmm_sharded_info(const mx_spmm_sharded *h, int *nshards, int *slot_rows, int *cuts /* nshards + 1 */, int *uses_rccl,
                          int *rccl_version);
int  mx_spmm_sharded_run(mx_spmm_sharded *h, int n, int dense_dtype, const void *B_host, size_t ldb, void *C_host, size_t ldc,
                         int colmajor_out);
int  mx_spmm_sharded_run_dev(mx_spmm_sharded *h, int n, int dense_dtype, const void *const *B_dev, size_t ldb, int flags);
int  mx_spmm_sharded_sync(mx_spmm_sharded *h);
int  mx_spmm_sharded_result(const mx_spmm_sharded *h, int k, void **C_dev, int *n, int *dense_dtype, int *device);
const char *mx_spmm_sharded_kernel(const mx_spmm_sharded *h, int k);
int  mx_spmm_sharded_destroy(mx_spmm_sharded *h);
/* raw device memory for callers that do not bring their own allocator */
int  mx_dev_malloc(void **dptr, size_t bytes);
int  mx_dev_free(void *dptr);
int  mx_dev_memset(void *dptr, int value, size_t bytes, void *stream);
/* asynchronous copies on `stream`: give them PINNED host memory (mx_host_register, hipHostMalloc).  With ordinary pageable
 * memory the HIP runtime pins the caller's pages on the fly; long randomised runs on the MI355X boxes ended now and then in a
 * GPU memory-access fault at a heap address with such copies in the mix — for ordinary memory use
 * mx_upload / mx_download below (synchronous; what every export-level call uses). */
int  mx_memcpy_h2d(void *dptr, const void *hptr, size_t bytes, void *stream);
int  mx_memcpy_d2h(void *hptr, const void *dptr, size_t bytes, void *stream);
int  mx_stream_sync(void *stream);
/* synchronous copies of ordinary (pageable) caller memory through the library's transfer engine (csrc/xfer.hip: large
 * buffers are registered and moved by one direct DMA — the destination of a download is first-touched by a team of host
 * threads; a pipeline over pinned slots when registration fails).  What the export-level calls use. */
int  mx_upload(void *dptr, const void *hptr, size_t bytes);
int  mx_download(void *hptr, const void *dptr, size_t bytes);
int  mx_host_register(void *hptr, size_t bytes);   /* pin caller memory for async H2D/D2H */
int  mx_host_unregister(void *hptr);

/* Device-side cache of the CSR operands that the export-level calls receive by host address: a later call with the same
 * three vectors (same addresses, nrows, nnz AND the same hash of every byte of their contents) skips the upload.  LRU,
 * capped at max_bytes (default 8 GiB or MXGPU_CSR_CACHE_MB; 0 disables and empties it).  With the default full hash an
 * operand changed in place is simply a miss; with MXGPU_CACHE_FINGERPRINT=sampled (a few 512-byte samples per array) a
 * caller that mutates a cached operand in place must call mx_cache_invalidate (host_ptr = any of the operand's three
 * vectors; NULL = everything). */
/* An entry also carries what has been derived from the operand on the device: the SpMM plan of the whole matrix (built
 * the first time a product that AUTO would plan finds the operand in the cache; every later product, and every block of a
 * pipelined call, skips the build) and, on request, an SpMV plan (option "spmv_planned").  Those arrays are about as large
 * as the CSR itself and are NOT counted against max_bytes; mx_cache_invalidate frees them with the entry.  When a device
 * allocation fails the library empties the cache, frees AUTO's per-thread plan and tries once more.  What a thread keeps
 * beyond that (export scratch for B and C, grow-only) is freed with mxd_release_workspaces() from that thread. */
int  mx_cache_configure(int64_t max_bytes);
int  mx_cache_invalidate(const void *host_ptr);
int  mx_cache_stats(int64_t *bytes, int *entries, int64_t *hits, int64_t *misses);

/* Run-time options (the environment variable in brackets is the default when the option was never set):
 *   "spmv_planned" [MXGPU_SPMV_PLANNED, default 0]  1: mx_matmul_csr_dvec_* on an operand found in the CSR cache builds / uses
 *                  an SpMV plan (mxd_spmv_plan_*: ~2x faster per product, equal to the reference to 1e-12 but not bitwise
 *                  and not bit-reproducible from run to run; the float32 kind rounds once from an f64 sum where the
 *                  reference accumulates in float, src/matmul.cpp:403).  0: every call runs the one-shot kernel — MX_SPMV_FLAT
 *                  for large operands: bit for bit the reference's loop, whatever the cache holds.
 *   "spmv_algo"    [MXGPU_SPMV_ALGO, default 0 = AUTO]  mx_spmv_algo for the one-shot products of the exports
 *   "spmv_planned_calls"  read-only: export-level products served by the planned kernel so far
 *   "pool_idle_bytes" / "pool_idle_blocks" / "pool_hits" / "pool_misses"  read-only: the export level's device blocks
 *                  (operands, results, kept plans) go back to a pool instead of hipFree — capped at MXGPU_POOL_MB, default
 *                  min(32 GiB, 1/8 of the device; 0 = every block is freed at once) — and mxd_release_workspaces() or a
 *                  failed allocation gives them back to the device (csrc/pool.hip: why)
 *   "small_calls"  read-only: export-level calls served by the small path — operands + result within ~2 MiB (SpMM 6 MiB,
 *                  merges 3 MiB: the measured crossovers against the regular path; MXGPU_SMALL_LIMIT_KB scales them) for SpMM,
 *                  SpMV, CSR (+) CSR, X[rows, ]: every input packed into one pinned block, one copy up, the same kernels, one copy
 *                  down, one synchronisation; no pool, no cache (MXGPU_SMALL_CALLS=0 switches it off)
 *   "pool_live_bytes" / "pool_live_blocks"  read-only: blocks handed out and not yet returned (operands in the CSR cache,
 *                  kept plans, thread scratch, results between begin and finish / discard) — a leaked mx_result shows here
 *   "offload_min_len" [MXGPU_OFFLOAD_MIN_LEN, default -1]  the operand length from which mx_should_offload says yes: -1 = the
 *                  measured per-routine defaults, 0 = always, n = that length for every routine
 * mx_last_call_phases: wall-clock phases of the calling thread's last SpMM export as "what;key=value;phase=ms;..."
 * (bench.py reports them for the cold / cached export calls). */
/* The offload gate, for BOTH integration options: should a call of the reference routine `routine`
 * ("tcrossprod_csr_dense_numeric", with or without the "_MatrixExtra_" prefix) whose longest argument vector has
 * `longest_len` elements (the entry count, or the dense operand) go to the GPU?  1 = yes, 0 = keep it on the host: one
 * export call costs 29-50 us whatever it does, 2.8-3.7x MatrixExtra's own routine at the reference's test sizes
 * (tests/testthat/test-matmul.R:108-114; profiles/r04_small_calls.json).  Defaults: products and CSR (+) CSR from 5e4, `X %*% v`
 * from 1e6, `X[rows, ]` / cbind / rbind (memcpys on the host) from 1e7.  Host arithmetic only; never touches the device.
 * The .Call shim (csrc/r_shim.cpp) asks it before every routine and, for a "no", calls MatrixExtra's own routine of the same
 * name when that DLL is loaded (R_FindSymbol); Option A keeps the original body as the host branch. */
int  mx_should_offload(const char *routine, int64_t longest_len);
int  mx_set_option(const char *name, int64_t value);
int  mx_get_option(const char *name, int64_t *value);
int  mx_last_call_phases(char *buf, size_t buflen);

/* ---- element types --------------------------------------------------------- */
typedef enum {
    MX_F64 = 0,      /* double (R numeric)                                   */
    MX_F32 = 1,      /* float  (float32@Data INTSXP reinterpreted)           */
    MX_I32 = 2,      /* R integer, NA_INTEGER = INT_MIN                      */
    MX_LGL = 3,      /* R logical, {0,1,NA_LOGICAL}                          */
    MX_NONE = 4      /* no values (ngRMatrix)                                */
} mx_dtype;

typedef enum {
    MX_OP_ADD = 0,   /* add_csr_elemwise(substract=false)   operators.cpp:539 */
    MX_OP_SUB = 1,   /* add_csr_elemwise(substract=true)                      */
    MX_OP_MUL = 2,   /* multiply_csr_elemwise               operators.cpp:209 */
    MX_OP_OR  = 3,   /* logicalor_csr_elemwise(xor=false)   operators.cpp:556 */
    MX_OP_XOR = 4,   /* logicalor_csr_elemwise(xor=true)                      */
    MX_OP_AND = 5,   /* logicaland_csr_elemwise             operators.cpp:224 */
    MX_OP_FIRST = 6  /* device level only (count / fill pair): union of the patterns, f64 values, the FIRST operand's value
                        where both hold a cell — the last step of the CSR (op) vector NA route, csrc/dvec_na.hip */
} mx_merge_op;

/* ========================================================================== */
/* (2) device level                                                           */
/* ========================================================================== */

/* SpMM  C = A * B,  A CSR m x K (int32 indptr[m+1], indices[nnz], f64 values),
 * B row-major K x n with leading dimension ldb (elements), C m x n.
 *   colmajor_out = 0 : C row-major, leading dim ldc  (gemm_csr_drm_as_drm,
 *                      src/matmul.cpp:118-142; exports start from zeroed C so
 *                      "C += A*B" == "C = A*B")
 *   colmajor_out = 1 : C column-major, leading dim ldc (gemm_csr_drm_as_dcm,
 *                      src/matmul.cpp:150-185)
 * dense_dtype MX_F64 or MX_F32; CSR values are f64 in both (narrowed per
 * nonzero for MX_F32 as at src/matmul.cpp:53-57).  Column indices need not be
 * sorted; duplicates accumulate.  Rows with no entries give zeros. */
int mxd_spmm_csr_dense(int m, int n,
                       const int32_t *indptr, const int32_t *indices, const double *values,
                       const void *B, size_t ldb,
                       void *C, size_t ldc,
                       int dense_dtype, int colmajor_out, void *stream);

/* Same product with an explicit kernel choice.  K = number of rows of B (= ncol A).
 *   algo MX_SPMM_ROWWAVE : one wavefront per row x 1-KiB column chunk (any operands)
 *   algo MX_SPMM_SLAB    : 128-byte column slabs dealt to the XCDs + column panels sized to one XCD's L2,
 *                          accumulators in registers (needs 16-B aligned rows of B; npanels > 1 needs
 *                          rows sorted by column — pass rows_sorted = 1 only when that is known, e.g. from
 *                          mxd_csr_rows_sorted)
 *   algo MX_SPMM_PLANNED : build a plan (see mxd_spmm_plan_create below) and run the planned panel-sweep kernel;
 *                          the plan is rebuilt on every call here — hold a plan yourself to amortise it
 *   algo MX_SPMM_AUTO    : PLANNED when B is larger than an XCD's L2, the operands qualify and there is enough
 *                          work to fill the chip, else ROWWAVE
 * Determinism: ROWWAVE and SLAB add a row's terms in CSR storage order (bitwise equal to a CPU loop with fused multiply-
 * add, and reproducible).  PLANNED adds per-panel partial sums in panel order (a regrouping: equal to 1e-12 for f64);
 * for matrices with very uneven row lengths some rows are shared by several lane groups of the ONE wavefront that owns
 * the rows' octet and folded with LDS atomics.  No other wavefront ever touches those sums, a wavefront's LDS
 * instructions execute in program order and the plan build is deterministic, so what is left open is only the order in
 * which the LDS unit serves lanes of one instruction that hit the same address — fixed on gfx950 as measured: bitwise
 * equal results over 320 runs with plan rebuilds and other kernels in between (tools/repro_probe.py; pinned by
 * tests/test_gpu_fullsize.py::test_planned_kernel_same_bits_run_to_run_on_skewed_rows), but not a promise of the ISA
 * manual.  Callers that want the guarantee by construction pass MX_SPMM_ROWWAVE (the exports: MXGPU_SPMM_ALGO=1).
 * npanels <= 0 / wg_per_cu <= 0 pick defaults.
 *   algo MX_SPMM_ROWSPLIT: short-and-fat products (few rows and / or long rows, narrow B — the reference's published
 *                          dense x CSC workload): one wavefront per row SEGMENT, G = 8 .. 64 lanes per row of B, partial
 *                          sums met in LDS in a fixed order (reproducible; bitwise the storage-order FMA chain when a row
 *                          is one segment, B's rows fill the wavefront and C is row-major, else a reassociation: 1e-12).
 *                          When B outgrows an XCD's L2 the product runs as `npanels` launches over column panels of
 *                          ~2.5 MB of B (rows sorted by column are cut at the panel bounds by a cursor kernel; a row that is
 *                          not sorted is taken whole by the first launch — always correct).  npanels = column panels
 *                          (1 .. 32), wg_per_cu = segments per row (1, 2, 4, 8); 0: chosen from the shape and the mean
 *                          row length.  wg_per_cu = -1: the ROW-GROUP form for many short rows against a narrow B (rows of
 *                          B of at most 512 bytes): the G lanes that own a row of B own one row of A, a wavefront carries
 *                          64 / G rows, every row is summed by its own group in storage order — bit for bit the reference's
 *                          FMA chain in both layouts; what wg_per_cu = 0 picks up to 56 / 40 / 24 entries per row with
 *                          8 / 16 / 32 lanes per row (falls back to one wavefront per row when B's rows are not 16-byte
 *                          aligned or fill the wavefront)
 *   algo MX_SPMM_TILE    : DENSE-ISH sparse operands (a row block of A comes back to every row of B several times: the
 *                          reference's published dense x CSC product at density .05, its test matrices at .4): workgroup =
 *                          row block of up to 300 rows x one 256- / 512-byte column slab; K-tiles of the slab of B (64 KB) are
 *                          brought into LDS by loader wavefronts with LDS-DMA (global_load_lds_dwordx4) while the compute
 *                          wavefronts sum their rows from the previous tile — B leaves L2 once per row block, not once per
 *                          entry.  A row is summed by one 16-lane group in storage order: bit for bit the reference's FMA chain
 *                          in BOTH layouts of C — with ONE exception (round 6): under a matrix profile (_ex3) that shows
 *                          rows several 32-entry windows per tile long, THOSE rows are cut into 2 / 4 / 8 interleaved parts,
 *                          summed side by side and added in order: the same bits on every run, the chain regrouped (1e-14
 *                          in f64); every other row keeps its bits; MXGPU_TILE_SPLIT=0 switches it off, the exports never
 *                          do it.  Rows of uneven length are dealt to the lane groups by length (same bits).  Needs 16-byte aligned rows of B (n a multiple of 2 / 4) and rows sorted by
 *                          column for the LDS sweep: rows_sorted = 1 = the caller vouches for it; otherwise a sortedness pass
 *                          flags every row (one read of the indices) and a row that is not sorted is summed whole from global
 *                          memory — always correct.  wg_per_cu = cpl + 4 * rg + 32 * small (cpl: 1 = 256-byte slabs, 2 = 512;
 *                          rg = rows per lane group 1 .. 5; small = 32 KB tiles with 16-entry windows), npanels = compute
 *                          wavefronts per workgroup (4 .. 15); 0 = chosen from the shape (csrc/spmm_tile.hip tile_geometry,
 *                          tile_est_us).  AUTO picks it when its cost model beats the row-split kernel's and the planned
 *                          sweep's by 10 % (tools/tile_map.py, profiles/r05_tile_map.json).
 * mxd_spmm_csr_dense_ex2: the same with nnz = the number of entries of A (indptr[m] - indptr[0]) when the caller knows it;
 * -1 = unknown (ROWSPLIT then reads indptr[m] from the device: one 4-byte copy and a stream sync). */
typedef enum { MX_SPMM_AUTO = 0, MX_SPMM_ROWWAVE = 1, MX_SPMM_SLAB = 2, MX_SPMM_PLANNED = 3, MX_SPMM_ROWSPLIT = 4,
               MX_SPMM_TILE = 5 } mx_spmm_algo;
int mxd_spmm_csr_dense_ex(int m, int n, int K,
                          const int32_t *indptr, const int32_t *indices, const double *values,
                          const void *B, size_t ldb, void *C, size_t ldc,
                          int dense_dtype, int colmajor_out, int algo, int rows_sorted,
                          int npanels, int wg_per_cu, void *stream);
int mxd_spmm_csr_dense_ex2(int m, int n, int K, int64_t nnz,
                           const int32_t *indptr, const int32_t *indices, const double *values,
                           const void *B, size_t ldb, void *C, size_t ldc,
                           int dense_dtype, int colmajor_out, int algo, int rows_sorted,
                           int npanels, int wg_per_cu, void *stream);

/* Planned SpMM (v3): a device-resident regrouping of A's entries by (row bundle, column panel), wave-interleaved,
 * so that the panel-sweep kernel reads every entry once, coalesced, while B's current slab-panel stays in L2.
 * The plan depends on A and npanels only; build it once per matrix and run it against any number of B.
 * mxd_spmm_plan_create: *plan = NULL creates, a previous plan re-uses its buffers (grow-only); one internal
 * stream sync (the padded size comes back to the host).  npanels <= 0 picks K*128 B / 2.5 MB.
 * mxd_spmm_plan_run: sync_mode 0 = free running, 1 = the waves of a CU's workgroup meet at every panel boundary,
 * 2 = 1 + one timing barrier per generation among the workgroups of an XCD group; -1 = default: 1 for rows of even
 * length, 2 once the lengths of the plan's 64-row octets vary by more than 10 % (mxd_spmm_plan_octet_cv: their
 * coefficient of variation, measured by the build).
 * Needs 16-B aligned rows of B; wg_per_cu is ignored (one 1024-thread workgroup per CU). */
typedef struct mx_spmm_plan mx_spmm_plan;
int mxd_spmm_plan_create(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                         int npanels, void *stream, mx_spmm_plan **plan);
int mxd_spmm_plan_destroy(mx_spmm_plan *plan);
int mxd_spmm_plan_info(const mx_spmm_plan *plan, int *npanels, int64_t *padded_entries);
int mxd_spmm_plan_octet_cv(const mx_spmm_plan *plan, double *cv);
/* longest work item (one octet x one 128-byte slab of B) over its share of the machine, for a B of n columns: above 2.5
 * (f32) / 4.5 (f64) AUTO runs the row-split kernel instead of this plan (rows sorted by length, a few giant rows against a narrow B) */
int mxd_spmm_plan_imbalance(const mx_spmm_plan *plan, int n, int dense_dtype, double *imbalance);
double mxd_spmm_plan_imbalance_limit(int dense_dtype);    /* the limit AUTO applies: 4.5 for f64 products, 2.5 for f32 (round 6) */
int mxd_spmm_plan_run(const mx_spmm_plan *plan, int n, const void *B, size_t ldb, void *C, size_t ldc,
                      int dense_dtype, int colmajor_out, int wg_per_cu, int sync_mode, void *stream);
/* rows [row0, row0 + nrows) of the planned matrix only (row0 a multiple of 64); C points at the block's first row, ldc is the
 * leading dimension of the whole result — what the export pipeline runs per row block once a matrix has a plan */
int mxd_spmm_plan_run_rows(const mx_spmm_plan *plan, int row0, int nrows, int n, const void *B, size_t ldb, void *C,
                           size_t ldc, int dense_dtype, int colmajor_out, int wg_per_cu, int sync_mode, void *stream);
/* What MX_SPMM_AUTO would run for these operands (*algo = MX_SPMM_PLANNED / _SLAB / _ROWWAVE; host arithmetic only), and
 * mxd_spmm_plan_create with AUTO's padding limit: *ready = 0 when the plan would hold more than 1.55x the CSR's entries
 * (very uneven rows) — the plan is then sized but not filled and the caller runs MX_SPMM_ROWWAVE, as AUTO does.  Together:
 * keep AUTO's decision AND the plan across products with one matrix (matrixextra_amd/device.py DeviceCSR, the CSR cache). */
int mxd_spmm_auto_algo(int m, int n, int K, int dense_dtype, const void *B, size_t ldb, const void *C, size_t ldc,
                       int colmajor_out, int *algo);
/* ... with the entry count (-1 = unknown: the rule above) and whether the caller keeps a plan across products: a cost model of
 * the row-split kernel and the planned sweep fitted to a map of 272 shapes (tools/auto_map.py, profiles/r04_auto_map.json;
 * csrc/spmm.hip spmm_auto_cost) chooses between MX_SPMM_ROWSPLIT, _PLANNED and, for one-slab products of very short rows,
 * _SLAB and (round 5) MX_SPMM_TILE for dense-ish operands; products below 2^22 multiply-adds with rows of at most 32 entries (mean)
 * stay on MX_SPMM_ROWWAVE.  keep_plan also says that the caller caches the matrix's sortedness (DeviceCSR, the CSR cache do).
 * mxd_spmm_auto_cost: the model's two estimates. */
int mxd_spmm_auto_algo2(int m, int n, int K, int64_t nnz /* -1 = unknown */, int keep_plan, int dense_dtype, const void *B,
                        size_t ldb, const void *C, size_t ldc, int colmajor_out, int *algo);
int mxd_spmm_auto_cost(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, double *rowsplit_us, double *planned_us,
                       int *panels);
/* ... and the LDS-tile kernel's estimate (round 5) with the slab width it would take (tile_cpl: 1 = 256 bytes, 2 = 512);
 * rows_sorted = 0 adds the sortedness pass the kernel then runs */
int mxd_spmm_auto_cost2(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, int colmajor_out, int rows_sorted,
                        double *rowsplit_us, double *planned_us, double *tile_us, int *panels, int *tile_cpl);
/* What AUTO has to know about a matrix beyond its sizes (round 5; csrc/profile.hip): real dgRMatrix data has power-law columns
 * and skewed row lengths.  mxd_csr_profile fills profile_host[MX_PROFILE_LEN]: [i], i = 0 .. 31 = the share of the entries
 * whose column is among the 2^i most frequent columns (two independent half-samples of <= 2^17 entries each: one ranks the
 * columns, the other measures them), [32] = coefficient of variation of the row lengths, [33] = longest row / mean row,
 * [34] = mean row length, [35] / [36] = entries / number of the rows longer than ~6 mean rows (what sizes the long-rows
 * scratch of the row-split kernel; only a size: a kernel checks on the device that the rows fit), [37] = 1 when [35] / [36]
 * are filled, [39] < 0 = "uniform columns".  ~40 us; one stream synchronisation (160 bytes come back): once per matrix, like
 * mxd_csr_rows_sorted — DeviceCSR and the CSR cache keep it.  workspace: mxd_csr_profile_workspace_bytes(K).
 * mxd_spmm_auto_algo3 / mxd_spmm_auto_cost3: AUTO's choice and estimates with the profile (NULL: uniform columns, equal rows
 * — the assumptions of mxd_spmm_auto_algo2): an XCD's L2 holds the hottest rows of B, so the gather kernels' hit rate is the
 * MASS of those columns, not their share of B's bytes; kernels that walk several rows in lockstep (row groups, the tile
 * kernel) run as long as the longest of their rows. */
#define MX_PROFILE_LEN 40
size_t mxd_csr_profile_workspace_bytes(int K);
int mxd_csr_profile(int m, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices, float *profile_host,
                    void *workspace, void *stream);
/* mxd_spmm_csr_dense_ex3 = _ex2 with the profile: AUTO's choice and the row-split kernel's panel count then follow it */
int mxd_spmm_csr_dense_ex3(int m, int n, int K, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                           const void *B, size_t ldb, void *C, size_t ldc, int dense_dtype, int colmajor_out, int algo,
                           int rows_sorted, int npanels, int wg_per_cu, const float *profile, void *stream);
int mxd_spmm_auto_algo3(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, const void *B, size_t ldb, const void *C,
                        size_t ldc, int colmajor_out, const float *profile, int *algo);
int mxd_spmm_auto_cost3(int m, int n, int K, int64_t nnz, int keep_plan, int dense_dtype, int colmajor_out, int rows_sorted,
                        const float *profile, double *rowsplit_us, double *planned_us, double *tile_us, int *panels, int *tile_cpl);
int mxd_spmm_plan_create_auto(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                              int npanels, void *stream, mx_spmm_plan **plan, int *ready);

/* AUTO's plan and the slab-major copy of B live in grow-only per-thread device buffers between calls; this frees them */
int mxd_release_workspaces(void);

/* HIP-event timing of the dominant kernel of every SpMM launch of this thread (events recorded on the launch stream
 * right around that kernel): enable, run, then read the per-launch milliseconds (the read synchronises). */
int mxd_spmm_kernel_timing(int enable);
int mxd_spmm_kernel_times(float *out_ms, int max_out, int *count);

/* name of the SpMM kernel the last mxd_spmm_csr_dense_ex call of this thread launched (reporting only) */
const char *mxd_spmm_last_kernel(void);

/* SpMV  y = A * v  (matmul_csr_dvec<>, src/matmul.cpp:381-419).
 * v_dtype MX_F64 / MX_I32 / MX_LGL -> y f64[m];  MX_F32 -> y f32[m]
 * (float accumulate).  NA_INTEGER / NA_LOGICAL entries contribute NA_REAL. */
int mxd_spmv_csr_dvec(int m, int64_t nnz /* lanes-per-row hint, -1 = unknown */,
                      const int32_t *indptr, const int32_t *indices, const double *values,
                      const void *v, int v_dtype, void *y, void *stream);
/* Same product with K = length of v (= ncol A), the exact nnz, and a kernel choice:
 *   MX_SPMV_GROUP : G lanes per row, v[j] gathered from L2 (any operands; what mxd_spmv_csr_dvec runs)
 *   MX_SPMV_FLAT  : 256-thread workgroups over equal slices of ~3.8 k entries (16-B loads, all in flight), v[j] gathered
 *                   from L2, rows summed one thread per row in storage order from separately rounded products — bit
 *                   for bit the reference's loop without FMA contraction for rows of up to 256 entries (needs the exact
 *                   nnz >= 4 and 16-B aligned indices / values)
 *   MX_SPMV_TILE  : one 1024-thread workgroup per ~24 k entries, v swept through LDS in 16 k-column panels instead of
 *                   gathered; same summation as FLAT (additionally needs 16-B aligned v and K <= 24 * 16384).  Wins when
 *                   v fits one panel or the rows are very uneven; see DESIGN.md §4.3
 *   MX_SPMV_AUTO  : FLAT when it applies, nnz >= 2^22 and there are at least 32k rows (round 4's map), else GROUP */
typedef enum { MX_SPMV_AUTO = 0, MX_SPMV_GROUP = 1, MX_SPMV_TILE = 2, MX_SPMV_FLAT = 3 } mx_spmv_algo;
int mxd_spmv_csr_dvec_ex(int m, int K, int64_t nnz,
                         const int32_t *indptr, const int32_t *indices, const double *values,
                         const void *v, int v_dtype, void *y, int algo, void *stream);
/* ... with the matrix profile (mxd_csr_profile, MX_PROFILE_LEN floats; NULL = sizes only): AUTO also takes FLAT from 2^20
 * entries on when the longest row has 16k entries or more (a tail for one lane group, nothing special for equal slices) */
int mxd_spmv_csr_dvec_ex2(int m, int K, int64_t nnz,
                          const int32_t *indptr, const int32_t *indices, const double *values,
                          const void *v, int v_dtype, void *y, int algo, const float *profile, void *stream);

/* Planned SpMV for repeated products with the same matrix (csrc/spmv_plan.hip): the plan regroups A's entries by
 * (block of 4096 rows, panel of 6144 columns) so that the kernel can keep the panel of v it needs in LDS instead of
 * gathering v[j] from L2 — the gather, not the (j, a) stream, bounds the one-shot kernels.  Build once per matrix
 * (about the cost of six one-shot products — ~1 ms at cfg3 —; one internal stream sync), run against any number of vectors of any of
 * the four kinds.  Row blocks hold at most 4096 rows and — rows of uneven length — about half a mean block's entries where a
 * block would be heavy (cut on the device).  Up to 64 * 6144 columns the panels of v sit in LDS; WIDER matrices (up to 2^28
 * columns; round 6) are regrouped by super-panels of 2^18 columns and run ONE LAUNCH PER SUPER-PANEL, v read from global memory
 * (every workgroup of a launch from the same 2 MB, which the L2s then hold), y added to from launch to launch — mxd_spmv_plan_run
 * then queues ceil(K / 2^18) launches.  Sums: panels in ascending order, entries of a row inside a panel added with
 * LDS atomics — equal to the reference to 1e-12 (f64) / 1e-5 (float32 kind), not bitwise and not bit-reproducible from
 * run to run; the float32 kind keeps f64 sums and rounds ONCE at the end (once per super-panel for wide matrices), where the
 * reference accumulates in float and rounds per term (src/matmul.cpp:403).  Integer / logical vectors: a row that meets an NA element yields NA_real_, a NaN
 * that comes out of the arithmetic stays an ordinary NaN (as the reference).  The one-shot MX_SPMV_FLAT kernel is the
 * bit-exact path. */
typedef struct mx_spmv_plan mx_spmv_plan;
int mxd_spmv_plan_create(int m, int K, const int32_t *indptr, const int32_t *indices, const double *values,
                         void *stream, mx_spmv_plan **plan);
int mxd_spmv_plan_info(const mx_spmv_plan *plan, int *npanels, int64_t *padded_entries);
int mxd_spmv_plan_run(const mx_spmv_plan *plan, const void *v, int v_dtype, void *y, void *stream);
int mxd_spmv_plan_destroy(mx_spmv_plan *plan);

/* diagnostic (tools/spmv_stamps.py): a device buffer of 8 x ceil(nnz / 23552) uint64 makes MX_SPMV_TILE run its stamped
 * build, which records the shader clock at its phase boundaries per workgroup; NULL switches back */
int mxd_debug_spmv_tile_stamps(void *stamps_dev);
/* likewise for the LDS-tile SpMM kernel (tools/tile_stamps.py): 2 x 16 x (workgroups) uint64 — per compute wavefront the cycles
 * spent at the tile barriers and in its whole sweep */
int mxd_debug_rowsplit_long_rows(long long *rows, long long *pieces);   /* last row-split product of this thread: rows / pieces
                                                                           handed to the long-rows path (0 / 0 = off); syncs */
int mxd_debug_spmm_tile_stamps(void *stamps_dev);
/* how the process's last LDS-tile product laid out its rows: 0 = consecutive rows, 1 = dealt to the lane groups by length,
 * | 2 = long rows cut into parts (those rows' sums regrouped), | 4 = row blocks made in one pass (csrc/spmm_tile.hip) */
int mxd_debug_spmm_tile_mode(void);
/* what the fit check found before that product: the rows longer than the piece, their pieces, and whether they fitted the
 * scratch sized from the profile's hint (fit 0: the product kernels summed them in line — slower, same answers); syncs */
int mxd_debug_rowsplit_long_fit(long long *rows_needed, long long *pieces_needed, int *fit);
/* the kernel family AUTO chose for the calling thread's last pipelined / sharded export product and the geometry every block
 * of it ran with, whatever thread ran the block (segments -1 = row groups; long_piece 0 = long-rows path off) */
int mx_debug_last_export_family(int *family, int *segments, int *panels, int *long_piece);

/* CSR (+) CSR, pass 1: per-row output lengths (union for ADD/SUB/OR/XOR,
 * intersection for MUL/AND) then exclusive scan into out_indptr[m+1].
 * Rows must be sorted ascending with unique column ids (precondition the R
 * callers establish, R/operators.R:58,64,748,754).  workspace: mxd_merge_workspace_bytes(m).
 * *nnz_out_host (pinned or pageable host int64) is written after an internal
 * stream sync — the one host round trip of the operation.  nnz1 / nnz2 only
 * steer the lanes-per-row choice (pass -1 when unknown). */
/* hint for the device-level merges of the calling thread (sticky): the operands' rows are of uneven length (coefficient of variation
 * above ~0.3) — the lane groups then take one width up (round 6: 14-15 % on log-normal rows; 10 % slower on rows of equal length,
 * hence not the default).  The export level reads the host row pointers and needs no hint. */
int mxd_csr_merge_rows_uneven(int on);
size_t mxd_merge_workspace_bytes(int m);
int mxd_csr_merge_count(int op, int m,
                        const int32_t *indptr1, const int32_t *indices1, int64_t nnz1,
                        const int32_t *indptr2, const int32_t *indices2, int64_t nnz2,
                        int32_t *out_indptr, void *workspace,
                        int64_t *nnz_out_host, void *stream);
/* pass 2: fill out_indices / out_values (f64 for ADD/SUB/MUL, int32 R logical
 * for OR/XOR/AND) at the offsets in out_indptr. */
int mxd_csr_merge_fill(int op, int m,
                       const int32_t *indptr1, const int32_t *indices1, const void *values1, int64_t nnz1,
                       const int32_t *indptr2, const int32_t *indices2, const void *values2, int64_t nnz2,
                       const int32_t *out_indptr, int32_t *out_indices, void *out_values,
                       void *stream);
/* The same merge in ONE pass over the inputs — OPT-IN (the exports use it only with MXGPU_MERGE_FUSED=1; bench.py and the
 * exports default to the count -> fill pair above, which is faster at cfg4: 1.07-1.14 ms against 1.40-1.6 ms, DESIGN.md
 * §4.4): every workgroup sizes its tile of rows from the registers it has just loaded, finds its place in the output with
 * a decoupled look-back over the tiles' totals, and places the entries.  out_indices / out_values must hold the UPPER BOUND of the result — nnz1 + nnz2
 * entries for ADD / SUB / OR / XOR, min(nnz1, nnz2) for MUL / AND, like the reference's scratch arrays
 * (operators.cpp:402-406, :139-143); the bound must fit int32 (else use the count -> fill pair).
 * workspace: mxd_merge_fused_workspace_bytes(m).  *nnz_out_host as above. */
size_t mxd_merge_fused_workspace_bytes(int m);
int mxd_csr_merge_fused(int op, int m,
                        const int32_t *indptr1, const int32_t *indices1, const void *values1, int64_t nnz1,
                        const int32_t *indptr2, const int32_t *indices2, const void *values2, int64_t nnz2,
                        int32_t *out_indptr, int32_t *out_indices, void *out_values,
                        void *workspace, int64_t *nnz_out_host, void *stream);
/* identical-pattern fast path (operators.cpp:104-132, :343-395): values only */
int mxd_values_elemwise(int op, int64_t nnz, const void *values1, const void *values2,
                        void *out_values, void *stream);

/* Row gather  out = A[rows_take, :]  (copy_csr_rows_template, src/slice.cpp:225-274)
 * pass 1: new_indptr[r+1] + total;  pass 2: copy.  value_dtype MX_F64 / MX_LGL / MX_NONE. */
size_t mxd_gather_workspace_bytes(int r);
int mxd_csr_gather_count(int r, const int32_t *indptr, const int32_t *rows_take,
                         int32_t *new_indptr, void *workspace,
                         int64_t *nnz_out_host, void *stream);
int mxd_csr_gather_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                        const int32_t *rows_take, const int32_t *new_indptr,
                        int32_t *new_indices, void *new_values, int value_dtype,
                        int64_t nnz_out /* lanes-per-row hint, -1 = unknown */, void *stream);

/* The same gather in ONE launch, for callers that can size the outputs from an estimate: new_indices / new_values hold
 * `capacity` entries.  new_indptr[r+1] and *nnz_out_host (read back once, behind all the work) are always exact; when
 * *nnz_out_host > capacity the entries beyond the capacity were not copied — allocate exactly and call mxd_csr_gather_fill
 * with the new_indptr this call produced.  avg_row_len (mean entries of a source row, <= 0 unknown) picks the lanes per row.
 * `workspace` is unused (may be NULL): the look-back state is kept by the library per thread and device; its words carry a
 * launch generation, so nothing is cleared between calls.  A thread may use any stream and switch streams between calls:
 * what the library keeps per thread (this state, the SpMV slice table, the row-split kernel's cursors, AUTO's plan, the
 * packed copy of B) is handed from one stream to the next through an event recorded behind its last user.  The size reaches
 * the host through a pinned word the kernel writes as soon as the last tile knows it: the call returns while the copies may
 * still be running on `stream` (stream order covers every later use of the outputs on the device). */
int mxd_csr_gather_fused(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                         const int32_t *rows_take, int32_t *new_indptr, int32_t *new_indices, void *new_values,
                         int value_dtype, int64_t capacity, double avg_row_len, void *workspace,
                         int64_t *nnz_out_host, void *stream);

/* Column-filtering slices.  Same count -> scan -> fill shape as the row gather; workspace of
 * mxd_gather_workspace_bytes(r).  avg_row_len (mean entries per source row) only steers the lanes-per-row choice.
 *   colrange: keep min_col <= col <= max_col, re-based to min_col, input order kept; values come out as f64
 *             whatever the input kind (copy_csr_rows_col_seq_template, src/slice.cpp:326-383)
 *   colmap:   arbitrary selector through a dense map built by mxd_colmap_build: start[ncol_map+1], pos[n] with
 *             pos[start[c] .. start[c+1]) = ascending positions of column c in cols_take
 *             (copy_csr_arbitrary_template, src/slice.cpp:449-578; re-order rows afterwards with mxd_csr_sort_rows
 *             unless cols_take is non-decreasing) */
int mxd_csr_colrange_count(int r, const int32_t *indptr, const int32_t *indices, const int32_t *rows_take,
                           int min_col, int max_col, double avg_row_len, int32_t *new_indptr,
                           void *workspace, int64_t *nnz_out_host, void *stream);
int mxd_csr_colrange_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                          int value_dtype, const int32_t *rows_take, int min_col, int max_col,
                          double avg_row_len, const int32_t *new_indptr, int32_t *new_indices,
                          double *new_values, void *stream);
size_t mxd_colmap_workspace_bytes(int ncol_map);
int mxd_colmap_build(const int32_t *cols_take, int64_t n, int ncol_map, int32_t *start, int32_t *pos,
                     void *workspace, void *stream);
int mxd_csr_colmap_count(int r, const int32_t *indptr, const int32_t *indices, const int32_t *rows_take,
                         int ncol_map, const int32_t *start, double avg_row_len, int32_t *new_indptr,
                         void *workspace, int64_t *nnz_out_host, void *stream);
int mxd_csr_colmap_fill(int r, const int32_t *indptr, const int32_t *indices, const void *values,
                        int value_dtype, const int32_t *rows_take, int ncol_map, const int32_t *start,
                        const int32_t *pos, double avg_row_len, const int32_t *new_indptr,
                        int32_t *new_indices, void *new_values, void *stream);
/* col -> ncol-1-col and each row reversed, in place (reverse_columns_inplace, src/slice.cpp:142-170) */
int mxd_csr_reverse_columns(int m, int64_t nnz, const int32_t *indptr, int32_t *indices, void *values,
                            int value_dtype, int ncol, void *stream);
int mxd_reversed_iota(int n, int32_t *out, void *stream);   /* out[i] = n-1-i: row list of reverse_rows */

/* cbind / rbind.
 * cbind: out row r = X row r followed by Y row r (Y's column ids already shifted by ncol X), rows past the end of
 * the shorter operand come from the longer one only (cbind_csr<>, src/cbind.cpp:4-99).  Outputs hold
 * max(nX, nY)+1 / nnzX+nnzY entries; no scan needed (offsets are Xp[r] + Yp[r]). */
int mxd_csr_cbind(int nX, int nY, const int32_t *Xp, const int32_t *Xj, const void *Xx, const int32_t *Yp,
                  const int32_t *Yj_plus_ncol, const void *Yx, int value_dtype, int64_t nnz_total,
                  int32_t *indptr, int32_t *indices, void *values, void *stream);
/* rbind: append one operand at (row_offset, entry_offset) with the value conversions of concat_csr_batch
 * (src/rbind.cpp:24-173).  in_kind 0 dgR, 1 lgR, 2 ngR, 3/4/5/6 d/i/l/n sparseVector (1-based indices, one row);
 * out_kind 0 dgR, 1 lgR, 2 ngR. */
int mxd_csr_rbind_append(int in_kind, const int32_t *indptr_in, const int32_t *indices_in, const void *values_in,
                         int nrows_in, int64_t nnz_in, int out_kind, int row_offset, int64_t entry_offset,
                         int32_t *out_indptr, int32_t *out_indices, void *out_values, void *stream);

/* SURVEY §8f rank 4.
 * CSR x sparse vector (matmul_csr_svec<>, src/matmul.cpp:486-641): y given as sorted 1-based indices + values;
 *   kind 0 numeric (f64), 1 integer, 2 logical, 3 binary (no values), 4 float32; out f64[m].
 * CSR (.) dense (multiply_csr_by_dense_elemwise<>, src/operators.cpp:239-334): dense column-major m x ncol;
 *   kind 0 double, 1 float32, 2 integer, 3 logical (f64 values in/out), 4 logical AND (int32 values in/out). */
int mxd_spmv_csr_svec(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const double *values,
                      const int32_t *y_indices_base1, int ny, const void *y_values, int kind, double *out,
                      void *stream);
int mxd_csr_by_dense_elemwise(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                              const void *values, const void *dense_colmajor, int kind, void *values_out,
                              void *stream);

/* CSR (op) dense vector with R's recycling (multiply_csr_by_dvec_no_NAs<>, src/operators.cpp:1604-2140): a
 * values-only transform, out[k] = values[k] op dvec[(row + col*m) mod dvec_len] (`recyle_pos`, :1478; the reference's
 * four length branches :1640,1773,1870,2033 all reduce to it).  op: R's * ^ / %% %/% on f64 values with the sparse
 * matrix on the left (x_is_lhs) or right, or R's 3-valued & on int32 logicals (MX_DV_LOGICAL_AND; values, dvec and out
 * int32).  ^ %% %/% follow R_pow / R_modulus / R_intdiv (:1482-1601).  nnz < 0 = unknown (launch shape only). */
typedef enum { MX_DV_MULTIPLY = 0, MX_DV_POWERTO = 1, MX_DV_DIVIDE = 2, MX_DV_DIVREST = 3, MX_DV_INTDIV = 4,
               MX_DV_LOGICAL_AND = 5 } mx_dvec_op;
int mxd_csr_by_dvec(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                    const void *values, const void *dvec, int64_t dvec_len, int op, int x_is_lhs,
                    void *values_out, void *stream);

/* The structure-changing route of CSR (op) dense vector (multiply_csr_by_dvec_with_NAs, src/operators.cpp:2258-2856): the
 * vector holds NA / NaN, zeros under / %% %/% ^, negatives under ^ or infinities under *, and every cell it makes special
 * becomes an explicit entry.  op MX_DV_MULTIPLY .. MX_DV_INTDIV, always `value op element`; rows sorted by column (the R
 * caller sorts first).  The result arrays are allocated by the library (free them with mx_dev_free): *out_indptr[m+1],
 * *out_indices / *out_values[*nnz_out].  *structure_unchanged = 1: no cell was added (lengths that neither divide nrows nor
 * cover the matrix only: the reference then returns its INPUT indptr / indices) — only *out_values[nnz] is set.
 * Synchronises `stream` (three sizes come back to the host).  See csrc/dvec_na.hip for how the result is put together
 * (values kernel + a CSR of the special cells + the MX_OP_FIRST union merge) and for the reference's quirks that are kept. */
int mxd_csr_by_dvec_with_NAs(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                             const double *values, const double *dvec, int64_t dvec_len, int op,
                             int32_t **out_indptr, int32_t **out_indices, double **out_values,
                             int64_t *nnz_out, int *structure_unchanged, void *stream);

/* check_is_seq / check_is_rev_seq (src/slice.cpp:25-47) on a device vector.
 * *flag_host receives 0/1 after an internal stream sync. */
int mxd_check_is_seq(const int32_t *idx, int64_t n, int reversed, int32_t *workspace4,
                     int *flag_host, void *stream);

/* STREAM-copy probe: dst[0..bytes) = src[0..bytes) with 16-byte loads / nontemporal stores (pointers and size multiples of
 * 16).  bench.py times it in-run: rooflines are quoted against the nominal 8 TB/s AND against this kernel's rate
 * (2 * bytes / time) on the box at hand. */
int mxd_stream_copy(void *dst, const void *src, size_t bytes, void *stream);

/* exclusive scan of int32 counts[n] -> out[n+1] (out[n] = total); total also
 * returned as int64 in *total_dev (device int64).  workspace: mxd_scan_workspace_bytes(n). */
size_t mxd_scan_workspace_bytes(int64_t n);
int mxd_exclusive_scan_i32(const int32_t *counts, int64_t n, int32_t *out,
                           int64_t *total_dev, void *workspace, void *stream);

/* What follows a merge and what precedes it in the R callers (widened after SURVEY §8f's four ranks):
 *   remove_zero_valued_csr<>  src/misc.cpp:553-664 (R `remove_zeros`, R/utils.R:286-312): the explicit zeros A - B leaves
 *     behind (operators.cpp:477-495) leave the matrix, with remove_NAs the missing values too.  value_dtype MX_F64 or
 *     MX_LGL.  Predicates exactly the reference's: zeros are +-0.0 / FALSE, NaN and NA are "true" and stay unless
 *     remove_NAs; for R logicals with remove_NAs only the NAs leave — the zeros stay (misc.cpp:636-647), kept as it is.
 *     count: out_indptr[m+1], *nnz_out_host, *dirty_host = is there anything the reference's first scan would have found
 *     (misc.cpp:562-584; 0: it returns its INPUT vectors); fill: ordered compaction.  `nnz` (< 0 unknown) only picks the
 *     lanes per row.  workspace: mxd_csr_drop_workspace_bytes(m).
 *   check_valid_csr_matrix  src/misc.cpp:970-1016 (R `check_sparse_matrix`, R/utils.R:448): *code_host = 0 valid,
 *     1 "Matrix has negative indices.", 2 "Matrix has invalid column indices.", 4 "Matrix has missing values in the index
 *     pointer.", 5 "Matrix index pointer is not monotonicaly increasing." — the reference's order of checks (its third one,
 *     NA among the indices, cannot fire: NA_INTEGER is negative).  nnz = length of `indices`; flags_dev: 4 ints of scratch. */
size_t mxd_csr_drop_workspace_bytes(int m);
int mxd_csr_drop_count(int m, int64_t nnz, const int32_t *indptr, const void *values, int value_dtype, int remove_NAs,
                       int32_t *out_indptr, void *workspace, int64_t *nnz_out_host, int *dirty_host, void *stream);
int mxd_csr_drop_fill(int m, int64_t nnz, const int32_t *indptr, const int32_t *indices, const void *values, int value_dtype,
                      int remove_NAs, const int32_t *out_indptr, int32_t *out_indices, void *out_values, void *stream);
int mxd_csr_check_valid(int m, int ncols, int64_t nnz, const int32_t *indptr, const int32_t *indices, int *flags_dev,
                        int *code_host, void *stream);

/* Next-row components: per-row sortedness check and
 * per-row index sort (check_is_sorted / sort_sparse_indices_known_ncol,
 * src/misc.cpp:118-128, :261-298). */
/* indptr[0] may be > 0 (a row-block view of a larger CSR that keeps absolute offsets into `indices`): only the entries
 * [indptr[0], indptr[m]) are looked at.  One pass over the indices + one over indptr, nothing read twice. */
int mxd_csr_rows_sorted(int m, const int32_t *indptr, const int32_t *indices,
                        int32_t *workspace4, int *flag_host, void *stream);
/* sorts every row by column id (stable), in place; tmp_indices / tmp_values are
 * caller scratch of nnz entries each (tmp_values unused for MX_NONE). */
int mxd_csr_sort_rows(int m, int64_t nnz, const int32_t *indptr, int32_t *indices, void *values,
                      int value_dtype, int32_t *tmp_indices, void *tmp_values, void *stream);

/* ========================================================================== */
/* (1) export level — host pointers, names follow the Rcpp exports            */
/* ========================================================================== */

/* tcrossprod_csr_dense_numeric  src/matmul.cpp:345-359  (RcppExports.cpp:547)
 * X CSR with nrows_X rows; Y_colmajor is nrow_Y x ncol_Y column-major (so it
 * is ncol_Y x nrow_Y row-major == B of the SpMM); out is nrows_X x nrow_Y
 * column-major, fully written.  nthreads is accepted and ignored. */
int mx_tcrossprod_csr_dense_numeric(const int32_t *X_indptr, const int32_t *X_indices,
                                    const double *X_values, int nrows_X,
                                    const double *Y_colmajor, int nrow_Y, int ncol_Y,
                                    int nthreads, double *out_colmajor);
/* tcrossprod_csr_dense_float32  src/matmul.cpp:361-375  (RcppExports.cpp:561) */
int mx_tcrossprod_csr_dense_float32(const int32_t *X_indptr, const int32_t *X_indices,
                                    const double *X_values, int nrows_X,
                                    const float *Y_colmajor, int nrow_Y, int ncol_Y,
                                    int nthreads, float *out_colmajor);
/* matmul_dense_csc_numeric  src/matmul.cpp:221-235  (RcppExports.cpp:489)
 * X_colmajor nrows_X x ncols_X; Y CSC with ncols_Y columns; out nrows_X x ncols_Y col-major. */
int mx_matmul_dense_csc_numeric(const double *X_colmajor, int nrows_X, int ncols_X,
                                const int32_t *Y_indptr, const int32_t *Y_indices,
                                const double *Y_values, int ncols_Y,
                                int nthreads, double *out_colmajor);
int mx_matmul_dense_csc_float32(const float *X_colmajor, int nrows_X, int ncols_X,
                                const int32_t *Y_indptr, const int32_t *Y_indices,
                                const double *Y_values, int ncols_Y,
                                int nthreads, float *out_colmajor);
/* tcrossprod_dense_csr_numeric  src/matmul.cpp:283-297  (RcppExports.cpp:517)
 * out is nrows_X x nrows_Y col-major; ncols_Y is unused by the reference too. */
int mx_tcrossprod_dense_csr_numeric(const double *X_colmajor, int nrows_X, int ncols_X,
                                    const int32_t *Y_indptr, const int32_t *Y_indices,
                                    const double *Y_values, int nrows_Y,
                                    int nthreads, int ncols_Y, double *out_colmajor);
int mx_tcrossprod_dense_csr_float32(const float *X_colmajor, int nrows_X, int ncols_X,
                                    const int32_t *Y_indptr, const int32_t *Y_indices,
                                    const double *Y_values, int nrows_Y,
                                    int nthreads, int ncols_Y, float *out_colmajor);

/* matmul_csr_dvec_{numeric,integer,logical,float32}  src/matmul.cpp:421-483
 * (RcppExports.cpp:575,589,603,617).  len_y = ncol(X). */
int mx_matmul_csr_dvec_numeric(const int32_t *X_indptr, const int32_t *X_indices,
                               const double *X_values, int nrows_X,
                               const double *y_dense, int len_y, int nthreads, double *out);
int mx_matmul_csr_dvec_integer(const int32_t *X_indptr, const int32_t *X_indices,
                               const double *X_values, int nrows_X,
                               const int32_t *y_dense, int len_y, int nthreads, double *out);
int mx_matmul_csr_dvec_logical(const int32_t *X_indptr, const int32_t *X_indices,
                               const double *X_values, int nrows_X,
                               const int32_t *y_dense, int len_y, int nthreads, double *out);
int mx_matmul_csr_dvec_float32(const int32_t *X_indptr, const int32_t *X_indices,
                               const double *X_values, int nrows_X,
                               const float *y_dense, int len_y, int nthreads, float *out);

/* Variable-size results: begin computes on the device and reports the sizes,
 * finish copies into caller-allocated vectors and releases the handle
 * (mx_result_discard releases without copying). */
typedef struct mx_result mx_result;
typedef struct {
    int64_t indptr_len;    /* length of the indptr vector to allocate           */
    int64_t nnz;           /* length of indices                                 */
    int64_t values_len;    /* length of values (nnz, or 0 when there are none)  */
    int     values_dtype;  /* MX_F64 / MX_LGL / MX_NONE                         */
    int     alias_structure; /* 1: reference returns the INPUT indptr1/indices1
                                objects themselves (operators.cpp:127-131,:390-394);
                                finish then fills only values.  2 (remove_zero_valued_csr
                                only): all three input vectors come back            */
} mx_result_info;

/* add_csr_elemwise / logicalor_csr_elemwise / multiply_csr_elemwise /
 * logicaland_csr_elemwise  (RcppExports.cpp:1287,1303,1192,1207).
 * op selects the export and its flag (substract / xor_op). nrows = len(indptr)-1. */
int mx_csr_elemwise_begin(int op, int nrows,
                          const int32_t *indptr1, const int32_t *indptr2,
                          const int32_t *indices1, const int32_t *indices2,
                          const void *values1, const void *values2,
                          int64_t nnz1, int64_t nnz2,
                          mx_result **res, mx_result_info *info);
/* copy_csr_rows_{numeric,logical,binary}  src/slice.cpp:276-324 (RcppExports.cpp:1886,1899,1912)
 * value_dtype MX_F64 / MX_LGL / MX_NONE; n_values = length of the values vector
 * (0 => no values copied, slice.cpp:246,257). */
int mx_copy_csr_rows_begin(const int32_t *indptr, int nrows,
                           const int32_t *indices, const void *values, int value_dtype,
                           int64_t n_values,
                           const int32_t *rows_take, int64_t n_take,
                           mx_result **res, mx_result_info *info);
/* copy_csr_rows_col_seq_{numeric,logical,binary}  src/slice.cpp:385-447 (RcppExports.cpp:1924,1939,1954):
 * cols_take is only used for its min / max (minus index1), as in the reference.  Result values are f64. */
int mx_copy_csr_rows_col_seq_begin(const int32_t *indptr, int nrows,
                                   const int32_t *indices, const void *values, int value_dtype, int64_t n_values,
                                   const int32_t *rows_take, int64_t n_take,
                                   const int32_t *cols_take, int64_t n_cols_take, int index1,
                                   mx_result **res, mx_result_info *info);
/* copy_csr_arbitrary_{numeric,logical,binary}  src/slice.cpp:580-634: rows_take, cols_take 0-based */
int mx_copy_csr_arbitrary_begin(const int32_t *indptr, int nrows,
                                const int32_t *indices, const void *values, int value_dtype, int64_t n_values,
                                const int32_t *rows_take, int64_t n_take,
                                const int32_t *cols_take, int64_t n_cols_take,
                                mx_result **res, mx_result_info *info);
/* reverse_rows_{numeric,logical,binary}  src/slice.cpp:98-140 */
int mx_reverse_rows_begin(const int32_t *indptr, int nrows, const int32_t *indices, const void *values,
                          int value_dtype, int64_t n_values, mx_result **res, mx_result_info *info);
/* reverse_columns_inplace_{numeric,logical,binary}  src/slice.cpp:172-221: modifies indices / values */
int mx_reverse_columns_inplace(const int32_t *indptr, int nrows, int32_t *indices, void *values,
                               int value_dtype, int64_t n_values, int ncol);
/* matmul_csr_svec_{numeric,integer,logical,binary,float32}  src/matmul.cpp:555-641 (kind as mxd_spmv_csr_svec) */
int mx_matmul_csr_svec(const int32_t *X_indptr, const int32_t *X_indices, const double *X_values, int nrows_X,
                       const int32_t *y_indices_base1, int64_t ny, const void *y_values, int kind, int nthreads,
                       double *out);
/* matmul_rowvec_by_csc / matmul_rowvec_by_cscbin  src/matmul.cpp:643-684 (float32 row vector %*% CsparseMatrix,
 * R/matmul.R:243-259,350-365,411-425): out[ncols_Y] float32, accumulated in float like the reference; values NULL = the
 * pattern kind (every stored entry counts as 1). */
int mx_matmul_rowvec_by_csc(const float *rowvec, int len, const int32_t *indptr, const int32_t *indices, const double *values,
                            int ncols_Y, float *out);
/* multiply_csr_by_dense_elemwise_{double,float32,int,bool} + logicaland_csr_by_dense_cpp  src/operators.cpp:289-334:
 * dense_mat column-major nrows x ncols; values_out has nnz entries (f64, or int32 for kind 4). */
int mx_multiply_csr_by_dense_elemwise(const int32_t *indptr, const int32_t *indices, const void *values, int nrows,
                                      const void *dense_mat, int64_t ncols, int kind, void *values_out);
/* multiply_csr_by_dvec_no_NAs_numeric  src/operators.cpp:2142-2175: exactly one of the five flags is set (as the R
 * caller passes them, R/operators.R:1134-1137); values_out f64[nnz]. */
int mx_multiply_csr_by_dvec_no_NAs_numeric(const int32_t *indptr, const int32_t *indices, const double *values,
                                           int nrows, const double *dvec, int64_t dvec_len, int ncols, int multiply,
                                           int powerto, int divide, int divrest, int intdiv, int X_is_LHS,
                                           double *values_out);
/* multiply_csr_by_dvec_with_NAs  src/operators.cpp:2258-2856 (RcppExports.cpp CallEntries: 11 arguments): the
 * structure-changing route, same flags.  info.alias_structure = 1 when the reference would hand back its input indptr /
 * indices (finish then fills only the values); errors as the reference's ("Unexpected error." for ^ / %% with the matrix
 * on the right, the int-overflow message). */
int mx_multiply_csr_by_dvec_with_NAs_begin(const int32_t *indptr, const int32_t *indices, const double *values, int nrows,
                                           const double *dvec, int64_t dvec_len, int ncols, int multiply, int powerto,
                                           int divide, int divrest, int intdiv, int X_is_LHS,
                                           mx_result **res, mx_result_info *info);
/* logicaland_csr_by_dvec_internal  src/operators.cpp:2177-2200: R logicals (int32), values_out int32[nnz] */
int mx_logicaland_csr_by_dvec_internal(const int32_t *indptr, const int32_t *indices, const int32_t *values,
                                       int nrows, const int32_t *dvec, int64_t dvec_len, int ncols,
                                       int32_t *values_out);
/* cbind_csr_{numeric,logical,binary}  src/cbind.cpp:101-157 (value_dtype MX_F64 / MX_LGL / MX_NONE) */
int mx_cbind_csr_begin(const int32_t *X_indptr, int nrows_X, const int32_t *X_indices, const void *X_values,
                       int64_t n_values_X, const int32_t *Y_indptr, int nrows_Y,
                       const int32_t *Y_indices_plus_ncol, const void *Y_values, int64_t n_values_Y,
                       int value_dtype, mx_result **res, mx_result_info *info);
/* concat_csr_batch  src/rbind.cpp:24-173 over plain arrays instead of S4 objects */
typedef struct {
    int kind;                  /* 0 dgR, 1 lgR, 2 ngR, 3 dsparseVector, 4 isparseVector, 5 lsparseVector, 6 nsparseVector */
    const int32_t *indptr;     /* matrices only */
    const int32_t *indices;    /* @j (0-based) or @i of a sparse vector (1-based) */
    const void *values;        /* f64 / int32 / NULL */
    int nrows;                 /* matrices: Dim[1]; vectors: ignored (one row) */
    int64_t nnz;
} mx_rbind_input;
int mx_concat_csr_batch_begin(const mx_rbind_input *objects, int n_inputs, int out_kind,
                              mx_result **res, mx_result_info *info);
/* remove_zero_valued_csr_{numeric,logical}  src/misc.cpp:667-698 (RcppExports.cpp CallEntries: 4 arguments each).
 * info.alias_structure = 2 when nothing has to leave: the reference returns its three INPUT vectors
 * (misc.cpp:586-590) — discard the handle and do the same. */
int mx_remove_zero_valued_csr_begin(const int32_t *indptr, const int32_t *indices, const void *values, int value_dtype,
                                    int nrows, int remove_NAs, mx_result **res, mx_result_info *info);
/* check_valid_csr_matrix  src/misc.cpp:970-1016: *code as mxd_csr_check_valid; message = the reference's text for it
 * ("" when valid), a static string. */
int mx_check_valid_csr_matrix(const int32_t *indptr, const int32_t *indices, int64_t n_indices, int nrows, int ncols,
                              int *code, const char **message);
int mx_result_finish(mx_result *res, int32_t *out_indptr, int32_t *out_indices, void *out_values);
int mx_result_discard(mx_result *res);

/* check_is_seq / check_is_rev_seq  src/slice.cpp:25-47 (RcppExports.cpp:1795,1805) */
int mx_check_is_seq(const int32_t *indices, int64_t n, int *result);
int mx_check_is_rev_seq(const int32_t *indices, int64_t n, int *result);

/* §8(f)-1: sort_sparse_indices (R/utils.R:22-161 -> src/misc.cpp:261-298,333-347) and
 * check_is_sorted (src/misc.cpp:118-128) for a CSR held in host memory; sorts in place. */
int mx_check_indices_are_sorted(const int32_t *indptr, const int32_t *indices, int nrows, int *result);
int mx_sort_sparse_indices(const int32_t *indptr, int32_t *indices, void *values,
                           int value_dtype, int nrows);

#ifdef __cplusplus
}
#endif
#endif /* MXGPU_H */
