/*
 * aesara_hip.h — C-ABI of libaesara_hip.so, the MI355X (gfx950) execution shim behind the
 * Aesara HIP linker.  Plain C linkage: pointers, sizes and integer codes only (no C++ or
 * torch types), loadable with ctypes/cffi/dlopen.
 *
 * Every entry point cites the reference code path it replaces (paths relative to the
 * reference tree, aesara-devs/aesara @ 2024-10-08).
 *
 * Conventions
 *   - return value: 0 = ok, <0 = error (AHIP_E*); ahip_last_error() returns a thread-local
 *     message for the last non-zero return on this thread.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Nothing here
 *     synchronises implicitly except where documented.
 *   - memory ownership: every device buffer is owned by the caller (PyTorch-ROCm allocations
 *     in the Python host); the shim never allocates or frees user-visible device memory.
 *     Workspaces are caller-provided.
 *   - strides are in ELEMENTS (not bytes); stride 0 == broadcast along that dim.
 *   - dtype codes: enum ahip_dtype (NumPy names).
 */
#ifndef AESARA_HIP_H
#define AESARA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AHIP_ABI_VERSION 1
#define AHIP_MAXD 6    /* max collapsed dims handed to a generated kernel            */
#define AHIP_MAXOPS 32 /* max operands (inputs + outputs) of one fused Elemwise kernel */

enum ahip_status {
  AHIP_OK = 0,
  AHIP_EINVAL = -1,   /* bad argument (-> ValueError)                    */
  AHIP_EHIP = -2,     /* HIP runtime error (-> RuntimeError)             */
  AHIP_ECOMPILE = -3, /* hiprtc compilation failed (log in last_error)   */
  AHIP_EINDEX = -4,   /* index out of bounds (-> IndexError)             */
  AHIP_ENOSUP = -5    /* unsupported dtype/layout combination            */
};

enum ahip_dtype {
  AHIP_BOOL = 0, AHIP_I8 = 1, AHIP_I16 = 2, AHIP_I32 = 3, AHIP_I64 = 4,
  AHIP_U8 = 5, AHIP_U16 = 6, AHIP_U32 = 7, AHIP_U64 = 8,
  AHIP_F32 = 9, AHIP_F64 = 10
};

typedef struct ahip_module_s* ahip_module_t;
typedef struct ahip_func_s* ahip_fn_t;
typedef struct ahip_graph_s* ahip_graph_t;
typedef struct ahip_event_s* ahip_event_t;
typedef struct ahip_list_s* ahip_list_t;
typedef struct ahip_comm_s* ahip_comm_t;

#define AHIP_COMM_ID_BYTES 128
enum ahip_red_op { AHIP_RED_SUM = 0, AHIP_RED_PROD = 1, AHIP_RED_MAX = 2, AHIP_RED_MIN = 3 };

/* Kernel-argument block of every GENERATED fused Elemwise(+CAReduce) kernel
 *   extern "C" __global__ void k(ahip_ew_args a);
 * (generator: aesara_amd/codegen.py).  Operands are ordered inputs then outputs.            */
typedef struct ahip_ew_args {
  int64_t n;                              /* total number of output elements (kept x reduced) */
  int64_t shape[AHIP_MAXD];               /* collapsed iteration shape, outermost first       */
  int64_t stride[AHIP_MAXOPS][AHIP_MAXD]; /* per-operand strides (elements), 0 = broadcast    */
  void* ptr[AHIP_MAXOPS];                 /* operand base pointers                            */
  void* ws;                               /* reduction workspace (partials), may be NULL      */
  void* out;                              /* reduction result pointer, may be NULL            */
  int64_t aux0;                           /* kernel-specific (e.g. #kept elements, #slices)   */
  int64_t aux1;
  int32_t nd;                             /* number of valid dims in shape/stride             */
  int32_t nops;
} ahip_ew_args;

/* Kernel-argument block of a HORIZONTALLY FUSED launch of the generated Elemwise + full-CAReduce
 * kernel: up to AHIP_HJOBS independent jobs of ONE kernel specialisation (same scalar program,
 * dtypes, flat contiguous operands) — each job its own operands, length and result — share one
 * grid: workgroups [wg0[j], wg0[j + 1]) belong to job j.
 *   extern "C" __global__ void k(ahip_ewh_args h);      (codegen.generate, spec.hjobs)        */
#define AHIP_HJOBS 16
#define AHIP_HOPS 6
typedef struct ahip_ewh_args {
  int64_t n[AHIP_HJOBS];                   /* elements of job j                                  */
  void* ptr[AHIP_HJOBS][AHIP_HOPS];        /* operand base pointers of job j (inputs, outputs)   */
  void* out[AHIP_HJOBS];                   /* reduction result of job j                          */
  uint32_t wg0[AHIP_HJOBS + 1];            /* first workgroup of job j; wg0[njobs] = grid size   */
  int32_t njobs;
  void* ws;                                /* shared reduction workspace                         */
  int64_t aux1;                            /* byte offset of the workspace tail (epochs, errors) */
} ahip_ewh_args;

/* Kernel-argument block of the GENERATED fused GEMV-chain + Elemwise epilogue kernels
 *   extern "C" __global__ void k(ahip_gv_args a);       (codegen.generate_gemv_epilogue)
 * y[m] = f(dot_0[m], .., dot_{D-1}[m], operands[m]),  dot_d[m] = sum_k A_d[m*a_rs + k*a_cs] * x_d[k*incx] */
#define AHIP_MAXDOTS 8
#define AHIP_GV_MAXOPS 16
#define AHIP_GV_MAXXIN 4
typedef struct ahip_gv_args {
  int64_t M;
  const void* A[AHIP_MAXDOTS]; int64_t a_rs[AHIP_MAXDOTS]; int64_t a_cs[AHIP_MAXDOTS];
  int64_t K[AHIP_MAXDOTS];
  const void* x[AHIP_MAXDOTS]; int64_t incx[AHIP_MAXDOTS];
  void* ptr[AHIP_GV_MAXOPS];      /* epilogue operands: inputs then outputs               */
  int64_t stride[AHIP_GV_MAXOPS]; /* element stride along m (0 = broadcast)               */
  int32_t ndots; int32_t nops;
  /* optional vector prologue of dot d: x_d = g(xin[d][0..3]) evaluated while it is loaded (unit-
   * stride vectors of length K_d or scalars); xout[d] != NULL: the computed x_d is also stored  */
  const void* xin[AHIP_MAXDOTS][AHIP_GV_MAXXIN];
  void* xout[AHIP_MAXDOTS];
} ahip_gv_args;

/* Kernel-argument block of the GENERATED single-pass "row program" kernels
 *   extern "C" __global__ void k(ahip_rp_args a);       (codegen.generate_rowpass)
 * per row m of the row-major N x K matrix X: d = X[m,:].w, a scalar program over d and the
 * row-wise operands, Sum partials of some of its values, and the column accumulation
 * g[k] += X[m,k] * r[m]; one partial per workgroup goes to col_ws [grid][K] / red_ws [grid][nred]. */
#define AHIP_RP_MAXOPS 16
typedef struct ahip_rp_args {
  int64_t N; int64_t K; const void* X; int64_t x_rs; const void* w;
  void* ptr[AHIP_RP_MAXOPS];      /* row-wise operands: inputs then materialised outputs   */
  int64_t stride[AHIP_RP_MAXOPS]; /* element stride along the row index (0 = broadcast)    */
  void* col_ws; void* red_ws;
  int32_t nops; int32_t nred;
} ahip_rp_args;

/* Kernel-argument block of the GENERATED "row-chain" kernels
 *   extern "C" __global__ void k(ahip_rc_args a);       (codegen.generate_rowchain)
 * rows of a [lshape..., K] space (K contiguous, N = prod(lshape) rows, up to 4 jointly-collapsed
 * leading dims); ptr: external operands then stored outputs; ls[k][d] = element stride of operand k
 * along leading dim d (0 = broadcast).                                                          */
#define AHIP_RC_MAXOPS 16
#define AHIP_RC_MAXLEAD 4
typedef struct ahip_rc_args {
  int64_t N; int64_t K;
  int64_t lshape[AHIP_RC_MAXLEAD];
  void* ptr[AHIP_RC_MAXOPS];
  int64_t ls[AHIP_RC_MAXOPS][AHIP_RC_MAXLEAD];
} ahip_rc_args;

/* Kernel-argument block of the GENERATED small-M "GEMM chain + epilogue" kernels
 *   extern "C" __global__ void k(ahip_ge_args a);       (codegen.generate_gemm_epilogue)
 * out[m,n] = f(A_0 @ B_0, ..., operands[m,n]); A_d is [M, K_d] k-contiguous (row stride a_rs),
 * B_d is [K_d, N] with element strides (b_rs, b_cs); ptr/rs/cs: epilogue operands then outputs
 * (element strides along m / n, 0 = broadcast).                                                 */
#define AHIP_GE_MAXDOTS 3
#define AHIP_GE_MAXOPS 12
typedef struct ahip_ge_args {
  int64_t M; int64_t N; int64_t K[AHIP_GE_MAXDOTS];
  const void* A[AHIP_GE_MAXDOTS]; int64_t a_rs[AHIP_GE_MAXDOTS];
  const void* B[AHIP_GE_MAXDOTS]; int64_t b_rs[AHIP_GE_MAXDOTS]; int64_t b_cs[AHIP_GE_MAXDOTS];
  void* ptr[AHIP_GE_MAXOPS]; int64_t rs[AHIP_GE_MAXOPS]; int64_t cs[AHIP_GE_MAXOPS];
} ahip_ge_args;

typedef struct ahip_device_info {
  int32_t device;
  int32_t cu_count;
  int32_t wavefront_size;
  int32_t max_threads_per_block;
  int64_t total_mem;
  int64_t lds_per_block;
  int32_t clock_khz;
  int32_t l2_bytes;
  char arch[64];
  char name[128];
} ahip_device_info;

/* ---- runtime ------------------------------------------------------------------------ */
/* replaces: nothing in the reference has a device (config.device accepts only "cpu",
 * configdefaults.py:331-336); this is the backend's analogue of cmodule.py's dlimport. */
int ahip_abi_version(void);
int ahip_init(int device_ordinal);
const char* ahip_last_error(void);
int ahip_get_device_info(ahip_device_info* out);
int ahip_stream_synchronize(void* stream);
/* launch-shape tunables: "stream_blocks_per_cu" (default 8), "reduce_blocks_per_cu" (default 8),
 * "gemm_small_max_tiles" (default 64: below that many 128x128 tiles ahip_gemm uses the
 * 16x16-tile split-K kernel that gives every CU work on small outputs), "gemm_skinny_nf",
 * "gemv_col_blocks_per_cu" (default 4) and "gemv_col_strip_lanes" (default 128: lanes x 16 B
 * of every row per workgroup in the transposed-matrix GEMV)                                    */
int ahip_set_param(const char* name, int64_t value);

/* ---- runtime compilation of generated kernels ------------------------------------------
 * replaces: link/c/cmodule.py:2047 GCC_compiler.compile_str + :1240 ModuleCache.module_from_key
 * (the reference compiles per-Op C++ with g++ and dlopens it; here HIP source -> gfx950 code
 * object via hiprtc, cached by the host keyed on sha256(source)).                            */
int ahip_compile(const char* source, const char* name, const char* const* options, int n_options,
                 void** code_out, size_t* size_out); /* works without a GPU (cross-compiles) */
int ahip_free_code(void* code);
int ahip_module_load(const void* code, size_t size, ahip_module_t* out);
int ahip_module_get_function(ahip_module_t m, const char* kernel_name, ahip_fn_t* out);
int ahip_module_unload(ahip_module_t m);
/* raw launch of a loaded kernel with an opaque kernarg block (a list that records such a launch
 * cannot be rebound: nothing says where its pointers are) */
int ahip_launch(ahip_fn_t f, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                uint32_t bz, uint32_t shmem_bytes, const void* kernarg, size_t kernarg_size,
                void* stream);
/* the same with the POINTER MAP of the kernarg block — byte offsets (multiples of 8) of its device
 * pointers, the only words ahip_list_bind_bases may ever re-point — and an optional cooperative
 * launch (hipModuleLaunchCooperativeKernel: the runtime refuses a grid that cannot be co-resident).
 * Used for the persistent one-kernel Scan loops (replaces scan/scan_perform.pyx:309-541).       */
int ahip_launch_p(ahip_fn_t f, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                  uint32_t bz, uint32_t shmem_bytes, const void* kernarg, size_t kernarg_size,
                  const uint16_t* ptr_offsets, int n_ptrs, int cooperative, void* stream);
/* workgroups of `f` (block_threads threads, dyn_lds_bytes of dynamic LDS) that fit on one CU, and
 * the CU count: a persistent kernel whose workgroups wait for each other is only launched when
 * grid <= blocks_per_cu * cu_count (the reference loop has no such hazard: scan_perform.pyx is
 * sequential; here co-residency is what makes the in-kernel exchange terminate)                 */
int ahip_occupancy(ahip_fn_t f, int block_threads, size_t dyn_lds_bytes, int* blocks_per_cu,
                   int* cu_count);

/* ---- K1/K3: fused broadcast Elemwise ----------------------------------------------------
 * replaces: tensor/elemwise.py:725 Elemwise.perform / :835 _c_all (C loop nest generated by
 * elemwise_cgen.py:228 make_loop / :305 make_reordered_loop) with the scalar body of
 * scalar/basic.py ScalarOp.c_code / :4250 Composite.c_code_template.
 * `k` must be a kernel generated for (nd, operand layout classes, vector width `vec`,
 * threads `block`); the shim packs ahip_ew_args, sizes the grid (grid-stride, capped at
 * cu_count*8 workgroups... see elemwise_launch.hip) and launches.                            */
int ahip_elemwise(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                  const int64_t* strides /* [nops*nd] */, int vec, int block, void* stream);

/* The same launch with the grid capped at `wg_per_cu` workgroups per CU instead of the library-wide
 * stream_blocks_per_cu (ahip_set_param): streams whose operands are read once (>= 96 MiB each,
 * non-temporal 16-byte accesses) run fastest with FEW wavefronts in flight — 2 x 256 threads per CU:
 * BASELINE config 1b 64.3-65.4 -> 62.0-62.4 us (profiles/r05_cfg1b_stream_sweep3.txt).             */
int ahip_elemwise_wg(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                     const int64_t* strides /* [nops*nd] */, int vec, int block, int wg_per_cu,
                     void* stream);

/* ---- K3t: Elemwise (+ optional full CAReduce) with transposed operands --------------------
 * replaces: the same Elemwise._c_all / CAReduce._c_all loop nests (tensor/elemwise.py:835/:1522,
 * elemwise_cgen.py:228-305 per-operand strides) when some input has its unit stride along
 * `tile_dim` instead of the last dim (a DimShuffle view, tensor/elemwise.py:39).  `k` is a kernel
 * generated for (nd, tile_dim, tile): it stages tile x tile blocks of those inputs through LDS.
 * `out`/`ws` both NULL: plain Elemwise, one workgroup per tile.  Both set: the scalar output
 * named by the kernel is reduced over all elements into `out` (workspace as for
 * ahip_elemwise_reduce_all).                                                                  */
int ahip_elemwise_tiled(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                        const int64_t* strides, int tile_dim, int tile, void* out, void* ws,
                        size_t ws_bytes, void* stream);

/* ---- K2: Elemwise fused into a full CAReduce (axis=None) --------------------------------
 * replaces: tensor/elemwise.py:1495 CAReduce.perform / :1522 _c_all (make_loop_careduce,
 * elemwise_cgen.py:502) applied to the output of the Elemwise above, without materialising
 * the intermediate.  ONE generated kernel: every workgroup publishes its partial into `ws` as
 * two epoch-tagged 8-byte granules and workgroup 0 collects them, folds them in index order
 * (deterministic) and stores the cast result to `out`.  `ws` (>= ahip_reduce_ws_bytes()) must be
 * ZERO-INITIALISED once by the caller; the kernel advances the epoch word, so the same
 * workspace serves every later launch on the same stream (graph replays included).  Layout:
 * [0, ahip_reduce_partials_bytes()) the granules; then a 4 KiB tail (epoch at +2048+64, the
 * epoch-tagged error word at +2048+128, at +2048+192 an optional pointer to a 4-byte flag in
 * device-visible host memory that a launch whose finalize timed out sets to 1); then 64 bytes
 * per workgroup for the time stamps of AESARA_HIP_EW_TRACE builds.  */
size_t ahip_reduce_ws_bytes(void);
size_t ahip_reduce_partials_bytes(void);
int ahip_elemwise_reduce_all(ahip_fn_t k, int nd, const int64_t* shape, int nops,
                             void* const* ptrs, const int64_t* strides, int vec, int block,
                             void* out, void* ws, size_t ws_bytes, void* stream);
/* Horizontal fusion: `njobs` (2..AHIP_HJOBS) independent full reductions of the SAME generated
 * kernel (spec.hjobs form) in ONE launch — flat contiguous operands only (`nops` <= AHIP_HOPS
 * per job, ptrs[j * nops + k]), n[j] elements (multiples of `vec`), results out[j].  The grid is
 * shared in proportion to the job sizes (at least one workgroup per job); every job runs its own
 * deterministic finalize (own partial slots and epoch word in `ws`).  What this buys: the fixed
 * cost of a launch (ramp, finalize hop, kernel boundary: ~5 us) is paid once, not per job.      */
int ahip_elemwise_reduce_all_multi(ahip_fn_t k, int njobs, int nops, void* const* ptrs,
                                   const int64_t* n, int vec, int block, void* const* outs, void* ws,
                                   size_t ws_bytes, void* stream);

/* ---- K2: axis CAReduce (optionally with a fused Elemwise producer) ----------------------
 * The iteration space is [kept dims (nk) | reduced dims (nr)], both already collapsed by the
 * host; shape/strides cover nk+nr dims in that order.  `k` is generated for (mode, vec,
 * lanes).  mode 0 ("row", unit stride inside the reduced group): `lanes` (power of two <= 64)
 * adjacent lanes per output element, each taking `vec`-element vectors of the reduced run.
 * mode 1 ("col", unit stride inside the kept group; `lanes` up to block): workgroups of `lanes` x (block/lanes)
 * threads, `vec` adjacent outputs per thread, block/lanes reduced rows in flight, folded in a
 * fixed order.  `nslices`>1 splits the reduced run over gridDim.y and writes
 * [nslices, n_kept] partials (accumulator dtype) into `out_or_ws` for a second pass.       */
int ahip_elemwise_reduce_axis(ahip_fn_t k, int mode, int nk, int nr, const int64_t* shape,
                              int nops, void* const* ptrs, const int64_t* strides, int nslices,
                              void* out_or_ws, int block, int vec, int lanes, void* stream);

/* ---- K4/K6: GEMM on MFMA ------------------------------------------------------------------
 * replaces: tensor/blas.py:518 GemmRelated / :872 Gemm (build_gemm_call :836 -> sgemm_/dgemm_
 * :767-820), :1659 Dot22, :1954 Dot22Scalar, :2179 BatchedDot (batch_gemm :2241) and the
 * NumPy-C-API fallback tensor/c_code/alt_blas_template.c.
 * C[M,N] = alpha * A[M,K] @ B[K,N] + beta * Cin[M,N]; all operands are addressed with explicit
 * (row, col) element strides so the 8 unit-stride layouts of encode_strides_in_unit (:719)
 * and arbitrary views are handled without copies.  beta == 0 ignores Cin (no NaN propagation,
 * like BLAS).  Cin may alias C.  dtype: AHIP_F32 or AHIP_F64.  batch strides in elements.     */
int ahip_gemm(int dtype, int64_t M, int64_t N, int64_t K, const void* alpha, const void* A,
              int64_t a_rs, int64_t a_cs, const void* B, int64_t b_rs, int64_t b_cs,
              const void* beta, const void* Cin, int64_t ci_rs, int64_t ci_cs, void* C,
              int64_t c_rs, int64_t c_cs, void* stream);
int ahip_gemm_batched(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K,
                      const void* alpha, const void* A, int64_t a_bs, int64_t a_rs, int64_t a_cs,
                      const void* B, int64_t b_bs, int64_t b_rs, int64_t b_cs, const void* beta,
                      const void* Cin, int64_t ci_bs, int64_t ci_rs, int64_t ci_cs, void* C,
                      int64_t c_bs, int64_t c_rs, int64_t c_cs, void* stream);
/* Integer / bool products (vector ALU, bit-exact with NumPy's wrap-around arithmetic):
 * replaces tensor/math.py:1879 Dot.perform (np.dot) and tensor/blas.py:2224 BatchedDot.perform for
 * AHIP_BOOL / AHIP_I8..AHIP_U64 operands of ONE dtype (mixed operands are cast by the caller, as
 * Dot.make_node :1903 upcasts).  C[b] = A[b] @ B[b]; element strides, zero batch stride = broadcast. */
int ahip_igemm_batched(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K, const void* A,
                       int64_t a_bs, int64_t a_rs, int64_t a_cs, const void* B, int64_t b_bs,
                       int64_t b_rs, int64_t b_cs, void* C, int64_t c_bs, int64_t c_rs, int64_t c_cs,
                       void* stream);
/* Split-K form for few output tiles and a very long K (weight gradients X.T @ dY): the K range runs
 * as S slices of the 128x128 kernel into the caller-provided workspace [S][M][N], then one pass
 * sums the slices in order and applies alpha / beta (deterministic).  ahip_gemm_ws_bytes returns
 * the bytes it wants (0 = this shape is not split); with ws == NULL / too small / 0 wanted,
 * ahip_gemm_splitk is exactly ahip_gemm.                                                         */
size_t ahip_gemm_ws_bytes(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K);
int ahip_gemm_splitk(int dtype, int64_t M, int64_t N, int64_t K, const void* alpha, const void* A,
                     int64_t a_rs, int64_t a_cs, const void* B, int64_t b_rs, int64_t b_cs,
                     const void* beta, const void* Cin, int64_t ci_rs, int64_t ci_cs, void* C,
                     int64_t c_rs, int64_t c_cs, void* ws, size_t ws_bytes, void* stream);

/* ---- K5: GEMV / GER (HBM-bound BLAS2) -----------------------------------------------------
 * replaces: tensor/blas.py:231 Gemv (perform :279), tensor/blas_c.py:611 CGemv (gemv_c_code
 * :369), tensor/blas.py:330 Ger / blas_c.py:328 CGer.
 * y_out[M] = alpha * A[M,N] @ x[N] + beta * y_in[M]  (beta == 0 ignores y_in; y_in may alias).
 * `ws` is scratch for the column-accumulate layout (a_rs > a_cs transposed views); size from
 * ahip_gemv_ws_bytes().                                                                        */
size_t ahip_gemv_ws_bytes(int dtype, int64_t M, int64_t N);
int ahip_gemv(int dtype, int64_t M, int64_t N, const void* alpha, const void* A, int64_t a_rs,
              int64_t a_cs, const void* x, int64_t incx, const void* beta, const void* y_in,
              int64_t incy_in, void* y_out, int64_t incy_out, void* ws, size_t ws_bytes,
              void* stream);
/* Fused GEMV chain + Elemwise epilogue (generated kernel `k`, one wavefront per output row):
 * replaces a chain of Gemv nodes (beta*y + alpha*A.x with y itself a Gemv, tensor/blas.py:231)
 * and the Elemwise consuming them (tensor/elemwise.py:304) — e.g. one GRU gate of the Scan
 * inner graph (scan/op.py:637) — without materialising the intermediate vectors.            */
int ahip_gemv_epilogue(ahip_fn_t k, const ahip_gv_args* args, int block, void* stream);
/* Single-pass GLM row program (generated kernel `k`): replaces Gemv(X, w) -> Elemwise / Sum ->
 * Gemv(X.T, r) (tensor/blas.py:231 twice + tensor/elemwise.py:304/1221 in between; BASELINE
 * config 5) reading X once.  `grid` workgroups each leave one partial in col_ws / red_ws, folded
 * afterwards by the ordinary reduction kernels (deterministic).  Returns the grid it will use
 * for (N, block) when k == NULL (to size the workspaces).                                      */
int ahip_rowpass_grid(int64_t N, int block, int rows_per_wave);
int ahip_rowpass(ahip_fn_t k, const ahip_rp_args* args, int block, int rows_per_wave,
                 size_t shmem_bytes, void* stream);
/* Row-chain kernel: a chain of last-axis CAReduce steps and the Elemwise steps between them in one
 * pass (each operand read once; intermediates in registers).  replaces e.g. Softmax.c_code
 * tensor/special.py:372-415 (three passes) / the CAReduce-DimShuffle-Elemwise node sequences of
 * tensor/elemwise.py:1495/:222/:725.  rows_per_wave = 64 / lanes-per-row of the generated kernel;
 * 0 = the long-row form (one workgroup per row, rows of tens of thousands of elements).          */
int ahip_rowchain(ahip_fn_t k, const ahip_rc_args* args, int block, int rows_per_wave, void* stream);
/* Small-M GEMM chain + Elemwise epilogue in one kernel (16 x 16*nf tile per workgroup, K split over
 * its 4 wavefronts, epilogue on the summed accumulators).  replaces Gemm / Dot22 nodes
 * (tensor/blas.py:872 / :1659) followed by the Elemwise that consumes them (tensor/elemwise.py:725). */
int ahip_gemm_epilogue(ahip_fn_t k, const ahip_ge_args* args, int nf, int waves /* 4, 8, 16: K slices */,
                       void* stream);
/* A_out[M,N] = A_in + alpha * x[M] y[N]^T */
int ahip_ger(int dtype, int64_t M, int64_t N, const void* alpha, const void* x, int64_t incx,
             const void* y, int64_t incy, const void* A_in, int64_t ai_rs, int64_t ai_cs,
             void* A_out, int64_t ao_rs, int64_t ao_cs, void* stream);

/* ---- K7/K8: fill and strided copy / set / inc ---------------------------------------------
 * replaces: tensor/basic.py:1389 Alloc (perform :1427, PyArray_CopyInto broadcast :1441-1490),
 * tensor/subtensor.py:1454 IncSubtensor (perform :1556), DimShuffle materialisation
 * (tensor/c_code/dimshuffle.c), compile/ops.py:149 DeepCopyOp, tensor/basic.py:2142 Join.
 * dst[i...] (op)= src[i...] over `shape` with per-side element strides (0 = broadcast src).
 * accumulate: 0 = set, 1 = += .  itemsize-only copies use dtype for the add.                  */
int ahip_copy_strided(int dtype, int nd, const int64_t* shape, const void* src,
                      const int64_t* sstrides, void* dst, const int64_t* dstrides, int accumulate,
                      void* stream);
int ahip_fill(int dtype, const void* value /* host scalar */, void* dst, int64_t n, void* stream);
/* tensor/basic.py:2867 ARange (perform :2937, np.arange(start, stop, step, dtype)): NumPy's fill
 * rule in the output dtype — dst[0] = first_next[0] = dtype(start), dst[1] = first_next[1] =
 * dtype(start + step), dst[i] = first + i*delta with delta = next - first, product and sum each
 * rounded (no fused multiply-add).  `first_next` (two values) and `delta` are host scalars of
 * `dtype`; n = ceil((stop - start) / step) is the caller's.                                      */
int ahip_arange(int dtype, const void* first_next, const void* delta, int64_t n, void* dst, void* stream);

/* ---- K9: integer row gather / scatter (bit-exact) ------------------------------------------
 * replaces: tensor/subtensor.py:1925 AdvancedSubtensor1 (perform :1953, x.take(idx, axis=0)) and
 * :2128 AdvancedIncSubtensor1 (np.add.at / set).  Rows are `row_elems` contiguous-or-strided
 * elements (src_rs/dst_rs = element stride between rows, inner dims must be contiguous).
 * Negative indices wrap once (idx + nrows); an index still out of range is reported through
 * *bad_index (device int64, caller-zeroed; first offending value+1 is stored) and the call
 * still returns 0 — the host raises IndexError after its next sync point.                      */
int ahip_take_rows(int dtype, const void* src, int64_t nrows, int64_t src_rs, int64_t row_elems,
                   const void* idx, int idx_dtype, int64_t nidx, int64_t idx_stride, void* dst,
                   int64_t dst_rs, int64_t* bad_index, void* stream);
int ahip_scatter_rows(int dtype, void* dst, int64_t nrows, int64_t dst_rs, int64_t row_elems,
                      const void* idx, int idx_dtype, int64_t nidx, int64_t idx_stride,
                      const void* src, int64_t src_rs, int accumulate, int64_t* bad_index,
                      void* stream);
/* Ordered floating-point scatter-add (float32 / float64): the same result as walking the index
 * list in order (np.add.at, AdvancedIncSubtensor1.perform tensor/subtensor.py:2128), bit for bit,
 * for every destination row that receives at most 64 contributions; rows receiving more are
 * accumulated with atomics (any order).  The list is bucketed by destination row (count, scan,
 * place) and one wavefront per row adds its sources in list order.  `ws` is a 16-byte aligned
 * workspace of ahip_scatter_add_ws_bytes(nrows, nidx) bytes (0 = extents beyond 2^31: use
 * ahip_scatter_rows).  Index and error handling as above.                                        */
size_t ahip_scatter_add_ws_bytes(int64_t nrows, int64_t nidx);
int ahip_scatter_add_rows_ordered(int dtype, void* dst, int64_t nrows, int64_t dst_rs,
                                  int64_t row_elems, const void* idx, int idx_dtype, int64_t nidx,
                                  int64_t idx_stride, const void* src, int64_t src_rs, void* ws,
                                  size_t ws_bytes, int64_t* bad_index, void* stream);
/* ---- K13: sort / argsort of rows ----------------------------------------------------------------
 * replaces: tensor/sort.py:29 SortOp (perform :48 np.sort) / :150 ArgSortOp (perform :184
 * np.argsort) along the last axis.  x is viewed as [rows, n] with element strides x_rs / x_cs;
 * keys_out (same dtype, may be NULL) and idx_out (int64, may be NULL) are C-contiguous [rows, n].
 * Ascending, NaN last, ties in input order (NumPy's stable order; its default introsort leaves
 * ties unspecified).  One workgroup per row, bitonic network in LDS: n <= ahip_sort_max_row().  */
int ahip_sort_max_row(int dtype);
int ahip_sort_rows(int dtype, const void* x, int64_t rows, int64_t n, int64_t x_rs, int64_t x_cs,
                   void* keys_out, int64_t* idx_out, void* stream);
/* rows LONGER than ahip_sort_max_row(): chunks of that many elements are sorted in LDS, then merged by
 * log2(n / chunk) rank-based merge passes (one binary search per element; ties to the left run: the
 * stable order).  `ws`: 16-byte aligned workspace of ahip_sort_large_ws_bytes() bytes (two key and two
 * int64 position images of the [rows, n] array).                                                   */
size_t ahip_sort_large_ws_bytes(int dtype, int64_t rows, int64_t n);
int ahip_sort_rows_large(int dtype, const void* x, int64_t rows, int64_t n, int64_t x_rs, int64_t x_cs,
                         void* keys_out, int64_t* idx_out, void* ws, size_t ws_bytes, void* stream);
/* Nonzero, replaces tensor/basic.py:845 Nonzero (perform :870 np.nonzero) and boolean-mask
 * indexing built on it.  `counts` is the inclusive running count of set entries over the
 * C-order flattened array (n entries; the caller makes it with ahip_cumulative and reads
 * counts[n-1] to size the outputs); entry i is set iff counts[i] != counts[i-1].  Writes the
 * coordinates of the set entries, in C order, to outs[0..nd-1] (int64, counts[n-1] each).       */
int ahip_nonzero_write(const int64_t* counts, int64_t n, int nd, const int64_t* shape,
                       int64_t* const* outs, void* stream);
/* N-d integer-array indexing, tensor/subtensor.py:2543 AdvancedSubtensor / :2647 AdvancedIncSubtensor
 * (perform :2607 / :2688) with integer index arrays only: out[j] = sum_d wrap(idx_d[j*stride_d]) * mults[d]
 * is the flat row index consumed by ahip_take_rows / ahip_scatter_rows; each index wraps once
 * (v + dims[d]); out-of-range values are reported through *bad_index (row 0 is used instead).       */
int ahip_linearize_indices(int nidx, const void* const* idx, const int* idx_dtypes,
                           const int64_t* idx_strides, const int64_t* dims, const int64_t* mults,
                           int64_t n, int64_t* out, int64_t* bad_index, void* stream);
/* tensor/extra_ops.py:102 SearchsortedOp (perform :144 np.searchsorted(x, v, side, sorter)): for
 * every element of the C-contiguous `v` (nv elements of `dtype`) the insertion point in the sorted
 * 1-d `x` (nx elements of the same dtype, element stride x_stride) — leftmost (right = 0: number of
 * elements < v[j]) or rightmost (right = 1: number of elements <= v[j]); NumPy's order, NaN above
 * every number.  `sorter` (int64, nx entries, may be NULL): x[sorter] is the sorted sequence.
 * out: int64[nv].                                                                                  */
int ahip_searchsorted(int dtype, const void* x, int64_t nx, int64_t x_stride, const void* v, int64_t nv,
                      int right, const int64_t* sorter, int64_t* out, void* stream);
/* ---- K12: cumulative sum / product along one axis --------------------------------------------
 * replaces: tensor/extra_ops.py:283 CumOp (perform :311 np.cumsum / np.cumprod).  x is viewed as
 * [outer, n, inner] with element strides (x_so, x_sn, x_si); out is C-contiguous [outer, n, inner].
 * mul = 0: sum, 1: product.  Integers wrap in `dtype`.  Few long lines are cut into chunks
 * (reduce, scan the chunk totals, scan with carry-in): that form needs a caller-provided
 * workspace of ahip_cumulative_ws_bytes() bytes (0 = single pass, `ws` may be NULL).            */
size_t ahip_cumulative_ws_bytes(int dtype, int64_t outer, int64_t n, int64_t inner);
int ahip_cumulative(int dtype, int mul, const void* x, int64_t outer, int64_t n, int64_t inner,
                    int64_t x_so, int64_t x_sn, int64_t x_si, void* out, void* ws,
                    size_t ws_bytes, void* stream);
/* Row argmax, replaces tensor/math.py:330 Argmax (perform :388: np.argmax over the reduced axes
 * moved last and flattened).  x is viewed as [nrows, k] with element strides x_rs / x_cs; out[r]
 * = index of the first maximum of row r; a NaN counts as the maximum (first NaN wins).  When
 * the outputs are adjacent in memory (x_rs == 1: argmax over axis 0) a column form slices the
 * reduced run over several workgroups; it takes an optional workspace of
 * ahip_argmax_ws_bytes() bytes for the per-slice partials (without it: one slice).              */
size_t ahip_argmax_ws_bytes(int dtype, int64_t nrows, int64_t k, int64_t x_rs, int64_t x_cs);
int ahip_argmax_rows(int dtype, const void* x, int64_t nrows, int64_t k, int64_t x_rs, int64_t x_cs,
                     int64_t* out, void* ws, size_t ws_bytes, void* stream);

/* ---- H1/K10: launch-list capture & replay (the CVM analogue) --------------------------------
 * replaces: link/vm.py:388 Loop.__call__ / link/c/c_code/lazylinker_c.c:752 CLazyLinker_call and
 * the per-step VM entry of scan/scan_perform.pyx:418.  Everything launched on `stream` between
 * begin and end is recorded into one hipGraph; replay costs one host call per eval.            */
/* Launch list: between begin and end, every kernel launch issued by ahip_* calls on THIS THREAD is
 * recorded (kernel, grid, kernarg copy) instead of executed.  ahip_list_run re-issues them with
 * plain launches from one host call (no per-node Python/ctypes cost, no graph-launch latency);
 * run it inside ahip_graph_begin/end to turn the same list into a hipGraph for long lists.    */
int ahip_list_begin(void);
int ahip_list_end(ahip_list_t* out);
int ahip_list_length(ahip_list_t l);
int ahip_list_run(ahip_list_t l, void* stream);
/* Zero-copy replay with rebound buffers (Function.__call__ binds NEW input arrays on every call,
 * compile/function/types.py:835-843; the reference's thunks read them through storage cells).
 * ahip_list_bind_bases: declare n address ranges [lo[k], hi[k]) (the plan inputs / output targets
 * of the recorded call; must not overlap); every word a launch site DECLARED as a device pointer
 * (the pointer map passed next to each argument block) that points into a range becomes a
 * relocation; returns their number (< 0: error; -2: a launch was recorded without a map).
 * ahip_list_run_rebased: patch the relocations for the new base addresses and re-issue. */
int ahip_list_bind_bases(ahip_list_t l, const uint64_t* lo, const uint64_t* hi, int n);
int ahip_list_run_rebased(ahip_list_t l, const uint64_t* bases, int n, void* stream);
int ahip_list_destroy(ahip_list_t l);
int ahip_graph_begin(void* stream);
int ahip_graph_end(void* stream, ahip_graph_t* out);
int ahip_graph_launch(ahip_graph_t g, void* stream);
int ahip_graph_destroy(ahip_graph_t g);

/* ---- the one collective of the path: RCCL all-reduce on the LAUNCH stream ----
 * replaces: nothing in the reference's data path (tensor/io.py:108-262 MPI send/recv is its only
 * communication code); it is what a CAReduce / contraction over a batch axis that has been split
 * over the GPUs of a node needs: local partials -> ONE all-reduce -> continue.  One process per
 * GPU.  Rank 0 creates the 128-byte id (ahip_comm_unique_id) and hands it to the other ranks out
 * of band (aesara_amd/dist.py: torch.distributed broadcast); every rank then calls
 * ahip_comm_init_rank.  ahip_allreduce enqueues on `stream`; while a launch list is being
 * recorded it becomes a list entry (replayed by ahip_list_run between the kernels around it).
 * RCCL itself is dlopen'ed at first use (ahip_comm_set_library / $AESARA_HIP_RCCL name a path). */
int ahip_comm_set_library(const char* path);
int ahip_comm_unique_id(void* id_out, size_t id_bytes /* >= AHIP_COMM_ID_BYTES */);
int ahip_comm_init_rank(const void* id, int nranks, int rank, ahip_comm_t* out);
int ahip_comm_size(ahip_comm_t c);
int ahip_comm_rank(ahip_comm_t c);
int ahip_allreduce(ahip_comm_t c, int dtype, int op /* ahip_red_op */, const void* sendbuf,
                   void* recvbuf, int64_t count, void* stream);
int ahip_comm_destroy(ahip_comm_t c);
/* ncclCommAbort: like destroy, but collectives stuck on the device (a rank that never arrived) are
 * terminated instead of waited for — what a watchdog calls before it falls back to another transport */
int ahip_comm_abort(ahip_comm_t c);

/* ---- timing on the launch stream (bench.py roofline leg) ----------------------------------- */
int ahip_event_create(ahip_event_t* out);
int ahip_event_record(ahip_event_t e, void* stream);
int ahip_event_elapsed_ms(ahip_event_t start, ahip_event_t stop, float* ms); /* syncs on stop */
int ahip_event_query(ahip_event_t e);   /* 0: everything before the record has finished, 1: not yet, < 0: error */
int ahip_event_destroy(ahip_event_t e);

#ifdef __cplusplus
}
#endif
#endif /* AESARA_HIP_H */
