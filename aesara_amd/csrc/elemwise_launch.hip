// Launch side of the generated fused Elemwise / CAReduce kernels (K1, K2, K3).
// The kernels themselves are produced at run time by aesara_amd/codegen.py and compiled with
// hiprtc (runtime.hip); this file packs their `ahip_ew_args` block, sizes the grid and
// launches.  Reference loops replaced: tensor/elemwise.py:835 Elemwise._c_all,
// :1522 CAReduce._c_all, elemwise_cgen.py:228/305/502.
#include "common.h"

#define AHIP_MAX_PARTIALS 4096


static int pack_args(ahip_ew_args* a, int nd, const int64_t* shape, int nops, void* const* ptrs,
                     const int64_t* strides) {
  AHIP_REQUIRE(nd >= 1 && nd <= AHIP_MAXD, "nd=%d outside [1,%d]", nd, AHIP_MAXD);
  AHIP_REQUIRE(nops >= 1 && nops <= AHIP_MAXOPS, "nops=%d outside [1,%d]", nops, AHIP_MAXOPS);
  memset(a, 0, sizeof(*a));
  int64_t n = 1;
  for (int d = 0; d < nd; ++d) {
    AHIP_REQUIRE(shape[d] >= 0, "negative extent");
    a->shape[d] = shape[d];
    n *= shape[d];
  }
  a->n = n;
  a->nd = nd;
  a->nops = nops;
  for (int k = 0; k < nops; ++k) {
    a->ptr[k] = ptrs[k];
    for (int d = 0; d < nd; ++d) a->stride[k][d] = strides[(size_t)k * nd + d];
  }
  return AHIP_OK;
}

static int launch(ahip_fn_t k, uint32_t gx, uint32_t gy, uint32_t block, const ahip_ew_args* a,
                  void* stream) {
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(a, po);
  return ahip_launch_module(k->fn, dim3(gx, gy, 1), dim3(block, 1, 1), 0, as_stream(stream), a,
                            sizeof(*a), po, np);
}

// Memory-bound streaming kernels: enough workgroups to fill 256 CUs x 8 resident blocks,
// grid-stride over the rest (guide: Guideline 11).
static int64_t g_stream_blocks_per_cu = 8;
static int64_t g_reduce_blocks_per_cu = 8;
static int64_t g_reduce_row_blocks_per_cu = 16;   // row-mode axis reductions: resident 256-thread workgroups per CU
// flat full reductions (ahip_elemwise_reduce_all): 16 wavefronts per CU = ONE 1024-thread workgroup.
// r04 timeline (tools/ew_trace.py): of two workgroups per CU the second is dispatched 1-3 us late
// (8192 wavefronts launch at ~20 cycles each per shader engine) and loses every issue arbitration
// to the older one (it finishes at 23 us, the first at 15): one workgroup per CU with twice the
// work per wavefront streams 128 MiB in 26.6-26.8 us instead of 27.0-27.3 (profiles/r04 sweeps 6, 7)
static int64_t g_reduce_flat_blocks_per_cu = 4;

static uint32_t stream_grid(int64_t items, int block) {
  int64_t want = (items + block - 1) / block;
  int64_t cap = (int64_t)ahip_cu_count() * g_stream_blocks_per_cu;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (uint32_t)want;
}

extern "C" {

size_t ahip_reduce_partials_bytes(void) { return (size_t)AHIP_MAX_PARTIALS * 16; }
// granules + 4 KiB tail (epoch, error word, host-flag pointer) + 64 B of trace stamps per workgroup
size_t ahip_reduce_ws_bytes(void) { return (size_t)AHIP_MAX_PARTIALS * (16 + 64) + 4096; }

int ahip_set_param(const char* name, int64_t value) {
  if (name != nullptr && !strcmp(name, "reduce_row_blocks_per_cu") && value > 0) { g_reduce_row_blocks_per_cu = value; return AHIP_OK; }
  AHIP_REQUIRE(name != nullptr && value > 0, "bad parameter");
  if (!strcmp(name, "stream_blocks_per_cu")) g_stream_blocks_per_cu = value;
  else if (!strcmp(name, "reduce_blocks_per_cu")) g_reduce_blocks_per_cu = g_reduce_flat_blocks_per_cu = value;
  else if (!strcmp(name, "reduce_flat_blocks_per_cu")) g_reduce_flat_blocks_per_cu = value;
  else if (!strcmp(name, "gemm_small_max_tiles")) ahip_gemm_set_small_max_tiles(value);
  else if (!strcmp(name, "gemm_skinny_nf")) ahip_gemm_set_skinny_nf(value);
  else if (!strcmp(name, "gemm_half_max_tiles")) ahip_gemm_set_half_max_tiles(value);
  else if (!strcmp(name, "gemm_half_min_tiles")) ahip_gemm_set_half_min_tiles(value);
  else if (!strcmp(name, "gemm_half_ksplit")) ahip_gemm_set_half_ksplit(value);
  else if (!strcmp(name, "copy_stream_bytes")) ahip_copy_set_stream_bytes(value);
  else if (!strcmp(name, "gemv_col_blocks_per_cu")) ahip_gemv_set_col_blocks_per_cu(value);
  else if (!strcmp(name, "gemv_col_strip_lanes")) ahip_gemv_set_col_strip_lanes(value);
  else if (!strcmp(name, "argmax_max_slices")) ahip_index_set_argmax_max_slices(value);
  else if (!strcmp(name, "gemm_group")) ahip_gemm_set_group(value);
  else { ahip_set_error("unknown parameter %s", name); return AHIP_EINVAL; }
  return AHIP_OK;
}

int ahip_elemwise(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                  const int64_t* strides, int vec, int block, void* stream) {
  AHIP_REQUIRE(k != nullptr, "null kernel");
  AHIP_REQUIRE(vec >= 1 && block >= 64 && block % 64 == 0, "bad vec/block");
  ahip_ew_args a;
  int rc = pack_args(&a, nd, shape, nops, ptrs, strides);
  if (rc) return rc;
  if (a.n == 0) return AHIP_OK;
  AHIP_REQUIRE(shape[nd - 1] % vec == 0, "inner extent %lld not divisible by vec %d",
               (long long)shape[nd - 1], vec);
  int64_t items = a.n / vec;
  return launch(k, stream_grid(items, block), 1, block, &a, stream);
}

int ahip_elemwise_wg(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                     const int64_t* strides, int vec, int block, int wg_per_cu, void* stream) {
  AHIP_REQUIRE(k != nullptr, "null kernel");
  AHIP_REQUIRE(vec >= 1 && block >= 64 && block % 64 == 0 && wg_per_cu >= 1, "bad vec/block/wg_per_cu");
  ahip_ew_args a;
  int rc = pack_args(&a, nd, shape, nops, ptrs, strides);
  if (rc) return rc;
  if (a.n == 0) return AHIP_OK;
  AHIP_REQUIRE(shape[nd - 1] % vec == 0, "inner extent %lld not divisible by vec %d",
               (long long)shape[nd - 1], vec);
  int64_t items = a.n / vec;
  int64_t want = (items + block - 1) / block;
  const int64_t cap = (int64_t)ahip_cu_count() * wg_per_cu;
  if (want > cap) want = cap;
  return launch(k, (uint32_t)(want < 1 ? 1 : want), 1, block, &a, stream);
}

int ahip_elemwise_reduce_all(ahip_fn_t k, int nd, const int64_t* shape, int nops,
                             void* const* ptrs, const int64_t* strides, int vec, int block,
                             void* out, void* ws, size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(k && out && ws, "null argument");
  AHIP_REQUIRE(ws_bytes >= ahip_reduce_ws_bytes(), "workspace too small");
  AHIP_REQUIRE(vec >= 1 && block >= 64 && block % 64 == 0, "bad vec/block");
  ahip_ew_args a;
  int rc = pack_args(&a, nd, shape, nops, ptrs, strides);
  if (rc) return rc;
  a.ws = ws;
  a.out = out;
  a.aux1 = (int64_t)AHIP_MAX_PARTIALS * 16;  // byte offset of the shard sums / epoch inside ws
  int64_t want = 1;
  if (a.n > 0) {
    AHIP_REQUIRE(shape[nd - 1] % vec == 0, "inner extent not divisible by vec");
    int64_t items = a.n / vec;
    // one partial per workgroup; a few workgroups of 256 threads per CU keep enough 16-byte
    // loads in flight (HBM-bound) while the in-kernel finalize stays tiny.
    want = (items + block - 1) / block;
    // reduce_blocks_per_cu counts 256-thread blocks: the resident thread count per CU stays
    // the same for larger workgroups (1024 threads -> 2 per CU -> 512 partials, which ONE
    // workgroup folds in a single level of the in-kernel finalize)
    int64_t per_cu = g_reduce_flat_blocks_per_cu * 256 / block;
    if (per_cu < 1) per_cu = 1;
    int64_t cap = (int64_t)ahip_cu_count() * per_cu;
    if (cap > AHIP_MAX_PARTIALS) cap = AHIP_MAX_PARTIALS;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
  } else {
    // an empty operand may have no storage at all: the kernel's first (clamped, unconditional)
    // loads read position 0 of every operand, so point them at readable memory
    for (int k = 0; k < nops; ++k) a.ptr[k] = ws;
  }
  return launch(k, (uint32_t)want, 1, block, &a, stream);
}

int ahip_elemwise_reduce_all_multi(ahip_fn_t k, int njobs, int nops, void* const* ptrs,
                                   const int64_t* n, int vec, int block, void* const* outs, void* ws,
                                   size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(k && ptrs && n && outs && ws, "null argument");
  AHIP_REQUIRE(njobs >= 1 && njobs <= AHIP_HJOBS, "njobs=%d outside [1,%d]", njobs, AHIP_HJOBS);
  AHIP_REQUIRE(nops >= 1 && nops <= AHIP_HOPS, "nops=%d outside [1,%d]", nops, AHIP_HOPS);
  AHIP_REQUIRE(ws_bytes >= ahip_reduce_ws_bytes(), "workspace too small");
  AHIP_REQUIRE(vec >= 1 && block >= 64 && block % 64 == 0, "bad vec/block");
  ahip_ewh_args h;
  memset(&h, 0, sizeof(h));
  h.njobs = njobs;
  h.ws = ws;
  h.aux1 = (int64_t)AHIP_MAX_PARTIALS * 16;
  int64_t per_cu = g_reduce_blocks_per_cu * 256 / block;
  if (per_cu < 1) per_cu = 1;
  int64_t cap = (int64_t)ahip_cu_count() * per_cu;
  if (cap > AHIP_MAX_PARTIALS) cap = AHIP_MAX_PARTIALS;
  if (cap < njobs) cap = njobs;
  int64_t total = 0;
  for (int j = 0; j < njobs; ++j) {
    AHIP_REQUIRE(n[j] >= 0 && n[j] % vec == 0, "job %d: %lld elements not divisible by vec %d", j,
                 (long long)n[j], vec);
    total += n[j] / vec;
  }
  // the grid is shared in proportion to the items of each job, one workgroup at least, and never
  // more workgroups than a job has block-sized pieces
  uint32_t at = 0;
  for (int j = 0; j < njobs; ++j) {
    const int64_t items = n[j] / vec;
    int64_t want = (items + block - 1) / block;
    int64_t share = total > 0 ? (cap * items + total - 1) / total : 1;
    if (want > share) want = share;
    if (want < 1) want = 1;
    h.n[j] = n[j];
    h.out[j] = outs[j];
    for (int kx = 0; kx < nops; ++kx) h.ptr[j][kx] = n[j] > 0 ? ptrs[(size_t)j * nops + kx] : ws;
    h.wg0[j] = at;
    at += (uint32_t)want;
  }
  AHIP_REQUIRE(at <= AHIP_MAX_PARTIALS, "too many workgroups");
  for (int j = njobs; j <= AHIP_HJOBS; ++j) h.wg0[j] = at;
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(&h, po);
  return ahip_launch_module(k->fn, dim3(at, 1, 1), dim3(block, 1, 1), 0, as_stream(stream), &h,
                            sizeof(h), po, np);
}

int ahip_elemwise_tiled(ahip_fn_t k, int nd, const int64_t* shape, int nops, void* const* ptrs,
                        const int64_t* strides, int tile_dim, int tile, void* out, void* ws,
                        size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(k != nullptr, "null kernel");
  AHIP_REQUIRE(nd >= 2 && tile_dim >= 0 && tile_dim < nd - 1, "bad tile_dim %d for nd=%d",
               tile_dim, nd);
  AHIP_REQUIRE(tile == 32 || tile == 64, "tile must be 32 or 64");
  AHIP_REQUIRE((out == nullptr) == (ws == nullptr), "out and ws go together");
  ahip_ew_args a;
  int rc = pack_args(&a, nd, shape, nops, ptrs, strides);
  if (rc) return rc;
  int64_t tiles = 1;
  for (int d = 0; d < nd; ++d)
    tiles *= (d == tile_dim || d == nd - 1) ? (shape[d] + tile - 1) / tile : shape[d];
  if (out == nullptr) {   // plain Elemwise: one workgroup per tile
    if (a.n == 0) return AHIP_OK;
    if (tiles > 0x7fffffffll) tiles = 0x7fffffffll;
    return launch(k, (uint32_t)tiles, 1, 256, &a, stream);
  }
  // + full reduction: one partial per workgroup, tiles taken grid-stride
  AHIP_REQUIRE(ws_bytes >= ahip_reduce_ws_bytes(), "workspace too small");
  a.ws = ws;
  a.out = out;
  a.aux1 = (int64_t)AHIP_MAX_PARTIALS * 16;
  int64_t cap = (int64_t)ahip_cu_count() * g_reduce_blocks_per_cu;
  if (cap > AHIP_MAX_PARTIALS) cap = AHIP_MAX_PARTIALS;
  if (tiles > cap) tiles = cap;
  if (tiles < 1) tiles = 1;
  return launch(k, (uint32_t)tiles, 1, 256, &a, stream);
}

int ahip_elemwise_reduce_axis(ahip_fn_t k, int mode, int nk, int nr, const int64_t* shape,
                              int nops, void* const* ptrs, const int64_t* strides, int nslices,
                              void* out_or_ws, int block, int vec, int lanes, void* stream) {
  AHIP_REQUIRE(k && out_or_ws, "null argument");
  AHIP_REQUIRE(nk >= 1 && nr >= 1 && nk + nr <= AHIP_MAXD, "bad nk/nr");
  AHIP_REQUIRE(block >= 64 && block % 64 == 0 && nslices >= 1 && nslices <= 65535,
               "bad block/nslices");
  AHIP_REQUIRE(vec >= 1 && lanes >= 1 && lanes <= (mode == 0 ? 64 : block) &&
               (lanes & (lanes - 1)) == 0, "bad vec/lanes");
  ahip_ew_args a;
  int rc = pack_args(&a, nk + nr, shape, nops, ptrs, strides);
  if (rc) return rc;
  int64_t nkept = 1, nred = 1;
  for (int d = 0; d < nk; ++d) nkept *= shape[d];
  for (int d = nk; d < nk + nr; ++d) nred *= shape[d];
  if (nkept == 0) return AHIP_OK;
  a.n = nkept;
  a.aux0 = nred;
  a.aux1 = nslices;
  a.out = out_or_ws;
  int64_t gx;
  if (mode == 0) {  // row: `lanes` adjacent lanes per output, vectors of `vec` along the run
    AHIP_REQUIRE(shape[nk + nr - 1] % vec == 0, "reduced inner extent not divisible by vec");
    int groups = block / lanes;
    gx = (nkept + groups - 1) / groups;
    // the kernel walks the outputs with a grid stride: a few resident workgroups per CU instead
    // of one tiny workgroup per `groups` outputs (wavefront dispatch, not HBM, bound the latter)
    const int64_t cap = (int64_t)ahip_cu_count() * g_reduce_row_blocks_per_cu;
    if (nslices == 1 && gx > cap) gx = cap;
  } else {          // col: `lanes` threads x `vec` adjacent outputs per workgroup
    AHIP_REQUIRE(shape[nk - 1] % vec == 0, "kept inner extent not divisible by vec");
    gx = (nkept / vec + lanes - 1) / lanes;
  }
  AHIP_REQUIRE(gx <= 0x7fffffffll, "grid too large");
  return launch(k, (uint32_t)gx, (uint32_t)nslices, block, &a, stream);
}

int ahip_gemv_epilogue(ahip_fn_t k, const ahip_gv_args* args, int block, void* stream) {
  AHIP_REQUIRE(k && args, "null argument");
  AHIP_REQUIRE(block >= 64 && block % 64 == 0, "bad block");
  AHIP_REQUIRE(args->ndots >= 1 && args->ndots <= AHIP_MAXDOTS, "bad ndots");
  AHIP_REQUIRE(args->nops >= 1 && args->nops <= AHIP_GV_MAXOPS, "bad nops");
  if (args->M <= 0) return AHIP_OK;
  int waves = block / 64;
  int64_t want = (args->M + waves - 1) / waves;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(args, po);
  return ahip_launch_module(k->fn, dim3((unsigned)want, 1, 1), dim3(block, 1, 1), 0,
                            as_stream(stream), args, sizeof(*args), po, np);
}

int ahip_rowpass_grid(int64_t N, int block, int rows_per_wave) {
  if (N <= 0 || block < 64 || rows_per_wave < 1) return 0;
  int waves = block / 64;
  int64_t want = (N + (int64_t)waves * rows_per_wave - 1) / ((int64_t)waves * rows_per_wave);
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  return (int)want;
}

int ahip_rowpass(ahip_fn_t k, const ahip_rp_args* args, int block, int rows_per_wave,
                 size_t shmem_bytes, void* stream) {
  AHIP_REQUIRE(k && args, "null argument");
  AHIP_REQUIRE(block >= 64 && block % 64 == 0, "bad block");
  AHIP_REQUIRE(args->nops >= 0 && args->nops <= AHIP_RP_MAXOPS, "bad nops");
  int grid = ahip_rowpass_grid(args->N, block, rows_per_wave);
  if (grid <= 0) return AHIP_OK;
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(args, po);
  return ahip_launch_module(k->fn, dim3((unsigned)grid, 1, 1), dim3(block, 1, 1), shmem_bytes,
                            as_stream(stream), args, sizeof(*args), po, np);
}

int ahip_rowchain(ahip_fn_t k, const ahip_rc_args* args, int block, int rows_per_wave,
                  void* stream) {
  AHIP_REQUIRE(k && args, "null argument");
  AHIP_REQUIRE(block >= 64 && block % 64 == 0 && rows_per_wave >= 0 && rows_per_wave <= 64,
               "bad block / rows_per_wave");
  if (args->N <= 0 || args->K <= 0) return AHIP_OK;
  int grid;
  if (rows_per_wave == 0) {   // long rows: one workgroup per row, grid-stride over the rows
    int64_t cap = (int64_t)ahip_cu_count() * (2048 / block > 0 ? 2048 / block : 1);
    grid = (int)(args->N < cap ? args->N : cap);
  } else {
    grid = ahip_rowpass_grid(args->N, block, rows_per_wave);
  }
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(args, po);
  return ahip_launch_module(k->fn, dim3((unsigned)grid, 1, 1), dim3(block, 1, 1), 0,
                            as_stream(stream), args, sizeof(*args), po, np);
}

int ahip_gemm_epilogue(ahip_fn_t k, const ahip_ge_args* args, int nf, int waves, void* stream) {
  AHIP_REQUIRE(k && args, "null argument");
  AHIP_REQUIRE(nf == 1 || nf == 2 || nf == 4, "nf must be 1, 2 or 4");
  AHIP_REQUIRE(waves == 4 || waves == 8 || waves == 16, "waves must be 4, 8 or 16");
  if (args->M <= 0 || args->N <= 0) return AHIP_OK;
  const int64_t gx = (args->N + 16 * nf - 1) / (16 * nf), gy = (args->M + 15) / 16;
  AHIP_REQUIRE(gy < 65536, "M too large for the small-M kernel");
  uint16_t po[AHIP_MAX_PTRS];
  const int np = ahip_ptrs(args, po);
  return ahip_launch_module(k->fn, dim3((unsigned)gx, (unsigned)gy, 1), dim3(64 * waves, 1, 1), 0,
                            as_stream(stream), args, sizeof(*args), po, np);
}

}  // extern "C"
