// K9 — integer row gather / scatter (bit-exact index handling).
//
// Replaces tensor/subtensor.py:1925 AdvancedSubtensor1 (perform :1953: x.take(idx, axis=0)) and
// :2128 AdvancedIncSubtensor1 (inc: np.add.at semantics, set: last write wins).
// Index handling mirrors NumPy: a negative index wraps once (idx + nrows); anything still out
// of range is reported through *bad_index and the row is skipped.
#include <type_traits>

#include "common.h"

namespace {

template <typename I>
__device__ __forceinline__ bool resolve(const I* idx, int64_t i, int64_t stride, int64_t nrows,
                                        int64_t* bad, int64_t* out) {
  if constexpr (sizeof(I) == 8 && !std::is_signed<I>::value) {
    // uint64 indices beyond int64 must not wrap into valid negative ones
    const unsigned long long u = (unsigned long long)idx[i * stride];
    if (u >= (unsigned long long)nrows) {
      const unsigned long long code = (u < 0x7FFFFFFFFFFFFFFEULL ? u : 0x7FFFFFFFFFFFFFFEULL) + 1ULL;
      atomicCAS(reinterpret_cast<unsigned long long*>(bad), 0ULL, code);
      return false;
    }
    *out = (int64_t)u;
    return true;
  }
  int64_t v = (int64_t)idx[i * stride];
  int64_t w = v < 0 ? v + nrows : v;
  if (w < 0 || w >= nrows) {
    unsigned long long code = (unsigned long long)(v >= 0 ? v + 1 : v);
    atomicCAS(reinterpret_cast<unsigned long long*>(bad), 0ULL, code);
    return false;
  }
  *out = w;
  return true;
}

struct IdxArgs {
  const void* src; void* dst; const void* idx;
  int64_t nrows, src_rs, dst_rs, row_elems, nidx, idx_stride;
  int64_t* bad;
  const int* hot_off;   // scatter-add only: when set, rows with <= ORDERED_MAX entries are skipped
  const int* nhot;      // with hot_off: number of such rows left (0: the kernel returns at once)
};
AHIP_PTRS_BEGIN(IdxArgs) AHIP_PTR1(src) AHIP_PTR1(dst) AHIP_PTR1(idx) AHIP_PTR1(bad) AHIP_PTR1(hot_off) AHIP_PTR1(nhot) AHIP_PTRS_END

constexpr int ORDERED_MAX = 64;   // entries per destination row the ordered form sums in index order
__device__ __forceinline__ bool cold_row(const int* off, int64_t r) {
  return off != nullptr && off[r + 1] - off[r] <= ORDERED_MAX;
}

template <typename T, typename I>
__global__ __launch_bounds__(256) void take_rows_kernel(IdxArgs a) {
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* __restrict__ dst = static_cast<T*>(a.dst);
  const int64_t total = a.nidx * a.row_elems;
  for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < total;
       f += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = f / a.row_elems, e = f - i * a.row_elems, r;
    if (resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r))
      dst[i * a.dst_rs + e] = src[r * a.src_rs + e];
  }
}

template <typename T> __device__ __forceinline__ void atomic_acc(T* p, T v) { atomicAdd(p, v); }
// 8/16-bit and 64-bit signed integers: CAS on the containing word (exact modular arithmetic)
template <typename T>
__device__ __forceinline__ void atomic_acc_small(T* p, T v) {
  uintptr_t addr = reinterpret_cast<uintptr_t>(p);
  unsigned* word = reinterpret_cast<unsigned*>(addr & ~uintptr_t(3));
  unsigned shift = (unsigned)(addr & 3) * 8;
  unsigned mask = (sizeof(T) == 1 ? 0xFFu : 0xFFFFu) << shift;
  unsigned old = *word, assumed;
  do {
    assumed = old;
    unsigned cur = (assumed & mask) >> shift;
    unsigned nv = (unsigned)(T)((T)cur + v) & (sizeof(T) == 1 ? 0xFFu : 0xFFFFu);
    old = atomicCAS(word, assumed, (assumed & ~mask) | (nv << shift));
  } while (old != assumed);
}
template <> __device__ __forceinline__ void atomic_acc<int8_t>(int8_t* p, int8_t v) { atomic_acc_small(p, v); }
template <> __device__ __forceinline__ void atomic_acc<uint8_t>(uint8_t* p, uint8_t v) { atomic_acc_small(p, v); }
template <> __device__ __forceinline__ void atomic_acc<int16_t>(int16_t* p, int16_t v) { atomic_acc_small(p, v); }
template <> __device__ __forceinline__ void atomic_acc<uint16_t>(uint16_t* p, uint16_t v) { atomic_acc_small(p, v); }
template <> __device__ __forceinline__ void atomic_acc<int64_t>(int64_t* p, int64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
template <> __device__ __forceinline__ void atomic_acc<uint64_t>(uint64_t* p, uint64_t v) {
  atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)v);
}
template <> __device__ __forceinline__ void atomic_acc<int32_t>(int32_t* p, int32_t v) { atomicAdd(p, v); }
template <> __device__ __forceinline__ void atomic_acc<uint32_t>(uint32_t* p, uint32_t v) { atomicAdd(p, v); }

// inc: fully parallel, one atomic per element (integers exact; floats commutative up to rounding)
template <typename T, typename I>
__global__ __launch_bounds__(256) void scatter_add_kernel(IdxArgs a) {
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* dst = static_cast<T*>(a.dst);
  if (a.nhot != nullptr && *a.nhot == 0) return;
  const int64_t total = a.nidx * a.row_elems;
  for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < total;
       f += (int64_t)gridDim.x * blockDim.x) {
    int64_t i = f / a.row_elems, e = f - i * a.row_elems, r;
    if (resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r) &&
        !cold_row(a.hot_off, r))
      atomic_acc<T>(dst + r * a.dst_rs + e, src[i * a.src_rs + e]);
  }
}

// set: NumPy semantics are sequential (the last duplicate wins), so each thread owns one column
// element and walks the index list in order.
template <typename T, typename I>
__global__ __launch_bounds__(256) void scatter_set_kernel(IdxArgs a) {
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* dst = static_cast<T*>(a.dst);
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < a.row_elems;
       e += (int64_t)gridDim.x * blockDim.x) {
    for (int64_t i = 0; i < a.nidx; ++i) {
      int64_t r;
      if (resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r))
        dst[r * a.dst_rs + e] = src[i * a.src_rs + e];
    }
  }
}

// ---- vector forms: rows are multiples of 16 bytes and 16-byte aligned -----------------------
// One wavefront (or, for rows shorter than 64 vectors, a power-of-two lane group) per index
// entry: the index is read and resolved ONCE per row, the row moves as 16-byte vectors with
// several loads in flight per lane, and no per-element 64-bit divide is left (the element form
// above does one per element; measured r01: gather of 65536 16-KiB rows 3.0 TB/s).
struct VecGeom { int64_t row_v, src_rs_v, dst_rs_v; int lg; };   // lg = log2(lanes per row)
struct IdxVecArgs { IdxArgs a; VecGeom g; };
AHIP_PTRS_BEGIN(IdxVecArgs) AHIP_PTR1(a.src) AHIP_PTR1(a.dst) AHIP_PTR1(a.idx) AHIP_PTR1(a.bad) AHIP_PTR1(a.hot_off) AHIP_PTR1(a.nhot) AHIP_PTRS_END

template <typename I>
__global__ __launch_bounds__(256) void take_rows_vec_kernel(IdxVecArgs w) {
  const IdxArgs& a = w.a;
  const VecGeom& g = w.g;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4* __restrict__ src = static_cast<const u4*>(a.src);
  u4* __restrict__ dst = static_cast<u4*>(a.dst);
  const int lane = threadIdx.x & 63;
  const int G = 1 << g.lg, sub = lane >> g.lg, l = lane & (G - 1), rpw = 64 >> g.lg;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw;
       i0 < a.nidx; i0 += nwaves * rpw) {
    const int64_t i = i0 + sub;
    int64_t r;
    if (i >= a.nidx || !resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r))
      continue;
    const u4* s = src + r * g.src_rs_v;
    u4* d = dst + i * g.dst_rs_v;
    int64_t u = l;
    for (; u + 3 * G < g.row_v; u += 4 * G) {
      const u4 v0 = s[u], v1 = s[u + G], v2 = s[u + 2 * G], v3 = s[u + 3 * G];
      d[u] = v0; d[u + G] = v1; d[u + 2 * G] = v2; d[u + 3 * G] = v3;
    }
    for (; u < g.row_v; u += G) d[u] = s[u];
  }
}

// scatter-add: one wavefront (or lane group) per index entry, index resolved once per row, but
// ELEMENT-wise atomics with consecutive lanes on consecutive elements: a wave-wide atomic then
// covers 256 contiguous bytes (2 cache lines).  Giving each lane a 16-byte unit spreads one
// instruction over 8 lines and measured 3.7x slower (0.88 ms vs 0.24 ms, 65536 x 1024 fp32).
template <typename T, typename I>
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(IdxVecArgs w) {
  const IdxArgs& a = w.a;
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* dst = static_cast<T*>(a.dst);
  if (a.nhot != nullptr && *a.nhot == 0) return;
  const int lane = threadIdx.x & 63;
  const int lg = w.g.lg;
  const int G = 1 << lg, sub = lane >> lg, l = lane & (G - 1), rpw = 64 >> lg;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t i0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw;
       i0 < a.nidx; i0 += nwaves * rpw) {
    const int64_t i = i0 + sub;
    int64_t r;
    if (i >= a.nidx || !resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r) ||
        cold_row(a.hot_off, r))
      continue;
    const T* s = src + i * a.src_rs;
    T* d = dst + r * a.dst_rs;
    int64_t e = l;
    for (; e + 3 * G < a.row_elems; e += 4 * G) {
      const T v0 = s[e], v1 = s[e + G], v2 = s[e + 2 * G], v3 = s[e + 3 * G];
      atomic_acc<T>(d + e, v0); atomic_acc<T>(d + e + G, v1);
      atomic_acc<T>(d + e + 2 * G, v2); atomic_acc<T>(d + e + 3 * G, v3);
    }
    for (; e < a.row_elems; e += G) atomic_acc<T>(d + e, s[e]);
  }
}


// ---- ordered float scatter-add ------------------------------------------------------------
// np.add.at (AdvancedIncSubtensor1.perform, tensor/subtensor.py:2128) walks the index list in
// order, so x[r] receives its contributions in increasing list position; floating-point atomics
// deliver them in any order (a rounding-level, run-to-run varying difference) and serialise on
// hot cache lines.  The ordered form buckets the list by destination row
//   count (hist) -> exclusive offsets (ahip_cumulative) -> place (atomic cursor per row)
// and then lets ONE wavefront own each destination row: it sorts the row's bucket by list
// position in registers (rank by 64 shuffles) and adds the source rows in that order — plain
// loads, one read-modify-write of the destination row, bit-exact with the reference.  Rows that
// collect more than ORDERED_MAX entries ("hot" rows) are left to the atomic kernel, which skips
// every other row (hot_off).
struct OrdArgs {
  IdxArgs a;
  int* cnt;      // [nrows + 1], zeroed; hist counts into cnt[r + 1]
  int* off;      // [nrows + 1] inclusive scan of cnt: off[r] = first bucket slot of row r
  int* cursor;   // [nrows], zeroed
  int* bucket;   // [nidx] list positions grouped by destination row
  int* nhot;     // [1] number of rows left to the atomic form (zeroed with cnt)
};
AHIP_PTRS_BEGIN(OrdArgs) AHIP_PTR1(a.src) AHIP_PTR1(a.dst) AHIP_PTR1(a.idx) AHIP_PTR1(a.bad) AHIP_PTR1(a.hot_off) AHIP_PTR1(a.nhot) AHIP_PTR1(cnt) AHIP_PTR1(off) AHIP_PTR1(cursor) AHIP_PTR1(bucket) AHIP_PTR1(nhot) AHIP_PTRS_END

__global__ __launch_bounds__(256) void ord_zero_kernel(OrdArgs o) {   // cnt and cursor
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i <= o.a.nrows;
       i += (int64_t)gridDim.x * blockDim.x) {
    o.cnt[i] = 0;
    if (i < o.a.nrows) o.cursor[i] = 0;
    if (i == 0) *o.nhot = 0;
  }
}

// inclusive scan of cnt[0..n) into off by ONE workgroup (n <= ORD_SCAN_MAX): thread t owns a
// contiguous chunk, the chunk totals are scanned through LDS
constexpr int ORD_SCAN_MAX = 1 << 17;
__global__ __launch_bounds__(1024) void ord_scan_kernel(OrdArgs o) {
  __shared__ int tot[1024];
  const int n = (int)o.a.nrows + 1, t = threadIdx.x;
  const int c = (n + 1023) / 1024, lo = t * c, hi = min(lo + c, n);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += o.cnt[i];
  tot[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const int v = t >= d ? tot[t - d] : 0;
    __syncthreads();
    tot[t] += v;
    __syncthreads();
  }
  int run = tot[t] - sum;
  int hot = 0;
  for (int i = lo; i < hi; ++i) {
    hot += o.cnt[i] > ORDERED_MAX;
    run += o.cnt[i];
    o.off[i] = run;
  }
  if (hot) atomicAdd(o.nhot, hot);
}

__global__ __launch_bounds__(256) void ord_count_hot_kernel(OrdArgs o) {   // large nrows only
  int hot = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < o.a.nrows;
       i += (int64_t)gridDim.x * blockDim.x)
    hot += o.cnt[i + 1] > ORDERED_MAX;
  if (hot) atomicAdd(o.nhot, hot);
}

template <typename I>
__global__ __launch_bounds__(256) void ord_hist_kernel(OrdArgs o) {
  const IdxArgs& a = o.a;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.nidx;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r;
    if (resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r))
      atomicAdd(o.cnt + r + 1, 1);
  }
}

template <typename I>
__global__ __launch_bounds__(256) void ord_place_kernel(OrdArgs o) {
  const IdxArgs& a = o.a;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.nidx;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r;
    if (resolve(static_cast<const I*>(a.idx), i, a.idx_stride, a.nrows, a.bad, &r))   // (reported by hist already)
      o.bucket[o.off[r] + atomicAdd(o.cursor + r, 1)] = (int)i;
  }
}

// wide rows: one wavefront per (destination row, tile of 64 * V elements); the waves of a row sort
// the same (L1-resident) bucket.  Four source rows are in flight per lane; the adds stay in list
// order.  Tiles rather than whole rows keep the work items short, so a row with many entries does
// not leave the chip idle behind it (measured r02, 65536 x 1024 fp32 into 8192 rows: row-per-wave
// 65 us).
template <typename T, int V>
__global__ __launch_bounds__(256) void ord_sum_wide_kernel(OrdArgs o) {
  const IdxArgs& a = o.a;
  typedef T TV __attribute__((ext_vector_type(V)));
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* dst = static_cast<T*>(a.dst);
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int64_t span = 64 * V;
  const int64_t ntiles = (a.row_elems + span - 1) / span, items = a.nrows * ntiles;
  const int wave0 = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  for (int64_t it = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave0; it < items; it += nwaves) {
    const int64_t r = it / ntiles, tile = it - r * ntiles;
    const int b0 = __builtin_amdgcn_readfirstlane(o.off[r]);
    const int s = __builtin_amdgcn_readfirstlane(o.off[r + 1]) - b0;
    if (s == 0 || s > ORDERED_MAX) continue;
    const int pos = lane < s ? o.bucket[b0 + lane] : 0x7FFFFFFF;
    int rank = 0;
    for (int m = 0; m < s; ++m) rank += __builtin_amdgcn_readlane(pos, m) < pos;
    int sorted = 0;                                   // lane k: the k-th smallest list position
    for (int k = 0; k < s; ++k) {
      const int from = __ffsll((unsigned long long)__ballot(rank == k)) - 1;
      const int v = __shfl(pos, from);
      sorted = lane == k ? v : sorted;
    }
    const int64_t e0 = tile * span + (int64_t)lane * V;
    const bool on = e0 < a.row_elems;                 // every lane stays in the loops below
    T* d = dst + r * a.dst_rs + e0;
    const T* sb = src + e0;
    TV acc;
    if (on) acc = *reinterpret_cast<const TV*>(d);
    int k = 0;
    for (; k + 3 < s; k += 4) {
      const int64_t p0 = __builtin_amdgcn_readlane(sorted, k), p1 = __builtin_amdgcn_readlane(sorted, k + 1);
      const int64_t p2 = __builtin_amdgcn_readlane(sorted, k + 2), p3 = __builtin_amdgcn_readlane(sorted, k + 3);
      if (on) {
        const TV v0 = *reinterpret_cast<const TV*>(sb + p0 * a.src_rs);
        const TV v1 = *reinterpret_cast<const TV*>(sb + p1 * a.src_rs);
        const TV v2 = *reinterpret_cast<const TV*>(sb + p2 * a.src_rs);
        const TV v3 = *reinterpret_cast<const TV*>(sb + p3 * a.src_rs);
        acc += v0; acc += v1; acc += v2; acc += v3;
      }
    }
    for (; k < s; ++k) {
      const int64_t p = __builtin_amdgcn_readlane(sorted, k);
      if (on) acc += *reinterpret_cast<const TV*>(sb + p * a.src_rs);
    }
    if (on) *reinterpret_cast<TV*>(d) = acc;
  }
}

// narrow rows: one thread per destination element; the row's (short) bucket is walked by
// repeated selection of the next larger list position (every lane of a row reads the same
// bucket words: broadcast loads).
template <typename T>
__global__ __launch_bounds__(256) void ord_sum_narrow_kernel(OrdArgs o) {
  const IdxArgs& a = o.a;
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* dst = static_cast<T*>(a.dst);
  const int64_t total = a.nrows * a.row_elems;
  for (int64_t f = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; f < total;
       f += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = f / a.row_elems, e = f - r * a.row_elems;
    const int b0 = o.off[r], s = o.off[r + 1] - b0;
    if (s == 0 || s > ORDERED_MAX) continue;
    T acc = dst[r * a.dst_rs + e];
    int last = -1;
    for (int k = 0; k < s; ++k) {
      int next = 0x7FFFFFFF;
      for (int m = 0; m < s; ++m) {
        const int p = o.bucket[b0 + m];
        next = (p > last && p < next) ? p : next;
      }
      acc += src[(int64_t)next * a.src_rs + e];
      last = next;
    }
    dst[r * a.dst_rs + e] = acc;
  }
}

template <typename T>
bool vec_geom(const IdxArgs& a, VecGeom* g) {
  const int64_t rb = a.row_elems * (int64_t)sizeof(T);
  auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (rb % 16 || rb < 16 || !al(a.src) || !al(a.dst) || (a.src_rs * (int64_t)sizeof(T)) % 16 ||
      (a.dst_rs * (int64_t)sizeof(T)) % 16)
    return false;
  g->row_v = rb / 16;
  g->src_rs_v = a.src_rs * (int64_t)sizeof(T) / 16;
  g->dst_rs_v = a.dst_rs * (int64_t)sizeof(T) / 16;
  int lg = 0;
  while (lg < 6 && (1 << lg) < g->row_v) ++lg;
  g->lg = lg;
  return true;
}

unsigned grid_for_rows(int64_t nidx, int lg) {
  const int64_t rows_per_block = 4 * (64 >> lg);
  int64_t want = (nidx + rows_per_block - 1) / rows_per_block;
  const int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  return (unsigned)(want < 1 ? 1 : want);
}

unsigned grid_for(int64_t items) {
  int64_t want = (items + 255) / 256;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (unsigned)want;
}

template <typename T, typename I>
int run(int which, IdxArgs& a, hipStream_t s) {
  IdxVecArgs w{a, {}};
  const bool vec = vec_geom<T>(a, &w.g);
  if (which == 0 && vec)
    AHIP_LAUNCH((take_rows_vec_kernel<I>), dim3(grid_for_rows(a.nidx, w.g.lg)), dim3(256), 0, s, w);
  else if (which == 0)
    AHIP_LAUNCH((take_rows_kernel<T, I>), dim3(grid_for(a.nidx * a.row_elems)), dim3(256), 0, s, a);
  else if (which == 1 && a.row_elems >= 8) {
    int lg = 0;
    while (lg < 6 && (1 << lg) < a.row_elems) ++lg;
    w.g.lg = lg;
    AHIP_LAUNCH((scatter_add_rows_kernel<T, I>), dim3(grid_for_rows(a.nidx, lg)), dim3(256), 0, s, w);
  } else if (which == 1)
    AHIP_LAUNCH((scatter_add_kernel<T, I>), dim3(grid_for(a.nidx * a.row_elems)), dim3(256), 0, s, a);
  else
    AHIP_LAUNCH((scatter_set_kernel<T, I>), dim3(grid_for(a.row_elems)), dim3(256), 0, s, a);
  return AHIP_OK;
}

template <typename T>
int by_index(int which, int idx_dtype, IdxArgs& a, hipStream_t s) {
  switch (idx_dtype) {
    case AHIP_I8: return run<T, int8_t>(which, a, s);
    case AHIP_I16: return run<T, int16_t>(which, a, s);
    case AHIP_I32: return run<T, int32_t>(which, a, s);
    case AHIP_I64: return run<T, int64_t>(which, a, s);
    case AHIP_U8: return run<T, uint8_t>(which, a, s);
    case AHIP_U16: return run<T, uint16_t>(which, a, s);
    case AHIP_U32: return run<T, uint32_t>(which, a, s);
    case AHIP_U64: return run<T, uint64_t>(which, a, s);
    default: ahip_set_error("index dtype %d is not an integer type", idx_dtype); return AHIP_EINVAL;
  }
}

int dispatch(int which, int dtype, int idx_dtype, IdxArgs& a, hipStream_t s) {
  if (which != 1) {  // pure data movement: by item size
    switch (ahip_itemsize(dtype)) {
      case 1: return by_index<uint8_t>(which, idx_dtype, a, s);
      case 2: return by_index<uint16_t>(which, idx_dtype, a, s);
      case 4: return by_index<uint32_t>(which, idx_dtype, a, s);
      case 8: return by_index<uint64_t>(which, idx_dtype, a, s);
      default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
    }
  }
  switch (dtype) {
    case AHIP_I8: return by_index<int8_t>(which, idx_dtype, a, s);
    case AHIP_I16: return by_index<int16_t>(which, idx_dtype, a, s);
    case AHIP_I32: return by_index<int32_t>(which, idx_dtype, a, s);
    case AHIP_I64: return by_index<int64_t>(which, idx_dtype, a, s);
    case AHIP_U8: return by_index<uint8_t>(which, idx_dtype, a, s);
    case AHIP_U16: return by_index<uint16_t>(which, idx_dtype, a, s);
    case AHIP_U32: return by_index<uint32_t>(which, idx_dtype, a, s);
    case AHIP_U64: return by_index<uint64_t>(which, idx_dtype, a, s);
    case AHIP_F32: return by_index<float>(which, idx_dtype, a, s);
    case AHIP_F64: return by_index<double>(which, idx_dtype, a, s);
    default: ahip_set_error("scatter-add unsupported for dtype %d", dtype); return AHIP_ENOSUP;
  }
}


// ---- ordered scatter-add: host side ----
struct OrdLayout { size_t cnt, off, cursor, bucket, nhot, scan_ws, total; };
inline size_t up16(size_t v) { return (v + 15) & ~size_t(15); }
OrdLayout ord_layout(int64_t nrows, int64_t nidx) {
  OrdLayout L;
  L.cnt = 0;
  L.cursor = up16((size_t)(nrows + 1) * 4);                 // cnt and cursor are zeroed together
  L.off = L.cursor + up16((size_t)nrows * 4);
  L.bucket = L.off + up16((size_t)(nrows + 1) * 4);
  L.nhot = L.bucket + up16((size_t)nidx * 4);
  L.scan_ws = L.nhot + 16;
  L.total = L.scan_ws + up16(ahip_cumulative_ws_bytes(AHIP_I32, 1, nrows + 1, 1));
  return L;
}

template <template <typename> class F>
int by_index_only(int idx_dtype, OrdArgs& o, hipStream_t s) {
  switch (idx_dtype) {
    case AHIP_I8: return F<int8_t>::go(o, s);
    case AHIP_I16: return F<int16_t>::go(o, s);
    case AHIP_I32: return F<int32_t>::go(o, s);
    case AHIP_I64: return F<int64_t>::go(o, s);
    case AHIP_U8: return F<uint8_t>::go(o, s);
    case AHIP_U16: return F<uint16_t>::go(o, s);
    case AHIP_U32: return F<uint32_t>::go(o, s);
    case AHIP_U64: return F<uint64_t>::go(o, s);
    default: ahip_set_error("index dtype %d is not an integer type", idx_dtype); return AHIP_EINVAL;
  }
}
template <typename I> struct HistGo {
  static int go(OrdArgs& o, hipStream_t s) {
    AHIP_LAUNCH((ord_hist_kernel<I>), dim3(grid_for(o.a.nidx)), dim3(256), 0, s, o);
    return AHIP_OK;
  }
};
template <typename I> struct PlaceGo {
  static int go(OrdArgs& o, hipStream_t s) {
    AHIP_LAUNCH((ord_place_kernel<I>), dim3(grid_for(o.a.nidx)), dim3(256), 0, s, o);
    return AHIP_OK;
  }
};

template <typename T>
int run_ord_sum(OrdArgs& o, hipStream_t s) {
  const IdxArgs& a = o.a;
  constexpr int V = 16 / sizeof(T);
  if (a.row_elems < 32) {
    AHIP_LAUNCH((ord_sum_narrow_kernel<T>), dim3(grid_for(a.nrows * a.row_elems)), dim3(256), 0, s, o);
    return AHIP_OK;
  }
  VecGeom g;
  const bool vec = vec_geom<T>(a, &g);
  const int64_t span = 64 * (vec ? V : 1);
  int64_t want = (a.nrows * ((a.row_elems + span - 1) / span) + 3) / 4;
  const int64_t cap = (int64_t)ahip_cu_count() * 32;
  if (want > cap) want = cap;
  if (vec)
    AHIP_LAUNCH((ord_sum_wide_kernel<T, V>), dim3((unsigned)want), dim3(256), 0, s, o);
  else
    AHIP_LAUNCH((ord_sum_wide_kernel<T, 1>), dim3((unsigned)want), dim3(256), 0, s, o);
  return AHIP_OK;
}

// ---- row argmax (np.argmax semantics: first maximum; a NaN is the maximum; first NaN wins) ----
template <typename T> __device__ __forceinline__ bool is_nan_(T) { return false; }
template <> __device__ __forceinline__ bool is_nan_<float>(float v) { return v != v; }
template <> __device__ __forceinline__ bool is_nan_<double>(double v) { return v != v; }

// (a, ia) beats (b, ib): larger value, NaN above everything, ties -> lower index
template <typename T>
__device__ __forceinline__ bool beats(T a, int64_t ia, T b, int64_t ib) {
  const bool na = is_nan_(a), nb = is_nan_(b);
  const bool nan_case = na & (!nb | (ia < ib));            // branch-free: both forms always evaluated
  const bool num_case = (a > b) | ((a == b) & (ia < ib));
  return (na | nb) ? nan_case : num_case;
}

// within ONE thread's scan the candidate index only grows, so "first maximum wins" needs no
// index comparison: a strictly larger value (or the first NaN) replaces the running best
template <typename T>
__device__ __forceinline__ bool better_later(T a, T b) {
  return (a > b) | (is_nan_(a) & !is_nan_(b));
}

struct ArgmaxArgs {
  const void* x; int64_t* out; int64_t nrows, k, x_rs, x_cs;
  void* pval; int64_t* pidx; int64_t nslices;   // column form: per-slice partial (value, index)
};
AHIP_PTRS_BEGIN(ArgmaxArgs) AHIP_PTR1(x) AHIP_PTR1(out) AHIP_PTR1(pval) AHIP_PTR1(pidx) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(ArgmaxArgs a) {
  const T* __restrict__ x = static_cast<const T*>(a.x);
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < a.nrows;
       r += nwaves) {
    const T* row = x + r * a.x_rs;
    T best = row[0];
    int64_t bi = 0;
    constexpr int VEC = 16 / sizeof(T);
    if (a.x_cs == 1 && a.k % VEC == 0 && reinterpret_cast<uintptr_t>(row) % 16 == 0) {
      // contiguous rows: 16-byte loads, two in flight per lane
      struct alignas(16) P { T v[VEC]; };
      const P* prow = reinterpret_cast<const P*>(row);
      const int64_t nv = a.k / VEC;
      int64_t j = lane;
      for (; j + 64 < nv; j += 128) {
        const P p0 = prow[j], p1 = prow[j + 64];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (better_later(p0.v[e], best)) { best = p0.v[e]; bi = j * VEC + e; }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (better_later(p1.v[e], best)) { best = p1.v[e]; bi = (j + 64) * VEC + e; }
      }
      for (; j < nv; j += 64) {
        const P p0 = prow[j];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (better_later(p0.v[e], best)) { best = p0.v[e]; bi = j * VEC + e; }
      }
    } else {
      for (int64_t j = lane; j < a.k; j += 64) {
        const T v = row[j * a.x_cs];
        if (better_later(v, best)) { best = v; bi = j; }
      }
    }
    for (int m = 32; m > 0; m >>= 1) {
      union { T t; int i[2]; } u; u.i[0] = u.i[1] = 0; u.t = best;
      union { int64_t t; int i[2]; } w; w.t = bi;
      u.i[0] = __shfl_xor(u.i[0], m, 64);
      if (sizeof(T) == 8) u.i[1] = __shfl_xor(u.i[1], m, 64);
      w.i[0] = __shfl_xor(w.i[0], m, 64); w.i[1] = __shfl_xor(w.i[1], m, 64);
      if (beats(u.t, w.t, best, bi)) { best = u.t; bi = w.t; }
    }
    if (lane == 0) a.out[r] = bi;
  }
}

// Column form (axis 0 of a matrix: the outputs are adjacent in memory, x_rs == 1): 128 lanes x V
// adjacent outputs per workgroup row, 2 rows of the reduced index in flight, gridDim.y slices of
// the reduced run; `beats` is a total order on (value, index), so any fold order gives np.argmax.
template <typename T, int V>
__global__ __launch_bounds__(256) void argmax_cols_kernel(ArgmaxArgs a) {
  struct alignas(sizeof(T) * V >= 16 ? 16 : sizeof(T) * V) P { T v[V]; };
  const T* __restrict__ x = static_cast<const T*>(a.x);
  constexpr int TX = 128, TY = 2;   // 2 KiB (fp32 x4) of every row per workgroup
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int64_t o0 = ((int64_t)blockIdx.x * TX + tx) * V;
  const bool valid = o0 < a.nrows;
  const int64_t per = (a.k + a.nslices - 1) / a.nslices;
  const int64_t kb = (int64_t)blockIdx.y * per, ke = (kb + per < a.k) ? kb + per : a.k;
  T best[V];
  int64_t bi[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { best[e] = (T)0; bi[e] = -1; }
  if (valid && kb + ty < ke) {
    int64_t j = kb + ty;
    {
      const P p = *reinterpret_cast<const P*>(x + j * a.x_cs + o0);
#pragma unroll
      for (int e = 0; e < V; ++e) { best[e] = p.v[e]; bi[e] = j; }
    }
#pragma unroll 8
    for (j += TY; j < ke; j += TY) {
      const P p = *reinterpret_cast<const P*>(x + j * a.x_cs + o0);
#pragma unroll
      for (int e = 0; e < V; ++e) {
        const bool take = better_later(p.v[e], best[e]);
        best[e] = take ? p.v[e] : best[e];
        bi[e] = take ? j : bi[e];
      }
    }
  }
  __shared__ T sv[TY][TX * V];
  __shared__ int64_t si[TY][TX * V];
#pragma unroll
  for (int e = 0; e < V; ++e) { sv[ty][tx * V + e] = best[e]; si[ty][tx * V + e] = bi[e]; }
  __syncthreads();
  if (ty != 0 || !valid) return;
#pragma unroll
  for (int e = 0; e < V; ++e) {
    for (int w = 1; w < TY; ++w) {
      const T v = sv[w][tx * V + e];
      const int64_t i = si[w][tx * V + e];
      const bool take = (i >= 0) & ((bi[e] < 0) | beats(v, i, best[e], bi[e]));
      best[e] = take ? v : best[e];
      bi[e] = take ? i : bi[e];
    }
    if (a.nslices == 1) {
      a.out[o0 + e] = bi[e];
    } else {
      static_cast<T*>(a.pval)[(int64_t)blockIdx.y * a.nrows + o0 + e] = best[e];
      a.pidx[(int64_t)blockIdx.y * a.nrows + o0 + e] = bi[e];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_fold_kernel(ArgmaxArgs a) {
  const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= a.nrows) return;
  T best = (T)0;
  int64_t bi = -1;
#pragma unroll 16
  for (int64_t s = 0; s < a.nslices; ++s) {   // independent loads: keep 16 slices in flight
    const T v = static_cast<const T*>(a.pval)[s * a.nrows + o];
    const int64_t i = a.pidx[s * a.nrows + o];
    const bool take = (i >= 0) & ((bi < 0) | beats(v, i, best, bi));
    best = take ? v : best;
    bi = take ? i : bi;
  }
  a.out[o] = bi;
}

int64_t g_argmax_max_slices = 128;   // measured r02: 64 -> 36.3 us, 128 -> 33.6 us, 256 -> 46 us (8192x4096 f32, axis 0)

int64_t argmax_slices(int itemsize, int64_t nrows, int64_t k, int64_t x_rs, int64_t x_cs) {
  if (x_rs != 1 || x_cs == 1 || nrows < 16 || k < 64) return 0;    // 0 = row form
  const int64_t per_block = 128 * (16 / itemsize);                   // outputs per workgroup (vector form)
  const int64_t bx = (nrows + per_block - 1) / per_block;
  int64_t want = (8 * (int64_t)ahip_cu_count() + bx - 1) / bx;
  if (want > g_argmax_max_slices) want = g_argmax_max_slices;
  if (want > k / 32) want = k / 32;
  return want < 1 ? 1 : want;
}

template <typename T>
int run_argmax(ArgmaxArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
  int64_t ns = argmax_slices((int)sizeof(T), a.nrows, a.k, a.x_rs, a.x_cs);
  if (ns == 0) {
    int64_t want = (a.nrows + 3) / 4, cap = (int64_t)ahip_cu_count() * 8;
    if (want > cap) want = cap;
    AHIP_LAUNCH((argmax_rows_kernel<T>), dim3((unsigned)want), dim3(256), 0, s, a);
    return AHIP_OK;
  }
  if (ns > 1 && (ws == nullptr || ws_bytes < (size_t)(ns * a.nrows) * 16)) ns = 1;
  a.nslices = ns;
  a.pval = ws;
  a.pidx = reinterpret_cast<int64_t*>(static_cast<char*>(ws) + (size_t)(ns * a.nrows) * 8);
  constexpr int VEC = 16 / sizeof(T);
  const bool vec = VEC > 1 && a.nrows % VEC == 0 && (a.x_cs * (int64_t)sizeof(T)) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(a.x) % 16 == 0;
  if (vec) {
    const int64_t bx = (a.nrows / VEC + 127) / 128;
    AHIP_LAUNCH((argmax_cols_kernel<T, VEC>), dim3((unsigned)bx, (unsigned)ns), dim3(256), 0, s, a);
  } else {
    const int64_t bx = (a.nrows + 127) / 128;
    AHIP_LAUNCH((argmax_cols_kernel<T, 1>), dim3((unsigned)bx, (unsigned)ns), dim3(256), 0, s, a);
  }
  if (ns > 1)
    AHIP_LAUNCH((argmax_fold_kernel<T>), dim3((unsigned)((a.nrows + 255) / 256)), dim3(256), 0, s, a);
  return AHIP_OK;
}

// ---- N-d integer indexing: flat row index of x[i0[j], i1[j], ...] (NumPy advanced indexing with
// only integer arrays: they address the leading dims; each index wraps once, like resolve()) ----
constexpr int LIN_MAX = 8;
struct LinArgs {
  const void* idx[LIN_MAX]; int dtype[LIN_MAX]; int64_t stride[LIN_MAX];
  int64_t dim[LIN_MAX]; int64_t mult[LIN_MAX];
  int64_t n; int64_t* out; int64_t* bad; int nidx;
};
AHIP_PTRS_BEGIN(LinArgs) AHIP_PTRA(idx, LIN_MAX) AHIP_PTR1(out) AHIP_PTR1(bad) AHIP_PTRS_END

__device__ __forceinline__ int64_t load_index(const void* p, int dtype, int64_t i) {
  switch (dtype) {
    case AHIP_I8: return static_cast<const int8_t*>(p)[i];
    case AHIP_I16: return static_cast<const int16_t*>(p)[i];
    case AHIP_I32: return static_cast<const int32_t*>(p)[i];
    case AHIP_I64: return static_cast<const int64_t*>(p)[i];
    case AHIP_U8: return static_cast<const uint8_t*>(p)[i];
    case AHIP_U16: return static_cast<const uint16_t*>(p)[i];
    case AHIP_U32: return static_cast<const uint32_t*>(p)[i];
    default: return (int64_t)static_cast<const uint64_t*>(p)[i];
  }
}

__global__ __launch_bounds__(256) void linearize_kernel(LinArgs a) {
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < a.n;
       j += (int64_t)gridDim.x * blockDim.x) {
    int64_t lin = 0;
    bool ok = true;
    for (int d = 0; d < a.nidx; ++d) {
      const int64_t v = load_index(a.idx[d], a.dtype[d], j * a.stride[d]);
      const int64_t w = v < 0 ? v + a.dim[d] : v;
      if (w < 0 || w >= a.dim[d]) {
        unsigned long long code = (unsigned long long)(v >= 0 ? v + 1 : v);
        atomicCAS(reinterpret_cast<unsigned long long*>(a.bad), 0ULL, code);
        ok = false;
      } else {
        lin += w * a.mult[d];
      }
    }
    a.out[j] = ok ? lin : 0;
  }
}

// ---- Nonzero: coordinates of the set entries of a flat inclusive count (compaction) ----
constexpr int NZ_MAXD = 8;
struct NzArgs { const int64_t* cnt; int64_t n; int nd; int64_t shape[NZ_MAXD]; int64_t* out[NZ_MAXD]; };
AHIP_PTRS_BEGIN(NzArgs) AHIP_PTR1(cnt) AHIP_PTRA(out, NZ_MAXD) AHIP_PTRS_END

__global__ __launch_bounds__(256) void nonzero_write_kernel(NzArgs a) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = a.cnt[i], prev = i ? a.cnt[i - 1] : 0;
    if (c == prev) continue;            // entry i is zero
    int64_t rem = i;
    for (int d = a.nd - 1; d >= 0; --d) {
      const int64_t q = rem / a.shape[d];
      a.out[d][c - 1] = rem - q * a.shape[d];
      rem = q;
    }
  }
}


// ---- Searchsorted: one binary search per element of v over the sorted 1-d x -------------------
// NumPy's order: NaN is larger than every number (np.sort puts NaNs last), so a < b is
// "a < b, or b is NaN and a is not".
struct SsArgs { const void* x; const void* v; const int64_t* sorter; int64_t* out; int64_t nx, xs, nv; int right; };
AHIP_PTRS_BEGIN(SsArgs) AHIP_PTR1(x) AHIP_PTR1(v) AHIP_PTR1(sorter) AHIP_PTR1(out) AHIP_PTRS_END

template <typename T> __device__ __forceinline__ bool ss_lt(T a, T b) { return a < b; }
template <> __device__ __forceinline__ bool ss_lt<float>(float a, float b) { return a < b || (b != b && a == a); }
template <> __device__ __forceinline__ bool ss_lt<double>(double a, double b) { return a < b || (b != b && a == a); }

template <typename T>
__global__ __launch_bounds__(256) void searchsorted_kernel(SsArgs a) {
  const T* x = static_cast<const T*>(a.x);
  const T* v = static_cast<const T*>(a.v);
  for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < a.nv;
       j += (int64_t)gridDim.x * blockDim.x) {
    const T key = v[j];
    int64_t lo = 0, hi = a.nx;
    while (lo < hi) {
      const int64_t mid = lo + ((hi - lo) >> 1);
      int64_t at = mid;
      if (a.sorter) {                       // NumPy leaves out-of-range sorter entries undefined-but-safe
        at = a.sorter[mid];
        at = at < 0 ? 0 : (at >= a.nx ? a.nx - 1 : at);
      }
      const T e = x[at * a.xs];
      const bool go_right = a.right ? !ss_lt<T>(key, e) : ss_lt<T>(e, key);
      if (go_right) lo = mid + 1; else hi = mid;
    }
    a.out[j] = lo;
  }
}

template <typename T>
static int run_searchsorted(const SsArgs& a, hipStream_t s) {
  AHIP_LAUNCH((searchsorted_kernel<T>), dim3(grid_for(a.nv)), dim3(256), 0, s, a);
  return AHIP_OK;
}

}  // namespace

void ahip_index_set_argmax_max_slices(int64_t v) { g_argmax_max_slices = v; }

extern "C" {

int ahip_take_rows(int dtype, const void* src, int64_t nrows, int64_t src_rs, int64_t row_elems,
                   const void* idx, int idx_dtype, int64_t nidx, int64_t idx_stride, void* dst,
                   int64_t dst_rs, int64_t* bad_index, void* stream) {
  AHIP_REQUIRE(nrows >= 0 && row_elems >= 0 && nidx >= 0, "negative extent");
  if (nidx == 0 || row_elems == 0) return AHIP_OK;
  AHIP_REQUIRE(idx && dst && bad_index, "null argument");
  AHIP_REQUIRE(src != nullptr || nrows == 0, "null src");
  IdxArgs a{src, dst, idx, nrows, src_rs, dst_rs, row_elems, nidx, idx_stride, bad_index};
  return dispatch(0, dtype, idx_dtype, a, as_stream(stream));
}

int ahip_scatter_rows(int dtype, void* dst, int64_t nrows, int64_t dst_rs, int64_t row_elems,
                      const void* idx, int idx_dtype, int64_t nidx, int64_t idx_stride,
                      const void* src, int64_t src_rs, int accumulate, int64_t* bad_index,
                      void* stream) {
  AHIP_REQUIRE(nrows >= 0 && row_elems >= 0 && nidx >= 0, "negative extent");
  if (nidx == 0 || row_elems == 0) return AHIP_OK;
  AHIP_REQUIRE(idx && src && bad_index, "null argument");
  AHIP_REQUIRE(dst != nullptr || nrows == 0, "null dst");
  IdxArgs a{src, dst, idx, nrows, src_rs, dst_rs, row_elems, nidx, idx_stride, bad_index};
  return dispatch(accumulate ? 1 : 2, dtype, idx_dtype, a, as_stream(stream));
}

size_t ahip_scatter_add_ws_bytes(int64_t nrows, int64_t nidx) {
  if (nrows <= 0 || nidx <= 0 || nrows >= 0x7FFFFFF0LL || nidx >= 0x7FFFFFF0LL) return 0;
  return ord_layout(nrows, nidx).total;
}

int ahip_scatter_add_rows_ordered(int dtype, void* dst, int64_t nrows, int64_t dst_rs,
                                  int64_t row_elems, const void* idx, int idx_dtype, int64_t nidx,
                                  int64_t idx_stride, const void* src, int64_t src_rs, void* ws,
                                  size_t ws_bytes, int64_t* bad_index, void* stream) {
  AHIP_REQUIRE(nrows >= 0 && row_elems >= 0 && nidx >= 0, "negative extent");
  if (nidx == 0 || row_elems == 0) return AHIP_OK;
  AHIP_REQUIRE(idx && src && bad_index, "null argument");
  AHIP_REQUIRE(dst != nullptr || nrows == 0, "null dst");
  AHIP_REQUIRE(dtype == AHIP_F32 || dtype == AHIP_F64, "ordered scatter-add is the floating-point form");
  if (nrows == 0) {   // every index is out of range: let the plain form report it
    IdxArgs a{src, dst, idx, nrows, src_rs, dst_rs, row_elems, nidx, idx_stride, bad_index};
    return dispatch(1, dtype, idx_dtype, a, as_stream(stream));
  }
  const size_t need = ahip_scatter_add_ws_bytes(nrows, nidx);
  AHIP_REQUIRE(need != 0, "ordered scatter-add: more than 2^31 rows or indices");
  AHIP_REQUIRE(ws != nullptr && ws_bytes >= need && reinterpret_cast<uintptr_t>(ws) % 16 == 0,
               "ordered scatter-add: workspace too small or misaligned");
  hipStream_t s = as_stream(stream);
  const OrdLayout L = ord_layout(nrows, nidx);
  char* w = static_cast<char*>(ws);
  OrdArgs o{{src, dst, idx, nrows, src_rs, dst_rs, row_elems, nidx, idx_stride, bad_index},
            reinterpret_cast<int*>(w + L.cnt), reinterpret_cast<int*>(w + L.off),
            reinterpret_cast<int*>(w + L.cursor), reinterpret_cast<int*>(w + L.bucket),
            reinterpret_cast<int*>(w + L.nhot)};
  AHIP_LAUNCH(ord_zero_kernel, dim3(grid_for(nrows + 1)), dim3(256), 0, s, o);
  int rc = by_index_only<HistGo>(idx_dtype, o, s);
  if (rc != AHIP_OK) return rc;
  if (nrows + 1 <= ORD_SCAN_MAX) {
    AHIP_LAUNCH(ord_scan_kernel, dim3(1), dim3(1024), 0, s, o);
  } else {
    rc = ahip_cumulative(AHIP_I32, 0, o.cnt, 1, nrows + 1, 1, 0, 1, 1, o.off, w + L.scan_ws,
                         L.total - L.scan_ws, stream);
    if (rc != AHIP_OK) return rc;
    AHIP_LAUNCH(ord_count_hot_kernel, dim3(grid_for(nrows)), dim3(256), 0, s, o);
  }
  rc = by_index_only<PlaceGo>(idx_dtype, o, s);
  if (rc != AHIP_OK) return rc;
  rc = dtype == AHIP_F32 ? run_ord_sum<float>(o, s) : run_ord_sum<double>(o, s);
  if (rc != AHIP_OK) return rc;
  IdxArgs hot = o.a;            // rows with more than ORDERED_MAX entries: atomics
  hot.hot_off = o.off;
  hot.nhot = o.nhot;
  return dispatch(1, dtype, idx_dtype, hot, s);
}

size_t ahip_argmax_ws_bytes(int dtype, int64_t nrows, int64_t k, int64_t x_rs, int64_t x_cs) {
  if (nrows <= 0 || k <= 0 || ahip_itemsize(dtype) <= 0) return 0;
  const int64_t ns = argmax_slices(ahip_itemsize(dtype), nrows, k, x_rs, x_cs);
  return ns > 1 ? (size_t)(ns * nrows) * 16 : 0;
}

int ahip_argmax_rows(int dtype, const void* x, int64_t nrows, int64_t k, int64_t x_rs, int64_t x_cs,
                     int64_t* out, void* ws, size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(nrows >= 0 && k >= 0, "negative extent");
  if (nrows == 0) return AHIP_OK;
  AHIP_REQUIRE(k > 0, "attempt to get argmax of an empty sequence");
  AHIP_REQUIRE(x && out, "null argument");
  ArgmaxArgs a{x, out, nrows, k, x_rs, x_cs, nullptr, nullptr, 1};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: return run_argmax<uint8_t>(a, ws, ws_bytes, s);
    case AHIP_I8: return run_argmax<int8_t>(a, ws, ws_bytes, s);
    case AHIP_I16: return run_argmax<int16_t>(a, ws, ws_bytes, s);
    case AHIP_I32: return run_argmax<int32_t>(a, ws, ws_bytes, s);
    case AHIP_I64: return run_argmax<int64_t>(a, ws, ws_bytes, s);
    case AHIP_U16: return run_argmax<uint16_t>(a, ws, ws_bytes, s);
    case AHIP_U32: return run_argmax<uint32_t>(a, ws, ws_bytes, s);
    case AHIP_U64: return run_argmax<uint64_t>(a, ws, ws_bytes, s);
    case AHIP_F32: return run_argmax<float>(a, ws, ws_bytes, s);
    case AHIP_F64: return run_argmax<double>(a, ws, ws_bytes, s);
    default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
  }
}

int ahip_nonzero_write(const int64_t* counts, int64_t n, int nd, const int64_t* shape,
                       int64_t* const* outs, void* stream) {
  AHIP_REQUIRE(nd >= 1 && nd <= NZ_MAXD, "1..8 dims");
  AHIP_REQUIRE(n >= 0, "negative extent");
  if (n == 0) return AHIP_OK;
  AHIP_REQUIRE(counts && shape && outs, "null argument");
  NzArgs a{};
  a.cnt = counts; a.n = n; a.nd = nd;
  for (int d = 0; d < nd; ++d) { a.shape[d] = shape[d]; a.out[d] = outs[d]; }
  AHIP_LAUNCH(nonzero_write_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a);
  return AHIP_OK;
}

int ahip_linearize_indices(int nidx, const void* const* idx, const int* idx_dtypes,
                           const int64_t* idx_strides, const int64_t* dims, const int64_t* mults,
                           int64_t n, int64_t* out, int64_t* bad_index, void* stream) {
  AHIP_REQUIRE(nidx >= 1 && nidx <= LIN_MAX, "1..8 index arrays");
  AHIP_REQUIRE(n >= 0, "negative extent");
  if (n == 0) return AHIP_OK;
  AHIP_REQUIRE(idx && idx_dtypes && idx_strides && dims && mults && out && bad_index, "null argument");
  LinArgs a{};
  for (int d = 0; d < nidx; ++d) {
    AHIP_REQUIRE(idx[d] != nullptr, "null index array");
    AHIP_REQUIRE(idx_dtypes[d] >= AHIP_I8 && idx_dtypes[d] <= AHIP_U64, "index dtype must be an integer type");
    a.idx[d] = idx[d]; a.dtype[d] = idx_dtypes[d]; a.stride[d] = idx_strides[d];
    a.dim[d] = dims[d]; a.mult[d] = mults[d];
  }
  a.n = n; a.out = out; a.bad = bad_index; a.nidx = nidx;
  AHIP_LAUNCH(linearize_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a);
  return AHIP_OK;
}

int ahip_searchsorted(int dtype, const void* x, int64_t nx, int64_t x_stride, const void* v, int64_t nv,
                      int right, const int64_t* sorter, int64_t* out, void* stream) {
  AHIP_REQUIRE(nx >= 0 && nv >= 0, "negative extent");
  if (nv == 0) return AHIP_OK;
  AHIP_REQUIRE(v && out && (x || nx == 0), "null argument");
  SsArgs a{x, v, sorter, out, nx, x_stride, nv, right ? 1 : 0};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: return run_searchsorted<uint8_t>(a, s);
    case AHIP_I8: return run_searchsorted<int8_t>(a, s);
    case AHIP_I16: return run_searchsorted<int16_t>(a, s);
    case AHIP_U16: return run_searchsorted<uint16_t>(a, s);
    case AHIP_I32: return run_searchsorted<int32_t>(a, s);
    case AHIP_U32: return run_searchsorted<uint32_t>(a, s);
    case AHIP_I64: return run_searchsorted<int64_t>(a, s);
    case AHIP_U64: return run_searchsorted<uint64_t>(a, s);
    case AHIP_F32: return run_searchsorted<float>(a, s);
    case AHIP_F64: return run_searchsorted<double>(a, s);
    default: ahip_set_error("searchsorted: unsupported dtype %d", dtype); return AHIP_EINVAL;
  }
}

}  // extern "C"
