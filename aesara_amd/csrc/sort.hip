// K13 — sort / argsort of rows (the last axis), one workgroup per row, bitonic network in LDS.
//
// Replaces tensor/sort.py:29 SortOp (perform :48 np.sort) and :150 ArgSortOp (perform :184
// np.argsort).  Keys are ordered like NumPy orders them: ascending, NaN after every number; ties
// (and NaNs among themselves) keep their input order — the order of NumPy's stable kinds, and the
// only one that is defined for its default introsort, which agrees whenever the keys are distinct.
// The row (padded to a power of two with "after everything" sentinels) lives in LDS as (key,
// position) pairs; rows longer than ahip_sort_max_row() do not fit and are refused.
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ bool nan_(T) { return false; }
template <> __device__ __forceinline__ bool nan_<float>(float v) { return v != v; }
template <> __device__ __forceinline__ bool nan_<double>(double v) { return v != v; }

// (a, ia) goes after (b, ib) in the sorted order
template <typename T>
__device__ __forceinline__ bool after(T a, int ia, T b, int ib) {
  const bool na = nan_(a), nb = nan_(b);
  const bool gt = (a > b) | (na & !nb);
  const bool eq = (a == b) | (na & nb);
  return gt | (eq & (ia > ib));
}

struct SortArgs {
  const void* x; void* keys_out; int64_t* idx_out;
  int64_t rows, n, x_rs, x_cs;     // element strides of the [rows, n] view; outputs are contiguous
  int n2;                          // n padded to a power of two
};
AHIP_PTRS_BEGIN(SortArgs) AHIP_PTR1(x) AHIP_PTR1(keys_out) AHIP_PTR1(idx_out) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void sort_rows_kernel(SortArgs a) {
  extern __shared__ unsigned char smem[];
  T* key = reinterpret_cast<T*>(smem);
  int* pos = reinterpret_cast<int*>(smem + (size_t)a.n2 * sizeof(T));
  const T* __restrict__ x = static_cast<const T*>(a.x);
  for (int64_t r = blockIdx.x; r < a.rows; r += gridDim.x) {
    const T* row = x + r * a.x_rs;
    for (int i = threadIdx.x; i < a.n2; i += blockDim.x) {
      key[i] = i < a.n ? row[(int64_t)i * a.x_cs] : (T)0;
      pos[i] = i < a.n ? i : 0x7fffffff;       // padding: sorts after every real entry
    }
    __syncthreads();
    for (int k = 2; k <= a.n2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < a.n2; i += blockDim.x) {
          const int p = i ^ j;
          if (p > i) {
            const T ki = key[i], kp = key[p];
            const int pi = pos[i], pp = pos[p];
            const bool pad_i = pi == 0x7fffffff, pad_p = pp == 0x7fffffff;
            // i-after-p with padding entries last
            const bool i_after_p = pad_i ? !pad_p : (pad_p ? false : after(ki, pi, kp, pp));
            const bool up = (i & k) == 0;
            if (i_after_p == up) { key[i] = kp; key[p] = ki; pos[i] = pp; pos[p] = pi; }
          }
        }
        __syncthreads();
      }
    }
    for (int i = threadIdx.x; i < a.n; i += blockDim.x) {
      if (a.keys_out) static_cast<T*>(a.keys_out)[r * a.n + i] = key[i];
      if (a.idx_out) a.idx_out[r * a.n + i] = pos[i];
    }
    __syncthreads();
  }
}

template <typename T>
int run_sort(SortArgs& a, hipStream_t s) {
  int n2 = 1;
  while (n2 < a.n) n2 <<= 1;
  a.n2 = n2;
  const size_t shmem = (size_t)n2 * (sizeof(T) + sizeof(int));
  AHIP_REQUIRE(shmem <= 64 * 1024, "sort: rows of %lld elements do not fit in LDS", (long long)a.n);
  int64_t want = a.rows, cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  AHIP_LAUNCH((sort_rows_kernel<T>), dim3((unsigned)want), dim3(256), shmem, s, a);
  return AHIP_OK;
}


// ---- rows longer than LDS: chunk sort + rank-based merge passes ---------------------------------
// A row is cut into chunks of C = ahip_sort_max_row() elements; every chunk is sorted in LDS as
// above (keys + GLOBAL positions out), then log2(n / C) merge passes double the run length: an
// element of the left run lands at  i + #{right elements strictly before it},  one of the right
// run at  j + #{left elements not after it}  (one binary search each) — ties go to the left run,
// whose positions are all smaller: the order stays the stable one.  Ping-pong between two key /
// position buffers in the caller's workspace; the last pass writes the outputs.
template <typename T> __device__ __forceinline__ bool lt_(T a, T b) { return (a < b) | (!nan_(a) & nan_(b)); }

struct ChunkArgs {
  const void* x; void* keys; int64_t* idx;
  int64_t rows, n, x_rs, x_cs;
  int C, nchunks;
};
AHIP_PTRS_BEGIN(ChunkArgs) AHIP_PTR1(x) AHIP_PTR1(keys) AHIP_PTR1(idx) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void sort_chunks_kernel(ChunkArgs a) {
  extern __shared__ unsigned char smem[];
  T* key = reinterpret_cast<T*>(smem);
  int* pos = reinterpret_cast<int*>(smem + (size_t)a.C * sizeof(T));
  const T* __restrict__ x = static_cast<const T*>(a.x);
  const int64_t total = a.rows * a.nchunks;
  for (int64_t w = blockIdx.x; w < total; w += gridDim.x) {
    const int64_t r = w / a.nchunks, c = w - r * a.nchunks;
    const int64_t base = c * a.C;
    const int len = (int)((a.n - base) < a.C ? (a.n - base) : a.C);
    const T* row = x + r * a.x_rs + base * a.x_cs;
    for (int i = threadIdx.x; i < a.C; i += blockDim.x) {
      key[i] = i < len ? row[(int64_t)i * a.x_cs] : (T)0;
      pos[i] = i < len ? i : 0x7fffffff;
    }
    __syncthreads();
    for (int k = 2; k <= a.C; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = threadIdx.x; i < a.C; i += blockDim.x) {
          const int p = i ^ j;
          if (p > i) {
            const T ki = key[i], kp = key[p];
            const int pi = pos[i], pp = pos[p];
            const bool pad_i = pi == 0x7fffffff, pad_p = pp == 0x7fffffff;
            const bool i_after_p = pad_i ? !pad_p : (pad_p ? false : after(ki, pi, kp, pp));
            const bool up = (i & k) == 0;
            if (i_after_p == up) { key[i] = kp; key[p] = ki; pos[i] = pp; pos[p] = pi; }
          }
        }
        __syncthreads();
      }
    }
    T* ko = static_cast<T*>(a.keys) + r * a.n + base;
    int64_t* io = a.idx + r * a.n + base;
    for (int i = threadIdx.x; i < len; i += blockDim.x) { ko[i] = key[i]; io[i] = base + pos[i]; }
    __syncthreads();
  }
}

struct MergeArgs {
  const void* kin; const int64_t* iin; void* kout; int64_t* iout;
  int64_t rows, n, run;
};
AHIP_PTRS_BEGIN(MergeArgs) AHIP_PTR1(kin) AHIP_PTR1(iin) AHIP_PTR1(kout) AHIP_PTR1(iout) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void merge_pass_kernel(MergeArgs a) {
  const T* __restrict__ kin = static_cast<const T*>(a.kin);
  T* __restrict__ kout = static_cast<T*>(a.kout);
  const int64_t total = a.rows * a.n;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = e / a.n, p = e - r * a.n;
    const int64_t pair = p / (2 * a.run) * (2 * a.run);           // start of this pair of runs
    const int64_t mid = pair + a.run < a.n ? pair + a.run : a.n;  // right run starts here
    const int64_t end = pair + 2 * a.run < a.n ? pair + 2 * a.run : a.n;
    const T* row = kin + r * a.n;
    const T v = row[p];
    int64_t dst;
    if (p < mid) {           // left run: elements of the right run strictly before v
      int64_t lo = mid, hi = end;
      while (lo < hi) { const int64_t m = lo + ((hi - lo) >> 1); if (lt_(row[m], v)) lo = m + 1; else hi = m; }
      dst = pair + (p - pair) + (lo - mid);
    } else {                 // right run: elements of the left run not after v
      int64_t lo = pair, hi = mid;
      while (lo < hi) { const int64_t m = lo + ((hi - lo) >> 1); if (!lt_(v, row[m])) lo = m + 1; else hi = m; }
      dst = pair + (lo - pair) + (p - mid);
    }
    if (kout) kout[r * a.n + dst] = v;
    if (a.iout) a.iout[r * a.n + dst] = a.iin[r * a.n + p];
  }
}

template <typename T>
int run_sort_large(const void* x, int64_t rows, int64_t n, int64_t x_rs, int64_t x_cs, void* keys_out,
                   int64_t* idx_out, void* ws, hipStream_t s, int C) {
  // workspace: keys A | keys B | positions A | positions B  (each rows * n)
  const int64_t tot = rows * n;
  char* base = static_cast<char*>(ws);
  const size_t kb = ((size_t)tot * sizeof(T) + 15) / 16 * 16;
  T* kA = reinterpret_cast<T*>(base);
  T* kB = reinterpret_cast<T*>(base + kb);
  int64_t* iA = reinterpret_cast<int64_t*>(base + 2 * kb);
  int64_t* iB = iA + tot;
  const int nchunks = (int)((n + C - 1) / C);
  ChunkArgs c{x, kA, iA, rows, n, x_rs, x_cs, C, nchunks};
  const size_t shmem = (size_t)C * (sizeof(T) + sizeof(int));
  int64_t want = rows * nchunks, cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  AHIP_LAUNCH((sort_chunks_kernel<T>), dim3((unsigned)want), dim3(256), shmem, s, c);
  const T* kin = kA; const int64_t* iin = iA;
  T* kout = kB; int64_t* iout = iB;
  int64_t blocks = (tot + 255) / 256;
  if (blocks > cap * 4) blocks = cap * 4;
  for (int64_t run = C; run < n; run *= 2) {
    const bool last = run * 2 >= n;
    MergeArgs m{kin, iin, last ? keys_out : (void*)kout, last ? idx_out : iout, rows, n, run};
    // (the last pass writes straight into the outputs; one that is not asked for is skipped)
    AHIP_LAUNCH((merge_pass_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, m);
    const T* tk = kin; kin = kout; kout = const_cast<T*>(tk);
    const int64_t* ti = iin; iin = iout; iout = const_cast<int64_t*>(ti);
  }
  return AHIP_OK;
}

}  // namespace

extern "C" {

int ahip_sort_max_row(int dtype) {
  const int sz = ahip_itemsize(dtype);
  if (sz <= 0) return 0;
  int n = 1;
  while (2 * n * (sz + 4) <= 64 * 1024) n *= 2;
  return n;
}

int ahip_sort_rows(int dtype, const void* x, int64_t rows, int64_t n, int64_t x_rs, int64_t x_cs,
                   void* keys_out, int64_t* idx_out, void* stream) {
  AHIP_REQUIRE(rows >= 0 && n >= 0, "negative extent");
  if (rows == 0 || n == 0) return AHIP_OK;
  AHIP_REQUIRE(x && (keys_out || idx_out), "null argument");
  AHIP_REQUIRE(n <= ahip_sort_max_row(dtype), "sort: rows of %lld elements do not fit in LDS",
               (long long)n);
  SortArgs a{x, keys_out, idx_out, rows, n, x_rs, x_cs, 0};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: return run_sort<uint8_t>(a, s);
    case AHIP_I8: return run_sort<int8_t>(a, s);
    case AHIP_I16: return run_sort<int16_t>(a, s);
    case AHIP_U16: return run_sort<uint16_t>(a, s);
    case AHIP_I32: return run_sort<int32_t>(a, s);
    case AHIP_U32: return run_sort<uint32_t>(a, s);
    case AHIP_I64: return run_sort<int64_t>(a, s);
    case AHIP_U64: return run_sort<uint64_t>(a, s);
    case AHIP_F32: return run_sort<float>(a, s);
    case AHIP_F64: return run_sort<double>(a, s);
    default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
  }
}

size_t ahip_sort_large_ws_bytes(int dtype, int64_t rows, int64_t n) {
  const int sz = ahip_itemsize(dtype);
  if (sz <= 0 || rows <= 0 || n <= 0) return 0;
  const size_t kb = ((size_t)(rows * n) * sz + 15) / 16 * 16;
  return 2 * kb + 2 * (size_t)(rows * n) * sizeof(int64_t);
}

int ahip_sort_rows_large(int dtype, const void* x, int64_t rows, int64_t n, int64_t x_rs, int64_t x_cs,
                         void* keys_out, int64_t* idx_out, void* ws, size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(rows >= 0 && n >= 0, "negative extent");
  if (rows == 0 || n == 0) return AHIP_OK;
  AHIP_REQUIRE(x && (keys_out || idx_out) && ws, "null argument");
  AHIP_REQUIRE(n > ahip_sort_max_row(dtype), "sort: rows that fit in LDS take ahip_sort_rows");
  AHIP_REQUIRE(ws_bytes >= ahip_sort_large_ws_bytes(dtype, rows, n), "sort: workspace too small");
  AHIP_REQUIRE(reinterpret_cast<uintptr_t>(ws) % 16 == 0, "sort: workspace must be 16-byte aligned");
  const int C = ahip_sort_max_row(dtype);
  hipStream_t s = as_stream(stream);
#define AHIP_SORT_LARGE(T) return run_sort_large<T>(x, rows, n, x_rs, x_cs, keys_out, idx_out, ws, s, C)
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: AHIP_SORT_LARGE(uint8_t);
    case AHIP_I8: AHIP_SORT_LARGE(int8_t);
    case AHIP_I16: AHIP_SORT_LARGE(int16_t);
    case AHIP_U16: AHIP_SORT_LARGE(uint16_t);
    case AHIP_I32: AHIP_SORT_LARGE(int32_t);
    case AHIP_U32: AHIP_SORT_LARGE(uint32_t);
    case AHIP_I64: AHIP_SORT_LARGE(int64_t);
    case AHIP_U64: AHIP_SORT_LARGE(uint64_t);
    case AHIP_F32: AHIP_SORT_LARGE(float);
    case AHIP_F64: AHIP_SORT_LARGE(double);
    default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
  }
#undef AHIP_SORT_LARGE
}

}  // extern "C"
