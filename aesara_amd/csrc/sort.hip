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

}  // extern "C"
