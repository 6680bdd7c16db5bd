// K5 — GEMV / GER: HBM-bound BLAS2 on plain VALU (2 flop per 4/8 bytes: never MFMA).
//
// Replaces tensor/blas.py:231 Gemv (perform :279 -> SciPy fblas [sd]gemv), tensor/blas_c.py:611
// CGemv (gemv_c_code :369), tensor/blas.py:330 Ger / blas_c.py:328 CGer.
//
//   y_out[m] = alpha * sum_n A[m*a_rs + n*a_cs] * x[n*incx] + beta * y_in[m*incy_in]
//
// Two memory layouts, chosen from the strides (no copies):
//   ROW  (a_cs == 1): the reduced axis n is contiguous. One wavefront per output row; lanes walk
//        the row with 16-byte loads, several rows in flight per wave loop; cross-lane reduce by
//        DPP/shuffle.  x is tiny relative to A and stays in L1/L2.
//   COL  (a_rs == 1): the output axis m is contiguous (transposed view, e.g. X.T @ r).  The
//        workgroup is a TM x TN thread grid: TM lanes cover m with 16-byte vectors (coalesced),
//        TN thread-rows take different n; grid.y slices the reduced axis over the chip and writes
//        [nslice][M] partials; a second tiny kernel folds the slices in a fixed order
//        (deterministic, no float atomics) and applies alpha/beta.
//   anything else takes the ROW kernel with scalar strided loads.
#include "common.h"

namespace {

template <typename T> struct V16;
template <> struct V16<float> { static constexpr int N = 4; using t = float __attribute__((ext_vector_type(4))); };
template <> struct V16<double> { static constexpr int N = 2; using t = double __attribute__((ext_vector_type(2))); };

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

struct GemvArgs {
  int64_t M, N;
  const void* A; int64_t a_rs, a_cs;
  const void* x; int64_t incx;
  const void* y_in; int64_t incy_in;
  void* y_out; int64_t incy_out;
  void* ws;
  double alpha, beta;
  int nslice; int tm; int64_t rows_per_slice;
};
AHIP_PTRS_BEGIN(GemvArgs) AHIP_PTR1(A) AHIP_PTR1(x) AHIP_PTR1(y_in) AHIP_PTR1(y_out) AHIP_PTR1(ws) AHIP_PTRS_END

// ---- ROW layout ----------------------------------------------------------------------------
template <typename T, bool VECLOAD>
__global__ __launch_bounds__(256) void gemv_row_kernel(GemvArgs g) {
  constexpr int VEC = V16<T>::N;
  using vec_t = typename V16<T>::t;
  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ x = static_cast<const T*>(g.x);
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  for (int64_t m = wave; m < g.M; m += nwaves) {
    const T* row = A + m * g.a_rs;
    T acc = 0;
    if constexpr (VECLOAD) {
      const int64_t nv = g.N / VEC;
      T acc2 = 0;
      int64_t v = lane;
      for (; v + 64 < nv; v += 128) {  // two independent 16-byte loads in flight per lane
        vec_t a0 = *reinterpret_cast<const vec_t*>(row + v * VEC);
        vec_t a1 = *reinterpret_cast<const vec_t*>(row + (v + 64) * VEC);
        vec_t x0 = *reinterpret_cast<const vec_t*>(x + v * VEC);
        vec_t x1 = *reinterpret_cast<const vec_t*>(x + (v + 64) * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) { acc += a0[e] * x0[e]; acc2 += a1[e] * x1[e]; }
      }
      for (; v < nv; v += 64) {
        vec_t a0 = *reinterpret_cast<const vec_t*>(row + v * VEC);
        vec_t x0 = *reinterpret_cast<const vec_t*>(x + v * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc += a0[e] * x0[e];
      }
      acc += acc2;
    } else {
      for (int64_t n = lane; n < g.N; n += 64) acc += row[n * g.a_cs] * x[n * g.incx];
    }
    acc = wave_sum(acc);
    if (lane == 0) {
      T v = alpha * acc;
      if (g.beta != 0.0) v += beta * static_cast<const T*>(g.y_in)[m * g.incy_in];
      static_cast<T*>(g.y_out)[m * g.incy_out] = v;
    }
  }
}

// ---- COL layout ----------------------------------------------------------------------------
// block = 256 threads = TM (lanes over m, VEC each) x TN (thread rows over n)
template <typename T, bool VECLOAD>
__global__ __launch_bounds__(256) void gemv_col_kernel(GemvArgs g) {
  constexpr int VEC = VECLOAD ? V16<T>::N : 1;
  const T* __restrict__ A = static_cast<const T*>(g.A);
  const T* __restrict__ x = static_cast<const T*>(g.x);
  __shared__ T red[256 * 4];
  const int TM = g.tm, TN = 256 / TM;
  const int tx = threadIdx.x % TM, ty = threadIdx.x / TM;
  const int64_t m = ((int64_t)blockIdx.x * TM + tx) * VEC;
  const int64_t n_begin = (int64_t)blockIdx.y * g.rows_per_slice;
  const int64_t n_end = (n_begin + g.rows_per_slice < g.N) ? n_begin + g.rows_per_slice : g.N;
  T acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0;
  if (m < g.M) {
    int64_t n = n_begin + ty;
    if constexpr (VECLOAD) {
      using vec_t = typename V16<T>::t;
      // 8, then 4 independent 16-byte loads in flight per lane
      for (; n + 7 * (int64_t)TN < n_end; n += 8 * (int64_t)TN) {
        vec_t a[8];
        T xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a[u] = *reinterpret_cast<const vec_t*>(A + (n + u * (int64_t)TN) * g.a_cs + m);
          xv[u] = x[(n + u * (int64_t)TN) * g.incx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int e = 0; e < VEC; ++e) acc[e] += a[u][e] * xv[u];
      }
      for (; n + 3 * (int64_t)TN < n_end; n += 4 * (int64_t)TN) {
        vec_t a0 = *reinterpret_cast<const vec_t*>(A + n * g.a_cs + m);
        vec_t a1 = *reinterpret_cast<const vec_t*>(A + (n + TN) * g.a_cs + m);
        vec_t a2 = *reinterpret_cast<const vec_t*>(A + (n + 2 * TN) * g.a_cs + m);
        vec_t a3 = *reinterpret_cast<const vec_t*>(A + (n + 3 * TN) * g.a_cs + m);
        T x0 = x[n * g.incx], x1 = x[(n + TN) * g.incx], x2 = x[(n + 2 * TN) * g.incx],
          x3 = x[(n + 3 * TN) * g.incx];
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          acc[e] += a0[e] * x0 + a1[e] * x1 + a2[e] * x2 + a3[e] * x3;
      }
      for (; n < n_end; n += TN) {
        vec_t a0 = *reinterpret_cast<const vec_t*>(A + n * g.a_cs + m);
        T x0 = x[n * g.incx];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += a0[e] * x0;
      }
    } else {
      for (; n < n_end; n += TN) acc[0] += A[n * g.a_cs + m * g.a_rs] * x[n * g.incx];
    }
  }
  // fold the TN thread rows through LDS (fixed order)
#pragma unroll
  for (int e = 0; e < VEC; ++e) red[(ty * TM + tx) * VEC + e] = acc[e];
  __syncthreads();
  if (ty == 0 && m < g.M) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      T s = 0;
      for (int k = 0; k < TN; ++k) s += red[(k * TM + tx) * VEC + e];
      if (m + e < g.M) {
        if (g.nslice == 1) {
          T v = (T)g.alpha * s;
          if (g.beta != 0.0) v += (T)g.beta * static_cast<const T*>(g.y_in)[(m + e) * g.incy_in];
          static_cast<T*>(g.y_out)[(m + e) * g.incy_out] = v;
        } else {
          static_cast<T*>(g.ws)[(int64_t)blockIdx.y * g.M + m + e] = s;
        }
      }
    }
  }
}

// second pass of the COL layout: one wavefront per output element folds the [nslice][M]
// partials (lanes stride over slices, fixed shuffle tree: deterministic), then alpha/beta.
template <typename T>
__global__ __launch_bounds__(256) void gemv_col_finalize(GemvArgs g) {
  const T* ws = static_cast<const T*>(g.ws);
  const int lane = threadIdx.x & 63;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); m < g.M;
       m += nwaves) {
    T s = 0;
    for (int k = lane; k < g.nslice; k += 64) s += ws[(int64_t)k * g.M + m];
    s = wave_sum(s);
    if (lane == 0) {
      T v = (T)g.alpha * s;
      if (g.beta != 0.0) v += (T)g.beta * static_cast<const T*>(g.y_in)[m * g.incy_in];
      static_cast<T*>(g.y_out)[m * g.incy_out] = v;
    }
  }
}

// y_out = beta * y_in (N == 0)
template <typename T>
__global__ void gemv_scale(GemvArgs g) {
  for (int64_t m = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; m < g.M;
       m += (int64_t)gridDim.x * blockDim.x) {
    T v = 0;
    if (g.beta != 0.0) v = (T)g.beta * static_cast<const T*>(g.y_in)[m * g.incy_in];
    static_cast<T*>(g.y_out)[m * g.incy_out] = v;
  }
}

constexpr int MAX_SLICES = 1024;
int64_t g_col_blocks_per_cu = 4;

int col_slices(int64_t M, int64_t N, int tm, int vec) {
  // enough workgroups to cover the chip ~4x, each with >= 64 reduced rows per thread row
  int64_t gx = (M + (int64_t)tm * vec - 1) / ((int64_t)tm * vec);
  int64_t want = ((int64_t)ahip_cu_count() * g_col_blocks_per_cu + gx - 1) / gx;
  int tn = 256 / tm;
  int64_t maxs = N / ((int64_t)tn * 16);
  if (want > maxs) want = maxs;
  if (want > MAX_SLICES) want = MAX_SLICES;
  if (want < 1) want = 1;
  return (int)want;
}

int64_t g_col_tm_max = 128;   // 2 KiB strips per row: measured 5.2 TB/s vs 4.0 (1 KiB) / 4.9 (4 KiB) on fp64 4096^2
int pick_tm(int64_t M, int vec) {
  int64_t need = (M + vec - 1) / vec;
  int tm = 1;
  while (tm < need && tm < g_col_tm_max) tm <<= 1;
  return tm;
}

template <typename T>
int gemv_dispatch(GemvArgs& g, size_t ws_bytes, hipStream_t s) {
  constexpr int VEC = V16<T>::N;
  if (g.M == 0) return AHIP_OK;
  if (g.N == 0 || g.alpha == 0.0) {
    unsigned blocks = (unsigned)((g.M + 255) / 256 < 1024 ? (g.M + 255) / 256 : 1024);
    AHIP_LAUNCH((gemv_scale<T>), dim3(blocks), dim3(256), 0, s, g);
    return AHIP_OK;
  }
  auto aligned = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  if (g.a_rs == 1 && g.a_cs != 1 && g.N > 1) {  // COL layout
    bool vec = aligned(g.A) && g.a_cs % VEC == 0 && g.M % VEC == 0;
    int v = vec ? VEC : 1;
    g.tm = pick_tm(g.M, v);
    g.nslice = col_slices(g.M, g.N, g.tm, v);
    if (g.nslice > 1 && (g.ws == nullptr || ws_bytes < (size_t)g.nslice * g.M * sizeof(T)))
      g.nslice = 1;  // no workspace: fall back to a single slice (still correct)
    g.rows_per_slice = (g.N + g.nslice - 1) / g.nslice;
    int64_t gx = (g.M + (int64_t)g.tm * v - 1) / ((int64_t)g.tm * v);
    dim3 grid((unsigned)gx, (unsigned)g.nslice);
    if (vec) AHIP_LAUNCH((gemv_col_kernel<T, true>), grid, dim3(256), 0, s, g);
    else AHIP_LAUNCH((gemv_col_kernel<T, false>), grid, dim3(256), 0, s, g);
    if (g.nslice > 1) {
      unsigned blocks = (unsigned)((g.M + 3) / 4 < 2048 ? (g.M + 3) / 4 : 2048);
      AHIP_LAUNCH((gemv_col_finalize<T>), dim3(blocks), dim3(256), 0, s, g);
    }
    return AHIP_OK;
  }
  // ROW layout (or generic strides)
  bool vec = g.a_cs == 1 && g.incx == 1 && aligned(g.A) && aligned(g.x) && g.a_rs % VEC == 0 &&
             g.N % VEC == 0;
  int64_t waves_needed = g.M;
  int64_t blocks = (waves_needed + 3) / 4;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (blocks > cap) blocks = cap;
  if (vec) AHIP_LAUNCH((gemv_row_kernel<T, true>), dim3((unsigned)blocks), dim3(256), 0, s, g);
  else AHIP_LAUNCH((gemv_row_kernel<T, false>), dim3((unsigned)blocks), dim3(256), 0, s, g);
  return AHIP_OK;
}

struct GerArgs {
  int64_t M, N;
  const void* x; int64_t incx;
  const void* y; int64_t incy;
  const void* A_in; int64_t ai_rs, ai_cs;
  void* A_out; int64_t ao_rs, ao_cs;
  double alpha;
};
AHIP_PTRS_BEGIN(GerArgs) AHIP_PTR1(x) AHIP_PTR1(y) AHIP_PTR1(A_in) AHIP_PTR1(A_out) AHIP_PTRS_END

template <typename T>
__global__ void ger_kernel(GerArgs g) {
  const int64_t total = g.M * g.N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / g.N, c = i - r * g.N;
    T v = static_cast<const T*>(g.A_in)[r * g.ai_rs + c * g.ai_cs] +
          (T)g.alpha * static_cast<const T*>(g.x)[r * g.incx] *
              static_cast<const T*>(g.y)[c * g.incy];
    static_cast<T*>(g.A_out)[r * g.ao_rs + c * g.ao_cs] = v;
  }
}

double host_scalar(int dtype, const void* p) {
  return dtype == AHIP_F32 ? (double)*static_cast<const float*>(p)
                           : *static_cast<const double*>(p);
}

}  // namespace

void ahip_gemv_set_col_blocks_per_cu(int64_t v) { g_col_blocks_per_cu = v; }
void ahip_gemv_set_col_strip_lanes(int64_t v) {
  if (v >= 1 && v <= 256 && (v & (v - 1)) == 0) g_col_tm_max = v;
}

extern "C" {

size_t ahip_gemv_ws_bytes(int dtype, int64_t M, int64_t N) {
  int isz = ahip_itemsize(dtype);
  if (isz == 0 || M <= 0 || N <= 0) return 16;
  size_t best = 16;
  const int vecs[2] = {16 / isz, 1};
  for (int k = 0; k < 2; ++k) {
    int tm = pick_tm(M, vecs[k]);
    size_t need = (size_t)col_slices(M, N, tm, vecs[k]) * (size_t)M * (size_t)isz;
    if (need > best) best = need;
  }
  return best;
}

int ahip_gemv(int dtype, int64_t M, int64_t N, const void* alpha, const void* A, int64_t a_rs,
              int64_t a_cs, const void* x, int64_t incx, const void* beta, const void* y_in,
              int64_t incy_in, void* y_out, int64_t incy_out, void* ws, size_t ws_bytes,
              void* stream) {
  AHIP_REQUIRE(dtype == AHIP_F32 || dtype == AHIP_F64, "gemv supports float32/float64 only");
  AHIP_REQUIRE(M >= 0 && N >= 0, "negative extent");
  AHIP_REQUIRE(alpha && beta, "null alpha/beta");
  GemvArgs g;
  memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.A = A; g.a_rs = a_rs; g.a_cs = a_cs; g.x = x; g.incx = incx;
  g.alpha = host_scalar(dtype, alpha);
  g.beta = host_scalar(dtype, beta);
  g.y_in = y_in; g.incy_in = incy_in; g.y_out = y_out; g.incy_out = incy_out;
  g.ws = ws; g.nslice = 1; g.tm = 64;
  if (M > 0) {
    AHIP_REQUIRE(y_out != nullptr, "null y_out");
    AHIP_REQUIRE(N == 0 || (A && x), "null A/x");
    AHIP_REQUIRE(g.beta == 0.0 || y_in != nullptr, "beta != 0 needs y_in");
  }
  return dtype == AHIP_F32 ? gemv_dispatch<float>(g, ws_bytes, as_stream(stream))
                           : gemv_dispatch<double>(g, ws_bytes, as_stream(stream));
}

int ahip_ger(int dtype, int64_t M, int64_t N, const void* alpha, const void* x, int64_t incx,
             const void* y, int64_t incy, const void* A_in, int64_t ai_rs, int64_t ai_cs,
             void* A_out, int64_t ao_rs, int64_t ao_cs, void* stream) {
  AHIP_REQUIRE(dtype == AHIP_F32 || dtype == AHIP_F64, "ger supports float32/float64 only");
  AHIP_REQUIRE(M >= 0 && N >= 0 && alpha, "bad argument");
  if (M == 0 || N == 0) return AHIP_OK;
  AHIP_REQUIRE(x && y && A_in && A_out, "null argument");
  GerArgs g{M, N, x, incx, y, incy, A_in, ai_rs, ai_cs, A_out, ao_rs, ao_cs,
            host_scalar(dtype, alpha)};
  int64_t total = M * N;
  int64_t blocks = (total + 255) / 256;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (blocks > cap) blocks = cap;
  if (dtype == AHIP_F32)
    AHIP_LAUNCH((ger_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
  else
    AHIP_LAUNCH((ger_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), g);
  return AHIP_OK;
}

}  // extern "C"
