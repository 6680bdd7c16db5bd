// K12 — cumulative sum / product along one axis.
//
// Replaces tensor/extra_ops.py:283 CumOp (perform :311: np.cumsum / np.cumprod; C :330-370
// PyArray_CumSum / PyArray_CumProd).  The input is viewed as [outer, n, inner] (element strides
// so, sn, si); the output is C-contiguous.  Two schedules:
//   * inner >= 64: one thread per (outer, inner) line, sequential over n — adjacent threads touch
//     adjacent addresses, so every step is one coalesced row of `inner` elements;
//   * inner small (the last-axis case): one wavefront per line, 64 elements per step scanned with
//     a shuffle-up Hillis-Steele pass plus a running carry.
// Integer arithmetic wraps in the output dtype (as the reference's C loops do); floating-point
// sums are accumulated in scan order per 64-element chunk (round-off level reordering only).
#include "common.h"

namespace {

struct CumArgs {
  const void* x; void* out;
  int64_t outer, n, inner, so, sn, si;
  int mul;
};

template <typename T> __device__ __forceinline__ T comb(T a, T b, int mul) { return mul ? a * b : a + b; }

template <typename T>
__global__ __launch_bounds__(256) void cum_lines_kernel(CumArgs a) {
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const int64_t lines = a.outer * a.inner;
  for (int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; l < lines;
       l += (int64_t)gridDim.x * blockDim.x) {
    const int64_t o = l / a.inner, i = l - o * a.inner;
    const T* px = x + o * a.so + i * a.si;
    T* po = out + o * a.n * a.inner + i;
    T acc = a.mul ? (T)1 : (T)0;
    for (int64_t k = 0; k < a.n; ++k) {
      acc = comb<T>(acc, px[k * a.sn], a.mul);
      po[k * a.inner] = acc;
    }
  }
}

template <typename T>
__device__ __forceinline__ T shfl_up_(T v, int d) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_up(u.i[0], d, 64); u.i[1] = __shfl_up(u.i[1], d, 64);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v;
    u.i = __shfl_up(u.i, d, 64);
    return u.t;
  } else {
    int w = (int)v;
    w = __shfl_up(w, d, 64);
    return (T)w;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cum_wave_kernel(CumArgs a) {
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const int lane = threadIdx.x & 63;
  const int64_t lines = a.outer * a.inner;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const T ident = a.mul ? (T)1 : (T)0;
  for (int64_t l = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); l < lines;
       l += nwaves) {
    const int64_t o = l / a.inner, i = l - o * a.inner;
    const T* px = x + o * a.so + i * a.si;
    T* po = out + o * a.n * a.inner + i;
    T carry = ident;
    for (int64_t k0 = 0; k0 < a.n; k0 += 64) {
      const int64_t k = k0 + lane;
      T v = (k < a.n) ? px[k * a.sn] : ident;
      for (int d = 1; d < 64; d <<= 1) {
        const T up = shfl_up_<T>(v, d);
        if (lane >= d) v = comb<T>(up, v, a.mul);
      }
      v = comb<T>(carry, v, a.mul);
      if (k < a.n) po[k * a.inner] = v;
      union { T t; int w[2]; } u;   // carry = lane 63's value
      u.w[0] = u.w[1] = 0;
      u.t = v;
      u.w[0] = __shfl(u.w[0], 63, 64);
      if (sizeof(T) == 8) u.w[1] = __shfl(u.w[1], 63, 64);
      carry = u.t;
    }
  }
}

template <typename T>
int run_cum(CumArgs& a, hipStream_t s) {
  const int64_t lines = a.outer * a.inner;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (a.inner >= 64) {
    int64_t want = (lines + 255) / 256;
    if (want > cap) want = cap;
    AHIP_LAUNCH((cum_lines_kernel<T>), dim3((unsigned)want), dim3(256), 0, s, a);
  } else {
    int64_t want = (lines + 3) / 4;
    if (want > cap) want = cap;
    AHIP_LAUNCH((cum_wave_kernel<T>), dim3((unsigned)want), dim3(256), 0, s, a);
  }
  return AHIP_OK;
}

}  // namespace

extern "C" {

int ahip_cumulative(int dtype, int mul, const void* x, int64_t outer, int64_t n, int64_t inner,
                    int64_t x_so, int64_t x_sn, int64_t x_si, void* out, void* stream) {
  AHIP_REQUIRE(outer >= 0 && n >= 0 && inner >= 0, "negative extent");
  if (outer == 0 || n == 0 || inner == 0) return AHIP_OK;
  AHIP_REQUIRE(x && out, "null argument");
  CumArgs a{x, out, outer, n, inner, x_so, x_sn, x_si, mul ? 1 : 0};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: return run_cum<uint8_t>(a, s);
    case AHIP_I8: return run_cum<int8_t>(a, s);
    case AHIP_I16: return run_cum<int16_t>(a, s);
    case AHIP_U16: return run_cum<uint16_t>(a, s);
    case AHIP_I32: return run_cum<int32_t>(a, s);
    case AHIP_U32: return run_cum<uint32_t>(a, s);
    case AHIP_I64: return run_cum<int64_t>(a, s);
    case AHIP_U64: return run_cum<uint64_t>(a, s);
    case AHIP_F32: return run_cum<float>(a, s);
    case AHIP_F64: return run_cum<double>(a, s);
    default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
  }
}

}  // extern "C"
