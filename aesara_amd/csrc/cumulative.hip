// K12 — cumulative sum / product along one axis.
//
// Replaces tensor/extra_ops.py:283 CumOp (perform :311: np.cumsum / np.cumprod; C :330-370
// PyArray_CumSum / PyArray_CumProd).  The input is viewed as [outer, n, inner] (element strides
// so, sn, si); the output is C-contiguous.  Two line schedules:
//   * inner >= 64: one thread per (outer, inner) line, sequential over n — adjacent threads touch
//     adjacent addresses, so every step is one coalesced row of `inner` elements;
//   * inner small (the last-axis case): one wavefront per line, 64 elements per step scanned with
//     a shuffle-up Hillis-Steele pass plus a running carry.
// Few long lines (a flat cumsum; axis 0 of a matrix) would leave the device idle, so the scanned
// axis is cut into chunks (reduce-then-scan): pass 1 reduces every chunk, pass 2 scans the chunk
// totals in place (the same line kernels on the [outer, nchunks, inner] totals), pass 3 scans
// every chunk starting from the total of the chunks before it.  The input is read twice and
// written once; the chunk count only depends on the shape, so results are reproducible.
// Integer arithmetic wraps in the output dtype (as the reference's C loops do); floating-point
// sums are accumulated in scan order within a chunk (round-off level reordering only).
#include "common.h"

namespace {

struct CumArgs {
  const void* x; void* out;
  int64_t outer, n, inner, so, sn, si;
  int mul;
  // chunked form: lines are (outer, chunk, inner); `carry` holds the inclusive scan of the chunk
  // totals [outer, nchunks, inner] (chunk c starts from carry[c - 1]); `totals` is pass 1's output
  int64_t chunk, nchunks;
  const void* carry; void* totals;
};
AHIP_PTRS_BEGIN(CumArgs) AHIP_PTR1(x) AHIP_PTR1(out) AHIP_PTR1(carry) AHIP_PTR1(totals) AHIP_PTRS_END

template <typename T> __device__ __forceinline__ T comb(T a, T b, int mul) { return mul ? a * b : a + b; }

template <typename T, bool TOTALS>
__global__ __launch_bounds__(256) void cum_lines_kernel(CumArgs a) {
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const T* __restrict__ carry = static_cast<const T*>(a.carry);
  T* __restrict__ totals = static_cast<T*>(a.totals);
  const int64_t lines = a.outer * a.nchunks * a.inner;
  for (int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; l < lines;
       l += (int64_t)gridDim.x * blockDim.x) {
    const int64_t oc = l / a.inner, i = l - oc * a.inner;
    const int64_t o = oc / a.nchunks, c = oc - o * a.nchunks;
    const int64_t k0 = c * a.chunk, k1 = (k0 + a.chunk < a.n) ? k0 + a.chunk : a.n;
    const T* px = x + o * a.so + i * a.si;
    T* po = out + o * a.n * a.inner + i;
    T acc = a.mul ? (T)1 : (T)0;
    if (TOTALS) {
#pragma unroll 8
      for (int64_t k = k0; k < k1; ++k) acc = comb<T>(acc, px[k * a.sn], a.mul);
      totals[l] = acc;
    } else {
      if (carry != nullptr && c > 0) acc = carry[l - a.inner];
#pragma unroll 8
      for (int64_t k = k0; k < k1; ++k) {
        acc = comb<T>(acc, px[k * a.sn], a.mul);
        po[k * a.inner] = acc;
      }
    }
  }
}

// inner >= 64, unit stride along the inner axis: each thread owns V ADJACENT lines and moves them
// with 16-byte loads / stores (the scalar form above moves 4-8 bytes per lane per step and
// measured 2.5 TB/s on 8192 x 4096 fp32, axis 0).
template <typename T, bool TOTALS>
__global__ __launch_bounds__(256) void cum_lines_vec_kernel(CumArgs a) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) P { T v[V]; };
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const T* __restrict__ carry = static_cast<const T*>(a.carry);
  T* __restrict__ totals = static_cast<T*>(a.totals);
  const int64_t iv = a.inner / V;
  const int64_t lines = a.outer * a.nchunks * iv;
  for (int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; l < lines;
       l += (int64_t)gridDim.x * blockDim.x) {
    const int64_t oc = l / iv, i = (l - oc * iv) * V;
    const int64_t o = oc / a.nchunks, c = oc - o * a.nchunks;
    const int64_t k0 = c * a.chunk, k1 = (k0 + a.chunk < a.n) ? k0 + a.chunk : a.n;
    const T* px = x + o * a.so + i;
    T* po = out + o * a.n * a.inner + i;
    P acc;
#pragma unroll
    for (int e = 0; e < V; ++e) acc.v[e] = a.mul ? (T)1 : (T)0;
    const int64_t tl = (o * a.nchunks + c) * a.inner + i;      // index into totals / carry
    if (TOTALS) {
#pragma unroll 8
      for (int64_t k = k0; k < k1; ++k) {
        const P p = *reinterpret_cast<const P*>(px + k * a.sn);
#pragma unroll
        for (int e = 0; e < V; ++e) acc.v[e] = comb<T>(acc.v[e], p.v[e], a.mul);
      }
      *reinterpret_cast<P*>(totals + tl) = acc;
    } else {
      if (carry != nullptr && c > 0) acc = *reinterpret_cast<const P*>(carry + tl - a.inner);
#pragma unroll 8
      for (int64_t k = k0; k < k1; ++k) {
        const P p = *reinterpret_cast<const P*>(px + k * a.sn);
#pragma unroll
        for (int e = 0; e < V; ++e) acc.v[e] = comb<T>(acc.v[e], p.v[e], a.mul);
        *reinterpret_cast<P*>(po + k * a.inner) = acc;
      }
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
__device__ __forceinline__ T shfl_xor_(T v, int d) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_xor(u.i[0], d, 64); u.i[1] = __shfl_xor(u.i[1], d, 64);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v;
    u.i = __shfl_xor(u.i, d, 64);
    return u.t;
  } else {
    int w = (int)v;
    w = __shfl_xor(w, d, 64);
    return (T)w;
  }
}

template <typename T, bool TOTALS>
__global__ __launch_bounds__(256) void cum_wave_kernel(CumArgs a) {
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const T* __restrict__ carry_in = static_cast<const T*>(a.carry);
  T* __restrict__ totals = static_cast<T*>(a.totals);
  const int lane = threadIdx.x & 63;
  const int64_t lines = a.outer * a.nchunks * a.inner;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const T ident = a.mul ? (T)1 : (T)0;
  for (int64_t l = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); l < lines;
       l += nwaves) {
    const int64_t oc = l / a.inner, i = l - oc * a.inner;
    const int64_t o = oc / a.nchunks, c = oc - o * a.nchunks;
    const int64_t kbeg = c * a.chunk, kend = (kbeg + a.chunk < a.n) ? kbeg + a.chunk : a.n;
    const T* px = x + o * a.so + i * a.si;
    T* po = out + o * a.n * a.inner + i;
    if (TOTALS) {
      T acc = ident;
#pragma unroll 4
      for (int64_t k = kbeg + lane; k < kend; k += 64) acc = comb<T>(acc, px[k * a.sn], a.mul);
      for (int m = 32; m > 0; m >>= 1) acc = comb<T>(acc, shfl_xor_<T>(acc, m), a.mul);
      if (lane == 0) totals[l] = acc;
      continue;
    }
    T carry = (carry_in != nullptr && c > 0) ? carry_in[l - a.inner] : ident;
    // 4 x 64 elements per trip: the four loads are independent and issued together, the scans
    // chain through `carry`
    for (int64_t k0 = kbeg; k0 < kend; k0 += 256) {
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + 64 * u + lane;
        v[u] = (k < kend) ? px[k * a.sn] : ident;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + 64 * u + lane;
        T w = v[u];
        for (int d = 1; d < 64; d <<= 1) {
          const T up = shfl_up_<T>(w, d);
          if (lane >= d) w = comb<T>(up, w, a.mul);
        }
        w = comb<T>(carry, w, a.mul);
        if (k < kend) po[k * a.inner] = w;
        union { T t; int q[2]; } c63;   // carry = lane 63's value
        c63.q[0] = c63.q[1] = 0;
        c63.t = w;
        c63.q[0] = __shfl(c63.q[0], 63, 64);
        if (sizeof(T) == 8) c63.q[1] = __shfl(c63.q[1], 63, 64);
        carry = c63.t;
      }
    }
  }
}

// scan axis contiguous (sn == 1, the flat / last-axis case): a lane owns V CONSECUTIVE elements
// per trip (one 16-byte load, serial scan of the V values, shuffle scan of the lane totals), 4
// trips in flight.
template <typename T, bool TOTALS>
__global__ __launch_bounds__(256) void cum_wave_vec_kernel(CumArgs a) {
  constexpr int V = 16 / sizeof(T);
  struct alignas(16) P { T v[V]; };
  const T* __restrict__ x = static_cast<const T*>(a.x);
  T* __restrict__ out = static_cast<T*>(a.out);
  const T* __restrict__ carry_in = static_cast<const T*>(a.carry);
  T* __restrict__ totals = static_cast<T*>(a.totals);
  const int lane = threadIdx.x & 63;
  const int64_t lines = a.outer * a.nchunks * a.inner;
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const T ident = a.mul ? (T)1 : (T)0;
  for (int64_t l = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); l < lines;
       l += nwaves) {
    const int64_t oc = l / a.inner, i = l - oc * a.inner;
    const int64_t o = oc / a.nchunks, c = oc - o * a.nchunks;
    const int64_t kbeg = c * a.chunk, kend = (kbeg + a.chunk < a.n) ? kbeg + a.chunk : a.n;
    const T* px = x + o * a.so + i * a.si;      // sn == 1, inner == 1 on this path
    T* po = out + o * a.n * a.inner + i;
    if (TOTALS) {
      T acc = ident;
#pragma unroll 4
      for (int64_t k = kbeg + (int64_t)lane * V; k < kend; k += 64 * V) {
        const P p = *reinterpret_cast<const P*>(px + k);
#pragma unroll
        for (int e = 0; e < V; ++e) acc = comb<T>(acc, p.v[e], a.mul);
      }
      for (int m = 32; m > 0; m >>= 1) acc = comb<T>(acc, shfl_xor_<T>(acc, m), a.mul);
      if (lane == 0) totals[l] = acc;
      continue;
    }
    T carry = (carry_in != nullptr && c > 0) ? carry_in[l - a.inner] : ident;
    for (int64_t k0 = kbeg; k0 < kend; k0 += 4 * 64 * V) {
      P v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + (int64_t)(64 * u + lane) * V;
        if (k < kend) v[u] = *reinterpret_cast<const P*>(px + k);
        else
#pragma unroll
          for (int e = 0; e < V; ++e) v[u].v[e] = ident;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + (int64_t)(64 * u + lane) * V;
#pragma unroll
        for (int e = 1; e < V; ++e) v[u].v[e] = comb<T>(v[u].v[e - 1], v[u].v[e], a.mul);
        T w = v[u].v[V - 1];                       // lane total -> exclusive prefix over lanes
        for (int d = 1; d < 64; d <<= 1) {
          const T up = shfl_up_<T>(w, d);
          if (lane >= d) w = comb<T>(up, w, a.mul);
        }
        T excl = shfl_up_<T>(w, 1);
        excl = (lane == 0) ? carry : comb<T>(carry, excl, a.mul);
#pragma unroll
        for (int e = 0; e < V; ++e) v[u].v[e] = comb<T>(excl, v[u].v[e], a.mul);
        if (k < kend) *reinterpret_cast<P*>(po + k) = v[u];
        union { T t; int q[2]; } c63;
        c63.q[0] = c63.q[1] = 0;
        c63.t = comb<T>(carry, w, a.mul);
        c63.q[0] = __shfl(c63.q[0], 63, 64);
        if (sizeof(T) == 8) c63.q[1] = __shfl(c63.q[1], 63, 64);
        carry = c63.t;
      }
    }
  }
}

// chunk count for a problem (depends on the shape only); 1 = single pass
int64_t cum_chunks(int64_t outer, int64_t n, int64_t inner) {
  const int64_t lines = outer * inner;
  if (lines <= 0 || n <= 0) return 1;
  int64_t want, max_chunks;
  if (inner >= 64) {          // thread per line: fill ~1024 threads per CU, chunks of >= 64
    // (wide inner extents take the 16-byte form: a thread owns up to 4 adjacent lines)
    const int64_t eff = inner >= 256 ? (lines + 3) / 4 : lines;
    want = ((int64_t)ahip_cu_count() * 1024 + eff - 1) / eff;
    max_chunks = n / 64;
  } else {                    // wave per line: ~32 waves per CU, chunks of >= 1024
    want = ((int64_t)ahip_cu_count() * 32 + lines - 1) / lines;
    max_chunks = n / 1024;
  }
  if (want > max_chunks) want = max_chunks;
  return want < 2 ? 1 : want;
}

// chunk length: a multiple of 64 elements so that chunk starts keep the 16-byte alignment the
// vector kernels need (any dtype: 64 * itemsize is a multiple of 16)
int64_t chunk_len(int64_t n, int64_t nch) {
  const int64_t c = (n + nch - 1) / nch;
  return (c + 63) / 64 * 64;
}

template <typename T, bool TOTALS>
int launch_cum(const CumArgs& a, hipStream_t s) {
  const int64_t lines = a.outer * a.nchunks * a.inner;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  constexpr int V = 16 / (int)sizeof(T);
  auto al = [](const void* p) { return p == nullptr || reinterpret_cast<uintptr_t>(p) % 16 == 0; };
  const bool ptrs_ok = al(a.x) && al(a.out) && al(a.carry) && al(a.totals);
  if (a.inner >= 64) {
    const bool vec = ptrs_ok && a.si == 1 && a.inner % V == 0 && a.so % V == 0 && a.sn % V == 0 &&
                     a.inner / V >= 64;
    int64_t want = ((vec ? lines / V : lines) + 255) / 256;
    if (want > cap) want = cap;
    if (vec) AHIP_LAUNCH((cum_lines_vec_kernel<T, TOTALS>), dim3((unsigned)want), dim3(256), 0, s, a);
    else AHIP_LAUNCH((cum_lines_kernel<T, TOTALS>), dim3((unsigned)want), dim3(256), 0, s, a);
  } else {
    // lane-owns-V form: every chunk start / line start must stay 16-byte aligned
    const bool vec = ptrs_ok && a.inner == 1 && a.sn == 1 && a.chunk % V == 0 && a.n % V == 0 &&
                     a.so % V == 0 && a.n >= 64 * V;
    int64_t want = (lines + 3) / 4;
    if (want > cap) want = cap;
    if (vec) AHIP_LAUNCH((cum_wave_vec_kernel<T, TOTALS>), dim3((unsigned)want), dim3(256), 0, s, a);
    else AHIP_LAUNCH((cum_wave_kernel<T, TOTALS>), dim3((unsigned)want), dim3(256), 0, s, a);
  }
  return AHIP_OK;
}

template <typename T>
int run_cum(CumArgs& a, void* ws, size_t ws_bytes, hipStream_t s) {
  const int64_t nch = cum_chunks(a.outer, a.n, a.inner);
  a.chunk = a.n; a.nchunks = 1; a.carry = nullptr; a.totals = nullptr;
  if (nch == 1) return launch_cum<T, false>(a, s);
  a.chunk = chunk_len(a.n, nch);
  a.nchunks = (a.n + a.chunk - 1) / a.chunk;
  const size_t need = (size_t)(a.outer * a.nchunks * a.inner) * sizeof(T);
  AHIP_REQUIRE(ws != nullptr && ws_bytes >= need, "cumulative: workspace of %zu bytes needed", need);
  a.totals = ws;
  int rc = launch_cum<T, true>(a, s);                    // pass 1: chunk totals
  if (rc) return rc;
  // pass 2: inclusive scan of the totals along the chunk axis, in place
  CumArgs t{ws, ws, a.outer, a.nchunks, a.inner, a.nchunks * a.inner, a.inner, 1, a.mul,
            a.nchunks, 1, nullptr, nullptr};
  rc = launch_cum<T, false>(t, s);
  if (rc) return rc;
  a.carry = ws; a.totals = nullptr;                       // pass 3: scan with carry-in
  return launch_cum<T, false>(a, s);
}

}  // namespace

extern "C" {

size_t ahip_cumulative_ws_bytes(int dtype, int64_t outer, int64_t n, int64_t inner) {
  if (outer <= 0 || n <= 0 || inner <= 0) return 0;
  const int64_t nch = cum_chunks(outer, n, inner);
  if (nch == 1) return 0;
  const int64_t chunk = chunk_len(n, nch);
  return (size_t)(outer * ((n + chunk - 1) / chunk) * inner) * (size_t)ahip_itemsize(dtype);
}

int ahip_cumulative(int dtype, int mul, const void* x, int64_t outer, int64_t n, int64_t inner,
                    int64_t x_so, int64_t x_sn, int64_t x_si, void* out, void* ws,
                    size_t ws_bytes, void* stream) {
  AHIP_REQUIRE(outer >= 0 && n >= 0 && inner >= 0, "negative extent");
  if (outer == 0 || n == 0 || inner == 0) return AHIP_OK;
  AHIP_REQUIRE(x && out, "null argument");
  CumArgs a{x, out, outer, n, inner, x_so, x_sn, x_si, mul ? 1 : 0, n, 1, nullptr, nullptr};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: case AHIP_U8: return run_cum<uint8_t>(a, ws, ws_bytes, s);
    case AHIP_I8: return run_cum<int8_t>(a, ws, ws_bytes, s);
    case AHIP_I16: return run_cum<int16_t>(a, ws, ws_bytes, s);
    case AHIP_U16: return run_cum<uint16_t>(a, ws, ws_bytes, s);
    case AHIP_I32: return run_cum<int32_t>(a, ws, ws_bytes, s);
    case AHIP_U32: return run_cum<uint32_t>(a, ws, ws_bytes, s);
    case AHIP_I64: return run_cum<int64_t>(a, ws, ws_bytes, s);
    case AHIP_U64: return run_cum<uint64_t>(a, ws, ws_bytes, s);
    case AHIP_F32: return run_cum<float>(a, ws, ws_bytes, s);
    case AHIP_F64: return run_cum<double>(a, ws, ws_bytes, s);
    default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
  }
}

}  // extern "C"
