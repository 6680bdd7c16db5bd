// K7/K8 — fill and strided copy / set / inc.
//
// Replaces tensor/basic.py:1389 Alloc (perform :1427; C PyArray_CopyInto broadcast :1441-1490),
// tensor/subtensor.py:1454 IncSubtensor (perform :1556), materialisation of DimShuffle views
// (tensor/c_code/dimshuffle.c), compile/ops.py:149 DeepCopyOp and tensor/basic.py:2142 Join.
// Views themselves are stride arithmetic on the host and never reach this file.
#include "common.h"

namespace {

struct CopyArgs {
  int64_t items;             // number of (vector) items
  int nd;
  int64_t shape[AHIP_MAXD];  // innermost extent counted in vectors
  int64_t ss[AHIP_MAXD];     // element strides (innermost already multiplied by VEC)
  int64_t ds[AHIP_MAXD];
  const void* src;
  void* dst;
};
AHIP_PTRS_BEGIN(CopyArgs) AHIP_PTR1(src) AHIP_PTR1(dst) AHIP_PTRS_END

template <typename T, int VEC> struct alignas(sizeof(T) * VEC) Pack { T v[VEC]; };

template <typename T> __device__ __forceinline__ T acc_add(T a, T b) { return a + b; }
template <> __device__ __forceinline__ bool acc_add<bool>(bool a, bool b) { return a || b; }

// NT: the source is read once and is too large to stay in the memory-side cache (the streaming
// policy of the generated kernels, exec_elemwise.BIG_STREAM): non-temporal 16-byte loads
template <typename T, int VEC, bool ACC, bool NT = false>
__global__ __launch_bounds__(256) void copy_kernel(CopyArgs a) {
  const T* __restrict__ src = static_cast<const T*>(a.src);
  T* __restrict__ dst = static_cast<T*>(a.dst);
  for (int64_t item = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; item < a.items;
       item += (int64_t)gridDim.x * blockDim.x) {
    int64_t rem = item, so = 0, dof = 0;
#pragma unroll 1
    for (int d = a.nd - 1; d > 0; --d) {
      int64_t q = rem / a.shape[d];
      int64_t r = rem - q * a.shape[d];
      so += r * a.ss[d];
      dof += r * a.ds[d];
      rem = q;
    }
    so += rem * a.ss[0];
    dof += rem * a.ds[0];
    if constexpr (VEC == 1) {
      T v = src[so];
      if constexpr (ACC) v = acc_add<T>(dst[dof], v);
      dst[dof] = v;
    } else {
      using P = Pack<T, VEC>;
      P v;
      if constexpr (NT && sizeof(P) == 16) {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 raw = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + so));
        v = __builtin_bit_cast(P, raw);
      } else {
        v = *reinterpret_cast<const P*>(src + so);
      }
      if constexpr (ACC) {
        P o = *reinterpret_cast<const P*>(dst + dof);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v.v[e] = acc_add<T>(o.v[e], v.v[e]);
      }
      *reinterpret_cast<P*>(dst + dof) = v;
    }
  }
}

struct FillArgs { void* dst; int64_t n; uint64_t bits; };
AHIP_PTRS_BEGIN(FillArgs) AHIP_PTR1(dst) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void fill_kernel(FillArgs f) {
  T* __restrict__ dst = static_cast<T*>(f.dst);
  const T value = (T)f.bits;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < f.n;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = value;
}

int64_t g_copy_stream_bytes = 96LL << 20;   // copies of at least this many bytes read their source with streaming loads

unsigned stream_grid(int64_t items) {
  int64_t want = (items + 255) / 256;
  int64_t cap = (int64_t)ahip_cu_count() * 8;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  return (unsigned)want;
}

template <typename T, bool ACC>
int launch_copy(CopyArgs& a, int vec, hipStream_t s) {
  unsigned grid = stream_grid(a.items);
  constexpr int MAXV = 16 / sizeof(T);
  if (vec == MAXV && MAXV > 1 && !ACC && a.items * 16 >= g_copy_stream_bytes)
    AHIP_LAUNCH((copy_kernel<T, MAXV, ACC, true>), dim3(grid), dim3(256), 0, s, a);
  else if (vec == MAXV && MAXV > 1)
    AHIP_LAUNCH((copy_kernel<T, MAXV, ACC>), dim3(grid), dim3(256), 0, s, a);
  else
    AHIP_LAUNCH((copy_kernel<T, 1, ACC>), dim3(grid), dim3(256), 0, s, a);
  return AHIP_OK;
}

template <typename T>
int copy_typed(CopyArgs& a, int vec, int accumulate, hipStream_t s) {
  return accumulate ? launch_copy<T, true>(a, vec, s) : launch_copy<T, false>(a, vec, s);
}

// ---- ARange: np.arange's fill rule (numpy/core/src/multiarray/arraytypes.c.src @TYPE@_fill):
// out[0] = first = dtype(start), out[1] = next = dtype(start + step), out[i] = first + i * delta
// with delta = next - first, the product and the sum each ROUNDED in the output dtype (no fused
// multiply-add: NumPy's baseline build has none there) ----
struct ArangeArgs { void* dst; int64_t n; double fstart, fdelta, fnext; int64_t istart, istep; };
AHIP_PTRS_BEGIN(ArangeArgs) AHIP_PTR1(dst) AHIP_PTRS_END

template <typename T, bool FLT>
__global__ __launch_bounds__(256) void arange_kernel(ArangeArgs a) {
  T* __restrict__ dst = static_cast<T*>(a.dst);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (FLT) {
#pragma clang fp contract(off)   // HIP's __fmul_rn / __fadd_rn are plain * and +: only this keeps the FMA out
      T v;
      if (i == 0) v = (T)a.fstart;
      else if (i == 1) v = (T)a.fnext;
      else { const T prod = (T)i * (T)a.fdelta; v = (T)a.fstart + prod; }
      dst[i] = v;
    } else dst[i] = (T)(a.istart + i * a.istep);
  }
}

}  // namespace

void ahip_copy_set_stream_bytes(int64_t v) { g_copy_stream_bytes = v; }

extern "C" {

int ahip_copy_strided(int dtype, int nd, const int64_t* shape, const void* src,
                      const int64_t* sstrides, void* dst, const int64_t* dstrides, int accumulate,
                      void* stream) {
  int isz = ahip_itemsize(dtype);
  AHIP_REQUIRE(isz > 0, "bad dtype %d", dtype);
  AHIP_REQUIRE(nd >= 0 && nd <= 32, "bad nd");
  // 1. drop size-1 dims, detect empty
  int64_t sh[32], ss[32], ds[32];
  int n = 0;
  int64_t total = 1;
  for (int d = 0; d < nd; ++d) {
    AHIP_REQUIRE(shape[d] >= 0, "negative extent");
    total *= shape[d];
    if (shape[d] == 1) continue;
    sh[n] = shape[d]; ss[n] = sstrides[d]; ds[n] = dstrides[d]; ++n;
  }
  if (total == 0) return AHIP_OK;
  AHIP_REQUIRE(src && dst, "null pointer");
  // 2. merge adjacent dims that are jointly contiguous on both sides
  int m = 0;
  for (int d = 0; d < n; ++d) {
    if (m > 0 && ss[m - 1] == ss[d] * sh[d] && ds[m - 1] == ds[d] * sh[d]) {
      sh[m - 1] *= sh[d]; ss[m - 1] = ss[d]; ds[m - 1] = ds[d];
    } else {
      sh[m] = sh[d]; ss[m] = ss[d]; ds[m] = ds[d]; ++m;
    }
  }
  if (m == 0) { sh[0] = 1; ss[0] = 1; ds[0] = 1; m = 1; }
  AHIP_REQUIRE(m <= AHIP_MAXD, "more than %d non-mergeable dims", AHIP_MAXD);
  // 3. vector width: 16-byte packs when the innermost dim is unit-stride on both sides
  int vec = 1;
  int maxv = 16 / isz;
  if (maxv > 1 && ss[m - 1] == 1 && ds[m - 1] == 1 && sh[m - 1] % maxv == 0 &&
      reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0) {
    bool ok = true;
    for (int d = 0; d < m - 1; ++d) ok = ok && ss[d] % maxv == 0 && ds[d] % maxv == 0;
    if (ok) vec = maxv;
  }
  CopyArgs a;
  memset(&a, 0, sizeof(a));
  a.nd = m;
  for (int d = 0; d < m; ++d) { a.shape[d] = sh[d]; a.ss[d] = ss[d]; a.ds[d] = ds[d]; }
  if (vec > 1) { a.shape[m - 1] /= vec; a.ss[m - 1] = vec; a.ds[m - 1] = vec; }
  a.items = total / vec;
  a.src = src; a.dst = dst;
  hipStream_t s = as_stream(stream);
  if (!accumulate) {  // bit copies by item size
    switch (isz) {
      case 1: return launch_copy<uint8_t, false>(a, vec, s);
      case 2: return launch_copy<uint16_t, false>(a, vec, s);
      case 4: return launch_copy<uint32_t, false>(a, vec, s);
      default: return launch_copy<uint64_t, false>(a, vec, s);
    }
  }
  switch (dtype) {
    case AHIP_BOOL: return launch_copy<bool, true>(a, vec, s);
    case AHIP_I8: return launch_copy<int8_t, true>(a, vec, s);
    case AHIP_I16: return launch_copy<int16_t, true>(a, vec, s);
    case AHIP_I32: return launch_copy<int32_t, true>(a, vec, s);
    case AHIP_I64: return launch_copy<int64_t, true>(a, vec, s);
    case AHIP_U8: return launch_copy<uint8_t, true>(a, vec, s);
    case AHIP_U16: return launch_copy<uint16_t, true>(a, vec, s);
    case AHIP_U32: return launch_copy<uint32_t, true>(a, vec, s);
    case AHIP_U64: return launch_copy<uint64_t, true>(a, vec, s);
    case AHIP_F32: return launch_copy<float, true>(a, vec, s);
    default: return launch_copy<double, true>(a, vec, s);
  }
}

int ahip_fill(int dtype, const void* value, void* dst, int64_t n, void* stream) {
  int isz = ahip_itemsize(dtype);
  AHIP_REQUIRE(isz > 0 && value, "bad argument");
  AHIP_REQUIRE(n >= 0, "negative size");
  if (n == 0) return AHIP_OK;
  AHIP_REQUIRE(dst != nullptr, "null dst");
  hipStream_t s = as_stream(stream);
  unsigned grid = stream_grid(n);
  FillArgs f{dst, n, 0};
  memcpy(&f.bits, value, (size_t)isz);  // little-endian: low bytes hold the value
  switch (isz) {
    case 1: AHIP_LAUNCH((fill_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, f); break;
    case 2: AHIP_LAUNCH((fill_kernel<uint16_t>), dim3(grid), dim3(256), 0, s, f); break;
    case 4: AHIP_LAUNCH((fill_kernel<uint32_t>), dim3(grid), dim3(256), 0, s, f); break;
    default: AHIP_LAUNCH((fill_kernel<uint64_t>), dim3(grid), dim3(256), 0, s, f); break;
  }
  return AHIP_OK;
}

int ahip_arange(int dtype, const void* first, const void* delta, int64_t n, void* dst, void* stream) {
  const void* start = first; const void* step = delta;
  AHIP_REQUIRE(start && step && n >= 0, "bad argument");
  if (n == 0) return AHIP_OK;
  AHIP_REQUIRE(dst != nullptr, "null dst");
  hipStream_t s = as_stream(stream);
  unsigned grid = stream_grid(n);
  ArangeArgs a{dst, n, 0.0, 0.0, 0.0, 0, 0};
  auto ival = [&](const void* p) -> int64_t {
    switch (dtype) {
      case AHIP_BOOL: case AHIP_U8: return *static_cast<const uint8_t*>(p);
      case AHIP_I8: return *static_cast<const int8_t*>(p);
      case AHIP_I16: return *static_cast<const int16_t*>(p);
      case AHIP_U16: return *static_cast<const uint16_t*>(p);
      case AHIP_I32: return *static_cast<const int32_t*>(p);
      case AHIP_U32: return *static_cast<const uint32_t*>(p);
      default: return *static_cast<const int64_t*>(p);
    }
  };
  if (dtype == AHIP_F32) {
    const float st = *static_cast<const float*>(start), sp = *static_cast<const float*>(step);
    a.fstart = st; a.fdelta = sp; a.fnext = static_cast<const float*>(start)[1];
    AHIP_LAUNCH((arange_kernel<float, true>), dim3(grid), dim3(256), 0, s, a);
  } else if (dtype == AHIP_F64) {
    const double st = *static_cast<const double*>(start), sp = *static_cast<const double*>(step);
    a.fstart = st; a.fdelta = sp; a.fnext = static_cast<const double*>(start)[1];
    AHIP_LAUNCH((arange_kernel<double, true>), dim3(grid), dim3(256), 0, s, a);
  } else {
    a.istart = ival(start); a.istep = ival(step);
    switch (ahip_itemsize(dtype)) {
      case 1: AHIP_LAUNCH((arange_kernel<uint8_t, false>), dim3(grid), dim3(256), 0, s, a); break;
      case 2: AHIP_LAUNCH((arange_kernel<uint16_t, false>), dim3(grid), dim3(256), 0, s, a); break;
      case 4: AHIP_LAUNCH((arange_kernel<uint32_t, false>), dim3(grid), dim3(256), 0, s, a); break;
      case 8: AHIP_LAUNCH((arange_kernel<uint64_t, false>), dim3(grid), dim3(256), 0, s, a); break;
      default: ahip_set_error("bad dtype %d", dtype); return AHIP_EINVAL;
    }
  }
  return AHIP_OK;
}

}  // extern "C"
