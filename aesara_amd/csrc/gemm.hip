// K4/K6 — GEMM on the CDNA4 matrix cores (gfx950), fp32 and fp64.
//
// Replaces the reference's tensor/blas.py:518 GemmRelated / :872 Gemm (C thunk -> sgemm_/dgemm_
// :767-820), :1659 Dot22, :1954 Dot22Scalar, :2179 BatchedDot and the NumPy fallback BLAS
// tensor/c_code/alt_blas_template.c.
//
// Design (MI355X-first, not a BLAS translation):
//   * exact-precision MFMA: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 (f32/f64 in, same
//     accumulate) — results are an ordinary k-ordered fma chain, so parity with a CPU BLAS is
//     roundoff-level; there is no TF32-like shortcut on gfx950.
//   * 128x128 workgroup tile, 4 wavefronts (2x2), each wave owns a 64x64 sub-tile as 4x4 MFMA
//     fragments (64 accumulator VGPRs); K is consumed in 128-byte slabs (BK = 32 f32 / 16 f64).
//   * both operands are staged through LDS in a [row][k] image with 144-byte rows (128 B + 16 B
//     pad): fragments are fetched with one conflict-free ds_read_b64 per lane.  Operands whose
//     contiguous axis is k are staged with 16-byte loads straight into that image; operands
//     whose contiguous axis is m/n (the "N" layout of B, transposed views of A) are loaded with
//     coalesced 16-byte vectors and transposed VECxVEC in registers before the ds_write_b128 —
//     all 8 unit-stride layouts of the reference (blas.py:719 encode_strides_in_unit) run
//     without a copy; arbitrary strides / ragged sizes take the scalar-staging instantiation.
//   * global->register prefetch of slab t+1 is issued before the MFMAs of slab t; two LDS
//     buffers, one barrier per slab.
//   * workgroup ids are remapped so the 8 XCDs (private L2s) each walk a contiguous band of
//     tiles (speed only, placement-independent).
#include <type_traits>

#include "common.h"

namespace {

template <typename T> struct Traits;
template <> struct Traits<float> {
  static constexpr int VEC = 4;    // elements per 16-byte vector
  static constexpr int BK = 32;    // k extent of one LDS slab (128 bytes)
  using vec_t = float __attribute__((ext_vector_type(4)));
  using acc_t = float __attribute__((ext_vector_type(4)));
};
template <> struct Traits<double> {
  static constexpr int VEC = 2;
  static constexpr int BK = 16;
  using vec_t = double __attribute__((ext_vector_type(2)));
  using acc_t = double __attribute__((ext_vector_type(4)));
};

constexpr int BM = 128, BN = 128;
constexpr int ROW_BYTES = 144;  // 128 B of k + 16 B pad -> conflict-free ds_read_b64 fragments
constexpr int THREADS = 256;

struct GemmArgs {
  int64_t M, N, K;
  const void* A; int64_t a_bs, a_rs, a_cs;
  const void* B; int64_t b_bs, b_rs, b_cs;
  const void* Cin; int64_t ci_bs, ci_rs, ci_cs;
  void* C; int64_t c_bs, c_rs, c_cs;
  double alpha, beta;
  int tiles_m, tiles_n;
};

// ---- staging ---------------------------------------------------------------------------
// MODE 0: k is the contiguous axis (unit stride along k, vectorisable)
// MODE 1: the row axis (m for A, n for B) is contiguous (register transpose)
// MODE 2: arbitrary strides / ragged extents (scalar loads, zero fill)
template <typename T, int MODE>
struct Stager {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  static constexpr int VEC = Tr::VEC;
  static constexpr int BK = Tr::BK;
  static constexpr int NV = 4;  // 16-byte vectors per thread per slab (128 rows * 8 vec / 256)
  vec_t r[NV];

  // base points at element (row0, 0) of the operand; rs = stride between rows (m or n),
  // ks = stride along k.  rows/kmax are the remaining extents for predication.
  __device__ __forceinline__ void load(const T* __restrict__ base, int64_t rs, int64_t ks,
                                       int64_t k0, int rows, int64_t K, int tid) {
    // interior slab (wave-uniform test): unconditional loads, no per-vector branches
    const bool full = (rows >= 128) && (k0 + BK <= K);
    if constexpr (MODE == 0) {
      if (full) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          int v = tid + THREADS * j;
          r[j] = *reinterpret_cast<const vec_t*>(base + (v >> 3) * rs + k0 + (v & 7) * VEC);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int v = tid + THREADS * j;
        int row = v >> 3, kv = v & 7;
        int64_t k = k0 + kv * VEC;
        vec_t val = 0;
        if (row < rows && k < K) val = *reinterpret_cast<const vec_t*>(base + row * rs + k);
        r[j] = val;
      }
    } else if constexpr (MODE == 1) {
      // micro-block = VEC k-rows x VEC contiguous rows; NV/VEC micro-blocks per thread.
      // Lane -> micro-block map: 8 consecutive lanes walk k (kq = 0..7) at the same row group, so
      // each global row is still read in full 128-byte lines AND the transposed ds_write_b128 of
      // an 8-lane group lands in ONE 128-byte LDS row segment (the n-fastest map measured 52 %
      // of LDS cycles lost to 4-way bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).
      if (full) {
#pragma unroll
        for (int b = 0; b < NV / VEC; ++b) {
          int mb = tid + THREADS * b;
          int kq = mb & 7, nq = mb >> 3;  // 8 lanes share a row: conflict-free ds_write_b128
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            r[b * VEC + i] = *reinterpret_cast<const vec_t*>(base + (k0 + kq * VEC + i) * ks +
                                                             nq * VEC);
        }
        return;
      }
#pragma unroll
      for (int b = 0; b < NV / VEC; ++b) {
        int mb = tid + THREADS * b;
        int kq = mb & 7, nq = mb >> 3;  // 8 lanes share a row: conflict-free ds_write_b128
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          int64_t k = k0 + kq * VEC + i;
          int row = nq * VEC;
          vec_t val = 0;
          if (row < rows && k < K) val = *reinterpret_cast<const vec_t*>(base + k * ks + row);
          r[b * VEC + i] = val;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int v = tid + THREADS * j;
        int row = v >> 3, kv = v & 7;
        vec_t val = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int64_t k = k0 + kv * VEC + e;
          if (row < rows && k < K) val[e] = base[row * rs + k * ks];
        }
        r[j] = val;
      }
    }
  }

  __device__ __forceinline__ void store(char* lds, int tid) const {
    if constexpr (MODE == 1) {
#pragma unroll
      for (int b = 0; b < NV / VEC; ++b) {
        int mb = tid + THREADS * b;
        int kq = mb & 7, nq = mb >> 3;  // 8 lanes share a row: conflict-free ds_write_b128
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) {
          vec_t t;
#pragma unroll
          for (int i = 0; i < VEC; ++i) t[i] = r[b * VEC + i][jj];
          *reinterpret_cast<vec_t*>(lds + (nq * VEC + jj) * ROW_BYTES + kq * 16) = t;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int v = tid + THREADS * j;
        int row = v >> 3, kv = v & 7;
        *reinterpret_cast<vec_t*>(lds + row * ROW_BYTES + kv * 16) = r[j];
      }
    }
  }
};

// ---- MFMA wrappers -------------------------------------------------------------------------
__device__ __forceinline__ void mma(Traits<float>::acc_t& c, float a, float b) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void mma(Traits<double>::acc_t& c, double a, double b) {
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// row of accumulator register r for lane l inside a 16x16 fragment
template <typename T> __device__ __forceinline__ int frag_row(int lane, int r);
template <> __device__ __forceinline__ int frag_row<float>(int lane, int r) {
  return (lane >> 4) * 4 + r;
}
template <> __device__ __forceinline__ int frag_row<double>(int lane, int r) {
  return (lane >> 4) + 4 * r;
}

template <typename T, int AMODE, int BMODE>
__global__ __launch_bounds__(THREADS, 2) void gemm_kernel(GemmArgs g) {
  using Tr = Traits<T>;
  using acc_t = typename Tr::acc_t;
  constexpr int BK = Tr::BK;
  constexpr int SLAB = (BM + BN) * ROW_BYTES;  // bytes per LDS buffer (A rows then B rows)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;  // 2x2 waves, 64x64 each

  // XCD-aware bijective remap of the linear workgroup id (speed only), then a grouped
  // row-major walk (8 tile-rows per group) for L2 reuse of the B panels.
  const int nwg = g.tiles_m * g.tiles_n;
  int bid = blockIdx.x;
  {
    int q = nwg / AHIP_NUM_XCD, rr = nwg % AHIP_NUM_XCD;
    int xcd = bid % AHIP_NUM_XCD;
    int base = (xcd < rr) ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q;
    bid = base + bid / AHIP_NUM_XCD;
  }
  constexpr int GROUP = 8;
  int group_sz = GROUP * g.tiles_n;
  int gid = bid / group_sz;
  int first_m = gid * GROUP;
  int gm = min(g.tiles_m - first_m, GROUP);
  int tm = first_m + (bid % group_sz) % gm;
  int tn = (bid % group_sz) / gm;

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t z = blockIdx.z;
  const T* A = static_cast<const T*>(g.A) + z * g.a_bs + m0 * g.a_rs;
  const T* B = static_cast<const T*>(g.B) + z * g.b_bs + n0 * g.b_cs;
  const int rows_a = (int)((g.M - m0) < BM ? (g.M - m0) : BM);
  const int rows_b = (int)((g.N - n0) < BN ? (g.N - n0) : BN);

  acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;

  Stager<T, AMODE> sa;
  Stager<T, BMODE> sb;
  const int nslab = (int)((g.K + BK - 1) / BK);

  if (nslab > 0) {
    sa.load(A, g.a_rs, g.a_cs, 0, rows_a, g.K, tid);
    sb.load(B, g.b_cs, g.b_rs, 0, rows_b, g.K, tid);
    sa.store(smem, tid);
    sb.store(smem + BM * ROW_BYTES, tid);
  }
  __syncthreads();

  // per-lane fragment addressing: lane (i = l & 15, kg = l >> 4) reads 8 bytes at
  // row*144 + kstep*32 + kg*8  (f32: k = 8*kstep + 2*kg + {0,1}; f64: k = 4*kstep + kg)
  const int frag_off = (lane & 15) * ROW_BYTES + (lane >> 4) * 8;

  // fp32 only: two-level summation.  Every FLUSH slabs (256 k) the MFMA accumulators are added
  // into a second accumulator set and cleared, so rounding error grows like
  // sqrt(256) + sqrt(K/256) instead of sqrt(K): at K = 4096 the Frobenius error vs an fp64
  // product drops from 1.15e-6 to the blocked-BLAS class (the reference's OpenBLAS sgemm also
  // sums in blocks), keeping parity inside the 1e-6 bar.  fp64 needs no such help.
  constexpr bool TWO_LEVEL = (sizeof(T) == 4);
  constexpr int FLUSH = 8;
  acc_t acc2[TWO_LEVEL ? 4 : 1][TWO_LEVEL ? 4 : 1];
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[i][j] = 0;
  }

  for (int t = 0; t < nslab; ++t) {
    const char* bufA = smem + (t & 1) * SLAB;
    const char* bufB = bufA + BM * ROW_BYTES;
    const bool more = (t + 1 < nslab);
    if (more) {
      sa.load(A, g.a_rs, g.a_cs, (int64_t)(t + 1) * BK, rows_a, g.K, tid);
      sb.load(B, g.b_cs, g.b_rs, (int64_t)(t + 1) * BK, rows_b, g.K, tid);
    }
    {
      // LDS fragments are double-buffered in registers: the ds_read_b64s of k-step ks+1 are
      // issued before the MFMAs of k-step ks, so the matrix pipe never waits on LDS latency
      using frag_t = typename std::conditional<sizeof(T) == 4,
                                               float __attribute__((ext_vector_type(2))), double>::type;
      constexpr int NB = sizeof(T) == 4 ? 2 : 1;  // fp64: 128 accumulator VGPRs leave no room
      frag_t fa[NB][4], fb[NB][4];
      const char* pa = bufA + wm * 64 * ROW_BYTES + frag_off;
      const char* pb = bufB + wn * 64 * ROW_BYTES + frag_off;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[0][i] = *reinterpret_cast<const frag_t*>(pa + i * 16 * ROW_BYTES);
        fb[0][i] = *reinterpret_cast<const frag_t*>(pb + i * 16 * ROW_BYTES);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = (NB == 2) ? (ks & 1) : 0, nxt = cur ^ 1;
        if constexpr (NB == 1) {
          if (ks > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              fa[0][i] = *reinterpret_cast<const frag_t*>(pa + i * 16 * ROW_BYTES + ks * 32);
              fb[0][i] = *reinterpret_cast<const frag_t*>(pb + i * 16 * ROW_BYTES + ks * 32);
            }
          }
        } else if (ks + 1 < 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            fa[nxt][i] = *reinterpret_cast<const frag_t*>(pa + i * 16 * ROW_BYTES + (ks + 1) * 32);
            fb[nxt][i] = *reinterpret_cast<const frag_t*>(pb + i * 16 * ROW_BYTES + (ks + 1) * 32);
          }
        }
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma(acc[i][j], fa[cur][i].x, fb[cur][j].x);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma(acc[i][j], fa[cur][i].y, fb[cur][j].y);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) mma(acc[i][j], fa[cur][i], fb[cur][j]);
        }
      }
    }
    if (more) {
      char* nb = smem + ((t + 1) & 1) * SLAB;
      sa.store(nb, tid);
      sb.store(nb + BM * ROW_BYTES, tid);
    }
    __syncthreads();
    if constexpr (TWO_LEVEL) {
      if ((t % FLUSH) == FLUSH - 1 && nslab > FLUSH) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc2[i][j] += acc[i][j];
            acc[i][j] = 0;
          }
      }
    }
  }
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += acc2[i][j];
  }

  // ---- epilogue: C = alpha*acc + beta*Cin ----------------------------------------------
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  T* C = static_cast<T*>(g.C) + z * g.c_bs;
  const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
  const bool use_cin = (g.beta != 0.0);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = n0 + wn * 64 + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wm * 64 + i * 16 + frag_row<T>(lane, r);
        if (row < g.M && col < g.N) {
          T v = alpha * acc[i][j][r];
          if (use_cin) v += beta * Cin[row * g.ci_rs + col * g.ci_cs];
          C[row * g.c_rs + col * g.c_cs] = v;
        }
      }
    }
}

// K == 0 or degenerate: C = beta*Cin
template <typename T>
__global__ void scale_kernel(GemmArgs g) {
  int64_t n = g.M * g.N;
  const int64_t z = blockIdx.z;
  T* C = static_cast<T*>(g.C) + z * g.c_bs;
  const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = i / g.N, c = i % g.N;
    T v = 0;
    if (g.beta != 0.0) v = (T)g.beta * Cin[r * g.ci_rs + c * g.ci_cs];
    C[r * g.c_rs + c * g.c_cs] = v;
  }
}

template <typename T>
int operand_mode(const void* p, int64_t rs, int64_t ks, int64_t rows, int64_t K, int64_t bs,
                 int64_t batch) {
  constexpr int VEC = Traits<T>::VEC;
  bool aligned = (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (batch <= 1 || bs % VEC == 0);
  if (ks == 1 && aligned && rs % VEC == 0 && K % VEC == 0) return 0;
  if (rs == 1 && aligned && ks % VEC == 0 && rows % VEC == 0) return 1;
  return 2;
}

template <typename T, int AM, int BMd>
int launch_gemm(const GemmArgs& g, int64_t batch, hipStream_t s) {
  constexpr size_t lds = 2 * (BM + BN) * ROW_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, AM, BMd>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      ahip_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return AHIP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)batch);
  AHIP_LAUNCH((gemm_kernel<T, AM, BMd>), grid, dim3(THREADS), lds, s, g);
  return AHIP_OK;
}

template <typename T>
int gemm_dispatch(GemmArgs& g, int64_t batch, hipStream_t s) {
  if (g.M == 0 || g.N == 0 || batch == 0) return AHIP_OK;
  if (g.K == 0 || g.alpha == 0.0) {
    int64_t n = g.M * g.N;
    unsigned blocks = (unsigned)(((n + 255) / 256) < 4096 ? ((n + 255) / 256) : 4096);
    AHIP_LAUNCH((scale_kernel<T>), dim3(blocks, 1, (unsigned)batch), dim3(256), 0, s, g);
    return AHIP_OK;
  }
  g.tiles_m = (int)((g.M + BM - 1) / BM);
  g.tiles_n = (int)((g.N + BN - 1) / BN);
  AHIP_REQUIRE((int64_t)g.tiles_m * g.tiles_n < (1LL << 31), "too many tiles");
  int am = operand_mode<T>(g.A, g.a_rs, g.a_cs, g.M, g.K, g.a_bs, batch);
  int bm = operand_mode<T>(g.B, g.b_cs, g.b_rs, g.N, g.K, g.b_bs, batch);
  switch (am * 3 + bm) {
    case 0: return launch_gemm<T, 0, 0>(g, batch, s);
    case 1: return launch_gemm<T, 0, 1>(g, batch, s);
    case 2: return launch_gemm<T, 0, 2>(g, batch, s);
    case 3: return launch_gemm<T, 1, 0>(g, batch, s);
    case 4: return launch_gemm<T, 1, 1>(g, batch, s);
    case 5: return launch_gemm<T, 1, 2>(g, batch, s);
    case 6: return launch_gemm<T, 2, 0>(g, batch, s);
    case 7: return launch_gemm<T, 2, 1>(g, batch, s);
    default: return launch_gemm<T, 2, 2>(g, batch, s);
  }
}

double host_scalar(int dtype, const void* p) {
  return dtype == AHIP_F32 ? (double)*static_cast<const float*>(p)
                           : *static_cast<const double*>(p);
}

}  // namespace

extern "C" {

int ahip_gemm_batched(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K,
                      const void* alpha, const void* A, int64_t a_bs, int64_t a_rs, int64_t a_cs,
                      const void* B, int64_t b_bs, int64_t b_rs, int64_t b_cs, const void* beta,
                      const void* Cin, int64_t ci_bs, int64_t ci_rs, int64_t ci_cs, void* C,
                      int64_t c_bs, int64_t c_rs, int64_t c_cs, void* stream) {
  AHIP_REQUIRE(dtype == AHIP_F32 || dtype == AHIP_F64, "gemm supports float32/float64 only");
  AHIP_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "negative extent");
  AHIP_REQUIRE(alpha && beta, "null alpha/beta");
  AHIP_REQUIRE(batch < 65536, "batch too large for grid.z");
  GemmArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.a_bs = a_bs; g.a_rs = a_rs; g.a_cs = a_cs;
  g.B = B; g.b_bs = b_bs; g.b_rs = b_rs; g.b_cs = b_cs;
  g.alpha = host_scalar(dtype, alpha);
  g.beta = host_scalar(dtype, beta);
  g.Cin = (g.beta != 0.0) ? Cin : C; g.ci_bs = ci_bs; g.ci_rs = ci_rs; g.ci_cs = ci_cs;
  g.C = C; g.c_bs = c_bs; g.c_rs = c_rs; g.c_cs = c_cs;
  g.tiles_m = g.tiles_n = 0;
  if (M > 0 && N > 0 && batch > 0) {
    AHIP_REQUIRE(C != nullptr, "null C");
    AHIP_REQUIRE(K == 0 || (A && B), "null A/B");
    AHIP_REQUIRE(g.beta == 0.0 || Cin != nullptr, "beta != 0 needs Cin");
  }
  return dtype == AHIP_F32 ? gemm_dispatch<float>(g, batch, as_stream(stream))
                           : gemm_dispatch<double>(g, batch, as_stream(stream));
}

int ahip_gemm(int dtype, int64_t M, int64_t N, int64_t K, const void* alpha, const void* A,
              int64_t a_rs, int64_t a_cs, const void* B, int64_t b_rs, int64_t b_cs,
              const void* beta, const void* Cin, int64_t ci_rs, int64_t ci_cs, void* C,
              int64_t c_rs, int64_t c_cs, void* stream) {
  return ahip_gemm_batched(dtype, 1, M, N, K, alpha, A, 0, a_rs, a_cs, B, 0, b_rs, b_cs, beta,
                           Cin, 0, ci_rs, ci_cs, C, 0, c_rs, c_cs, stream);
}

}  // extern "C"
