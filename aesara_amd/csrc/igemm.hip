// Integer / bool matrix products on the vector ALU (no matrix cores: there is no integer MFMA whose
// results wrap like NumPy's, and these products are rare on this path).
// replaces: tensor/math.py:1879 Dot.perform (np.dot) and tensor/blas.py:2224 BatchedDot.perform for
// bool / (u)int8..64 operands.  NumPy computes an integer product in the operands' own dtype with
// wrap-around (arithmetic mod 2^bits); wrap-around is a ring homomorphism, so accumulating in a
// wider unsigned type and truncating once gives the same bits: 8/16/32-bit types accumulate in
// uint32, 64-bit types in uint64.  bool is np.dot's OR of ANDs.
// C[b, m, n] = sum_k A[b, m, k] * B[b, k, n]; every operand addressed with explicit element strides
// (views, transposes, zero batch strides for broadcast operands: no copies).
// 64 x 64 output tile per 256-thread workgroup, 4 x 4 per thread, K in slabs of 16 through LDS.
#include "common.h"

namespace {

struct igemm_args {
  const void* A; const void* B; void* C;
  int64_t M, N, K;
  int64_t a_bs, a_rs, a_cs, b_bs, b_rs, b_cs, c_bs, c_rs, c_cs;
  int64_t mt, nt;          // tiles along M / N: blockIdx.x = (batch * mt + my) * nt + nx
};
AHIP_PTRS_BEGIN(igemm_args) AHIP_PTR1(A) AHIP_PTR1(B) AHIP_PTR1(C) AHIP_PTRS_END

constexpr int TM = 64, TN = 64, TK = 16;

template <typename T, typename Acc, bool IS_BOOL>
__global__ __launch_bounds__(256) void igemm_kernel(igemm_args a) {
  __shared__ Acc sA[TK][TM + 1];
  __shared__ Acc sB[TK][TN + 1];
  // one linear grid axis (up to 2^31 - 1 workgroups): neither the batch (a 2-d BatchedDot with
  // 100k rows is batch = 100k) nor the row tiles are held to gridDim.y/z's 65535
  const int64_t lin = blockIdx.x, nx = lin % a.nt, rest = lin / a.nt, my = rest % a.mt, bz = rest / a.mt;
  const T* __restrict__ A = (const T*)a.A + bz * a.a_bs;
  const T* __restrict__ B = (const T*)a.B + bz * a.b_bs;
  T* __restrict__ C = (T*)a.C + bz * a.c_bs;
  const int64_t m0 = my * TM, n0 = nx * TN;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 x 16 threads, 4 x 4 outputs each
  Acc acc[4][4] = {};
  for (int64_t k0 = 0; k0 < a.K; k0 += TK) {
    // stage A[m0:m0+64, k0:k0+16] and B[k0:k0+16, n0:n0+64] (1024 elements each, 4 per thread)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = threadIdx.x + 256 * i;
      {
        const int kk = e & 15, mm = e >> 4;                     // lanes along k
        const int64_t m = m0 + mm, k = k0 + kk;
        Acc v = 0;
        if (m < a.M && k < a.K) v = (Acc)A[m * a.a_rs + k * a.a_cs];
        sA[kk][mm] = IS_BOOL ? (Acc)(v != 0) : v;
      }
      {
        const int nn = e & 63, kk = e >> 6;                     // lanes along n
        const int64_t n = n0 + nn, k = k0 + kk;
        Acc v = 0;
        if (n < a.N && k < a.K) v = (Acc)B[k * a.b_rs + n * a.b_cs];
        sB[kk][nn] = IS_BOOL ? (Acc)(v != 0) : v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      Acc av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = sA[kk][ty * 4 + i]; bv[i] = sB[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (IS_BOOL) acc[i][j] |= av[i] & bv[j];
          else acc[i][j] += av[i] * bv[j];
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t m = m0 + ty * 4 + i, n = n0 + tx * 4 + j;
      if (m < a.M && n < a.N) C[m * a.c_rs + n * a.c_cs] = IS_BOOL ? (T)(acc[i][j] != 0) : (T)acc[i][j];
    }
}

template <typename T, typename Acc, bool IS_BOOL = false>
int run(const igemm_args& a, int64_t batch, hipStream_t s) {
  dim3 grid((unsigned)(a.nt * a.mt * batch));
  AHIP_LAUNCH((igemm_kernel<T, Acc, IS_BOOL>), grid, dim3(256), 0, s, a);
  return AHIP_OK;
}

}  // namespace

extern "C" int ahip_igemm_batched(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K,
                                  const void* A, int64_t a_bs, int64_t a_rs, int64_t a_cs,
                                  const void* B, int64_t b_bs, int64_t b_rs, int64_t b_cs, void* C,
                                  int64_t c_bs, int64_t c_rs, int64_t c_cs, void* stream) {
  AHIP_REQUIRE(batch >= 0 && M >= 0 && N >= 0 && K >= 0, "negative extent");
  if (batch == 0 || M == 0 || N == 0) return AHIP_OK;
  AHIP_REQUIRE(A && B && C, "null operand");
  const int64_t mt = (M + TM - 1) / TM, nt = (N + TN - 1) / TN;
  AHIP_REQUIRE((double)mt * (double)nt * (double)batch < 2147483647.0, "igemm: more than 2^31 - 1 output tiles");
  igemm_args a{A, B, C, M, N, K, a_bs, a_rs, a_cs, b_bs, b_rs, b_cs, c_bs, c_rs, c_cs, mt, nt};
  hipStream_t s = as_stream(stream);
  switch (dtype) {
    case AHIP_BOOL: return run<unsigned char, unsigned, true>(a, batch, s);
    case AHIP_I8: case AHIP_U8: return run<unsigned char, unsigned>(a, batch, s);
    case AHIP_I16: case AHIP_U16: return run<unsigned short, unsigned>(a, batch, s);
    case AHIP_I32: case AHIP_U32: return run<unsigned, unsigned>(a, batch, s);
    case AHIP_I64: case AHIP_U64: return run<unsigned long long, unsigned long long>(a, batch, s);
    default:
      ahip_set_error("ahip_igemm_batched: dtype %d is not an integer / bool type", dtype);
      return AHIP_EINVAL;
  }
}
