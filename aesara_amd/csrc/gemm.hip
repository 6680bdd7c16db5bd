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
//   * 128x128 workgroup tile, 8 wavefronts (2x4), each wave owns a 64x32 sub-tile as 4x2 MFMA
//     fragments; K is consumed in 128-byte slabs (BK = 32 f32 / 16 f64).  <= 128 VGPRs per
//     wave -> two workgroups = 16 waves per CU (4 per SIMD).  Measured on the 4-wave / 64x64
//     predecessor: 1/8 of the slab loads miss the 4 MiB XCD L2 by construction (8x8 concurrent
//     tiles per XCD), their latency exceeded one slab of MFMA time and the matrix pipe idled
//     26 % (SQ_WAIT_ANY ~1900 of ~5500 cycles per wave-slab).  Four waves per SIMD plus a
//     two-slab-deep register prefetch cover that latency.
//   * both operands are staged through LDS in a [row][k] image with 144-byte rows (128 B + 16 B
//     pad) whose 8-byte units are swapped pairwise in rows with bit 3 set (Stage::swz): the
//     compiler fetches fragments with ds_read2_b64 (16-lane groups over 32 banks), for which the
//     plain image made rows i / i+8 collide (rocprofv3 r02: SQ_LDS_BANK_CONFLICT /
//     SQ_LDS_IDX_ACTIVE = 0.43); with the swizzle the counter reads 0.00
//     (profiles/r02_gemm_pmc_summary.json; 4096^3 fp32 NN 115 -> 120, TN / TT
//     123 -> 126 TFLOP/s: the LDS was 34 % busy, not the limiter).  Waves 0-3 stage
//     A, waves 4-7 stage B (4 x 16-byte vectors per thread per slab) through per-thread 32-bit
//     offsets from a wave-uniform base (SGPR base + VGPR offset addressing: no per-slab address
//     arithmetic).  Operands whose contiguous axis is k go straight into the image; operands
//     whose contiguous axis is m/n (the "N" layout of B, transposed views of A) are loaded with
//     coalesced 16-byte vectors and transposed VECxVEC in registers before the ds_write_b128 —
//     all 8 unit-stride layouts of the reference (blas.py:719 encode_strides_in_unit) run
//     without a copy; ragged sizes take the predicated (EDGE) instantiation, arbitrary strides
//     the scalar-staging one.
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
  using frag_t = float __attribute__((ext_vector_type(2)));
};
template <> struct Traits<double> {
  static constexpr int VEC = 2;
  static constexpr int BK = 16;
  using vec_t = double __attribute__((ext_vector_type(2)));
  using acc_t = double __attribute__((ext_vector_type(4)));
  using frag_t = double;
};

constexpr int BM = 128, BN = 128;
// tile shapes: H = 0 the 128x128 / 8-wave tile described above; H = 1 its half-size sibling
// (64x64, 4 wavefronts as 2x2, 32x32 = 2x2 fragments per wave, waves 0-1 stage A, 2-3 stage B)
// for problems that give the big tile fewer workgroups than the chip has CUs
template <int H> struct Cfg {
  static constexpr int BM = H ? 64 : 128, BN = H ? 64 : 128;
  static constexpr int THREADS = H ? 256 : 512;
  static constexpr int STG = THREADS / 2;                    // threads staging one operand
  static constexpr int WN = H ? 2 : 4;                       // waves along n (2 along m)
  static constexpr int FM = BM / 2 / 16, FN = BN / WN / 16;  // 16x16 fragments per wave
};
constexpr int ROW_BYTES = 144;  // 128 B of k + 16 B pad (+ unit swizzle, Stage::swz) -> conflict-free fragments
constexpr int NV = 4;           // 16-byte vectors per staging thread per slab (128*8/256)

struct GemmArgs {
  int64_t M, N, K;
  const void* A; int64_t a_bs, a_rs, a_cs;
  const void* B; int64_t b_bs, b_rs, b_cs;
  const void* Cin; int64_t ci_bs, ci_rs, ci_cs;
  void* C; int64_t c_bs, c_rs, c_cs;
  double alpha, beta;
  int tiles_m, tiles_n;
  int group;             // tile-rows per group of the workgroup walk (L2 reuse of the B panels)
  int64_t a_lim, b_lim;  // bytes from A / B (per batch item) to the end of their valid extent
};
AHIP_PTRS_BEGIN(GemmArgs) AHIP_PTR1(A) AHIP_PTR1(B) AHIP_PTR1(Cin) AHIP_PTR1(C) AHIP_PTRS_END

void set_limits(GemmArgs& g, int64_t isz);

// ---- staging ---------------------------------------------------------------------------
// MODE 0: k is the contiguous axis (unit stride along k, vectorisable)
// MODE 1: the row axis (m for A, n for B) is contiguous (register transpose)
// MODE 2: arbitrary strides (scalar loads, zero fill; always predicated)
//
// `rs` = element stride between rows of the operand (m for A, n for B), `ks` = stride along k.
// A staging thread (stid in [0,256)) owns NV vectors per slab; their byte offsets from the
// wave-uniform slab base are loop invariant and kept as 32-bit VGPRs.
template <typename T, int MODE, int STG = 256>
struct Stage {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  static constexpr int VEC = Tr::VEC;

  // vector j of thread stid: (row, k) inside the 128 x BK slab
  static __device__ __forceinline__ void coords(int stid, int j, int& row, int& k) {
    if constexpr (MODE == 1) {
      // micro-block = VEC k-rows x VEC contiguous rows, NV/VEC micro-blocks per thread; 8
      // consecutive lanes walk k (kq) at the same row group: global rows are read in full
      // 128-byte lines and the transposed ds_write_b128 of an 8-lane group lands in ONE
      // 128-byte LDS row segment (conflict-free)
      int mb = stid + STG * (j / VEC);
      int kq = mb & 7, nq = mb >> 3;
      row = nq * VEC;
      k = kq * VEC + (j % VEC);
    } else {
      int v = stid + STG * j;
      row = v >> 3;
      k = (v & 7) * VEC;
    }
  }

  static __device__ __forceinline__ void offsets(unsigned (&off)[NV], int stid, int64_t rs,
                                                 int64_t ks) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      int row, k;
      coords(stid, j, row, k);
      off[j] = (unsigned)(((int64_t)row * rs + (int64_t)k * ks) * (int64_t)sizeof(T));
    }
  }

  // 16-byte load through a buffer resource: SGPR descriptor (wave-uniform base) + 32-bit VGPR
  // byte offset (guide T8) — no 64-bit per-lane addresses to compute, keep or spill
  static __device__ __forceinline__ vec_t bload(const char* base, unsigned off,
                                                int records = 0x7FFFFFFF) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, records, 0x00020000);
    u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
    return __builtin_bit_cast(vec_t, v);
  }

  // base = byte address of element (row0, k0) of the operand (wave uniform)
  // EM 0: interior tile, no checks.  EM 1: every vector predicated on (row, k).  EM 2 (ragged
  // M / N, K a multiple of BK): unpredicated loads through a descriptor whose num_records ends
  // at the operand's last valid byte — the hardware returns 0 beyond it, rows / columns past
  // M / N read neighbouring (valid) data that only reaches outputs which are never stored.
  template <int EM>
  static __device__ __forceinline__ void load(vec_t (&r)[NV], const char* __restrict__ base,
                                              const unsigned (&off)[NV], int stid, int64_t rs,
                                              int64_t ks, int rows, int64_t krem, int records) {
    if constexpr (MODE == 2) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int row, k;
        coords(stid, j, row, k);
        vec_t val = 0;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (row < rows && k + e < krem)
            val[e] = reinterpret_cast<const T*>(base)[(int64_t)row * rs + (int64_t)(k + e) * ks];
        r[j] = val;
      }
    } else if constexpr (EM == 0) {
#pragma unroll
      for (int j = 0; j < NV; ++j) r[j] = bload(base, off[j]);
    } else if constexpr (EM == 2) {
#pragma unroll
      for (int j = 0; j < NV; ++j) r[j] = bload(base, off[j], records);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int row, k;
        coords(stid, j, row, k);
        vec_t val = 0;
        if (row < rows && k < krem) val = bload(base, off[j]);
        r[j] = val;
      }
    }
  }

  // 8-byte-unit swizzle: in rows with bit 3 set the two halves of every 16-byte vector trade
  // places (unit u is stored at u ^ 1).  The fragment reads are emitted as ds_read2_b64, which
  // the LDS serves in groups of 16 lanes over 32 banks: with the plain image rows i and i + 8
  // of a fragment (pitch 36 dwords: 36 * 8 = 0 mod 32) hit the same bank pair — measured
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.43 (profiles/r02_gemm_pmc_before_lds_swizzle.json); the
  // readers apply the same XOR to their k-group (frag_off), so rows i and i + 8 now read
  // different halves and the 16 lanes of a group cover 32 distinct banks.
  static __device__ __forceinline__ vec_t swz(vec_t t, bool sw) {
    vec_t u;
    if constexpr (VEC == 4) {
      u.x = sw ? t.z : t.x; u.y = sw ? t.w : t.y; u.z = sw ? t.x : t.z; u.w = sw ? t.y : t.w;
    } else {
      u.x = sw ? t.y : t.x; u.y = sw ? t.x : t.y;
    }
    return u;
  }

  static __device__ __forceinline__ void store(const vec_t (&r)[NV], char* lds, int stid) {
    if constexpr (MODE == 1) {
#pragma unroll
      for (int b = 0; b < NV / VEC; ++b) {
        int mb = stid + STG * b;
        int kq = mb & 7, nq = mb >> 3;
#pragma unroll
        for (int jj = 0; jj < VEC; ++jj) {
          vec_t t;
#pragma unroll
          for (int i = 0; i < VEC; ++i) t[i] = r[b * VEC + i][jj];
          const int row = nq * VEC + jj;
          *reinterpret_cast<vec_t*>(lds + row * ROW_BYTES + kq * 16) = swz(t, (row & 8) != 0);
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        int v = stid + STG * j;
        *reinterpret_cast<vec_t*>(lds + (v >> 3) * ROW_BYTES + (v & 7) * 16) =
            swz(r[j], ((v >> 3) & 8) != 0);
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

// KS > 1 (64x64 tile only): K is cut into KS contiguous ranges, one per GROUP of 4 wavefronts inside
// the workgroup (its own pair of LDS slab buffers); the groups' accumulators are summed through LDS
// in group order (deterministic) and group 0 stores the tile.  For problems that give the chip one
// 4-wave workgroup per CU (1024^3, 16384x64x1024): 2-4 waves per SIMD instead of 1, so the load
// latency of a slab hides behind the other groups' MFMAs.  Needs K % (BK * KS) == 0.
template <typename T, int AMODE, int BMODE, int EM, int H = 0, int KS = 1>
__global__ __launch_bounds__(Cfg<H>::THREADS * KS, 4) void gemm_kernel(GemmArgs g) {
  constexpr bool EDGE = (EM != 0);
  constexpr int BM = Cfg<H>::BM, BN = Cfg<H>::BN, STG = Cfg<H>::STG, WN = Cfg<H>::WN;
  constexpr int FM = Cfg<H>::FM, FN = Cfg<H>::FN;
  using StA = Stage<T, AMODE, STG>;
  using StB = Stage<T, BMODE, STG>;
  using Tr = Traits<T>;
  using acc_t = typename Tr::acc_t;
  using vec_t = typename Tr::vec_t;
  using frag_t = typename Tr::frag_t;
  constexpr int BK = Tr::BK;
  constexpr int SLAB = (BM + BN) * ROW_BYTES;  // bytes per LDS buffer (A rows then B rows)
  extern __shared__ __attribute__((aligned(16))) char smem_all[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int GW = 2 * WN;                 // wavefronts per K group (= all of them for KS == 1)
  const int grp = KS > 1 ? wave_all / GW : 0;
  const int wave = KS > 1 ? wave_all % GW : wave_all;
  const int wm = wave / WN, wn = wave % WN;  // 2 x WN waves, (16 FM) x (16 FN) outputs each
  const bool stage_a = wave < WN;           // wave-uniform: which operand this wave stages
  const int stid = tid & (STG - 1);
  char* const smem = smem_all + (KS > 1 ? grp * 2 * SLAB : 0);
  const int64_t Kg = KS > 1 ? g.K / KS : g.K;            // this group's K extent ...
  const int64_t kg0 = KS > 1 ? (int64_t)grp * Kg : 0;    // ... and where it starts

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
  const int GROUP = g.group;
  int group_sz = GROUP * g.tiles_n;
  int gid = bid / group_sz;
  int first_m = gid * GROUP;
  int gm = min(g.tiles_m - first_m, GROUP);
  int tm = first_m + (bid % group_sz) % gm;
  int tn = (bid % group_sz) / gm;

  const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
  const int64_t z = blockIdx.z;
  const int rows_a = (int)((g.M - m0) < BM ? (g.M - m0) : BM);
  const int rows_b = (int)((g.N - n0) < BN ? (g.N - n0) : BN);

  // this wave's staging operand: base of its tile rows, strides, slab step in bytes
  const int64_t s_rs = stage_a ? g.a_rs : g.b_cs;  // stride between tile rows (m / n)
  const int64_t s_ks = stage_a ? g.a_cs : g.b_rs;  // stride along k
  const char* sbase = stage_a
      ? reinterpret_cast<const char*>(static_cast<const T*>(g.A) + z * g.a_bs + m0 * g.a_rs + kg0 * s_ks)
      : reinterpret_cast<const char*>(static_cast<const T*>(g.B) + z * g.b_bs + n0 * g.b_cs + kg0 * s_ks);
  const int s_rows = stage_a ? rows_a : rows_b;
  const int64_t slab_bytes = (int64_t)BK * s_ks * (int64_t)sizeof(T);
  // end of this operand's valid bytes (EM 2: bound of the load descriptors)
  const char* send = stage_a
      ? reinterpret_cast<const char*>(static_cast<const T*>(g.A) + z * g.a_bs) + g.a_lim
      : reinterpret_cast<const char*>(static_cast<const T*>(g.B) + z * g.b_bs) + g.b_lim;
  unsigned off[NV];
  if (stage_a) StA::offsets(off, stid, s_rs, s_ks);
  else StB::offsets(off, stid, s_rs, s_ks);

  acc_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) acc[i][j] = 0;

  // two register staging sets (slab parity): loads of slab t+2 are issued at the start of slab
  // t and stored to LDS at the end of slab t+1 (load-to-use distance: two slabs)
  vec_t r0[NV], r1[NV];
  const int nslab = (int)((Kg + BK - 1) / BK);

  auto gload = [&](vec_t (&r)[NV], int t) {
    const char* base = sbase + (int64_t)t * slab_bytes;
    const int64_t krem = Kg - (int64_t)t * BK;
    int records = 0x7FFFFFFF;
    if constexpr (EM == 2) {
      const int64_t rem = send - base;
      records = rem <= 0 ? 0 : (rem > 0x7FFFFFFF ? 0x7FFFFFFF : (int)rem);
    }
    if (stage_a)
      StA::template load<EM>(r, base, off, stid, s_rs, s_ks, s_rows, krem, records);
    else
      StB::template load<EM>(r, base, off, stid, s_rs, s_ks, s_rows, krem, records);
  };
  auto lstore = [&](const vec_t (&r)[NV], int t) {
    char* nb = smem + (t & 1) * SLAB;
    // the LDS addresses are a few shifts of stid: recompute them per slab instead of letting
    // LICM park 4-8 loop-invariant address VGPRs (the kernel sits at the 128-VGPR budget that
    // buys 4 waves per SIMD; a spilled address is a scratch reload whose s_waitcnt vmcnt also
    // drains the two-slab-deep prefetch)
    int s2 = stid;
    asm volatile("" : "+v"(s2));
    if (stage_a) StA::store(r, nb, s2);
    else StB::store(r, nb + BM * ROW_BYTES, s2);
  };

  if (nslab > 0) {
    gload(r0, 0);
    gload(r1, nslab > 1 ? 1 : 0);   // unconditional, like the prefetch in slab()
    lstore(r0, 0);
  }
  __syncthreads();

  // per-lane fragment addressing: lane (i = l & 15, kg = l >> 4) reads 8 bytes at
  // row*144 + kstep*32 + kg*8  (f32: k = 8*kstep + 2*kg + {0,1}; f64: k = 4*kstep + kg)
  // (fragment base rows are multiples of 16, so bit 3 of the tile row is bit 3 of the lane:
  // rows with that bit set hold their 8-byte units swapped pairwise, see Stage::swz)
  const int frag_off = (lane & 15) * ROW_BYTES + ((lane >> 4) ^ ((lane >> 3) & 1)) * 8;
  const int pa_off = wm * (16 * FM) * ROW_BYTES + frag_off;
  const int pb_off = BM * ROW_BYTES + wn * (16 * FN) * ROW_BYTES + frag_off;

  // fp32 only: chunked summation.  A 4096-long fp32 fma chain has a Frobenius error of 1.15e-6
  // against an fp64 product — outside the 1e-6 parity bar (the reference's OpenBLAS sgemm sums in
  // blocks).  Instead of a second accumulator set (32 more VGPRs would cost the 4-waves-per-SIMD
  // budget; scratch-resident sums measured -25 % throughput), every CHUNK slabs (2048 k) the
  // accumulators are folded into the C tile in memory (each lane owns its C elements, so this is
  // a private read-modify-write) and cleared: chains stay <= 2048 long (0.81e-6 at K = 4096) for
  // one extra pass over C per 2048 k (+3 % at K = 4096).  fp64 needs no such help.
  constexpr bool CHUNKED = (sizeof(T) == 4);
  constexpr int CHUNK = 64;
  bool flushed = false;

  const T alpha = (T)g.alpha, beta = (T)g.beta;
  T* C = static_cast<T*>(g.C) + z * g.c_bs;
  const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
  const bool use_cin = (g.beta != 0.0);
  // C = alpha*acc + (first fold: beta*Cin, later folds: the partial already in C)
  auto fold = [&](bool first) {
    // launder the lane id: the 32 element addresses must be recomputed HERE, not hoisted out of
    // the slab loop as loop invariants (that cost 64+ VGPRs and 111 spills)
    int l2 = lane;
    asm volatile("" : "+v"(l2));
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int64_t col = n0 + wn * (16 * FN) + j * 16 + (l2 & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t row = m0 + wm * (16 * FM) + i * 16 + frag_row<T>(l2, r);
          if (!EDGE || (row < g.M && col < g.N)) {
            T v = alpha * acc[i][j][r];
            if (first) {
              if (use_cin) v += beta * Cin[row * g.ci_rs + col * g.ci_cs];
            } else {
              v += C[row * g.c_rs + col * g.c_cs];
            }
            C[row * g.c_rs + col * g.c_cs] = v;
          }
          acc[i][j][r] = 0;
        }
      }
  };

  // one slab: prefetch slab t+2 into lr, MFMAs on LDS buffer t&1, stage slab t+1 from ur into
  // the other LDS buffer, one barrier
  auto slab = [&](int t, vec_t (&lr)[NV], const vec_t (&ur)[NV]) {
    const char* buf = smem + (t & 1) * SLAB;
    // always issued (the last two slabs re-read the final slab): a prefetch inside a branch
    // makes the compiler's s_waitcnt at the LDS stores below count only the loads it is sure of,
    // i.e. wait for vmcnt(3..0) — the just-issued prefetch included — instead of vmcnt(7..4)
    gload(lr, t + 2 < nslab ? t + 2 : nslab - 1);
    // fragment fetches run ONE k-step ahead of the MFMAs that consume them (two register sets,
    // scheduling barriers pin the order): the ds_read latency of step ks+1 hides behind the 16
    // MFMAs of step ks instead of stalling the wave at the top of every step pair
    frag_t fa[2][FM], fb[2][FN];
    auto fetch = [&](int set, int ks) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
        fa[set][i] = *reinterpret_cast<const frag_t*>(buf + pa_off + i * 16 * ROW_BYTES + ks * 32);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        fb[set][j] = *reinterpret_cast<const frag_t*>(buf + pb_off + j * 16 * ROW_BYTES + ks * 32);
    };
    auto mmas = [&](int set) {
      if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma(acc[i][j], fa[set][i].x, fb[set][j].x);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma(acc[i][j], fa[set][i].y, fb[set][j].y);
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma(acc[i][j], fa[set][i], fb[set][j]);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks + 1 < 4) fetch((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mmas(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (t + 1 < nslab) lstore(ur, t + 1);
    __syncthreads();
    if constexpr (CHUNKED) {
      if ((t % CHUNK) == CHUNK - 1 && t + 1 < nslab) {
        fold(!flushed);
        flushed = true;
      }
    }
  };

  for (int t = 0; t < nslab; t += 2) {
    slab(t, r0, r1);                         // loads slab t+2 (even set), stages t+1
    if (t + 1 < nslab) slab(t + 1, r1, r0);  // loads slab t+3 (odd set), stages t+2
  }
  if constexpr (KS > 1) {
    // the groups' partial tiles through LDS (the slab buffers are free: the loop ended on a
    // barrier), summed by group 0 in group order
    T* red = reinterpret_cast<T*>(smem_all);
    constexpr int PER_WAVE = FM * FN * 4 * 64;
    if (grp > 0) {
      T* dst = red + ((grp - 1) * GW + wave) * PER_WAVE + lane;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[((i * FN + j) * 4 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 1; g2 < KS; ++g2) {
      const T* src = red + ((g2 - 1) * GW + wave) * PER_WAVE + lane;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[i][j][r] += src[((i * FN + j) * 4 + r) * 64];
    }
  }
  // ---- epilogue: C = alpha*acc + beta*Cin (or + the partial already folded into C) --------
  fold(!flushed);
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


// ---- small-output GEMM: 16x16 tile per workgroup, K split over its 4 wavefronts ----------------
// When M*N is too small to give every CU a 128x128 tile (a batch-64 recurrent step is 64 x 1024:
// 8 big tiles on 256 CUs) the big kernel leaves the chip idle and pays a full K loop per tile.
// Here every 16x16 output block is its own workgroup (M=64, N=1024 -> 256 workgroups), each
// wavefront runs the MFMA chain over a quarter of K straight from global memory (the operands are
// L2-resident at these sizes; fragments are loaded in MFMA layout, no LDS staging), and the four
// partial tiles are summed in wave order through LDS (deterministic).  Arbitrary strides.
int64_t g_gemm_group = 8;          // tile-rows per walk group (ahip_set_param "gemm_group")
// 64x64 tile (Cfg<1>) window, measured r02 fp32 (default -> half tile): 1024^3 36.8 -> 32.1 us,
// 2048x1024x1024 64.2 -> 48.4, 3072x1024x1024 91.6 -> 65.6, 3072x2048x1024 (384 big tiles:
// 1.5 rounds of the chip) 144 -> 123; it loses below ~48 big tiles (768^2x1024: 26 -> 33 us, the
// 16-row kernels win) and from two full rounds of big tiles on (4096x2048x1024: 148 -> 157)
int64_t g_half_max_tiles = 448;    // used below this many 128x128 tiles ...
int64_t g_half_min_tiles = 192;    // ... when the problem has at least this many 64x64 tiles
int64_t g_half_ksplit = -1;        // K groups inside a 64x64-tile workgroup: -1 = by workgroup count, 1 / 2 / 4 forced
int64_t g_small_max_tiles = 64;    // scalar-load form: below this many 128x128 tiles (x4 for the
                                   // vector-load form, see gemm_dispatch)

template <typename T>
__global__ __launch_bounds__(256) void gemm_small_kernel(GemmArgs g) {
  using acc_t = typename Traits<T>::acc_t;
  __shared__ T part[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int64_t z = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * 16, n0 = (int64_t)blockIdx.x * 16;
  const T* __restrict__ A = static_cast<const T*>(g.A) + z * g.a_bs;
  const T* __restrict__ B = static_cast<const T*>(g.B) + z * g.b_bs;
  // wave w owns k in [w*kq, min(K, (w+1)*kq)), kq a multiple of 4
  const int64_t kq = ((g.K + 15) / 16) * 4;
  const int64_t kbeg = wave * kq;
  const int64_t kend = (kbeg + kq < g.K) ? kbeg + kq : g.K;
  const bool mok = m0 + r < g.M, nok = n0 + r < g.N;
  const T* ap = A + (mok ? m0 + r : 0) * g.a_rs;
  const T* bp = B + (nok ? n0 + r : 0) * g.b_cs;
  acc_t acc = {0, 0, 0, 0}, tot = {0, 0, 0, 0};
  constexpr int U = 8;  // k-steps (of 4) per unrolled group: 16 loads in flight per lane
  int64_t k = kbeg, next_fold = kbeg + 512;   // two-level sum, see gemm_skinny_kernel
  for (; k + 4 * U <= kend; k += 4 * U) {
    T a[U], b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = k + 4 * u + kg;
      a[u] = mok ? ap[kk * g.a_cs] : (T)0;
      b[u] = nok ? bp[kk * g.b_rs] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mma(acc, a[u], b[u]);
    if (k + 4 * U >= next_fold) {
      tot += acc;
      acc = acc_t{0, 0, 0, 0};
      next_fold += 512;
    }
  }
  for (; k < kend; k += 4) {
    const int64_t kk = k + kg;
    const bool kok = kk < kend;
    const T a = (mok && kok) ? ap[kk * g.a_cs] : (T)0;
    const T b = (nok && kok) ? bp[kk * g.b_rs] : (T)0;
    mma(acc, a, b);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) part[wave][frag_row<T>(lane, i) * 16 + r] = tot[i] + acc[i];
  __syncthreads();
  const int e = threadIdx.x, er = e >> 4, ec = e & 15;
  if (m0 + er < g.M && n0 + ec < g.N) {
    T sum = part[0][e];
    sum += part[1][e];
    sum += part[2][e];
    sum += part[3][e];
    T v = (T)g.alpha * sum;
    if (g.beta != 0.0) {
      const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
      v += (T)g.beta * Cin[(m0 + er) * g.ci_rs + (n0 + ec) * g.ci_cs];
    }
    T* C = static_cast<T*>(g.C) + z * g.c_bs;
    C[(m0 + er) * g.c_rs + (n0 + ec) * g.c_cs] = v;
  }
}


// Vector-load variant of the small-output kernel for the common layouts: A k-contiguous (row-major
// x), B either n-contiguous (x @ W, BKC = false) or k-contiguous (x @ W.T, BKC = true).  Tile
// 16 x (16*NF): a lane loads one 16-byte vector of A along k (VEC consecutive k = VEC MFMA steps)
// and, per step, NF consecutive columns of B (BKC = false; the NF elements feed NF column
// fragments, fragment f holding columns n0 + NF*r + f) or one k-vector per column fragment
// (BKC = true, fragment f = columns n0 + 16 f + r).  Two-group software pipeline on the loads.
int64_t g_skinny_nf = 0;           // 0 = choose by grid size; 1/2/4 forces the tile width

template <typename T, int NF, bool BKC, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(GemmArgs g) {
  using acc_t = typename Traits<T>::acc_t;
  constexpr int VEC = Traits<T>::VEC;
  constexpr int G = 4 * VEC;  // k extent of one vector group (4 lane groups x VEC)
  struct alignas(sizeof(T) * VEC) KV { T v[VEC]; };
  struct alignas(sizeof(T) * NF) NV_ { T v[NF]; };
  __shared__ T part[NW][256 * NF];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int64_t z = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * 16, n0 = (int64_t)blockIdx.x * (16 * NF);
  const T* __restrict__ A = static_cast<const T*>(g.A) + z * g.a_bs;
  const T* __restrict__ B = static_cast<const T*>(g.B) + z * g.b_bs;
  const int64_t kq = ((g.K + NW * G - 1) / (NW * G)) * G;   // K split over the NW wavefronts
  const int64_t kbeg = wave * kq;
  const int64_t kend = (kbeg + kq < g.K) ? kbeg + kq : g.K;
  const bool mok = m0 + r < g.M;
  const T* ap = A + (mok ? m0 + r : 0) * g.a_rs + VEC * kg;
  acc_t acc[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) acc[f] = acc_t{0, 0, 0, 0};

  struct Frag { KV a; KV bk[NF]; NV_ bn[VEC]; };
  auto load = [&](int64_t k0, Frag& fr) {
    const bool kok = k0 + VEC * kg < kend;
    fr.a = (mok && kok) ? *reinterpret_cast<const KV*>(ap + k0) : KV{};
    if constexpr (BKC) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int64_t nn = n0 + 16 * f + r;
        fr.bk[f] = (nn < g.N && kok)
            ? *reinterpret_cast<const KV*>(B + nn * g.b_cs + k0 + VEC * kg) : KV{};
      }
    } else {
      const int64_t nn = n0 + NF * r;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        fr.bn[j] = (nn < g.N && kok)
            ? *reinterpret_cast<const NV_*>(B + (k0 + VEC * kg + j) * g.b_rs + nn) : NV_{};
    }
  };
  auto compute = [&](const Frag& fr) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
      for (int f = 0; f < NF; ++f)
        mma(acc[f], fr.a.v[j], BKC ? fr.bk[f].v[j] : fr.bn[j].v[f]);
  };
  // fp32: a long k-ordered FMA chain drifts past the 1e-6 Frobenius bar (1.15e-6 at 4096), so the
  // running accumulators are folded into a second register set every FOLD k (two-level sum)
  constexpr int64_t FOLD = 512;
  acc_t tot[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) tot[f] = acc_t{0, 0, 0, 0};
  Frag f0, f1;
  int64_t k0 = kbeg, next_fold = kbeg + FOLD;
  constexpr int DEPTH = 64 / G;
  if (kend - kbeg <= DEPTH * G) {
    // short K slice: issue every load before the first MFMA (one memory round trip per wave)
    Frag fr[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load(kbeg + u * G, fr[u]);   // beyond kend: zeros, no access
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) compute(fr[u]);
    k0 = kend;
  }
  if (k0 < kend) load(k0, f0);
  for (; k0 < kend; k0 += 2 * G) {
    if (k0 + G < kend) load(k0 + G, f1);
    compute(f0);
    if (k0 + G < kend) {
      if (k0 + 2 * G < kend) load(k0 + 2 * G, f0);
      compute(f1);
    }
    if (k0 + 2 * G >= next_fold) {
#pragma unroll
      for (int f = 0; f < NF; ++f) { tot[f] += acc[f]; acc[f] = acc_t{0, 0, 0, 0}; }
      next_fold += FOLD;
    }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      part[wave][(frag_row<T>(lane, i) * 16 + r) * NF + f] = tot[f][i] + acc[f][i];
  __syncthreads();
  if (threadIdx.x >= 256) return;
  const int e = threadIdx.x, er = e >> 4, ec = e & 15;
  if (m0 + er < g.M) {
    T* C = static_cast<T*>(g.C) + z * g.c_bs;
    const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int64_t col = BKC ? n0 + 16 * f + ec : n0 + NF * ec + f;
      if (col >= g.N) continue;
      T sum = part[0][e * NF + f];
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += part[w][e * NF + f];
      T v = (T)g.alpha * sum;
      if (g.beta != 0.0) v += (T)g.beta * Cin[(m0 + er) * g.ci_rs + col * g.ci_cs];
      C[(m0 + er) * g.c_rs + col * g.c_cs] = v;
    }
  }
}


// Weight-gradient layout (A = X.T, m-contiguous; B n-contiguous): C[M,N] = sum_k A[k][m] B[k][n] with
// few output tiles and a very long K (X.T @ dY of a classifier: 1024 x 1000 x 32768 is 64 big
// tiles).  Tile (16 VEC) x (16 VEC) per workgroup, K split over its 4 wavefronts; per k-step a
// lane loads ONE 16-byte vector of A (VEC consecutive rows = VEC row fragments, fragment e holding
// rows m0 + VEC*i + e) and ONE of B (VEC column fragments): 2 loads feed VEC*VEC MFMAs.
template <typename T>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs g) {
  using acc_t = typename Traits<T>::acc_t;
  constexpr int VEC = Traits<T>::VEC;
  constexpr int TILE = 16 * VEC;
  constexpr int U = 4;  // k-steps per load group
  struct alignas(sizeof(T) * VEC) PV { T v[VEC]; };
  __shared__ T part[3][TILE * TILE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kg = lane >> 4;
  const int64_t z = blockIdx.z;
  const int64_t m0 = (int64_t)blockIdx.y * TILE, n0 = (int64_t)blockIdx.x * TILE;
  const T* __restrict__ A = static_cast<const T*>(g.A) + z * g.a_bs;
  const T* __restrict__ B = static_cast<const T*>(g.B) + z * g.b_bs;
  const int64_t kq = ((g.K + 4 * 4 * U - 1) / (4 * 4 * U)) * (4 * U);
  const int64_t kbeg = wave * kq;
  const int64_t kend = (kbeg + kq < g.K) ? kbeg + kq : g.K;
  const bool mok = m0 + VEC * r < g.M, nok = n0 + VEC * r < g.N;   // M % VEC == N % VEC == 0
  const T* ap = A + (mok ? m0 + VEC * r : 0);       // a_rs == 1: rows contiguous
  const T* bp = B + (nok ? n0 + VEC * r : 0);       // b_cs == 1: columns contiguous
  acc_t acc[VEC][VEC], tot[VEC][VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e)
#pragma unroll
    for (int f = 0; f < VEC; ++f) { acc[e][f] = acc_t{0, 0, 0, 0}; tot[e][f] = acc_t{0, 0, 0, 0}; }
  struct Grp { PV a[U], b[U]; };
  auto load = [&](int64_t k0, Grp& gr) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = k0 + 4 * u + kg;
      const bool kok = kk < kend;
      gr.a[u] = (mok && kok) ? *reinterpret_cast<const PV*>(ap + kk * g.a_cs) : PV{};
      gr.b[u] = (nok && kok) ? *reinterpret_cast<const PV*>(bp + kk * g.b_rs) : PV{};
    }
  };
  auto compute = [&](const Grp& gr) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int f = 0; f < VEC; ++f) mma(acc[e][f], gr.a[u].v[e], gr.b[u].v[f]);
  };
  constexpr int64_t GK = 4 * U;
  Grp g0, g1;
  int64_t k0 = kbeg, next_fold = kbeg + 512;
  if (k0 < kend) load(k0, g0);
  for (; k0 < kend; k0 += 2 * GK) {
    if (k0 + GK < kend) load(k0 + GK, g1);
    compute(g0);
    if (k0 + GK < kend) {
      if (k0 + 2 * GK < kend) load(k0 + 2 * GK, g0);
      compute(g1);
    }
    if (k0 + 2 * GK >= next_fold) {   // two-level sum (see gemm_skinny_kernel)
#pragma unroll
      for (int e = 0; e < VEC; ++e)
#pragma unroll
        for (int f = 0; f < VEC; ++f) { tot[e][f] += acc[e][f]; acc[e][f] = acc_t{0, 0, 0, 0}; }
      next_fold += 512;
    }
  }
  // element (row i, fragment e; column r, fragment f) -> tile position (VEC*i + e, VEC*r + f)
  auto tpos = [&](int i, int e, int f) { return (VEC * frag_row<T>(lane, i) + e) * TILE + VEC * r + f; };
  if (wave > 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int f = 0; f < VEC; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wave - 1][tpos(i, e, f)] = tot[e][f][i] + acc[e][f][i];
  }
  __syncthreads();
  if (wave == 0) {
    T* C = static_cast<T*>(g.C) + z * g.c_bs;
    const T* Cin = static_cast<const T*>(g.Cin) + z * g.ci_bs;
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int f = 0; f < VEC; ++f)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int p = tpos(i, e, f);
          const int64_t row = m0 + p / TILE, col = n0 + p % TILE;
          if (row >= g.M || col >= g.N) continue;
          T sum = tot[e][f][i] + acc[e][f][i];
          sum += part[0][p];
          sum += part[1][p];
          sum += part[2][p];
          T v = (T)g.alpha * sum;
          if (g.beta != 0.0) v += (T)g.beta * Cin[row * g.ci_rs + col * g.ci_cs];
          C[row * g.c_rs + col * g.c_cs] = v;
        }
  }
}

template <typename T>
int launch_tn(const GemmArgs& g, int64_t batch, hipStream_t s) {
  constexpr int TILE = 16 * Traits<T>::VEC;
  const int64_t gx = (g.N + TILE - 1) / TILE, gy = (g.M + TILE - 1) / TILE;
  AHIP_LAUNCH((gemm_tn_kernel<T>), dim3((unsigned)gx, (unsigned)gy, (unsigned)batch), dim3(256), 0,
              s, g);
  return AHIP_OK;
}

template <typename T, int NF, bool BKC>
int launch_skinny(const GemmArgs& g, int64_t batch, hipStream_t s) {
  const int64_t gx = (g.N + 16 * NF - 1) / (16 * NF), gy = (g.M + 15) / 16;
  const dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)batch);
  // small grids are bound by the memory round trips of each wavefront's K chain, not by MFMA
  // rate: split K over more wavefronts (batch-64 recurrent step: 12.7 -> 9.7 ms per 512 steps)
  const int64_t wgs = gx * gy * batch;
  if (NF <= 2 && wgs <= 512 && g.K >= 1024) {
    AHIP_LAUNCH((gemm_skinny_kernel<T, NF, BKC, 16>), grid, dim3(1024), 0, s, g);
  } else if (wgs <= 1024 && g.K >= 512) {
    AHIP_LAUNCH((gemm_skinny_kernel<T, NF, BKC, 8>), grid, dim3(512), 0, s, g);
  } else {
    AHIP_LAUNCH((gemm_skinny_kernel<T, NF, BKC, 4>), grid, dim3(256), 0, s, g);
  }
  return AHIP_OK;
}

template <typename T>
int launch_small(const GemmArgs& g, int64_t batch, hipStream_t s) {
  const int64_t gx = (g.N + 15) / 16, gy = (g.M + 15) / 16;
  AHIP_REQUIRE(gy < 65536 && batch < 65536, "grid too large for the small-tile GEMM");
  // vector-load variant when the layouts allow it
  constexpr int VEC = Traits<T>::VEC;
  auto al = [](const void* p, size_t b) { return reinterpret_cast<uintptr_t>(p) % b == 0; };
  const bool a_ok = g.a_cs == 1 && al(g.A, 16) && g.a_rs % VEC == 0 && g.K % VEC == 0 &&
                    (batch <= 1 || g.a_bs % VEC == 0);
  const bool a_mc = g.a_rs == 1 && al(g.A, 16) && g.a_cs % VEC == 0 && g.M % VEC == 0 &&
                    (batch <= 1 || g.a_bs % VEC == 0);
  const bool b_nc = g.b_cs == 1 && al(g.B, 16) && g.b_rs % VEC == 0 && g.N % VEC == 0 &&
                    (batch <= 1 || g.b_bs % VEC == 0);
  if (!a_ok && a_mc && b_nc) return launch_tn<T>(g, batch, s);
  if (a_ok) {
    const bool bk_ok = g.b_rs == 1 && al(g.B, 16) && g.b_cs % VEC == 0 &&
                       (batch <= 1 || g.b_bs % VEC == 0);
    int nf = VEC;
    if (g_skinny_nf > 0) nf = (int)(g_skinny_nf < VEC ? g_skinny_nf : VEC);
    else
      while (nf > 1 && gy * ((g.N + 16 * nf - 1) / (16 * nf)) * batch < ahip_cu_count()) nf /= 2;
    auto bn_ok = [&](int f) {
      return g.b_cs == 1 && al(g.B, sizeof(T) * f) && g.b_rs % f == 0 && g.N % f == 0 &&
             (batch <= 1 || g.b_bs % f == 0);
    };
    if (bk_ok) {
      if (nf >= 4 && VEC >= 4) return launch_skinny<T, (VEC >= 4 ? 4 : VEC), true>(g, batch, s);
      if (nf >= 2) return launch_skinny<T, 2, true>(g, batch, s);
      return launch_skinny<T, 1, true>(g, batch, s);
    }
    while (nf > 1 && !bn_ok(nf)) nf /= 2;
    if (bn_ok(nf)) {
      if (nf >= 4 && VEC >= 4) return launch_skinny<T, (VEC >= 4 ? 4 : VEC), false>(g, batch, s);
      if (nf >= 2) return launch_skinny<T, 2, false>(g, batch, s);
      return launch_skinny<T, 1, false>(g, batch, s);
    }
  }
  AHIP_LAUNCH((gemm_small_kernel<T>), dim3((unsigned)gx, (unsigned)gy, (unsigned)batch), dim3(256),
              0, s, g);
  return AHIP_OK;
}

// staging mode of one operand (rs = stride between its tile rows, ks = stride along k) and
// whether 32-bit byte offsets inside a tile are safe
template <typename T>
int operand_mode(const void* p, int64_t rs, int64_t ks, int64_t rows, int64_t K, int64_t bs,
                 int64_t batch) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int BK = Traits<T>::BK;
  bool aligned = (reinterpret_cast<uintptr_t>(p) % 16 == 0) && (batch <= 1 || bs % VEC == 0);
  auto fits32 = [&](int64_t span) { return span >= 0 && span * (int64_t)sizeof(T) < (1LL << 31); };
  if (ks == 1 && aligned && rs % VEC == 0 && K % VEC == 0 && rs >= 0 && fits32(128 * rs + BK))
    return 0;
  if (rs == 1 && aligned && ks % VEC == 0 && rows % VEC == 0 && ks >= 0 && fits32(BK * ks + 128))
    return 1;
  return 2;
}

template <typename T, int AM, int BMd, int EDGE, int H = 0, int KS = 1>
int launch_gemm(const GemmArgs& g, int64_t batch, hipStream_t s) {
  constexpr size_t lds = KS * 2 * (Cfg<H>::BM + Cfg<H>::BN) * ROW_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(gemm_kernel<T, AM, BMd, EDGE, H, KS>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      ahip_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
      return AHIP_EHIP;
    }
    attr_set = true;
  }
  dim3 grid((unsigned)(g.tiles_m * g.tiles_n), 1, (unsigned)batch);
  AHIP_LAUNCH((gemm_kernel<T, AM, BMd, EDGE, H, KS>), grid, dim3(Cfg<H>::THREADS * KS), lds, s, g);
  return AHIP_OK;
}

template <typename T, int EDGE>
int dispatch_modes(const GemmArgs& g, int am, int bm, int64_t batch, hipStream_t s) {
  switch (am * 3 + bm) {
    case 0: return launch_gemm<T, 0, 0, EDGE>(g, batch, s);
    case 1: return launch_gemm<T, 0, 1, EDGE>(g, batch, s);
    case 3: return launch_gemm<T, 1, 0, EDGE>(g, batch, s);
    case 4: return launch_gemm<T, 1, 1, EDGE>(g, batch, s);
    default: break;
  }
  // any scalar-staged operand: always the predicated instantiation
  switch (am * 3 + bm) {
    case 2: return launch_gemm<T, 0, 2, 1>(g, batch, s);
    case 5: return launch_gemm<T, 1, 2, 1>(g, batch, s);
    case 6: return launch_gemm<T, 2, 0, 1>(g, batch, s);
    case 7: return launch_gemm<T, 2, 1, 1>(g, batch, s);
    default: return launch_gemm<T, 2, 2, 1>(g, batch, s);
  }
}


// which instantiation: 0 interior, 2 ragged M / N with K a multiple of BK (bounded descriptors,
// predicated stores only), 1 fully predicated
template <typename T>
int edge_mode(const GemmArgs& g, int bm_ = BM, int bn_ = BN) {
  const bool kfull = g.K % Traits<T>::BK == 0;
  if (kfull && g.M % bm_ == 0 && g.N % bn_ == 0) return 0;
  return kfull ? 2 : 1;
}

// the 64x64 tile: vector-staged operands and K a multiple of the slab only (everything else
// keeps the 128x128 / 16-row kernels)
template <typename T, int EDGE, int KS = 1>
int dispatch_half(const GemmArgs& g, int am, int bm, int64_t batch, hipStream_t s) {
  switch (am * 3 + bm) {
    case 0: return launch_gemm<T, 0, 0, EDGE, 1, KS>(g, batch, s);
    case 1: return launch_gemm<T, 0, 1, EDGE, 1, KS>(g, batch, s);
    case 3: return launch_gemm<T, 1, 0, EDGE, 1, KS>(g, batch, s);
    default: return launch_gemm<T, 1, 1, EDGE, 1, KS>(g, batch, s);
  }
}

template <typename T>
int run_big(const GemmArgs& g, int am, int bm, int64_t batch, hipStream_t s) {
  switch (edge_mode<T>(g)) {
    case 0: return dispatch_modes<T, 0>(g, am, bm, batch, s);
    case 2: return dispatch_modes<T, 2>(g, am, bm, batch, s);
    default: return dispatch_modes<T, 1>(g, am, bm, batch, s);
  }
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
  g.group = (int)g_gemm_group;
  g.tiles_n = (int)((g.N + BN - 1) / BN);
  AHIP_REQUIRE((int64_t)g.tiles_m * g.tiles_n < (1LL << 31), "too many tiles");
  int am = operand_mode<T>(g.A, g.a_rs, g.a_cs, g.M, g.K, g.a_bs, batch);
  int bm = operand_mode<T>(g.B, g.b_cs, g.b_rs, g.N, g.K, g.b_bs, batch);
  {
    // mid-size problems: fewer 128x128 tiles than CUs, but enough 64x64 tiles to fill the chip
    const int64_t t128 = (int64_t)g.tiles_m * g.tiles_n * batch;
    const int64_t tm64 = (g.M + 63) / 64, tn64 = (g.N + 63) / 64;
    // (exactly one big tile per CU is one full round of the chip: the big tile keeps it)
    // ... or whose extents leave most of a 128x128 tile empty (a batch of 64x64 products: the big
    // tile would compute 4x the work): the 64-tiling wastes less than 2/3 of what the 128-tiling does
    const double eff128 = (double)g.M * g.N / ((double)g.tiles_m * BM * g.tiles_n * BN);
    const double eff64 = (double)g.M * g.N / ((double)tm64 * 64 * tn64 * 64);
    const bool mostly_empty = eff128 * 1.5 < eff64;
    if (am < 2 && bm < 2 && g.K % Traits<T>::BK == 0 &&
        ((t128 < g_half_max_tiles && t128 != 256) || mostly_empty) &&
        tm64 * tn64 * batch >= g_half_min_tiles && tm64 * tn64 < (1LL << 31)) {
      g.tiles_m = (int)tm64;
      g.tiles_n = (int)tn64;
      // few workgroups (at most two per CU) and a long K: K groups inside the workgroup
      const int64_t wgs = tm64 * tn64 * batch;
      const int64_t cus = ahip_cu_count() > 0 ? ahip_cu_count() : 256;
      int ks = (int)g_half_ksplit;
      if (ks < 0) ks = wgs <= cus ? 4 : (wgs <= 2 * cus ? 2 : 1);
      while (ks > 1 && (g.K % (Traits<T>::BK * ks) != 0 || g.K / ks > 2048 || g.K / ks < 4 * Traits<T>::BK))
        ks >>= 1;
      const bool interior = edge_mode<T>(g, 64, 64) == 0;
      if (ks >= 4) return interior ? dispatch_half<T, 0, 4>(g, am, bm, batch, s)
                                   : dispatch_half<T, 2, 4>(g, am, bm, batch, s);
      if (ks == 2) return interior ? dispatch_half<T, 0, 2>(g, am, bm, batch, s)
                                   : dispatch_half<T, 2, 2>(g, am, bm, batch, s);
      return interior ? dispatch_half<T, 0>(g, am, bm, batch, s)
                      : dispatch_half<T, 2>(g, am, bm, batch, s);
    }
  }
  // below one 128x128 tile per CU the 16-row kernels win when their vector-load form applies
  // (measured fp32: 1024^3 = 64 tiles 86 -> 36 us, 2048x1024x1024 = 128 tiles 88 -> 65 us,
  // 2048x2048x512 = 256 tiles: big 53 vs 64 us); the scalar-load form only below 64 tiles
  {
    constexpr int VEC = Traits<T>::VEC;
    const bool a_vec = g.a_cs == 1 && reinterpret_cast<uintptr_t>(g.A) % 16 == 0 &&
                       g.a_rs % VEC == 0 && g.K % VEC == 0 && (batch <= 1 || g.a_bs % VEC == 0);
    const bool b_vec = reinterpret_cast<uintptr_t>(g.B) % 16 == 0 &&
                       ((g.b_rs == 1 && g.b_cs % VEC == 0) ||
                        (g.b_cs == 1 && g.b_rs % VEC == 0 && g.N % VEC == 0)) &&
                       (batch <= 1 || g.b_bs % VEC == 0);
    const bool tn_vec = g.a_rs == 1 && reinterpret_cast<uintptr_t>(g.A) % 16 == 0 &&
                        g.a_cs % VEC == 0 && g.M % VEC == 0 && g.b_cs == 1 && g.b_rs % VEC == 0 &&
                        g.N % VEC == 0 && reinterpret_cast<uintptr_t>(g.B) % 16 == 0 &&
                        (batch <= 1 || (g.a_bs % VEC == 0 && g.b_bs % VEC == 0));
    const int64_t limit = ((a_vec && b_vec) || tn_vec) ? (g_small_max_tiles < 64 ? g_small_max_tiles
                                                                       : 4 * g_small_max_tiles)
                                           : g_small_max_tiles;
    if ((int64_t)g.tiles_m * g.tiles_n * batch < limit && (g.M + 15) / 16 < 65536 && batch < 65536)
      return launch_small<T>(g, batch, s);
  }
  return run_big<T>(g, am, bm, batch, s);
}


// ---- split-K for few-tile / long-K problems (weight gradients) ---------------------------------
// The K range is cut into S slices that run as S "batches" of the big 128x128 kernel into a
// caller-provided workspace [S][M][N]; splitk_reduce_kernel sums the slices in order and applies
// the alpha / beta epilogue (deterministic; no float atomics).
struct SplitArgs { const void* ws; const void* Cin; void* C; int64_t M, N, S, ci_rs, ci_cs, c_rs, c_cs;
                   double alpha, beta; };
AHIP_PTRS_BEGIN(SplitArgs) AHIP_PTR1(ws) AHIP_PTR1(Cin) AHIP_PTR1(C) AHIP_PTRS_END

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitArgs a) {
  const T* __restrict__ ws = static_cast<const T*>(a.ws);
  const int64_t n = a.M * a.N;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    T sum = ws[i];
    for (int64_t sidx = 1; sidx < a.S; ++sidx) sum += ws[sidx * n + i];
    const int64_t r = i / a.N, c = i - r * a.N;
    T v = (T)a.alpha * sum;
    if (a.beta != 0.0) v += (T)a.beta * static_cast<const T*>(a.Cin)[r * a.ci_rs + c * a.ci_cs];
    static_cast<T*>(a.C)[r * a.c_rs + c * a.c_cs] = v;
  }
}

// number of K slices for (M, N, K), 0 = do not split
int64_t splitk_slices(int64_t M, int64_t N, int64_t K, int64_t batch, int bk) {
  if (batch != 1 || M < BM || N < BN || K < 8192) return 0;
  const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (tiles >= 128) return 0;
  int64_t S = 16;
  while (S > 1 && (tiles * S > 512 || K % (S * bk) != 0 || K / S < 1024)) S /= 2;
  return S > 1 ? S : 0;
}

template <typename T>
int gemm_splitk(const GemmArgs& g, int64_t S, void* ws, hipStream_t s) {
  GemmArgs p = g;
  p.K = g.K / S;
  p.a_bs = p.K * g.a_cs;
  p.b_bs = p.K * g.b_rs;
  p.C = ws; p.c_bs = g.M * g.N; p.c_rs = g.N; p.c_cs = 1;
  p.Cin = ws; p.ci_bs = p.c_bs; p.ci_rs = g.N; p.ci_cs = 1;
  p.alpha = 1.0; p.beta = 0.0;
  set_limits(p, (int64_t)sizeof(T));
  p.tiles_m = (int)((g.M + BM - 1) / BM);
  p.group = (int)g_gemm_group;
  p.tiles_n = (int)((g.N + BN - 1) / BN);
  int am = operand_mode<T>(p.A, p.a_rs, p.a_cs, p.M, p.K, p.a_bs, S);
  int bm = operand_mode<T>(p.B, p.b_cs, p.b_rs, p.N, p.K, p.b_bs, S);
  int rc = run_big<T>(p, am, bm, S, s);
  if (rc) return rc;
  SplitArgs r{ws, g.Cin, g.C, g.M, g.N, S, g.ci_rs, g.ci_cs, g.c_rs, g.c_cs, g.alpha, g.beta};
  int64_t n = g.M * g.N;
  unsigned blocks = (unsigned)(((n + 255) / 256) < 2048 ? ((n + 255) / 256) : 2048);
  AHIP_LAUNCH((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, s, r);
  return AHIP_OK;
}

// bytes from the operand's first element to one past its last valid one (non-negative strides)
void set_limits(GemmArgs& g, int64_t isz) {
  auto ext = [&](int64_t r, int64_t rs, int64_t c, int64_t cs) {
    if (r <= 0 || c <= 0) return (int64_t)0;
    const int64_t last = (r - 1) * (rs > 0 ? rs : 0) + (c - 1) * (cs > 0 ? cs : 0);
    return (last + 1) * isz;
  };
  g.a_lim = ext(g.M, g.a_rs, g.K, g.a_cs);
  g.b_lim = ext(g.K, g.b_rs, g.N, g.b_cs);
}

double host_scalar(int dtype, const void* p) {
  return dtype == AHIP_F32 ? (double)*static_cast<const float*>(p)
                           : *static_cast<const double*>(p);
}

}  // namespace

void ahip_gemm_set_small_max_tiles(int64_t v) { g_small_max_tiles = v; }
void ahip_gemm_set_half_max_tiles(int64_t v) { g_half_max_tiles = v; }
void ahip_gemm_set_half_min_tiles(int64_t v) { g_half_min_tiles = v; }
void ahip_gemm_set_group(int64_t v) { g_gemm_group = v; }
void ahip_gemm_set_skinny_nf(int64_t v) { g_skinny_nf = v; }
void ahip_gemm_set_half_ksplit(int64_t v) { g_half_ksplit = v; }

extern "C" {

int ahip_gemm_batched(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K,
                      const void* alpha, const void* A, int64_t a_bs, int64_t a_rs, int64_t a_cs,
                      const void* B, int64_t b_bs, int64_t b_rs, int64_t b_cs, const void* beta,
                      const void* Cin, int64_t ci_bs, int64_t ci_rs, int64_t ci_cs, void* C,
                      int64_t c_bs, int64_t c_rs, int64_t c_cs, void* stream) {
  AHIP_REQUIRE(dtype == AHIP_F32 || dtype == AHIP_F64, "gemm supports float32/float64 only");
  AHIP_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "negative extent");
  AHIP_REQUIRE(alpha && beta, "null alpha/beta");
  GemmArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.a_bs = a_bs; g.a_rs = a_rs; g.a_cs = a_cs;
  g.B = B; g.b_bs = b_bs; g.b_rs = b_rs; g.b_cs = b_cs;
  g.alpha = host_scalar(dtype, alpha);
  g.beta = host_scalar(dtype, beta);
  g.Cin = (g.beta != 0.0) ? Cin : C; g.ci_bs = ci_bs; g.ci_rs = ci_rs; g.ci_cs = ci_cs;
  g.C = C; g.c_bs = c_bs; g.c_rs = c_rs; g.c_cs = c_cs;
  g.tiles_m = g.tiles_n = 0;
  set_limits(g, dtype == AHIP_F32 ? 4 : 8);
  if (M > 0 && N > 0 && batch > 0) {
    AHIP_REQUIRE(C != nullptr, "null C");
    AHIP_REQUIRE(K == 0 || (A && B), "null A/B");
    AHIP_REQUIRE(g.beta == 0.0 || Cin != nullptr, "beta != 0 needs Cin");
  }
  // grid.z carries the batch index: more than 65535 items (matmul over many small matrices)
  // run as consecutive launches over slices of the batch
  const int64_t isz = dtype == AHIP_F32 ? 4 : 8;
  for (int64_t b0 = 0; b0 < batch || b0 == 0; b0 += 65535) {
    const int64_t nb = (batch - b0) < 65535 ? (batch - b0) : 65535;
    GemmArgs h = g;
    h.A = static_cast<const char*>(g.A) + b0 * a_bs * isz;
    h.B = static_cast<const char*>(g.B) + b0 * b_bs * isz;
    h.Cin = static_cast<const char*>(g.Cin) + b0 * (g.beta != 0.0 ? ci_bs : c_bs) * isz;
    h.C = static_cast<char*>(g.C) + b0 * c_bs * isz;
    int rc = dtype == AHIP_F32 ? gemm_dispatch<float>(h, nb, as_stream(stream))
                               : gemm_dispatch<double>(h, nb, as_stream(stream));
    if (rc) return rc;
    if (batch == 0) break;
  }
  return AHIP_OK;
}

int ahip_gemm(int dtype, int64_t M, int64_t N, int64_t K, const void* alpha, const void* A,
              int64_t a_rs, int64_t a_cs, const void* B, int64_t b_rs, int64_t b_cs,
              const void* beta, const void* Cin, int64_t ci_rs, int64_t ci_cs, void* C,
              int64_t c_rs, int64_t c_cs, void* stream) {
  return ahip_gemm_batched(dtype, 1, M, N, K, alpha, A, 0, a_rs, a_cs, B, 0, b_rs, b_cs, beta,
                           Cin, 0, ci_rs, ci_cs, C, 0, c_rs, c_cs, stream);
}

size_t ahip_gemm_ws_bytes(int dtype, int64_t batch, int64_t M, int64_t N, int64_t K) {
  if (dtype != AHIP_F32 && dtype != AHIP_F64) return 0;
  const int bk = dtype == AHIP_F32 ? Traits<float>::BK : Traits<double>::BK;
  const int64_t S = splitk_slices(M, N, K, batch, bk);
  return (size_t)(S * M * N) * (dtype == AHIP_F32 ? 4 : 8);
}

int ahip_gemm_splitk(int dtype, int64_t M, int64_t N, int64_t K, const void* alpha, const void* A,
                     int64_t a_rs, int64_t a_cs, const void* B, int64_t b_rs, int64_t b_cs,
                     const void* beta, const void* Cin, int64_t ci_rs, int64_t ci_cs, void* C,
                     int64_t c_rs, int64_t c_cs, void* ws, size_t ws_bytes, void* stream) {
  const size_t need = ahip_gemm_ws_bytes(dtype, 1, M, N, K);
  if (need == 0 || ws == nullptr || ws_bytes < need)
    return ahip_gemm(dtype, M, N, K, alpha, A, a_rs, a_cs, B, b_rs, b_cs, beta, Cin, ci_rs, ci_cs,
                     C, c_rs, c_cs, stream);
  AHIP_REQUIRE(alpha && beta && A && B && C, "null argument");
  GemmArgs g;
  g.M = M; g.N = N; g.K = K;
  g.A = A; g.a_bs = 0; g.a_rs = a_rs; g.a_cs = a_cs;
  g.B = B; g.b_bs = 0; g.b_rs = b_rs; g.b_cs = b_cs;
  g.alpha = host_scalar(dtype, alpha);
  g.beta = host_scalar(dtype, beta);
  AHIP_REQUIRE(g.beta == 0.0 || Cin != nullptr, "beta != 0 needs Cin");
  g.Cin = (g.beta != 0.0) ? Cin : C; g.ci_bs = 0; g.ci_rs = ci_rs; g.ci_cs = ci_cs;
  g.C = C; g.c_bs = 0; g.c_rs = c_rs; g.c_cs = c_cs;
  g.tiles_m = g.tiles_n = 0;
  set_limits(g, dtype == AHIP_F32 ? 4 : 8);
  const int bk = dtype == AHIP_F32 ? Traits<float>::BK : Traits<double>::BK;
  const int64_t S = splitk_slices(M, N, K, 1, bk);
  return dtype == AHIP_F32 ? gemm_splitk<float>(g, S, ws, as_stream(stream))
                           : gemm_splitk<double>(g, S, ws, as_stream(stream));
}

}  // extern "C"
