"""HIP source generator for the fused broadcast Elemwise / CAReduce kernels (K1, K2, K3).

Plays the role of the reference's per-Op C generators — ``Elemwise._c_all``
(tensor/elemwise.py:835), ``elemwise_cgen.make_loop`` / ``make_reordered_loop`` (:228/:305),
``CAReduce._c_all`` (:1522) and ``Composite.c_code_template`` (scalar/basic.py:4250) — but emits
one gfx950 kernel per fused group instead of a CPU loop nest:

* one lane handles ``VEC`` consecutive elements of the innermost (collapsed) dimension with
  16-byte global loads; outer dimensions are index arithmetic over at most ``AHIP_MAXD``
  collapsed dims; a grid-stride loop covers the rest (HBM-bound streaming shape);
* broadcast operands (stride 0 on the inner dim) are loaded once per lane, not per element;
* a CAReduce consumer is fused into the same kernel: per-lane accumulation in the reference's
  accumulator dtype (``CAReduce._acc_dtype`` :1371), wavefront reduction with cross-lane
  shuffles, one partial per workgroup, and a deterministic fixed-order finalize.

The scalar bodies restate the ``c_code`` of the reference ScalarOps (scalar/basic.py,
scalar/math.py) for the supported dtypes (bool, (u)int8-64, float32/64).
"""
from __future__ import annotations

import hashlib

import numpy as np

from . import knobs
from ._lib import AHIP_MAXD, AHIP_MAXOPS

CTYPE = {
    "bool": "unsigned char", "int8": "signed char", "int16": "short", "int32": "int",
    "int64": "long long", "uint8": "unsigned char", "uint16": "unsigned short",
    "uint32": "unsigned int", "uint64": "unsigned long long", "float32": "float",
    "float64": "double",
}
# type used for values held in registers (bool is a real C++ bool there)
RTYPE = dict(CTYPE, bool="bool")

PRELUDE = r"""
typedef long long i64;
#ifndef NAN
#define NAN __builtin_nanf("")
#endif
#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif
#define AHIP_MAXD %d
#define AHIP_MAXOPS %d
struct Args {
  i64 n; i64 shape[AHIP_MAXD]; i64 stride[AHIP_MAXOPS][AHIP_MAXD]; void* ptr[AHIP_MAXOPS];
  void* ws; void* out; i64 aux0; i64 aux1; int nd; int nops;
};
// horizontally fused full reductions (ahip_ewh_args): jobs share one grid
#define AHIP_HJOBS 16
#define AHIP_HOPS 6
struct ArgsH {
  i64 n[AHIP_HJOBS]; void* ptr[AHIP_HJOBS][AHIP_HOPS]; void* out[AHIP_HJOBS];
  unsigned wg0[AHIP_HJOBS + 1]; int njobs; void* ws; i64 aux1;
};
template <typename T, int N> struct alignas((sizeof(T) * N) >= 16 ? 16 : (sizeof(T) * N)) Pack { T v[N]; };

template <typename T, int N> __device__ __forceinline__ Pack<T, N> nt_load(const Pack<T, N>* p) {
  Pack<T, N> r;
  if constexpr (sizeof(T) * N >= 16) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    const u4* q = (const u4*)p;
    u4* d = (u4*)&r;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) * N / 16; ++i) d[i] = __builtin_nontemporal_load(q + i);
  } else {
    r = *p;
  }
  return r;
}

template <typename T, int N> __device__ __forceinline__ void nt_store(Pack<T, N>* p, const Pack<T, N>& r) {
  if constexpr (sizeof(T) * N >= 16) {
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4* q = (u4*)p;
    const u4* d = (const u4*)&r;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) * N / 16; ++i) __builtin_nontemporal_store(d[i], q + i);
  } else {
    *p = r;
  }
}

// ---- scalar helpers (reference: aesara/scalar/basic.py, scalar/math.py c_code) ----
template <typename T> __device__ __forceinline__ T idiv_floor(T x, T y) {  // FloorDivide :2039
  if (y == 0) return 0;
  T q = x / y;
  if ((x %% y != 0) && ((x < 0) != (y < 0))) q -= 1;
  return q;
}
template <typename T> __device__ __forceinline__ T udiv_floor(T x, T y) { return y == 0 ? 0 : x / y; }
template <typename T> __device__ __forceinline__ T imod_py(T x, T y) {      // Mod :2144
  if (y == 0) return 0;
  T r = x %% y;
  if (r != 0 && ((r < 0) != (y < 0))) r += y;
  return r;
}
template <typename T> __device__ __forceinline__ T umod(T x, T y) { return y == 0 ? 0 : x %% y; }
__device__ __forceinline__ float fmod_py(float x, float y) {
  if (y == 0.0f) return fmodf(x, y);
  float r = fmodf(x, y);
  if (r != 0.0f && ((r < 0.0f) != (y < 0.0f))) r += y;
  return r;
}
__device__ __forceinline__ double fmod_py(double x, double y) {
  if (y == 0.0) return fmod(x, y);
  double r = fmod(x, y);
  if (r != 0.0 && ((r < 0.0) != (y < 0.0))) r += y;
  return r;
}
template <typename T> __device__ __forceinline__ T ipow(T b, T e) {
  T r = 1;
  if (e < 0) return (b == 1) ? 1 : ((b == (T)-1) ? ((e & 1) ? (T)-1 : 1) : 0);
  while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
  return r;
}
template <typename T> __device__ __forceinline__ T upow(T b, T e) {
  T r = 1;
  while (e) { if (e & 1) r *= b; b *= b; e >>= 1; }
  return r;
}
// ScalarMaximum/ScalarMinimum c_code (:1745): NaN propagates
template <typename T> __device__ __forceinline__ T fmax_nan(T x, T y) { return (y > x) ? y : ((x >= y) ? x : (T)NAN); }
template <typename T> __device__ __forceinline__ T fmin_nan(T x, T y) { return (y < x) ? y : ((x <= y) ? x : (T)NAN); }
template <typename T> __device__ __forceinline__ T imax(T x, T y) { return x > y ? x : y; }
// MulWithoutZeros.c_code (tensor/math.py:2731): zeros are skipped, 0 is the identity
template <typename T> __device__ __forceinline__ T mwz_(T x, T y) { return x == 0 ? y : (y == 0 ? x : (T)(y * x)); }
template <typename T> __device__ __forceinline__ T imin(T x, T y) { return x < y ? x : y; }
// x / c for a loop-invariant c with r = 1/c precomputed: Markstein refinement gives the correctly
// rounded quotient when c and r are normal numbers (ok: hoisted, wave-uniform) and nothing
// overflowed (a non-finite q or residual makes res non-finite); otherwise the full division.
__device__ __forceinline__ bool recip_ok(double c, double r) {
  return fabs(c) >= 2.2250738585072014e-308 && fabs(c) < INFINITY &&
         fabs(r) >= 2.2250738585072014e-308 && fabs(r) < INFINITY;
}
__device__ __forceinline__ bool recip_ok(float c, float r) {
  return fabsf(c) >= 1.17549435e-38f && fabsf(c) < INFINITY &&
         fabsf(r) >= 1.17549435e-38f && fabsf(r) < INFINITY;
}
__device__ __forceinline__ double fdiv_inv(double x, double c, double r, bool ok) {
  const double q = x * r;
  double res = fma(fma(-q, c, x), r, q);
  if (__builtin_expect(!(ok && fabs(res) < INFINITY), 0)) {
    asm volatile("" ::: "memory");   // keep the full division out of line (no if-conversion)
    res = x / c;
  }
  return res;
}
__device__ __forceinline__ float fdiv_inv(float x, float c, float r, bool ok) {
  const float q = x * r;
  float res = fmaf(fmaf(-q, c, x), r, q);
  if (__builtin_expect(!(ok && fabsf(res) < INFINITY), 0)) {
    asm volatile("" ::: "memory");
    res = x / c;
  }
  return res;
}
// (K * y) / c with K = +-2^k a literal, correctly rounded like fdiv_inv: the real number is
// y / (c / K); c / K and K * r = RN(1 / (c / K)) are exact scalings, loop invariant, so the scaling
// multiply of every element goes away and the refinement runs on (y, c / K, K * r).  Full division
// of the scaled operands when any of c, r, c / K, K * r is not a normal number.
template <typename T> __device__ __forceinline__ T fdiv_inv_s(T y, T K, T c, T r, bool ok) {
  const T cK = c * ((T)1 / K), rK = r * K;
  const T q = y * rK;
  T res = fma(fma(-q, cK, y), rK, q);
  if (__builtin_expect(!(ok && recip_ok(cK, rK) && fabs(res) < (T)INFINITY), 0)) {
    asm volatile("" ::: "memory");
    res = (K * y) / c;
  }
  return res;
}
// tolerance mode (AESARA_HIP_FASTDIV=1): x * (1/c) without the refinement — at most 1.5 ulp from
// the quotient (north_star's bar is 1e-6 rel); the full division when c or 1/c is not a normal number
template <typename T> __device__ __forceinline__ T fdiv_rcp_s(T y, T K, T c, T r, bool ok) {
  T res = y * (K * r);            // (K * y) / c, K a power of two: K * r is exact and loop invariant
  if (__builtin_expect(!ok, 0)) {
    asm volatile("" ::: "memory");
    res = (K * y) / c;
  }
  return res;
}
template <typename T> __device__ __forceinline__ T fdiv_rcp(T x, T c, T r, bool ok) {
  T res = x * r;
  if (__builtin_expect(!ok, 0)) {
    asm volatile("" ::: "memory");
    res = x / c;
  }
  return res;
}
__device__ __forceinline__ float sigmoid_(float x) { return 1.0f / (1.0f + expf(-x)); }   // Sigmoid :1110
__device__ __forceinline__ double sigmoid_(double x) { return 1.0 / (1.0 + exp(-x)); }
__device__ __forceinline__ float softplus_(float x) {                                      // Softplus :1173
  return x < -37.0f ? expf(x) : (x < 18.0f ? log1pf(expf(x)) : (x < 33.3f ? x + expf(-x) : x));
}
__device__ __forceinline__ double softplus_(double x) {
  return x < -37.0 ? exp(x) : (x < 18.0 ? log1p(exp(x)) : (x < 33.3 ? x + exp(-x) : x));
}
// Psi :361 — the reference's C body: Bernardo (1976), Algorithm AS 103 (0 for x <= 0, like it)
__device__ __forceinline__ double psi_as103(double x) {
  double y = x, psi = 0.0;
  if (y <= 0.0) return psi;
  if (y <= 1.0e-5) return -0.5772156649 - 1.0 / y;
  while (y < 8.5) { psi = psi - 1.0 / y; y = y + 1; }
  double R = 1.0 / y;
  psi = psi + log(y) - .5 * R;
  R = R * R;
  psi = psi - R * (8.333333333e-2 - R * (8.333333333e-3 - R * 3.968253968e-3));
  return psi;
}
// TriGamma :454 — the reference's C body: Algorithm AS 121 (0 for x <= 0, like it)
__device__ __forceinline__ double trigamma_as121(double x) {
  if (x <= 0) return 0.0;
  if (x <= 0.0001) return 1.0 / x / x;
  double value = 0.0, z = x;
  while (z < 5.0) { value += 1.0 / z / z; z += 1.0; }
  const double y = 1.0 / z / z;
  value += 0.5 * y + (1.0 + y * (0.1666666667 + y * (-0.03333333333 + y * (0.02380952381 + y * -0.03333333333)))) / z;
  return value;
}
// Gamma :283 (C: tgamma): exact at the small integers like the C library's (ocml's tgamma(1) is one
// ulp low: truncating it into an integer output — gamma_inplace on an int64 array — gave 0)
__device__ inline double gamma_(double x) {
  if (x == floor(x) && x >= 1.0 && x <= 23.0) {
    double p = 1.0;
    for (int i = 2; i < (int)x; ++i) p *= (double)i;
    return p;
  }
  return tgamma(x);
}
__device__ inline float gamma_(float x) { return (float)gamma_((double)x); }
// I0 :1064 / I1 :1038 (scipy.special.i0 / i1): even / odd in x; the device library's routines are
// evaluated on |x|
__device__ inline double bessel_i0_(double x) { return cyl_bessel_i0(fabs(x)); }
__device__ inline float bessel_i0_(float x) { return cyl_bessel_i0f(fabsf(x)); }
__device__ inline double bessel_i1_(double x) { return copysign(cyl_bessel_i1(fabs(x)), x); }
__device__ inline float bessel_i1_(float x) { return copysignf(cyl_bessel_i1f(fabsf(x)), x); }
// Regularised incomplete gamma functions (GammaInc :580 / GammaIncC :629 / Chi2SF :538 / GammaU :836
// / GammaL :877: the reference's C bodies call GammaP / GammaQ / upperGamma / lowerGamma of
// scalar/c_code/gamma.c): the power series for x < k + 1, the continued fraction (modified Lentz)
// otherwise, both scaled by exp(k log x - x - lgamma(k)).  NaN for k <= 0 or x < 0 like the reference.
__device__ inline double igam_series_(double k, double x) {
  double term = 1.0 / k, sum = term, n = k;
  for (int i = 0; i < 1024; ++i) {
    n += 1.0; term *= x / n; sum += term;
    if (fabs(term) < fabs(sum) * 2.2204460492503131e-16) break;
  }
  return sum;
}
__device__ inline double igam_cfrac_(double k, double x) {
  const double tiny = 2.2204460492503131e-16 * 2.2204460492503131e-16 * 2.2204460492503131e-16;   // gamma.c TINY
  double b = x + 1.0 - k, c = 1.0 / tiny, d = 1.0 / b, f = d;
  for (int i = 1; i < 1024; ++i) {
    const double a = -(double)i * ((double)i - k);
    b += 2.0;
    d = a * d + b; if (fabs(d) < tiny) d = tiny;
    c = b + a / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    const double e = d * c;
    f *= e;
    if (fabs(e - 1.0) < 2.2204460492503131e-16) break;
  }
  return f;
}
// upperGamma / lowerGamma of gamma.c: ALWAYS the continued fraction / the series (whatever x is)
__device__ inline double gamma_upper_(double k, double x) {
  if (!(k > 0.0) || !(x > 0.0)) return NAN;
  return igam_cfrac_(k, x) * exp(k * log(x) - x);
}
__device__ inline double gamma_lower_(double k, double x) {
  if (!(k > 0.0) || !(x > 0.0)) return NAN;
  return igam_series_(k, x) * exp(k * log(x) - x);
}
__device__ inline double gamma_p_(double k, double x) {
  if (!(k > 0.0) || !(x >= 0.0)) return NAN;
  if (x == 0.0) return 0.0;
  const double w = exp(k * log(x) - x - lgamma(k));
  return x < k + 1.0 ? igam_series_(k, x) * w : 1.0 - igam_cfrac_(k, x) * w;
}
__device__ inline double gamma_q_(double k, double x) {
  if (!(k > 0.0) || !(x >= 0.0)) return NAN;
  if (x == 0.0) return 1.0;
  const double w = exp(k * log(x) - x - lgamma(k));
  return x < k + 1.0 ? 1.0 - igam_series_(k, x) * w : igam_cfrac_(k, x) * w;
}
__device__ __forceinline__ float log1mexp_(float x) { return x < -0.6931471805599453f ? log1pf(-expf(x)) : logf(-expm1f(x)); }
__device__ __forceinline__ double log1mexp_(double x) { return x < -0.6931471805599453 ? log1p(-exp(x)) : log(-expm1(x)); }
__device__ __forceinline__ float round_away(float x) { return x < 0 ? ceilf(x - 0.5f) : floorf(x + 0.5f); }
__device__ __forceinline__ double round_away(double x) { return x < 0 ? ceil(x - 0.5) : floor(x + 0.5); }

// ---- float64 exp through a 64-entry table (Tang's scheme; tools/gen_exp_table.py prints the
// constants): x = (64 k + j) ln2/64 + r with |r| <= ln2/128, exp(x) = 2^k * T[j] * (1 + p(r)).
// T = 2^(j/64) correctly rounded, one copy per wavefront in LDS (no barrier: a wave reads only
// what it wrote).  |x| >= 708, infinities and NaN: the argument is clamped first and 2^k applied
// by v_ldexp_f64 (denormal results, overflow, underflow), behind a rare branch.  Measured
// against expl over 2e7 arguments: <= 1.02 ulp (the C library: 0.51, ocml's exp: 1).  Replaces
// Exp.c_code (scalar/basic.py:3102) `exp(x)` for float64 only.
__device__ const double AHIP_EXP2_64[64] = {
  0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
  0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
  0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
  0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
  0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
  0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
  0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
  0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
  0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
  0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
  0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
  0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
  0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
  0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
  0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
  0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double exp_tbl64(double x, const double* tbl) {
  const bool big = ((unsigned)__double2hiint(x) & 0x7fffffffu) >= 0x40862000u;   // |x| >= 708, inf, NaN
  double xc = x;
  if (__builtin_expect(big, 0)) {
    asm volatile("" ::: "memory");   // rare: keep it a branch (no if-conversion into the hot path)
    xc = fmin(fmax(x, -1000.0), 1000.0);
  }
  double s;                                                     // x * 64/ln2 + 1.5 * 2^52: k lands in the low word
  asm("v_fma_f64 %%0, %%1, %%2, %%3" : "=v"(s) : "v"(xc), "v"(0x1.71547652b82fep+6), "s"(0x1.8p+52));
  const int ki = __double2loint(s);
  const double kd = s - 0x1.8p+52;
  double r = fma(kd, -0x1.62e42ff000000p-7, xc);                // k * C1 is exact (33-bit C1)
  r = fma(kd, 0x1.718432a1b0e26p-41, r);
  const double T = tbl[ki & 63];
  const double r2 = r * r;
  // Horner steps with a constant addend as three-address v_fma_f64 with the constant in an
  // SGPR pair (the compiler's two-address v_fmac_f64 copies the constant into the destination
  // first: one v_mov_b64 per step)
  double q;
  asm("v_fma_f64 %%0, %%1, %%2, %%3" : "=v"(q) : "v"(r), "s"(0x1.11111d8fbe766p-7), "v"(0x1.55556b3304ec0p-5));
  asm("v_fma_f64 %%0, %%1, %%2, %%3" : "=v"(q) : "v"(q), "v"(r), "s"(0x1.5555555555255p-3));
  asm("v_fma_f64 %%0, %%1, %%2, %%3" : "=v"(q) : "v"(q), "v"(r), "s"(0x1.ffffffffff57fp-2));
  const double p = fma(r2, q, r);
  const double m = fma(T, p, T);                                // T * e^r, in [0.99, 2.01)
  // |x| < 708: the result is a normal number and 2^k is an add into the exponent field
  int hi;
  asm("v_lshl_add_u32 %%0, %%1, 14, %%2" : "=v"(hi) : "v"(ki & ~63), "v"(__double2hiint(m)));
  double res = __hiloint2double(hi, __double2loint(m));
  if (__builtin_expect(big, 0)) {
    asm volatile("" ::: "memory");
    res = ldexp(m, ki >> 6);        // denormal results, overflow to inf, underflow to 0
    if (x != x) res = x;
  }
  return res;
}

// ---- cross-lane reduction plumbing (64-wide wavefronts) ----
template <typename T> __device__ __forceinline__ T shfl_xor_(T v, int m) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __shfl_xor(u.i[0], m, 64); u.i[1] = __shfl_xor(u.i[1], m, 64);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __shfl_xor(u.i, m, 64); return u.t;
  } else {
    int i = (int)v; i = __shfl_xor(i, m, 64); return (T)i;
  }
}
// DPP move of a whole value (32-bit pieces): ctrl 0xB1 / 0x4E = quad_perm [1,0,3,2] / [2,3,0,1],
// 0x141 = row_half_mirror, 0x140 = row_mirror.  After combining with these four in turn every lane
// of a 16-lane row holds the row's fold; lane_get_ then reads the four row leaders (uniform).
template <typename T, int CTRL> __device__ __forceinline__ T dpp_mov_(T v) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __builtin_amdgcn_update_dpp(u.i[0], u.i[0], CTRL, 0xf, 0xf, false);
    u.i[1] = __builtin_amdgcn_update_dpp(u.i[1], u.i[1], CTRL, 0xf, 0xf, false);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v;
    u.i = __builtin_amdgcn_update_dpp(u.i, u.i, CTRL, 0xf, 0xf, false);
    return u.t;
  } else {
    int i = (int)v; i = __builtin_amdgcn_update_dpp(i, i, CTRL, 0xf, 0xf, false); return (T)i;
  }
}
template <typename T> __device__ __forceinline__ T lane_get_(T v, int lane) {
  if constexpr (sizeof(T) == 8) {
    union { T t; int i[2]; } u; u.t = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane); u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
    return u.t;
  } else if constexpr (sizeof(T) == 4) {
    union { T t; int i; } u; u.t = v; u.i = __builtin_amdgcn_readlane(u.i, lane); return u.t;
  } else {
    int i = (int)v; i = __builtin_amdgcn_readlane(i, lane); return (T)i;
  }
}
""" % (AHIP_MAXD, AHIP_MAXOPS)

_FLOAT_FN = {
    "sqrt": "sqrt", "exp": "exp", "exp2": "exp2", "expm1": "expm1", "log": "log",
    "log2": "log2", "log10": "log10", "log1p": "log1p", "sin": "sin", "cos": "cos",
    "tan": "tan", "arcsin": "asin", "arccos": "acos", "arctan": "atan", "sinh": "sinh",
    "cosh": "cosh", "tanh": "tanh", "arcsinh": "asinh", "arccosh": "acosh", "arctanh": "atanh",
    "ceil": "ceil", "floor": "floor", "trunc": "trunc", "round_half_to_even": "rint",
    "erf": "erf", "erfc": "erfc",
    # scalar/math.py: Gamma :283 (tgamma), GammaLn :317 (lgamma), Erfcx :108, Erfinv :173,
    # Erfcinv :219, J0 :978 / J1 :947 (libm j0 / j1), I0 :1064 / I1 :1038 (scipy.special.i0 / i1)
    "gamma": "gamma_", "gammaln": "lgamma", "erfcx": "erfcx", "erfinv": "erfinv",
    "erfcinv": "erfcinv", "j0": "j0", "j1": "j1", "i0": "bessel_i0_", "i1": "bessel_i1_",
}

_IDENT = {  # reduction identities
    "add": lambda dt: "0", "mul": lambda dt: "1", "or": lambda dt: "0", "xor": lambda dt: "0",
    "mul_without_zeros": lambda dt: "0",      # MulWithoutZeros.identity (tensor/math.py:2720)
    "and": lambda dt: "true" if dt == "bool" else "(%s)~(%s)0" % (RTYPE[dt], RTYPE[dt]),
}


def _is_float(dt):
    return dt in ("float32", "float64")


def _is_uint(dt):
    return dt.startswith("uint")


def _lit(value, dt):
    """C literal for a scalar constant of dtype ``dt``."""
    if dt == "bool":
        return "true" if value else "false"
    if _is_float(dt):
        v = float(value)
        if np.isnan(v):
            return "(%s)NAN" % RTYPE[dt]
        if np.isinf(v):
            return "(%s)(%sINFINITY)" % (RTYPE[dt], "-" if v < 0 else "")
        r = repr(float(np.float32(v))) if dt == "float32" else repr(v)
        if "e" not in r and "." not in r:
            r += ".0"
        return r + ("f" if dt == "float32" else "")
    v = int(value)
    if dt == "int64":
        return "(%dLL)" % v if v > -(2 ** 63) else "(-9223372036854775807LL - 1)"
    if dt == "uint64":
        return "(%dULL)" % v
    return "((%s)%d)" % (RTYPE[dt], v)


def _cast(expr, src_dt, dst_dt):
    if src_dt == dst_dt:
        return expr
    if dst_dt == "bool":
        return "((%s) != 0)" % expr
    return "((%s)(%s))" % (RTYPE[dst_dt], expr)


def _fname(base, dt):
    if base.endswith("_"):          # an overloaded wrapper of the preamble (float and double forms)
        return base
    return base + ("f" if dt == "float32" else "")


def scalar_node_expr(op, ins, in_dts, dt):
    """C++ expression for one scalar node.  ``ins``: C expressions of the inputs (already in
    their own dtypes ``in_dts``); result must have register type ``RTYPE[dt]``."""
    T = RTYPE[dt]
    c = [_cast(e, d, dt) for e, d in zip(ins, in_dts)]  # inputs cast to the output dtype
    if op in ("add", "mul"):
        if dt == "bool":
            return "(" + (" || " if op == "add" else " && ").join(c) + ")"
        return "(" + (" + " if op == "add" else " * ").join(c) + ")"
    if op == "sub":
        return "(%s)(%s - %s)" % (T, c[0], c[1]) if dt != "bool" else "(%s != %s)" % (c[0], c[1])
    if op == "neg":
        return "(%s)(-%s)" % (T, c[0])
    if op == "true_div":
        return "(%s / %s)" % (c[0], c[1])
    if op == "int_div":
        if _is_float(dt):
            return "%s(%s / %s)" % (_fname("floor", dt), c[0], c[1])
        return "%s<%s>(%s, %s)" % ("udiv_floor" if _is_uint(dt) or dt == "bool" else "idiv_floor",
                                   T, c[0], c[1])
    if op == "mod":
        if _is_float(dt):
            return "fmod_py(%s, %s)" % (c[0], c[1])
        return "%s<%s>(%s, %s)" % ("umod" if _is_uint(dt) or dt == "bool" else "imod_py", T,
                                   c[0], c[1])
    if op == "pow":
        if _is_float(dt):
            return "%s(%s, %s)" % (_fname("pow", dt), c[0], c[1])
        return "%s<%s>(%s, %s)" % ("upow" if _is_uint(dt) or dt == "bool" else "ipow", T, c[0], c[1])
    if op in ("maximum", "minimum"):
        if _is_float(dt):
            f = "fmax_nan" if op == "maximum" else "fmin_nan"
        else:
            f = "imax" if op == "maximum" else "imin"
        e = c[0]
        for x in c[1:]:
            e = "%s<%s>(%s, %s)" % (f, T, e, x)
        return e
    if op in ("lt", "gt", "le", "ge", "eq", "neq"):
        sym = {"lt": "<", "gt": ">", "le": "<=", "ge": ">=", "eq": "==", "neq": "!="}[op]
        ct = np.result_type(*[np.dtype(d) for d in in_dts]).name
        if ct not in RTYPE:
            ct = "float64"
        a, b = (_cast(e, d, ct) for e, d in zip(ins, in_dts))
        return _cast("(%s %s %s)" % (a, sym, b), "bool", dt)
    if op in ("and", "or", "xor"):
        if dt == "bool":
            sym = {"and": "&&", "or": "||", "xor": "!="}[op]
        else:
            sym = {"and": "&", "or": "|", "xor": "^"}[op]
        return "(%s)(" % T + (" %s " % sym).join(c) + ")"
    if op == "invert":
        return "(!%s)" % c[0] if dt == "bool" else "(%s)(~%s)" % (T, c[0])
    if op == "abs":
        if _is_float(dt):
            return "%s(%s)" % (_fname("fabs", dt), c[0])
        if _is_uint(dt) or dt == "bool":
            return c[0]
        return "(%s)(%s < 0 ? -%s : %s)" % (T, c[0], c[0], c[0])
    if op == "sgn":
        if _is_uint(dt) or dt == "bool":
            return "(%s)(%s != 0)" % (T, c[0])
        if _is_float(dt):
            # Sgn.c_code scalar/basic.py:2620: NaN stays NaN (np.sign does the same)
            return "(%s)(%s > 0 ? 1 : (%s < 0 ? -1 : (%s != %s ? NAN : 0)))" % (T, c[0], c[0], c[0], c[0])
        return "(%s)((%s > 0) - (%s < 0))" % (T, c[0], c[0])
    if op == "sqr":
        return "(%s)(%s * %s)" % (T, c[0], c[0]) if dt != "bool" else c[0]
    if op in _FLOAT_FN and _is_float(dt):
        return "%s(%s)" % (_fname(_FLOAT_FN[op], dt), c[0])
    if op in ("ceil", "floor", "trunc", "round_half_to_even", "round_half_away_from_zero") \
            and not _is_float(dt):
        return c[0]
    if op == "round_half_away_from_zero":
        return "round_away(%s)" % c[0]
    if op == "reciprocal":
        return "((%s)1 / %s)" % (T, c[0])
    if op == "sigmoid":
        return "sigmoid_(%s)" % c[0]
    if op == "softplus":
        return "softplus_(%s)" % c[0]
    if op == "log1mexp":
        return "log1mexp_(%s)" % c[0]
    if op == "softsign" and _is_float(dt):
        return "(%s / ((%s)1 + %s(%s)))" % (c[0], T, _fname("fabs", dt), c[0])
    if op == "ultra_fast_sigmoid" and _is_float(dt):
        # UltraFastScalarSigmoid.c_code tensor/nnet/sigm.py:54 (a piecewise tanh approximation): x and z
        # are variables of the OUTPUT type, the arithmetic between them is double (its constants are)
        return ("({ %(T)s ux_ = (%(T)s)(0.5 * (double)%(x)s); const double ua_ = ux_ >= (%(T)s)0 ? (double)ux_ "
                ": (double)(%(T)s)(-ux_); double uz_ = ua_ < 1.7 ? (1.5 * ua_ / (1 + ua_)) : (ua_ < 3 ? "
                "(0.935409070603099 + 0.0458812946797165 * (ua_ - 1.7)) : 0.99505475368673); "
                "const %(T)s uzt_ = (%(T)s)(ux_ >= (%(T)s)0 ? uz_ : -uz_); (%(T)s)(0.5 * ((double)uzt_ + 1.)); })"
                % {"T": T, "x": c[0]})
    if op == "xlogx" and _is_float(dt):          # XlogX.c_code tensor/xlogx.py:27
        return "(%s == (%s)0 ? (%s)0 : %s * %s(%s))" % (c[0], T, T, c[0], _fname("log", dt), c[0])
    if op == "xlogy0" and _is_float(dt):         # XlogY0.c_code tensor/xlogx.py:58
        return "(%s == (%s)0 ? (%s)0 : %s * %s(%s))" % (c[0], T, T, c[0], _fname("log", dt), c[1])
    if op in ("gammainc", "gammaincc") and _is_float(dt):
        return "(%s)%s((double)%s, (double)%s)" % (T, "gamma_p_" if op == "gammainc" else "gamma_q_", c[0], c[1])
    if op == "chi2sf" and _is_float(dt):          # Chi2SF.c_code: 1 - GammaP(k / 2, x / 2), inputs (x, k)
        return "(%s)gamma_q_((double)%s * 0.5, (double)%s * 0.5)" % (T, c[1], c[0])
    if op in ("gammau", "gammal") and _is_float(dt):
        return "(%s)%s((double)%s, (double)%s)" % (T, "gamma_upper_" if op == "gammau" else "gamma_lower_",
                                                    c[0], c[1])
    if op == "psi" and _is_float(dt):
        return "(%s)psi_as103((double)%s)" % (T, c[0])
    if op == "tri_gamma" and _is_float(dt):
        return "(%s)trigamma_as121((double)%s)" % (T, c[0])
    if op == "deg2rad":
        return "(%s * (%s)0.017453292519943295)" % (c[0], T)
    if op == "rad2deg":
        return "(%s * (%s)57.29577951308232)" % (c[0], T)
    if op == "arctan2":
        return "%s(%s, %s)" % (_fname("atan2", dt), c[0], c[1])
    if op in ("identity", "cast"):
        return c[0]
    if op == "second":
        return c[1]
    if op == "switch":
        return "((%s) ? %s : %s)" % (_cast(ins[0], in_dts[0], "bool"), c[1], c[2])
    if op == "clip":
        return "(%s < %s ? %s : (%s > %s ? %s : %s))" % (c[0], c[1], c[1], c[0], c[2], c[2], c[0])
    if op == "isnan":
        return _cast("(%s != %s)" % (ins[0], ins[0]) if _is_float(in_dts[0]) else "false", "bool", dt)
    if op == "isinf":
        if _is_float(in_dts[0]):
            return _cast("(%s(%s) == INFINITY)" % (_fname("fabs", in_dts[0]), ins[0]), "bool", dt)
        return _cast("false", "bool", dt)
    raise NotImplementedError(f"HIP codegen: scalar op {op!r} for dtype {dt}")


def invariant_nodes(scalar, inv_inputs):
    """Indices of scalar nodes that depend only on loop-invariant operands (scalar inputs with
    all-zero strides, constants, other invariant nodes)."""
    inv = set()

    def is_inv(r):
        return r[0] == "c" or (r[0] == "i" and inv_inputs[r[1]]) or (r[0] == "t" and r[1] in inv)

    for k, n in enumerate(scalar["nodes"]):
        if all(is_inv(r) for r in n["in"]) and n["op"] != "second":
            inv.add(k)
    return inv


# ops through which a perturbation of <= 1.5 ulp stays a perturbation of a few ulp (continuous, no
# jumps, no integer results): what a quotient may pass through on its way to a float sum for the
# reciprocal form of a division to be admissible (``sum_only_nodes``)
_CONTINUOUS = {"add", "sub", "mul", "neg", "exp", "exp2", "expm1", "log", "log2", "log10", "log1p",
               "sqr", "sqrt", "sin", "cos", "tanh", "sinh", "cosh", "arctan", "sigmoid", "softplus",
               "true_div", "reciprocal", "identity", "abs", "erf", "erfc"}


def sum_only_nodes(scalar, red, stored_refs):
    """Scalar nodes whose value reaches memory ONLY as a term of the kernel's own floating-point
    SUM (through continuous functions, never through an element-wise output, a comparison, a
    rounding op or an integer cast).  Such a kernel's result already depends on the order of
    summation at the 1e-16 level, so a quotient in that set may be formed as x * (1/c) (<= 1.5 ulp
    from the IEEE quotient); every other division stays the correctly rounded one."""
    if red is None or red.get("op") != "add" or not _is_float(red.get("acc", "")):
        return set()
    nodes, outs = scalar["nodes"], [list(o) for o in scalar["out"]]
    stored_t = {outs[r][1] for r in stored_refs if outs[r][0] == "t"}
    sink = outs[red["ref"]]
    other_out_t = {o[1] for j, o in enumerate(outs) if o[0] == "t" and j != red["ref"]} | stored_t
    ok = {}
    for k in range(len(nodes) - 1, -1, -1):
        good = k not in other_out_t and _is_float(nodes[k]["dtype"])
        if good:
            for m in range(k + 1, len(nodes)):
                if any(r[0] == "t" and r[1] == k for r in nodes[m]["in"]):
                    if nodes[m]["op"] not in _CONTINUOUS or not ok.get(m, False):
                        good = False
                        break
        if good and not any(any(r[0] == "t" and r[1] == k for r in nodes[m]["in"])
                            for m in range(k + 1, len(nodes))) and sink != ["t", k]:
            good = False             # feeds nothing: leave it alone
        ok[k] = good
    return {k for k, g in ok.items() if g}


def _is_pow2(v):
    try:
        m, _e = np.frexp(float(v))
        return abs(m) == 0.5 and np.isfinite(float(v))
    except (TypeError, ValueError):
        return False


def emit_scalar_body(scalar, in_exprs, in_dts, indent="      ", suffix="", hoisted=None,
                     only=None, exp_tbl=None, sum_only=()):
    """Lines computing the temporaries of a plan scalar expression; returns (lines, out_exprs,
    out_dtypes).  ``hoisted``: {node index: (name, recip_name | None)} of temporaries already
    computed before the loop (loop-invariant sub-expressions); ``only``: restrict emission to
    that set of nodes (used to emit the invariant prologue itself); ``exp_tbl``: name of the
    wave's 2^(j/64) table in LDS — float64 ``exp`` nodes then go through ``exp_tbl64``."""
    lines = []
    tdt = [n["dtype"] for n in scalar["nodes"]]
    hoisted = hoisted or {}

    def ref(r):
        if r[0] == "i":
            return in_exprs[r[1]], in_dts[r[1]]
        if r[0] == "t":
            if r[1] in hoisted:
                return hoisted[r[1]][0], tdt[r[1]]
            return "t%d%s" % (r[1], suffix), tdt[r[1]]
        return _lit(r[1], r[2]), r[2]

    uses = {}
    for n in scalar["nodes"]:
        for r in n["in"]:
            if r[0] == "t":
                uses[r[1]] = uses.get(r[1], 0) + 1
    for k, n in enumerate(scalar["nodes"]):
        if k in hoisted or (only is not None and k not in only):
            continue
        refs = [ref(r) for r in n["in"]]
        dt = n["dtype"]
        div = n["in"][1] if n["op"] == "true_div" else None
        if (div is not None and _is_float(dt) and div[0] == "t" and div[1] in hoisted
                and hoisted[div[1]][1] and refs[1][1] == dt):
            # divisor is loop invariant: correctly-rounded division from its hoisted
            # reciprocal (q = x*r; q += r*fma(-q, c, x)) instead of the full v_div_* sequence
            # (AESARA_HIP_FASTDIV=1, tolerance mode: the rounded product x * r alone, <= 1.5 ulp)
            e = None
            num = n["in"][0]
            # AESARA_HIP_FASTDIV: 0 never / 1 always / 2 (default) only for a quotient that reaches
            # memory solely as a term of this kernel's float sum (``sum_only_nodes``)
            fd = knobs.get("FASTDIV")
            fast = fd == 1 or (fd == 2 and k in sum_only)
            if num[0] == "t" and num[1] not in hoisted \
                    and uses.get(num[1], 0) == 1 and list(num) not in [list(o) for o in scalar["out"]]:
                # (K * y) / c with K = +-2^k a literal: scaling by a power of two is exact, so the
                # quotient is y * (K * r) — K * r is loop invariant (the compiler hoists it), the
                # scaling multiply of every element goes away (config 2: -0.5 * sqr(x - mu))
                m = scalar["nodes"][num[1]]
                if m["op"] == "mul" and len(m["in"]) == 2 and m["dtype"] == dt:
                    for ci in (0, 1):
                        c_, y_ = m["in"][ci], m["in"][1 - ci]
                        if c_[0] == "c" and _is_pow2(c_[1]) and ref(y_)[1] == dt \
                                and (fast or 2.0 ** -8 <= abs(float(c_[1])) <= 2.0 ** 8):
                            e = "%s(%s, %s, %s, %s)" % (
                                "fdiv_rcp_s" if fast else "fdiv_inv_s", ref(y_)[0],
                                _lit(c_[1], dt), refs[1][0], hoisted[div[1]][1])
                            break
            if e is None:
                e = "%s(%s, %s, %s)" % ("fdiv_rcp" if fast else "fdiv_inv",
                                        _cast(refs[0][0], refs[0][1], dt), refs[1][0], hoisted[div[1]][1])
        elif exp_tbl and n["op"] == "exp" and dt == "float64":
            e = "exp_tbl64(%s, %s)" % (_cast(refs[0][0], refs[0][1], dt), exp_tbl)
        else:
            e = scalar_node_expr(n["op"], [x[0] for x in refs], [x[1] for x in refs], dt)
        lines.append("%sconst %s t%d%s = %s;" % (indent, RTYPE[dt], k, suffix, e))
    outs = [ref(r) for r in scalar["out"]]
    return lines, [o[0] for o in outs], [o[1] for o in outs]


def red_combine(op, acc_dt, a, b):
    T = RTYPE[acc_dt]
    # bool: NO short-circuit operators — `b` is often a cross-lane shuffle that every lane must
    # execute (a lane that skipped it would hand its partner an undefined value)
    if op == "add":
        return "(bool)((int)%s | (int)%s)" % (a, b) if acc_dt == "bool" else "(%s)(%s + %s)" % (T, a, b)
    if op == "mul":
        return "(bool)((int)%s & (int)%s)" % (a, b) if acc_dt == "bool" else "(%s)(%s * %s)" % (T, a, b)
    if op == "mul_without_zeros":
        return "mwz_<%s>(%s, %s)" % (T, a, b)
    if op == "maximum":
        return "%s<%s>(%s, %s)" % ("fmax_nan" if _is_float(acc_dt) else "imax", T, a, b)
    if op == "minimum":
        return "%s<%s>(%s, %s)" % ("fmin_nan" if _is_float(acc_dt) else "imin", T, a, b)
    sym = {"and": "&", "or": "|", "xor": "^"}[op]
    if acc_dt == "bool":
        return "(bool)((int)%s %s (int)%s)" % (a, sym, b)
    return "(%s)(%s %s %s)" % (T, a, sym, b)


def red_identity(op, acc_dt):
    if op in _IDENT:
        return _IDENT[op](acc_dt)
    info_max = op == "minimum"
    if _is_float(acc_dt):
        return "(%s)(%sINFINITY)" % (RTYPE[acc_dt], "" if info_max else "-")
    if acc_dt == "bool":
        return "true" if info_max else "false"
    ii = np.iinfo(acc_dt)
    return _lit(ii.max if info_max else ii.min, acc_dt)


# ----------------------------------------------------------------------------------------
# kernel specs
# ----------------------------------------------------------------------------------------
# ---- spec-key memo ---------------------------------------------------------------------------
# A spec key is a SHA-256 over the JSON of the scalar program and the layout fields; computing it
# for every launch of every call cost ~50 us of Python per step.  The scalar programs are
# long-lived objects owned by the plan / the fused steps, so their identity plus the (small,
# hashable) layout fields memoises the key; the memo keeps the objects alive so ids stay valid.
_KEY_MEMO = {}


def _hashable(x):
    if isinstance(x, dict):
        return tuple(sorted((k, _hashable(v)) for k, v in x.items()))
    if isinstance(x, (list, tuple)):
        return tuple(_hashable(v) for v in x)
    return x


def _memo_key(scalars, fields, compute):
    sig = (tuple(id(sc) for sc in scalars), _hashable(fields))
    hit = _KEY_MEMO.get(sig)
    if hit is None:
        if len(_KEY_MEMO) > 8192:       # ad-hoc scalar programs (casts) come and go
            _KEY_MEMO.clear()
        hit = (compute(), scalars)
        _KEY_MEMO[sig] = hit
    return hit[0]


class KernelSpec:
    """Everything that determines the generated source of one fused kernel.

    scalar      : plan scalar expression (dict)
    in_dtypes   : dtypes of the Elemwise inputs
    out_dtypes  : dtypes of the materialised Elemwise outputs (may be empty for pure reduce)
    out_refs    : indices into scalar["out"] that are stored
    inner       : per-operand (inputs then stored outputs) inner-dim class: 'c' unit stride,
                  'b' broadcast (stride 0), 's' arbitrary stride (forces vec == 1)
    nd          : number of collapsed dims (elemwise / reduce_all) or nk + nr (reduce_axis)
    vec, block  : elements per lane along the inner dim; threads per workgroup
    idx64       : use 64-bit index arithmetic
    reduce      : None | dict(kind='all'|'row'|'col', op=, acc=, out=, ref=<scalar out index>,
                  nk=, nr=)
    """

    def __init__(self, scalar, in_dtypes, out_dtypes, out_refs, inner, nd, vec, block=256,
                 idx64=False, reduce=None, unroll=1, nt=False, invariant=None, tile_dim=None,
                 early=None, blocked=None, trace=None, fast_exp=None, hjobs=False):
        self.scalar = scalar
        # horizontal fusion: the kernel takes an ArgsH block — several independent jobs of this
        # one specialisation in one grid (flat full reductions only)
        self.hjobs = bool(hjobs)
        flat_all = (reduce is not None and reduce.get("kind") == "all" and tile_dim is None
                    and nd == 1 and vec > 1)
        # flat full reductions: the first group of loads is issued before the invariant prologue
        # (its dependent scalar loads and the reciprocal would otherwise delay them ~0.3 us)
        self.early = bool(knobs.get("EARLY") if early is None else early) and flat_all
        # one contiguous chunk of the stream per workgroup instead of a grid-stride walk
        self.blocked = int(knobs.get("RED_BLOCKED") if blocked is None else blocked) if flat_all else 0
        # plain flat Elemwise streams (no reduction): the same walks, off unless measured better
        if reduce is None and tile_dim is None and nd == 1 and vec > 1:
            # (plain Elemwise streams keep the grid-stride walk: the blocked walks lose there,
            # profiles/r04_cfg1b_stream_walks.txt — the STREAM_BLOCKED switch is gone)
            self.blocked = int(0 if blocked is None else blocked)
        if self.hjobs:
            assert flat_all and len(in_dtypes) + len(out_dtypes) <= 6, "hjobs: flat full reductions only"
            self.blocked = 1                    # a contiguous chunk per workgroup inside its job
        # per-workgroup s_memrealtime stamps into the reduce workspace (tools/ew_trace.py)
        self.trace = bool(knobs.get("EW_TRACE") if trace is None else trace) and \
            reduce is not None and reduce.get("kind") == "all" and tile_dim is None
        # float64 exp through the LDS table (exp_tbl64); only where the scalar program has one
        self.fast_exp = bool(knobs.get("FASTEXP") if fast_exp is None else fast_exp) and \
            tile_dim is None and any(n["op"] == "exp" and n["dtype"] == "float64"
                                     for n in scalar["nodes"])
        self.in_dtypes = list(in_dtypes)
        self.out_dtypes = list(out_dtypes)
        self.out_refs = list(out_refs)
        self.inner = list(inner)
        self.nd = nd
        self.vec = vec
        self.block = block
        self.idx64 = idx64
        self.reduce = reduce
        self.unroll = unroll   # independent vectors in flight per lane (flat 1-d shape only)
        self.nt = nt           # non-temporal (streaming) loads for read-once operands
        # per-input flag: operand is a true scalar (all strides zero) -> loop invariant
        self.invariant = list(invariant) if invariant else [False] * len(self.in_dtypes)
        # tiled form (generate_tiled): dim whose 64-element runs are staged through LDS for the
        # operands of class 't' (unit stride along tile_dim instead of along the last dim)
        self.tile_dim = tile_dim
        assert len(self.inner) == len(self.in_dtypes) + len(self.out_dtypes)
        assert 1 <= nd <= AHIP_MAXD and len(self.inner) <= AHIP_MAXOPS

    def key(self):
        fields = [self.in_dtypes, self.out_dtypes, self.out_refs, self.inner, self.nd, self.vec,
                  self.block, self.idx64, self.reduce, self.unroll, self.nt, self.invariant,
                  self.tile_dim, "v11" if self.tile_dim else "v10b", self._variant()]
        return _memo_key([self.scalar], fields, self._key)

    def _variant(self):
        return "r4%d%d%d%d%d%s%s" % (self.early, self.blocked, self.trace, self.fast_exp, 0,
                                     "H" if self.hjobs else "", "D%d" % knobs.get("FASTDIV"))

    def _key(self):
        import json
        blob = json.dumps([self.scalar, self.in_dtypes, self.out_dtypes, self.out_refs,
                           self.inner, self.nd, self.vec, self.block, self.idx64, self.reduce,
                           self.unroll, self.nt, self.invariant, "v10", self._variant()] +
                          ([["tile2", self.tile_dim]] if self.tile_dim is not None else []),
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def _offset_code(spec, nops, nd_lo, nd_hi, var, idx_t, inner_vecs=None):
    """Index decomposition of `var` over dims [nd_lo, nd_hi) (outermost first), accumulating
    per-operand offsets into off<k>.  If inner_vecs, the innermost dim is counted in vectors
    and its index is left in `inner` (not multiplied into the offsets)."""
    L = []
    dims = list(range(nd_lo, nd_hi))
    L.append("      %s rem = %s;" % (idx_t, var))
    for pos, d in enumerate(reversed(dims)):
        last = pos == len(dims) - 1
        is_inner = (d == nd_hi - 1) and inner_vecs
        ext = inner_vecs if is_inner else "(%s)a.shape[%d]" % (idx_t, d)
        if last:
            L.append("      { const %s r = rem;" % idx_t)
        else:
            L.append("      { const %s q = rem / %s; const %s r = rem - q * %s; rem = q;" %
                     (idx_t, ext, idx_t, ext))
        if is_inner:
            L.append("        inner = r;")
        else:
            for k in range(nops):
                L.append("        off%d += (i64)r * a.stride[%d][%d];" % (k, k, d))
        L.append("      }")
    return L


TRACE_SLOTS = 8          # 8-byte stamps per workgroup of an EW_TRACE build (after the 4 KiB tail of ws)
TRACE_HALF = 2048        # workgroup slots per half of the trace area (even / odd launch epochs)


def _kernel_prologue(spec, name, L, mid=None, pre=None):
    """Kernel head shared by generate / generate_tiled: operand pointers, accumulator, and the
    loop-invariant part (scalar operands, sub-expressions of them, reciprocals of invariant
    divisors).  ``pre(L)`` / ``mid(L)`` may emit code right after the operand pointers (index
    arithmetic on kernel arguments) and after the small loads of the head (the early first loads
    of a flat reduction).  Returns (hoisted, inv_in) for emit_scalar_body."""
    nin = len(spec.in_dtypes)
    nout = len(spec.out_dtypes)
    nops = nin + nout
    V = spec.vec
    red = spec.reduce
    L.append(PRELUDE)
    hj = getattr(spec, "hjobs", False)
    if hj:
        # horizontally fused form: find this workgroup's job, then build the single-job view `a`
        # the rest of the kernel is written against (flat operands: shape[0] = n, no strides)
        L.append('extern "C" __global__ __launch_bounds__(%d) void %s(ArgsH h) {' % (spec.block, name))
        L.append("  unsigned job_ = 0;")
        L.append("  for (int j = 1; j < h.njobs; ++j) if (blockIdx.x >= h.wg0[j]) job_ = j;")
        L.append("  const unsigned lb_ = blockIdx.x - h.wg0[job_], gj_ = h.wg0[job_ + 1] - h.wg0[job_];")
        L.append("  const unsigned slot0_ = h.wg0[job_];")
        L.append("  struct { i64 n; i64 shape[1]; void* ptr[AHIP_HOPS]; void* ws; void* out; i64 aux1; } a;")
        L.append("  a.n = h.n[job_]; a.shape[0] = a.n; a.ws = h.ws; a.out = h.out[job_]; a.aux1 = h.aux1;")
        L.append("  for (int k = 0; k < %d; ++k) a.ptr[k] = h.ptr[job_][k];" % nops)
    else:
        L.append('extern "C" __global__ __launch_bounds__(%d) void %s(Args a) {' % (spec.block, name))
    for k in range(nin):
        L.append("  const %s* __restrict__ p%d = (const %s*)a.ptr[%d];" %
                 (CTYPE[spec.in_dtypes[k]], k, CTYPE[spec.in_dtypes[k]], k))
    for k in range(nout):
        L.append("  %s* __restrict__ p%d = (%s*)a.ptr[%d];" %
                 (CTYPE[spec.out_dtypes[k]], nin + k, CTYPE[spec.out_dtypes[k]], nin + k))
    if any(c == "s" for c in spec.inner):
        assert V == 1
        for k in range(nops):
            if spec.inner[k] == "s":
                L.append("  const i64 is%d = a.stride[%d][%d];" % (k, k, spec.nd - 1))

    if getattr(spec, "trace", False):
        # stamps are kept in registers until the launch epoch is known: even and odd epochs write
        # to two halves of the trace area, so the stamps of two CONSECUTIVE launches survive
        # (end of one launch -> first wavefront of the next, on one clock)
        L.append("  unsigned long long tr_s_[8] = {0, 0, 0, 0, 0, 0, 0, 0};")
        L.append("  if (threadIdx.x == 0) { tr_s_[0] = __builtin_readcyclecounter(); tr_s_[1] = wall_clock64(); "
                 "unsigned hw_; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\" : \"=s\"(hw_)); "
                 "unsigned xcc_; asm volatile(\"s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\" : \"=s\"(xcc_)); "
                 "tr_s_[7] = ((unsigned long long)xcc_ << 32) | hw_; }")
    # every load the head of the kernel depends on is ISSUED before anything waits: the small
    # ones first (exp-table entry, scalar operands, launch epoch: they come back from the
    # memory-side cache), then — ``mid`` — the first group of a flat reduction's stream, so the
    # invariant arithmetic below runs while the stream's first bytes are in flight
    if pre is not None:
        pre(L)
    fast_exp = getattr(spec, "fast_exp", False)
    if fast_exp:
        L.append("  const double etv_ = AHIP_EXP2_64[threadIdx.x & 63];")
    hoisted = {}
    inv_in = {}
    if any(spec.invariant):
        for k in range(nin):
            if spec.invariant[k]:
                e = "xinv%d" % k
                L.append("  const %s %s = p%d[0];" % (CTYPE[spec.in_dtypes[k]], e, k))
                inv_in[k] = "(%s != 0)" % e if spec.in_dtypes[k] == "bool" else e
    if red is not None and red["kind"] == "all":
        # launch epoch of the finalize (read early: its latency hides under the streaming loop);
        # a fused launch keeps one epoch word per job (a job's collector advances it when all of
        # THAT job's workgroups have published, i.e. have read it)
        L.append("  unsigned* const epochp = (unsigned*)((char*)a.ws + a.aux1 + 2048 + %s);" %
                 ("256 + 4 * job_" if hj else "64"))
        L.append("  const unsigned ep0 = __hip_atomic_load(epochp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
    if mid is not None:
        mid(L)
    if fast_exp:
        # one copy of the 2^(j/64) table per wavefront: lane j writes entry j of its wave's copy
        # and the wave reads only that copy, so there is no workgroup barrier (LDS operations of
        # one wave complete in order; the asm keeps the compiler from moving reads above it)
        L.append("  __shared__ double exptbl_[%d];" % spec.block)
        L.append("  exptbl_[threadIdx.x] = etv_;")
        L.append("  asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");")
        L.append("  const double* const etbl_ = exptbl_ + (threadIdx.x & ~63u);")
    if red is not None:
        acc_t = RTYPE[red["acc"]]
        L.append("  %s acc = %s;" % (acc_t, red_identity(red["op"], red["acc"])))

    # loop-invariant prologue: sub-expressions that depend only on scalar operands are computed
    # once per thread, and reciprocals of invariant divisors are hoisted
    if any(spec.invariant):
        inv_nodes = invariant_nodes(spec.scalar, spec.invariant)
        if inv_nodes:
            ins0 = [inv_in.get(k, "0") for k in range(nin)]
            lines, _, _ = emit_scalar_body(spec.scalar, ins0, spec.in_dtypes, indent="  ",
                                           suffix="_inv", only=inv_nodes)
            L.extend(lines)
            divisors = {n["in"][1][1] for n in spec.scalar["nodes"]
                        if n["op"] == "true_div" and n["in"][1][0] == "t"}
            for k in sorted(inv_nodes):
                dt = spec.scalar["nodes"][k]["dtype"]
                rname = None
                if k in divisors and _is_float(dt):
                    rname = "r%d_inv, ok%d_inv" % (k, k)
                    L.append("  const %s r%d_inv = (%s)1 / t%d_inv;" % (RTYPE[dt], k, RTYPE[dt], k))
                    L.append("  const bool ok%d_inv = recip_ok(t%d_inv, r%d_inv);" % (k, k, k))
                hoisted[k] = ("t%d_inv" % k, rname)
    return hoisted, inv_in


REDUCE_ERR_OFF = 128     # error word of the in-kernel finalize: bytes past the shard sums (epoch at +64)
REDUCE_HOSTFLAG_OFF = 192  # 8-byte pointer to a host-mapped (pinned) flag the failing launch also raises


def wave_fold_lines(acc_t, comb, var="acc", indent="    "):
    """Fold `var` over the 64 lanes of a wavefront in a fixed tree, without LDS round trips: four
    DPP steps leave every lane of a 16-lane row with the row's fold, the four row leaders are
    then combined in lane order (uniform reads).  Every lane ends with the wave's fold."""
    out = []
    for ctrl in ("0xB1", "0x4E", "0x141", "0x140"):
        out.append("%s%s = %s;" % (indent, var, comb(var, "dpp_mov_<%s, %s>(%s)" % (acc_t, ctrl, var))))
    rows = ["lane_get_<%s>(%s, %d)" % (acc_t, var, 16 * r) for r in range(4)]
    out.append("%s%s = %s;" % (indent, var, comb(comb(comb(rows[0], rows[1]), rows[2]), rows[3])))
    return out


COLLECT_K = 4            # partials per lane per polling round of the finalize (8 granule loads in flight;
#                          8 per lane would put the kernel above 64 VGPRs = one 1024-thread workgroup per CU)


def _reduce_all_finalize(spec, red, L):
    """Deterministic in-kernel finalize of a full reduction (appended after the streaming loop:
    `acc` holds the thread's partial).

    ONE hop on a static tree, no tickets and no fences: every workgroup folds its threads (DPP
    inside a wavefront, the waves in order through LDS) and publishes its partial as two
    epoch-tagged 8-byte granules {hi32 | epoch}, {lo32 | epoch} (agent-scope write-through
    stores, single-copy atomic).  Workgroup 0 then collects: wavefront w takes the partials
    [256 w, 256 w + 256), four per lane and all eight granule loads of a lane in flight at
    once, re-reading until every tag carries this launch's epoch; lanes fold their four in index
    order, the wave folds by DPP, the collecting waves in order through LDS (the other waves of
    workgroup 0 have exited: the barrier counts live waves only).  With the default 1024-thread
    workgroups a launch has <= 512 partials: two wavefronts collect side by side and the critical
    path after the last workgroup has streamed is one store -> load visibility latency.  The
    workspace is zero-initialised once; epoch 0 never matches a live tag; every spin is bounded
    and a partial that never arrives raises the error word (a float result is NaN)."""
    acc_t = RTYPE[red["acc"]]
    sm_t = acc_t if acc_t != "bool" else "unsigned char"
    comb = lambda a_, b_: red_combine(red["op"], red["acc"], a_, b_)  # noqa: E731
    nw = spec.block // 64
    AG = "__ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT"
    K = COLLECT_K
    ident = red_identity(red["op"], red["acc"])
    tr = getattr(spec, "trace", False)

    L.append("  __shared__ %s sm[%d];" % (sm_t, nw))
    L.append("  unsigned long long* wsp = (unsigned long long*)a.ws;")
    L.append("  unsigned* errp = (unsigned*)((char*)a.ws + a.aux1 + 2048 + %d);" % REDUCE_ERR_OFF)
    L.append("  const unsigned ep = ep0 + 1u;")
    L.append("  const unsigned wv_ = threadIdx.x >> 6, ln_ = threadIdx.x & 63u;")
    # workgroup partial: wave folds, then the waves in order
    L.append("  {")
    L.extend(wave_fold_lines(acc_t, comb))
    if nw > 1:
        L.append("    if (ln_ == 0) sm[wv_] = acc;")
        L.append("    __syncthreads();")
    L.append("    if (threadIdx.x == 0) {")
    if nw > 1:
        L.append("      %s r = sm[0];" % acc_t)
        L.append("      for (int w = 1; w < %d; ++w) r = %s;" % (nw, comb("r", "(%s)sm[w]" % acc_t)))
    else:
        L.append("      %s r = acc;" % acc_t)
    L.append("      union { unsigned long long u; %s v; } cv; cv.u = 0; cv.v = r;" % acc_t)
    L.append("      unsigned long long* slot = wsp + 2 * (size_t)blockIdx.x;")
    L.append("      __hip_atomic_store(slot, ((cv.u >> 32) << 32) | ep, %s);" % AG)
    L.append("      __hip_atomic_store(slot + 1, (cv.u << 32) | ep, %s);" % AG)
    if tr:
        L.append("      tr_s_[4] = wall_clock64();")
        L.append("      unsigned long long* const tr_ = (unsigned long long*)((char*)a.ws + a.aux1 + 4096) + "
                 "%d * ((size_t)blockIdx.x + (ep & 1u) * %d);" % (TRACE_SLOTS, TRACE_HALF))
        L.append("      for (int q = 0; q < 8; ++q) if (q < 5 || q == 7) tr_[q] = tr_s_[q];")
    L.append("    }")
    L.append("  }")
    hj = getattr(spec, "hjobs", False)
    L.append("  if (%s != 0) return;" % ("lb_" if hj else "blockIdx.x"))
    # ---- workgroup 0 (of the job): collect
    L.append("  const unsigned G_ = %s;" % ("gj_" if hj else "gridDim.x"))
    if hj:
        L.append("  wsp += 2 * (size_t)slot0_;             // this job's partial slots")
    L.append("  const unsigned ncol_ = (G_ + %du) / %du < %du ? (G_ + %du) / %du : %du;   // collecting waves" %
             (64 * K - 1, 64 * K, nw, 64 * K - 1, 64 * K, nw))
    L.append("  const bool one_wave = ncol_ <= 1u;")
    L.append("  if (wv_ >= ncol_) return;")
    if nw > 1:
        L.append("  if (!one_wave) __syncthreads();        // sm[] is reused below (live waves only)")
    L.append("  acc = %s;" % ident)
    L.append("  for (unsigned c0 = wv_ * %du; c0 < G_; c0 += %du) {" % (64 * K, 64 * K * nw))
    L.append("    unsigned long long g0[%d], g1[%d];" % (K, K))
    L.append("    bool seen = false;")
    L.append("    for (int spin = 0; spin < (1 << 24); ++spin) {")
    L.append("      bool all_ = true;")
    L.append("#pragma unroll")
    L.append("      for (int k = 0; k < %d; ++k) {" % K)
    L.append("        const unsigned idx = c0 + (unsigned)k * 64u + ln_;")
    L.append("        if (idx < G_) {")
    L.append("          g0[k] = __hip_atomic_load(wsp + 2 * (size_t)idx, %s);" % AG)
    L.append("          g1[k] = __hip_atomic_load(wsp + 2 * (size_t)idx + 1, %s);" % AG)
    L.append("        }")
    L.append("      }")
    L.append("#pragma unroll")
    L.append("      for (int k = 0; k < %d; ++k) {" % K)
    L.append("        const unsigned idx = c0 + (unsigned)k * 64u + ln_;")
    L.append("        if (idx < G_) all_ = all_ && (unsigned)g0[k] == ep && (unsigned)g1[k] == ep;")
    L.append("      }")
    L.append("      if (all_) { seen = true; break; }")
    L.append("      __builtin_amdgcn_s_sleep(1);")
    L.append("    }")
    # a partial that never arrived (the producing workgroup starved for ~2^24 polls: a device
    # shared with something that never yields) must not become a silently wrong sum: the launch
    # tags the device error word with ITS epoch (so only its own float result becomes NaN: later
    # launches carry other epochs and nothing has to be cleared) and raises a flag in pinned host
    # memory, which the executor looks at after every call without touching the device
    L.append("    if (!seen) {")
    L.append("      __hip_atomic_store(errp, ep, %s);" % AG)
    L.append("      unsigned* hostp = *(unsigned* volatile*)((char*)a.ws + a.aux1 + 2048 + %d);" % REDUCE_HOSTFLAG_OFF)
    L.append("      if (hostp) __hip_atomic_store(hostp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);")
    L.append("    }")
    L.append("#pragma unroll")
    L.append("    for (int k = 0; k < %d; ++k) {" % K)
    L.append("      const unsigned idx = c0 + (unsigned)k * 64u + ln_;")
    L.append("      if (idx < G_) {")
    L.append("        union { unsigned long long u; %s v; } cv; cv.u = ((g0[k] >> 32) << 32) | (g1[k] >> 32);" % acc_t)
    L.append("        acc = %s;" % comb("acc", "cv.v"))
    L.append("      }")
    L.append("    }")
    L.append("  }")
    if tr:
        L.append("  if (threadIdx.x == 0) tr_s_[5] = wall_clock64();")
    L.append("  {")
    L.extend(wave_fold_lines(acc_t, comb))
    if nw > 1:
        L.append("    if (!one_wave) {")
        L.append("      if (ln_ == 0) sm[wv_] = acc;")
        L.append("      __syncthreads();")
        L.append("      acc = sm[0];")
        L.append("      for (unsigned w = 1; w < ncol_; ++w) acc = %s;" % comb("acc", "(%s)sm[w]" % acc_t))
        L.append("    }")
    L.append("    if (threadIdx.x == 0) {")
    L.append("      %s r = acc;" % acc_t)
    if _is_float(red["acc"]):
        L.append("      if (__hip_atomic_load(errp, %s) == ep) r = (%s)__builtin_nan(\"\");" % (AG, acc_t))
    L.append("      *(%s*)a.out = %s;" % (CTYPE[red["out"]], _store_val("r", red["acc"], red["out"])))
    L.append("      __hip_atomic_store(epochp, ep, %s);" % AG)
    if tr:
        L.append("      tr_s_[6] = wall_clock64();")
        L.append("      unsigned long long* const tr_ = (unsigned long long*)((char*)a.ws + a.aux1 + 4096) + "
                 "%d * ((size_t)blockIdx.x + (ep & 1u) * %d);" % (TRACE_SLOTS, TRACE_HALF))
        L.append("      tr_[5] = tr_s_[5]; tr_[6] = tr_s_[6];")
    L.append("    }")
    L.append("  }")


def generate_tiled(spec: KernelSpec):
    """K3t — Elemwise (optionally + full reduction) with TRANSPOSED operands.

    The reference walks such operands with its strided loop nest (`elemwise_cgen.py:228-305`
    make_loop with per-operand strides); a lane-per-element GPU loop would fetch one element per
    cache line from them.  Here the last dim and `tile_dim` are cut into T x T tiles: operands of
    class 't' (unit stride along tile_dim) are read with the lanes running along tile_dim — full
    lines — into a padded LDS tile, then the scalar body runs with the lanes along the last dim,
    taking those operands from LDS (conflict-free thanks to the odd row pitch) and every other
    operand / output directly (coalesced for class 'c', broadcast for 'b').  One workgroup of 256
    threads per tile; remaining dims are decomposed from the tile index."""
    nin = len(spec.in_dtypes)
    nout = len(spec.out_dtypes)
    nops = nin + nout
    nd = spec.nd
    td, T = spec.tile_dim
    red = spec.reduce
    assert spec.vec == 1 and spec.block == 256 and T in (32, 64) and 0 <= td < nd - 1
    assert red is None or red["kind"] == "all"
    LP = 256 // T          # tile lines covered per pass
    P = T // LP            # passes
    name = "ewt_" + spec.key()
    L = []
    hoisted, inv_in = _kernel_prologue(spec, name, L)
    tk = [k for k in range(nin) if spec.inner[k] == "t" and k not in inv_in]
    assert tk and all(spec.inner[k] != "t" for k in range(nin, nops))
    for k in tk:
        L.append("  __shared__ %s tile%d[%d][%d];" % (CTYPE[spec.in_dtypes[k]], k, T, T + 1))
    outer = [d for d in range(nd - 1) if d != td]
    L.append("  const i64 R = a.shape[%d], C = a.shape[%d];" % (td, nd - 1))
    L.append("  const i64 tr = (R + %d) / %d, tc = (C + %d) / %d;" % (T - 1, T, T - 1, T))
    L.append("  const i64 ntiles = tr * tc%s;" % "".join(" * a.shape[%d]" % d for d in outer))
    L.append("  const int ta = threadIdx.x %% %d, tb = threadIdx.x / %d;" % (T, T))
    L.append("  for (i64 tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {")
    L.append("    i64 rem = tix;")
    L.append("    const i64 jc = rem % tc; rem /= tc;")
    L.append("    const i64 jr = rem % tr; rem /= tr;")
    live = [k for k in range(nops) if k not in inv_in]
    L.append("    i64 " + ", ".join("off%d = 0" % k for k in live) + ";")
    for d in reversed(outer):
        L.append("    { const i64 q = rem / a.shape[%d]; const i64 r = rem - q * a.shape[%d]; rem = q;"
                 % (d, d))
        for k in live:
            L.append("      off%d += r * a.stride[%d][%d];" % (k, k, d))
        L.append("    }")
    L.append("    const i64 r0 = jr * %d, c0 = jc * %d;" % (T, T))
    # every load of the tile is issued up front with clamped (always valid) coordinates — no
    # branches between them, so P x (operands) requests per lane are in flight at once; only the
    # compute / store / accumulate is guarded at ragged edges.
    direct = [k for k in range(nin) if k not in inv_in and k not in tk]

    def addr(k, r, c):
        inner = {"c": " + %s" % c, "b": "", "s": " + %s * a.stride[%d][%d]" % (c, k, nd - 1)}[
            spec.inner[k]]
        return "off%d + %s * a.stride[%d][%d]%s" % (k, r, k, td, inner)

    L.append("    const i64 rl = (r0 + ta < R) ? r0 + ta : R - 1;   // phase 1: lanes along tile_dim")
    L.append("    const i64 cq = (c0 + ta < C) ? c0 + ta : C - 1;   // phase 2: lanes along the last dim")
    for k in tk + direct:
        L.append("    %s v%d[%d];" % (CTYPE[spec.in_dtypes[k]], k, P))
    L.append("#pragma unroll")
    L.append("    for (int j = 0; j < %d; ++j) {" % P)
    L.append("      const i64 cl = (c0 + tb + j * %d < C) ? c0 + tb + j * %d : C - 1;" % (LP, LP))
    for k in tk:
        L.append("      v%d[j] = p%d[off%d + rl + cl * a.stride[%d][%d]];" % (k, k, k, k, nd - 1))
    L.append("    }")
    if direct:
        L.append("#pragma unroll")
        L.append("    for (int j = 0; j < %d; ++j) {" % P)
        L.append("      const i64 rq = (r0 + tb + j * %d < R) ? r0 + tb + j * %d : R - 1;" % (LP, LP))
        for k in direct:
            L.append("      v%d[j] = p%d[%s];" % (k, k, addr(k, "rq", "cq")))
        L.append("    }")
    L.append("#pragma unroll")
    L.append("    for (int j = 0; j < %d; ++j) {" % P)
    for k in tk:
        L.append("      tile%d[tb + j * %d][ta] = v%d[j];" % (k, LP, k))
    L.append("    }")
    L.append("    __syncthreads();")
    L.append("#pragma unroll")
    L.append("    for (int j = 0; j < %d; ++j) {" % P)
    L.append("      const int rr = tb + j * %d;" % LP)
    L.append("      const i64 r = r0 + rr, c = c0 + ta;")
    L.append("      if (r < R && c < C) {")
    ins = []
    for k in range(nin):
        if k in inv_in:
            ins.append(inv_in[k])
            continue
        ct = CTYPE[spec.in_dtypes[k]]
        if k in tk:
            L.append("        const %s x%d = tile%d[ta][rr];" % (ct, k, k))
        else:
            L.append("        const %s x%d = v%d[j];" % (ct, k, k))
        ins.append("(x%d != 0)" % k if spec.in_dtypes[k] == "bool" else "x%d" % k)
    lines, outs, odts = emit_scalar_body(spec.scalar, ins, spec.in_dtypes, indent="        ",
                                         suffix="_t", hoisted=hoisted)
    L.extend(lines)
    for k, ri in enumerate(spec.out_refs):
        val = _store_val(outs[ri], odts[ri], spec.out_dtypes[k])
        L.append("        p%d[%s] = %s;" % (nin + k, addr(nin + k, "r", "c"), val))
    if red is not None:
        val = _cast(outs[red["ref"]], odts[red["ref"]], red["acc"])
        L.append("        acc = %s;" % red_combine(red["op"], red["acc"], "acc", val))
    L.append("      }")
    L.append("    }")
    L.append("    __syncthreads();")
    L.append("  }")
    if red is not None:
        _reduce_all_finalize(spec, red, L)
    L.append("}")
    return "\n".join(L) + "\n", (name,)


def generate(spec: KernelSpec):
    """Return (source, kernel_names) for a spec (one kernel per spec)."""
    if spec.tile_dim is not None:
        return generate_tiled(spec)
    nin = len(spec.in_dtypes)
    nout = len(spec.out_dtypes)
    nops = nin + nout
    V = spec.vec
    U = spec.unroll
    idx_t = "i64" if spec.idx64 else "int"
    red = spec.reduce
    name = "ew_" + spec.key()
    L = []
    etbl = "etbl_" if spec.fast_exp else None
    flat_u = (red is None or red["kind"] == "all") and spec.nd == 1 and U > 1
    early = spec.early and flat_u
    inv_set = {k for k in range(nin) if spec.invariant[k]}

    def index_setup(L_):
        # the walk over the items (vectors of V elements): grid-stride, or (blocked) one
        # contiguous chunk per workgroup, a whole number of vectors per thread
        L_.append("  const %s inner_vecs = (%s)(a.shape[%d] / %d);" % (idx_t, idx_t, spec.nd - 1, V))
        if spec.blocked:
            L_.append("  const %s items_all = (%s)(a.n / %d);" % (idx_t, idx_t, V))
            grd = "gj_" if spec.hjobs else "gridDim.x"
            L_.append("  const %s chunk_ = ((items_all + (%s)%s - 1) / (%s)%s + %d) / %d * %d;" %
                      (idx_t, idx_t, grd, idx_t, grd, spec.block - 1, spec.block, spec.block))
            if spec.blocked == 2:
                # workgroup b runs on XCD b % 8 (observed dispatch order, a speed hint only): give
                # every XCD one contiguous eighth of the stream, so an XCD's L2 / TLB sees 1/8 of
                # the pages instead of all of them
                L_.append("  const unsigned vb_ = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + "
                          "blockIdx.x / 8u : blockIdx.x;")
            else:
                L_.append("  const unsigned vb_ = %s;" % ("lb_" if spec.hjobs else "blockIdx.x"))
            L_.append("  const %s beg_ = (%s)vb_ * chunk_;" % (idx_t, idx_t))
            L_.append("  const %s items = beg_ + chunk_ < items_all ? beg_ + chunk_ : items_all;" % idx_t)
            L_.append("  const %s step = %d;" % (idx_t, spec.block))
            L_.append("  %s item = beg_ + threadIdx.x;" % idx_t)
        else:
            L_.append("  const %s items = (%s)(a.n / %d);" % (idx_t, idx_t, V))
            L_.append("  const %s step = (%s)gridDim.x * %d;" % (idx_t, idx_t, spec.block))
            L_.append("  %s item = (%s)blockIdx.x * %d + threadIdx.x;" % (idx_t, idx_t, spec.block))

    flat = [{"c": "(i64)%%s * %d" % V, "b": "0", "s": "(i64)%%s * is%d" % k}[spec.inner[k]]
            for k in range(nops)]

    def loads(elem_off_exprs, sfx="", decl=True):
        B = []
        for k in range(nin):
            ct = CTYPE[spec.in_dtypes[k]]
            cls = spec.inner[k]
            if k in inv_set:
                continue
            if cls == "c" and V > 1:
                ptr = "(const Pack<%s, %d>*)(p%d + %s)" % (ct, V, k, elem_off_exprs[k])
                head = "const Pack<%s, %d> " % (ct, V) if decl else ""
                if int(spec.nt) & 1:
                    B.append("      %sx%d%s = nt_load(%s);" % (head, k, sfx, ptr))
                else:
                    B.append("      %sx%d%s = *%s;" % (head, k, sfx, ptr))
            else:
                B.append("      %sx%d%s = p%d[%s];" % ("const %s " % ct if decl else "", k, sfx, k,
                                                     elem_off_exprs[k]))
        return B

    def early_loads(L_):
        # the first group of U vectors per lane, issued straight after the kernel arguments: the
        # invariant prologue below (dependent scalar loads of mu / sigma, a full-precision
        # reciprocal, the exp table) then runs while they are in flight
        for u in range(U):
            for k in range(nin):
                if k in inv_set:
                    continue
                ct = CTYPE[spec.in_dtypes[k]]
                L_.append("  %s x%d_e%d;" % ("Pack<%s, %d>" % (ct, V) if spec.inner[k] == "c" and V > 1
                                              else ct, k, u))
        # issued unconditionally, at clamped positions (no branch: the compiler then knows how
        # many loads are outstanding and waits for the exp table / scalars only); an empty
        # operand is re-pointed at the workspace by the launcher, so position 0 is always readable
        L_.append("  const bool first_ = item + %d * step < items;" % (U - 1))
        L_.append("  const %s last_ = items > 0 ? items - 1 : 0;" % idx_t)
        for u in range(U):
            L_.append("  const %s ie%d_ = item + %d * step < last_ ? item + %d * step : last_;" %
                      (idx_t, u, u, u))
            it = "ie%d_" % u
            L_.extend(loads([f % it if "%s" in f else f for f in flat], "_e%d" % u, decl=False))

    hoisted, inv_in = _kernel_prologue(spec, name, L, mid=early_loads if early else None,
                                       pre=index_setup if early else None)

    def compute(elem_off_exprs, sfx="", accs=None):
        B = []
        accs = accs or ["acc"] * V
        for k in range(nout):
            if V > 1:
                B.append("      Pack<%s, %d> y%d%s;" % (CTYPE[spec.out_dtypes[k]], V, k, sfx))
        for v in range(V):
            ins = []
            for k in range(nin):
                if k in inv_in:
                    ins.append(inv_in[k])
                    continue
                if spec.inner[k] == "c" and V > 1:
                    e = "x%d%s.v[%d]" % (k, sfx, v)
                else:
                    e = "x%d%s" % (k, sfx)
                if spec.in_dtypes[k] == "bool":
                    e = "(%s != 0)" % e
                ins.append(e)
            lines, outs, odts = emit_scalar_body(spec.scalar, ins, spec.in_dtypes,
                                                 suffix="_%d%s" % (v, sfx), hoisted=hoisted,
                                                 exp_tbl=etbl,
                                                 sum_only=sum_only_nodes(spec.scalar, red, spec.out_refs))
            B.extend(lines)
            for k, ri in enumerate(spec.out_refs):
                val = _cast(outs[ri], odts[ri], spec.out_dtypes[k])
                if spec.out_dtypes[k] == "bool":
                    val = "(unsigned char)(%s)" % val
                if V > 1:
                    B.append("      y%d%s.v[%d] = %s;" % (k, sfx, v, val))
                else:
                    B.append("      p%d[%s] = %s;" % (nin + k, elem_off_exprs[nin + k], val))
            if red is not None:
                val = _cast(outs[red["ref"]], odts[red["ref"]], red["acc"])
                B.append("      %s = %s;" % (accs[v], red_combine(red["op"], red["acc"], accs[v], val)))
        if V > 1:
            for k in range(nout):
                if int(spec.nt) & 2:
                    B.append("      nt_store((Pack<%s, %d>*)(p%d + %s), y%d%s);" %
                             (CTYPE[spec.out_dtypes[k]], V, nin + k, elem_off_exprs[nin + k], k, sfx))
                else:
                    B.append("      *(Pack<%s, %d>*)(p%d + %s) = y%d%s;" %
                             (CTYPE[spec.out_dtypes[k]], V, nin + k, elem_off_exprs[nin + k], k, sfx))
        return B

    def elem_offsets(inner="inner", off_sfx=""):
        out = []
        for k in range(nops):
            cls = spec.inner[k]
            if cls == "c":
                out.append("off%d%s + (i64)%s * %d" % (k, off_sfx, inner, V))
            elif cls == "b":
                out.append("off%d%s" % (k, off_sfx))
            else:
                out.append("off%d%s + (i64)%s * is%d" % (k, off_sfx, inner, k))
        return out

    tstamp = (lambda k: L.append("  if (threadIdx.x == 0) tr_s_[%d] = wall_clock64();" % k)) \
        if spec.trace else (lambda k: None)
    if red is None or red["kind"] == "all":
        nd = spec.nd
        if not early:
            index_setup(L)
        if nd == 1 and U > 1:
            # flat streaming shape: U independent vectors in flight per lane, loads first
            if early:
                L.append("  if (first_) {")
                for u in range(U):
                    it = "(item + %d * step)" % u
                    L.extend(compute([f % it if "%s" in f else f for f in flat], "_e%d" % u))
                L.append("    item += %d * step;" % U)
                L.append("  }")
                tstamp(2)
            L.append("  for (; item + %d * step < items; item += %d * step) {" % (U - 1, U))
            for u in range(U):
                it = "(item + %d * step)" % u
                L.extend(loads([f % it if "%s" in f else f for f in flat], "_u%d" % u))
            for u in range(U):
                it = "(item + %d * step)" % u
                L.extend(compute([f % it if "%s" in f else f for f in flat], "_u%d" % u))
            L.append("  }")
        L.append("  for (; item < items; item += step) {")
        L.append("      i64 " + ", ".join("off%d = 0" % k for k in range(nops)) + ";")
        L.append("      %s inner = 0;" % idx_t)
        L.extend(_offset_code(spec, nops, 0, nd, "item", idx_t, inner_vecs="inner_vecs"))
        eo = elem_offsets()
        L.extend(loads(eo))
        L.extend(compute(eo))
        L.append("  }")
        tstamp(3)
    else:
        # K2 axis reductions.  Dims are [kept (nk) | reduced (nr)]; `lanes` threads cooperate:
        #   row: the unit stride is in the reduced group.  `lanes` (<= 64, power of two) adjacent
        #        lanes share one output, each taking vectors of V elements of the flattened
        #        reduced index; gridDim.y slices long reductions (partials + fold pass).
        #   col: the unit stride is in the kept group.  A workgroup is TX x TY threads: TX =
        #        `lanes` along the kept index (V adjacent outputs per thread), TY rows of the
        #        reduced index walked concurrently; folded with shuffles, then across the waves
        #        through LDS, in a fixed order.
        nk, nr = red["nk"], red["nr"]
        G = red.get("lanes", 64 if red["kind"] == "row" else spec.block)
        acc_t = RTYPE[red["acc"]]
        comb = lambda a_, b_: red_combine(red["op"], red["acc"], a_, b_)  # noqa: E731
        vec_cls = {"c": " + (i64)inner * %d" % V, "b": "", "s": ""}
        if red["kind"] == "row":
            assert G <= 64 and 64 % G == 0
            L.append("  const int gl = threadIdx.x %% %d;" % G)
            # a workgroup walks the outputs with a grid stride (the launcher caps the grid at a
            # few workgroups per CU): millions of short rows as one output per thread group would
            # be bound by the rate at which wavefronts are DISPATCHED (max over 4 194 304 rows of
            # 8: 32 768 workgroups of one 16-byte load per thread, 32 us = 0.52 of the HBM peak)
            if red.get("short"):
                # every output's reduced run is at most one vector per lane (rows of 8 ... 256
                # elements): the inner loop below runs at most once, so the compiler may keep the
                # loads of several outputs in flight
                L.append("#pragma unroll 4")
            L.append("  for (i64 ob = (i64)blockIdx.x * %d; ob < a.n; ob += (i64)gridDim.x * %d) {" %
                     (spec.block // G, spec.block // G))
            L.append("  const i64 o = ob + threadIdx.x / %d;" % G)
            L.append("  const bool valid = o < a.n;")
            L.append("  acc = %s;" % red_identity(red["op"], red["acc"]))
            L.append("  i64 " + ", ".join("base%d = 0" % k for k in range(nops)) + ";")
            L.append("  if (valid) {")
            L.append("      i64 " + ", ".join("off%d = 0" % k for k in range(nops)) + ";")
            L.extend(_offset_code(spec, nops, 0, nk, "o", idx_t))
            L.append("      " + " ".join("base%d = off%d;" % (k, k) for k in range(nops)))
            L.append("  }")
            L.append("  const %s inner_vecs = (%s)(a.shape[%d] / %d);" % (idx_t, idx_t, nk + nr - 1, V))
            L.append("  const i64 nredv = a.aux0 / %d;" % V)
            L.append("  const i64 per = (nredv + a.aux1 - 1) / a.aux1;")
            L.append("  const i64 rbeg = (i64)blockIdx.y * per;")
            L.append("  const i64 rend = !valid ? 0 : ((rbeg + per < nredv) ? rbeg + per : nredv);")
            # consecutive iterations are independent: unrolling keeps several loads in flight
            if red.get("short"):
                L.append("  if (rbeg + gl < rend) { const i64 r0 = rbeg + gl;")
            else:
                L.append("#pragma unroll %d" % (U if U > 1 else 8))
                L.append("  for (i64 r0 = rbeg + gl; r0 < rend; r0 += %d) {" % G)
            L.append("      i64 " + ", ".join("off%d = base%d" % (k, k) for k in range(nops)) + ";")
            L.append("      %s inner = 0;" % idx_t)
            if V > 1:
                L.extend(_offset_code(spec, nops, nk, nk + nr, "(%s)r0" % idx_t, idx_t,
                                      inner_vecs="inner_vecs"))
                eo = ["off%d%s" % (k, vec_cls[spec.inner[k]]) for k in range(nops)]
            else:
                L.extend(_offset_code(spec, nops, nk, nk + nr, "(%s)r0" % idx_t, idx_t))
                eo = ["off%d" % k for k in range(nops)]
            L.extend(loads(eo))
            L.extend(compute(eo))
            L.append("  }")
            L.append("  for (int m = %d; m > 0; m >>= 1) acc = %s;" %
                     (G // 2, comb("acc", "shfl_xor_<%s>(acc, m)" % acc_t)))
            L.append("  if (valid && gl == 0) {")
            L.append("    if (a.aux1 == 1) ((%s*)a.out)[o] = %s;" %
                     (CTYPE[red["out"]], _store_val("acc", red["acc"], red["out"])))
            L.append("    else ((%s*)a.out)[(i64)blockIdx.y * a.n + o] = acc;" % CTYPE[red["acc"]])
            L.append("  }")
            L.append("  }")      # (grid-stride walk over the outputs)
        else:
            TX = G
            assert (TX <= 64 and 64 % TX == 0 or TX % 64 == 0) and spec.block % TX == 0
            TY = spec.block // TX
            # rows of the LDS fold: the waves (TX <= 64: a wave holds 64 / TX reduced rows, folded
            # by shuffles first) or the TY thread rows (TX > 64: every wave is part of one row)
            nw = spec.block // 64 if TX <= 64 else TY
            accs = ["acc"] + ["acc_%d" % v for v in range(1, V)]
            for nm in accs[1:]:
                L.append("  %s %s = %s;" % (acc_t, nm, red_identity(red["op"], red["acc"])))
            L.append("  const int tx = threadIdx.x %% %d, ty = threadIdx.x / %d;" % (TX, TX))
            L.append("  const i64 nkeptv = a.n / %d;" % V)
            L.append("  const i64 ov = (i64)blockIdx.x * %d + tx;" % TX)
            L.append("  const bool valid = ov < nkeptv;")
            L.append("  i64 " + ", ".join("base%d = 0" % k for k in range(nops)) + ";")
            L.append("  if (valid) {")
            L.append("      i64 " + ", ".join("off%d = 0" % k for k in range(nops)) + ";")
            if V > 1:
                L.append("      const %s inner_vecs = (%s)(a.shape[%d] / %d);" % (idx_t, idx_t, nk - 1, V))
                L.append("      %s inner = 0;" % idx_t)
                L.extend(_offset_code(spec, nops, 0, nk, "(%s)ov" % idx_t, idx_t,
                                      inner_vecs="inner_vecs"))
                L.append("      " + " ".join("base%d = off%d%s;" % (k, k, vec_cls[spec.inner[k]])
                                             for k in range(nops)))
            else:
                L.extend(_offset_code(spec, nops, 0, nk, "(%s)ov" % idx_t, idx_t))
                L.append("      " + " ".join("base%d = off%d;" % (k, k) for k in range(nops)))
            L.append("  }")
            L.append("  const i64 nred = a.aux0;")
            L.append("  const i64 per = (nred + a.aux1 - 1) / a.aux1;")
            L.append("  const i64 rbeg = (i64)blockIdx.y * per;")
            L.append("  const i64 rend = !valid ? 0 : ((rbeg + per < nred) ? rbeg + per : nred);")
            L.append("#pragma unroll %d" % (U if U > 1 else 8))
            L.append("  for (i64 r0 = rbeg + ty; r0 < rend; r0 += %d) {" % TY)
            L.append("      i64 " + ", ".join("off%d = base%d" % (k, k) for k in range(nops)) + ";")
            L.extend(_offset_code(spec, nops, nk, nk + nr, "(%s)r0" % idx_t, idx_t))
            eo = ["off%d" % k for k in range(nops)]
            L.extend(loads(eo))
            L.extend(compute(eo, accs=accs))
            L.append("  }")
            # fold over ty: lanes of a wave that share tx (xor masks >= TX), then the waves in order
            sm_t = acc_t if acc_t != "bool" else "unsigned char"
            if TX < 64:
                for nm in accs:
                    L.append("  for (int m = 32; m >= %d; m >>= 1) %s = %s;" %
                             (TX, nm, comb(nm, "shfl_xor_<%s>(%s, m)" % (acc_t, nm))))
            if nw > 1:
                L.append("  __shared__ %s sm[%d][%d];" % (sm_t, nw, TX * V))
                if TX <= 64:
                    L.append("  if ((threadIdx.x & 63) < %d) {" % TX)
                    row = "threadIdx.x >> 6"
                else:
                    L.append("  {")
                    row = "ty"
                for v, nm in enumerate(accs):
                    L.append("    sm[%s][tx * %d + %d] = %s;" % (row, V, v, nm))
                L.append("  }")
                L.append("  __syncthreads();")
            L.append("  if (valid && threadIdx.x < %d) {" % TX)
            for v, nm in enumerate(accs):
                if nw > 1:
                    L.append("    %s r%d = sm[0][tx * %d + %d];" % (acc_t, v, V, v))
                    L.append("    for (int w = 1; w < %d; ++w) r%d = %s;" %
                             (nw, v, comb("r%d" % v, "(%s)sm[w][tx * %d + %d]" % (acc_t, V, v))))
                else:
                    L.append("    %s r%d = %s;" % (acc_t, v, nm))
                L.append("    if (a.aux1 == 1) ((%s*)a.out)[ov * %d + %d] = %s;" %
                         (CTYPE[red["out"]], V, v, _store_val("r%d" % v, red["acc"], red["out"])))
                L.append("    else ((%s*)a.out)[(i64)blockIdx.y * a.n + ov * %d + %d] = r%d;" %
                         (CTYPE[red["acc"]], V, v, v))
            L.append("  }")

    if red is not None and red["kind"] == "all":
        _reduce_all_finalize(spec, red, L)
    L.append("}")
    return "\n".join(L) + "\n", (name,)


def _store_val(expr, src_dt, dst_dt):
    v = _cast(expr, src_dt, dst_dt)
    if dst_dt == "bool":
        v = "(unsigned char)(%s)" % v
    return v


IDENTITY_SCALAR = {"n_in": 1, "nodes": [], "out": [["i", 0]]}


# ----------------------------------------------------------------------------------------
# fused GEMV-chain + Elemwise epilogue (ROW layout): one wavefront per output row
# ----------------------------------------------------------------------------------------
AHIP_MAXDOTS = 8
AHIP_GV_MAXOPS = 16

GV_STRUCT = r"""
#define AHIP_MAXDOTS %d
#define AHIP_GV_MAXOPS %d
struct GvArgs {
  i64 M;
  const void* A[AHIP_MAXDOTS]; i64 a_rs[AHIP_MAXDOTS]; i64 a_cs[AHIP_MAXDOTS]; i64 K[AHIP_MAXDOTS];
  const void* x[AHIP_MAXDOTS]; i64 incx[AHIP_MAXDOTS];
  void* ptr[AHIP_GV_MAXOPS]; i64 stride[AHIP_GV_MAXOPS];
  int ndots; int nops;
  const void* xin[AHIP_MAXDOTS][4]; void* xout[AHIP_MAXDOTS];
};
""" % (AHIP_MAXDOTS, AHIP_GV_MAXOPS)


class GemvEpiSpec:
    """y[m] = f(dot_0[m], ..., dot_{D-1}[m], operands[m]) with dot_d[m] = A_d[m, :] . x_d.

    Replaces chains of ``Gemv`` nodes (tensor/blas.py:231; ``beta*y + alpha*A.x`` with y another
    Gemv) and the ``Elemwise`` that consumes them — e.g. one GRU gate
    ``sigmoid(W.T x + U.T h) * h`` of BASELINE config 4 — by a single HBM/L2-bound kernel: each
    wavefront owns output rows, streams the D matrix rows with 16-byte loads, reduces with
    cross-lane shuffles and evaluates the scalar epilogue in registers.

    dtype     : float32 | float64 (all matrices / vectors of the dots)
    dot_vec   : per dot, True when rows can be read with 16-byte vectors
    scalar    : plan scalar expression; its first D inputs are the dot results
    in_dtypes : dtypes of the non-dot epilogue operands; out_dtypes/out_refs as in KernelSpec
    """

    def __init__(self, dtype, dot_vec, scalar, in_dtypes, out_dtypes, out_refs, block=256,
                 rpw=1, kvs=None, xprogs=None, nt=False):
        # nt: the matrix rows are read with non-temporal 16-byte loads (a matrix of half the
        # memory-side cache or more is read once as far as the caches go, exec_elemwise.BIG_STREAM)
        self.nt = bool(nt)
        # xprogs: per dot None or {"scalar", "cls": ["v" | "s", ...], "out_ref", "store"}: the
        # dot's vector is an Elemwise of <= 4 vectors / scalars, evaluated while it is loaded
        # (and stored by the first wavefront when something else reads it).  Needs kvs.
        self.xprogs = list(xprogs) if xprogs and any(xprogs) else None
        assert not self.xprogs or kvs
        self.rpw = rpw  # rows per wavefront iteration (4 for short rows, 1 for long rows)
        # kvs: per dot, 16-byte vectors per lane (K = 64 * VEC * kv) when every row length is
        # such a multiple and small: the kernel is then specialised on the lengths and issues
        # ALL row loads of ALL dots before the first FMA (one memory round trip per wavefront
        # instead of one per loop iteration per dot — short kernels are latency-, not
        # bandwidth-bound: BASELINE config 4 step kernels 4.5 -> see DESIGN §3.3)
        self.kvs = list(kvs) if kvs else None
        self.dtype = dtype
        self.dot_vec = list(dot_vec)
        self.scalar = scalar
        self.in_dtypes = list(in_dtypes)
        self.out_dtypes = list(out_dtypes)
        self.out_refs = list(out_refs)
        self.block = block
        assert 1 <= len(self.dot_vec) <= AHIP_MAXDOTS
        assert len(self.in_dtypes) + len(self.out_dtypes) <= AHIP_GV_MAXOPS

    def key(self):
        xp = [None if x is None else [x["cls"], x["out_ref"], x["store"]] for x in (self.xprogs or [])]
        fields = ["gv4", self.dtype, self.dot_vec, self.in_dtypes, self.out_dtypes, self.out_refs,
                  self.block, self.rpw, self.kvs, xp] + (["nt"] if self.nt else [])
        return _memo_key([self.scalar] + [x["scalar"] for x in (self.xprogs or []) if x],
                         fields, self._key)

    def _key(self):
        import json
        blob = json.dumps(["gv4", self.dtype, self.dot_vec, self.scalar, self.in_dtypes,
                           self.out_dtypes, self.out_refs, self.block, self.rpw, self.kvs,
                           self.xprogs] + (["nt"] if self.nt else []),
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def generate_gemv_epilogue(spec: GemvEpiSpec):
    T = RTYPE[spec.dtype]
    V = 4 if spec.dtype == "float32" else 2
    D = len(spec.dot_vec)
    R = spec.rpw
    nin, nout = len(spec.in_dtypes), len(spec.out_dtypes)
    name = "gv_" + spec.key()
    waves = spec.block // 64
    L = [PRELUDE, GV_STRUCT]
    LD = "nt_load(%s)" if spec.nt else "*%s"          # how a 16-byte piece of a matrix row is read
    L.append('extern "C" __global__ __launch_bounds__(%d) void %s(GvArgs a) {' % (spec.block, name))
    L.append("  const int lane = threadIdx.x & 63;")
    L.append("  const i64 nwaves = (i64)gridDim.x * %d;" % waves)
    # each wavefront owns R consecutive output rows per iteration: for short rows (K*itemsize of
    # a few KB) this keeps R independent 16-byte loads in flight per lane instead of one
    L.append("  for (i64 m0 = ((i64)blockIdx.x * %d + (threadIdx.x >> 6)) * %d; m0 < a.M; "
             "m0 += nwaves * %d) {" % (waves, R, R))
    if spec.kvs:
        # ---- fixed lengths: loads of every dot first, then FMAs, then interleaved reductions
        for d in range(D):
            L.append("    const %s* __restrict__ xv%d = (const %s*)a.x[%d];" % (T, d, T, d))
            for r in range(R):
                L.append("    const %s* __restrict__ row%d_%d = (const %s*)a.A[%d] + "
                         "((m0 + %d < a.M) ? (m0 + %d) : (a.M - 1)) * a.a_rs[%d];"
                         % (T, d, r, T, d, r, r, d))
        for d in range(D):
            xp = spec.xprogs[d] if spec.xprogs else None
            for j in range(spec.kvs[d]):
                for r in range(R):
                    L.append("    const Pack<%s, %d> a%d_%d_%d = %s;" % (T, V, d, r, j, LD % (
                        "(const Pack<%s, %d>*)(row%d_%d + (%d + lane) * %d)" % (T, V, d, r, j * 64, V))))
                if xp is None:
                    L.append("    const Pack<%s, %d> x%d_%d = *(const Pack<%s, %d>*)(xv%d + (%d + lane) * %d);"
                             % (T, V, d, j, T, V, d, j * 64, V))
                else:
                    for q, c in enumerate(xp["cls"]):
                        if c == "v":
                            L.append("    const Pack<%s, %d> xi%d_%d_%d = *(const Pack<%s, %d>*)"
                                     "((const %s*)a.xin[%d][%d] + (%d + lane) * %d);"
                                     % (T, V, d, q, j, T, V, T, d, q, j * 64, V))
                        elif j == 0:
                            L.append("    const %s xs%d_%d = *(const %s*)a.xin[%d][%d];" % (T, d, q, T, d, q))
        # vector prologues: x_d evaluated from its operands, stored once if something reads it
        for d in range(D):
            xp = spec.xprogs[d] if spec.xprogs else None
            if xp is None:
                continue
            for j in range(spec.kvs[d]):
                L.append("    Pack<%s, %d> x%d_%d;" % (T, V, d, j))
                for e in range(V):
                    ins_ = ["xi%d_%d_%d.v[%d]" % (d, q, j, e) if c == "v" else "xs%d_%d" % (d, q)
                            for q, c in enumerate(xp["cls"])]
                    lines, oe, od = emit_scalar_body(xp["scalar"], ins_, [spec.dtype] * len(ins_),
                                                     indent="    ", suffix="_xp%d_%d_%d" % (d, j, e))
                    L.extend(lines)
                    L.append("    x%d_%d.v[%d] = %s;" % (d, j, e, _cast(oe[xp["out_ref"]],
                                                                         od[xp["out_ref"]], spec.dtype)))
                if xp["store"]:
                    L.append("    if (m0 == 0) *(Pack<%s, %d>*)((%s*)a.xout[%d] + (%d + lane) * %d) = x%d_%d;"
                             % (T, V, T, d, j * 64, V, d, j))
        for d in range(D):
            for r in range(R):
                terms = ["a%d_%d_%d.v[%d] * x%d_%d.v[%d]" % (d, r, j, e, d, j, e)
                         for j in range(spec.kvs[d]) for e in range(V)]
                # two interleaved accumulation chains per dot (as the generic loop does)
                L.append("    %s d%d_%d = (%s) + (%s);" % (T, d, r, " + ".join(terms[0::2]),
                                                          " + ".join(terms[1::2]) if terms[1::2] else "0"))
        L.append("    for (int s = 32; s > 0; s >>= 1) {")
        for d in range(D):
            for r in range(R):
                L.append("      d%d_%d += shfl_xor_<%s>(d%d_%d, s);" % (d, r, T, d, r))
        L.append("    }")
    for d in (range(D) if not spec.kvs else []):
        for r in range(R):
            L.append("    %s d%d_%d = 0;" % (T, d, r))
        L.append("    {")
        L.append("      const %s* __restrict__ xv = (const %s*)a.x[%d];" % (T, T, d))
        L.append("      const i64 K = a.K[%d];" % d)
        for r in range(R):
            # rows past M are clamped to the last row (their results are never stored)
            L.append("      const %s* __restrict__ row%d = (const %s*)a.A[%d] + "
                     "((m0 + %d < a.M) ? (m0 + %d) : (a.M - 1)) * a.a_rs[%d];" % (T, r, T, d, r, r, d))
        if spec.dot_vec[d]:
            L.append("      const i64 nv = K / %d;" % V)
            if R == 1:
                L.append("      %s e0 = 0, e1 = 0;" % T)
                L.append("      i64 v = lane;")
                L.append("      for (; v + 64 < nv; v += 128) {")
                L.append("        const Pack<%s, %d> a0 = %s;" % (T, V, LD % ("(const Pack<%s, %d>*)(row0 + v * %d)" % (T, V, V))))
                L.append("        const Pack<%s, %d> a1 = %s;" % (T, V, LD % ("(const Pack<%s, %d>*)(row0 + (v + 64) * %d)" % (T, V, V))))
                L.append("        const Pack<%s, %d> x0 = *(const Pack<%s, %d>*)(xv + v * %d);" % (T, V, T, V, V))
                L.append("        const Pack<%s, %d> x1 = *(const Pack<%s, %d>*)(xv + (v + 64) * %d);" % (T, V, T, V, V))
                for e in range(V):
                    L.append("        e0 += a0.v[%d] * x0.v[%d]; e1 += a1.v[%d] * x1.v[%d];" % (e, e, e, e))
                L.append("      }")
                L.append("      for (; v < nv; v += 64) {")
                L.append("        const Pack<%s, %d> a0 = %s;" % (T, V, LD % ("(const Pack<%s, %d>*)(row0 + v * %d)" % (T, V, V))))
                L.append("        const Pack<%s, %d> x0 = *(const Pack<%s, %d>*)(xv + v * %d);" % (T, V, T, V, V))
                for e in range(V):
                    L.append("        e0 += a0.v[%d] * x0.v[%d];" % (e, e))
                L.append("      }")
                L.append("      d%d_0 = e0 + e1;" % d)
            else:
                L.append("      for (i64 v = lane; v < nv; v += 64) {")
                L.append("        const Pack<%s, %d> x0 = *(const Pack<%s, %d>*)(xv + v * %d);" % (T, V, T, V, V))
                for r in range(R):
                    L.append("        const Pack<%s, %d> a%d = %s;" % (T, V, r, LD % (
                        "(const Pack<%s, %d>*)(row%d + v * %d)" % (T, V, r, V))))
                for r in range(R):
                    for e in range(V):
                        L.append("        d%d_%d += a%d.v[%d] * x0.v[%d];" % (d, r, r, e, e))
                L.append("      }")
        else:
            L.append("      const i64 cs = a.a_cs[%d], ix = a.incx[%d];" % (d, d))
            L.append("      for (i64 k = lane; k < K; k += 64) {")
            L.append("        const %s xk = xv[k * ix];" % T)
            for r in range(R):
                L.append("        d%d_%d += row%d[k * cs] * xk;" % (d, r, r))
            L.append("      }")
        for r in range(R):
            L.append("      for (int s = 32; s > 0; s >>= 1) d%d_%d += shfl_xor_<%s>(d%d_%d, s);"
                     % (d, r, T, d, r))
        L.append("    }")
    for r in range(R):
        L.append("    if (m0 + %d < a.M) {" % r)
        L.append("      const i64 m = m0 + %d;" % r)
        ins = ["d%d_%d" % (d, r) for d in range(D)]
        in_dts = [spec.dtype] * D
        for k in range(nin):
            ct = CTYPE[spec.in_dtypes[k]]
            L.append("      const %s x%d = ((const %s*)a.ptr[%d])[m * a.stride[%d]];" % (ct, k, ct, k, k))
            ins.append("(x%d != 0)" % k if spec.in_dtypes[k] == "bool" else "x%d" % k)
            in_dts.append(spec.in_dtypes[k])
        lines, outs, odts = emit_scalar_body(spec.scalar, ins, in_dts, indent="      ")
        L.extend(lines)
        L.append("      if (lane == 0) {")
        for k, ri in enumerate(spec.out_refs):
            val = _store_val(outs[ri], odts[ri], spec.out_dtypes[k])
            L.append("        ((%s*)a.ptr[%d])[m * a.stride[%d]] = %s;" %
                     (CTYPE[spec.out_dtypes[k]], nin + k, nin + k, val))
        L.append("      }")
        L.append("    }")
    L.append("  }")
    L.append("}")
    return "\n".join(L) + "\n", (name,)


# ----------------------------------------------------------------------------------------
# single-pass "row program": y = X.w -> row-wise Elemwise / Sum -> X.T.r, X read ONCE
# ----------------------------------------------------------------------------------------
RP_MAXOPS = 16
RP_MAXRED = 8

RP_STRUCT = r"""
#define RP_MAXOPS %d
struct RpArgs {
  i64 N; i64 K; const void* X; i64 x_rs; const void* w;
  void* ptr[RP_MAXOPS]; i64 stride[RP_MAXOPS];
  void* col_ws; void* red_ws;
  int nops; int nred;
};
""" % RP_MAXOPS


class RowPassSpec:
    """One pass over a row-major matrix X (N x K) computing, per row m:
    ``d = X[m,:] . w``; a scalar program over ``d`` and row-wise operands; full ``Sum``
    reductions of some of its values (accumulator dtype per CAReduce._acc_dtype); materialised
    row-wise outputs; and the column accumulation ``g[k] += X[m,k] * r[m]`` (the ``X.T @ r`` of
    the gradient) while the row is still in registers.

    Replaces, for GLM-shaped graphs such as BASELINE config 5 (logistic logp + grad), the
    reference's two BLAS2 passes over X (tensor/blas.py:231 Gemv on X and on X.T) plus the
    Elemwise/Sum nodes between them (X is read twice, 2 x 16 GiB at N = 16M).

    kv        : 16-byte vectors of a row held per lane (K = 64 * VEC * kv)
    scalar    : merged scalar program; input 0 is the dot result, then the row operands
    reds      : [(scalar out index, acc dtype)] — Sum reductions
    col_ref   : scalar out index of r (the column-accumulation weight), or None
    """

    def __init__(self, dtype, kv, scalar, in_dtypes, out_dtypes, out_refs, reds, col_ref,
                 rpw=2, block=256, nt=False):
        self.nt = bool(nt)      # rows of X read with non-temporal loads (exec_elemwise.BIG_STREAM)
        self.dtype, self.kv, self.scalar = dtype, kv, scalar
        self.in_dtypes, self.out_dtypes, self.out_refs = list(in_dtypes), list(out_dtypes), list(out_refs)
        self.reds, self.col_ref, self.rpw, self.block = [list(r) for r in reds], col_ref, rpw, block
        assert len(self.in_dtypes) + len(self.out_dtypes) <= RP_MAXOPS and len(self.reds) <= RP_MAXRED

    def key(self):
        fields = ["rp2", self.dtype, self.kv, self.in_dtypes, self.out_dtypes, self.out_refs,
                  self.reds, self.col_ref, self.rpw, self.block] + (["nt"] if self.nt else [])
        return _memo_key([self.scalar], fields, self._key)

    def _key(self):
        import json
        blob = json.dumps(["rp2", self.dtype, self.kv, self.scalar, self.in_dtypes, self.out_dtypes,
                           self.out_refs, self.reds, self.col_ref, self.rpw, self.block] +
                          (["nt"] if self.nt else []), sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def generate_rowpass(spec: RowPassSpec):
    """Wave-level schedule: a wavefront takes R = rpw consecutive rows per iteration (R a power
    of two <= 16), keeps them in registers, forms R per-lane partial dots, and REDUCE-SCATTERS
    them across the wave (log2 R halving steps + an all-reduce over the remaining lane bits) so
    that lane l ends up with the full dot of row (l >> (6 - log2 R)).  Each lane then runs the
    scalar program for ITS row only (not 64 redundant copies), leaders store/accumulate, and the
    R column-accumulation weights are broadcast back with v_readlane."""
    T = RTYPE[spec.dtype]
    V = 4 if spec.dtype == "float32" else 2
    KV, R = spec.kv, spec.rpw
    P = R.bit_length() - 1
    assert 1 << P == R and P <= 4
    SH = 6 - P                     # lanes per row group = 1 << SH
    nin, nout = len(spec.in_dtypes), len(spec.out_dtypes)
    waves = spec.block // 64
    name = "rp_" + spec.key()
    L = [PRELUDE, RP_STRUCT]
    L.append('extern "C" __global__ __launch_bounds__(%d) void %s(RpArgs a) {' % (spec.block, name))
    L.append("  const int lane = threadIdx.x & 63;")
    L.append("  const int wave = threadIdx.x >> 6;")
    L.append("  const i64 nwaves = (i64)gridDim.x * %d;" % waves)
    L.append("  const bool leader = (lane & %d) == 0;" % ((1 << SH) - 1))
    L.append("  const %s* __restrict__ X = (const %s*)a.X;" % (T, T))
    for v in range(KV):
        L.append("  const Pack<%s, %d> w%d = *(const Pack<%s, %d>*)((const %s*)a.w + (%d * 64 + lane) * %d);"
                 % (T, V, v, T, V, T, v, V))
    if spec.col_ref is not None:
        for v in range(KV):
            for e in range(V):
                L.append("  %s g%d_%d = 0;" % (T, v, e))
    for j, (_, acc) in enumerate(spec.reds):
        L.append("  %s racc%d = 0;" % (RTYPE[acc], j))
    L.append("  for (i64 m0 = ((i64)blockIdx.x * %d + wave) * %d; m0 < a.N; m0 += nwaves * %d) {"
             % (waves, R, R))
    for r in range(R):
        L.append("    const %s* __restrict__ row%d = X + ((m0 + %d < a.N) ? (m0 + %d) : (a.N - 1)) * a.x_rs;"
                 % (T, r, r, r))
        for v in range(KV):
            L.append("    const Pack<%s, %d> x%d_%d = %s((const Pack<%s, %d>*)(row%d + (%d * 64 + lane) * %d));"
                     % (T, V, r, v, "nt_load" if spec.nt else "*", T, V, r, v, V))
    for r in range(R):
        terms = " + ".join("x%d_%d.v[%d] * w%d.v[%d]" % (r, v, e, v, e)
                           for v in range(KV) for e in range(V))
        L.append("    %s p%d = %s;" % (T, r, terms))
    # reduce-scatter over the top P lane bits
    cur = ["p%d" % r for r in range(R)]
    for step in range(P):
        mask = 32 >> step
        half = len(cur) // 2
        L.append("    const bool up%d = (lane & %d) != 0;" % (step, mask))
        nxt = []
        for i in range(half):
            nm = "q%d_%d" % (step, i)
            L.append("    const %s %s = (up%d ? %s : %s) + shfl_xor_<%s>(up%d ? %s : %s, %d);"
                     % (T, nm, step, cur[i + half], cur[i], T, step, cur[i], cur[i + half], mask))
            nxt.append(nm)
        cur = nxt
    L.append("    %s d = %s;" % (T, cur[0]))
    for step in range(P, 6):
        mask = 32 >> step
        L.append("    d += shfl_xor_<%s>(d, %d);" % (T, mask))
    # lane-local epilogue for row (lane >> SH)
    L.append("    const i64 m = m0 + (lane >> %d);" % SH)
    L.append("    const bool valid = m < a.N;")
    L.append("    const i64 mc = valid ? m : (a.N - 1);")
    ins, in_dts = ["d"], [spec.dtype]
    for k in range(nin):
        ct = CTYPE[spec.in_dtypes[k]]
        L.append("    const %s o%d = ((const %s*)a.ptr[%d])[mc * a.stride[%d]];" % (ct, k, ct, k, k))
        ins.append("(o%d != 0)" % k if spec.in_dtypes[k] == "bool" else "o%d" % k)
        in_dts.append(spec.in_dtypes[k])
    lines, outs, odts = emit_scalar_body(spec.scalar, ins, in_dts, indent="    ")
    L.extend(lines)
    if nout:
        L.append("    if (leader && valid) {")
        for k, ri in enumerate(spec.out_refs):
            L.append("      ((%s*)a.ptr[%d])[m * a.stride[%d]] = %s;" %
                     (CTYPE[spec.out_dtypes[k]], nin + k, nin + k,
                      _store_val(outs[ri], odts[ri], spec.out_dtypes[k])))
        L.append("    }")
    for j, (ri, acc) in enumerate(spec.reds):
        L.append("    if (leader && valid) racc%d += %s;" % (j, _cast(outs[ri], odts[ri], acc)))
    if spec.col_ref is not None:
        L.append("    const %s rmine = %s;" % (T, _cast(outs[spec.col_ref], odts[spec.col_ref], spec.dtype)))
        for r in range(R):
            src = r << SH
            if spec.dtype == "float32":
                L.append("    const float rr%d = __builtin_bit_cast(float, __builtin_amdgcn_readlane("
                         "__builtin_bit_cast(int, rmine), %d));" % (r, src))
            else:
                L.append("    double rr%d; { union { double dd; int ii[2]; } u; u.dd = rmine; "
                         "u.ii[0] = __builtin_amdgcn_readlane(u.ii[0], %d); "
                         "u.ii[1] = __builtin_amdgcn_readlane(u.ii[1], %d); rr%d = u.dd; }"
                         % (r, src, src, r))
            L.append("    if (m0 + %d < a.N) {" % r)
            for v in range(KV):
                for e in range(V):
                    L.append("      g%d_%d += x%d_%d.v[%d] * rr%d;" % (v, e, r, v, e, r))
            L.append("    }")
    L.append("  }")
    # ---- fold the waves of the workgroup (fixed order), one partial per workgroup ----
    L.append("  extern __shared__ __attribute__((aligned(16))) char smem[];")
    if spec.col_ref is not None:
        L.append("  %s* sg = (%s*)smem;" % (T, T))
        for v in range(KV):
            for e in range(V):
                L.append("  sg[(i64)wave * a.K + (%d * 64 + lane) * %d + %d] = g%d_%d;" % (v, V, e, v, e))
    L.append("  double* sr = (double*)(smem + %d * a.K * sizeof(%s));" % (waves, T))
    for j, (_, acc) in enumerate(spec.reds):
        L.append("  { double t = (double)racc%d; for (int s = 32; s > 0; s >>= 1) "
                 "t += shfl_xor_<double>(t, s); if (lane == 0) sr[wave * %d + %d] = t; }"
                 % (j, RP_MAXRED, j))
    L.append("  __syncthreads();")
    if spec.col_ref is not None:
        L.append("  for (i64 k = threadIdx.x; k < a.K; k += %d) {" % spec.block)
        L.append("    %s s = sg[k];" % T)
        L.append("    for (int wv = 1; wv < %d; ++wv) s += sg[(i64)wv * a.K + k];" % waves)
        L.append("    ((%s*)a.col_ws)[(i64)blockIdx.x * a.K + k] = s;" % T)
        L.append("  }")
    if spec.reds:
        L.append("  if (threadIdx.x < %d) {" % len(spec.reds))
        L.append("    double s = sr[threadIdx.x];")
        L.append("    for (int wv = 1; wv < %d; ++wv) s += sr[wv * %d + threadIdx.x];" % (waves, RP_MAXRED))
        L.append("    ((double*)a.red_ws)[(i64)blockIdx.x * %d + threadIdx.x] = s;" % len(spec.reds))
        L.append("  }")
    L.append("}")
    return "\n".join(L) + "\n", (name,)


# ----------------------------------------------------------------------------------------
# row-chain kernels: a chain of last-axis reductions and the Elemwise steps between them
# (softmax, log-softmax, softmax gradient, mean/variance normalisation ...) in ONE pass
# ----------------------------------------------------------------------------------------
RC_MAXOPS = 16

RC_MAXLEAD = 4

RC_STRUCT = r"""
#define RC_MAXOPS %d
#define RC_MAXLEAD %d
struct RcArgs { i64 N; i64 K; i64 lshape[RC_MAXLEAD]; void* ptr[RC_MAXOPS]; i64 ls[RC_MAXOPS][RC_MAXLEAD]; };
""" % (RC_MAXOPS, RC_MAXLEAD)


class RowChainSpec:
    """Rows of an [N, K] space (K = the last, contiguous axis) processed by sub-wave groups of
    ``L`` lanes (L a power of two <= 64, 64 / L rows per wavefront); a lane keeps ``nch`` packs
    of ``V`` consecutive elements of every full operand in registers, so each operand is read
    from HBM exactly once and every intermediate between the reductions stays in registers.

    Replaces the separate passes the reference makes for such chains — e.g. Softmax.c_code
    (tensor/special.py:372-415: max pass, exp+sum pass, scale pass over the output) or the
    CAReduce / DimShuffle / Elemwise node sequence a hand-written normalisation lowers to
    (tensor/elemwise.py:1495, :222, :725).

    ext      : [(dtype, cls)] external operands; cls "f" full [N, K], "r" per-row [N, 1],
               "c" per-column [1, K], "s" scalar
    members  : the chain in execution order; each {"scalar", "ins", "reduce", "stores"} with
               ins[i] = ["e", k] | ["f", member, scalar-out index] | ["r", member];
               reduce = None | {"op", "acc", "out", "ref", "slot"}; stores = [[out index, dtype,
               slot]] (slots index RcArgs.ptr after the external operands)
    """

    def __init__(self, ext, members, L, V, nch, lnd=1, block=256, nt=False):
        self.ext = [list(e) for e in ext]
        self.members, self.L, self.V, self.nch, self.block = members, L, V, nch, block
        self.lnd = lnd          # jointly-collapsed leading dims (row index -> coordinates)
        self.nt = bool(nt)      # streaming (non-temporal) loads of the full operands: read-once rows
        assert L in (1, 2, 4, 8, 16, 32, 64) and block % 64 == 0 and 1 <= lnd <= RC_MAXLEAD

    def key(self):
        fields = ["rc2" + ("n" if self.nt else ""), self.ext, self.L, self.V, self.nch, self.lnd, self.block,
                  [[m["ins"], m.get("reduce"), m.get("stores"), m.get("rowlike")]
                   for m in self.members]]
        return _memo_key([m["scalar"] for m in self.members], fields, self._key)

    def _key(self):
        import json
        blob = json.dumps(["rc2" + ("n" if self.nt else ""), self.ext, self.members, self.L, self.V, self.nch,
                           self.lnd, self.block],
                          sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def generate_rowchain(spec: RowChainSpec):
    L_, V, NCH = spec.L, spec.V, spec.nch
    rpw = 64 // L_
    waves = spec.block // 64
    name = "rc_" + spec.key()
    S = [PRELUDE, RC_STRUCT]
    S.append('extern "C" __global__ __launch_bounds__(%d) void %s(RcArgs a) {' % (spec.block, name))
    S.append("  const int lane = threadIdx.x & 63;")
    S.append("  const int sub = lane & %d;" % (L_ - 1))
    S.append("  const int grp = lane >> %d;" % (L_.bit_length() - 1))
    S.append("  const i64 nwaves = (i64)gridDim.x * %d;" % waves)
    for c in range(NCH):
        S.append("  const i64 col%d = ((i64)%d + sub) * %d;" % (c, c * L_, V))
        S.append("  const bool ok%d = col%d < a.K;" % (c, c))
    for k, (dt, cls) in enumerate(spec.ext):
        ct = CTYPE[dt]
        if cls == "s":
            S.append("  const %s sc%d = *(const %s*)a.ptr[%d];" % (ct, k, ct, k))
        elif cls == "c":
            for c in range(NCH):
                S.append("  Pack<%s, %d> co%d_%d = {}; if (ok%d) co%d_%d = *(const Pack<%s, %d>*)"
                         "((const %s*)a.ptr[%d] + col%d);" % (ct, V, k, c, c, k, c, ct, V, ct, k, c))
    S.append("  for (i64 rb = ((i64)blockIdx.x * %d + (threadIdx.x >> 6)) * %d; rb < a.N; "
             "rb += nwaves * %d) {" % (waves, rpw, rpw))
    S.append("    const i64 row = rb + grp;")
    S.append("    const bool rv = row < a.N;")
    S.append("    const i64 rr = rv ? row : a.N - 1;")
    # row index -> coordinates over the collapsed leading dims (one div/mod per extra dim)
    rem = "rr"
    for d in range(spec.lnd - 1, 0, -1):
        S.append("    const i64 q%d = %s / a.lshape[%d];" % (d, rem, d))
        S.append("    const i64 lc%d = %s - q%d * a.lshape[%d];" % (d, rem, d, d))
        rem = "q%d" % d
    S.append("    const i64 lc0 = %s;" % rem)

    def row_off(k):
        return " + ".join("lc%d * a.ls[%d][%d]" % (d, k, d) for d in range(spec.lnd))
    for k, (dt, cls) in enumerate(spec.ext):
        ct = CTYPE[dt]
        if cls == "f":
            S.append("    const %s* __restrict__ xp%d = (const %s*)a.ptr[%d] + %s;"
                     % (ct, k, ct, k, row_off(k)))
            for c in range(NCH):
                S.append("    Pack<%s, %d> x%d_%d = {}; if (ok%d) x%d_%d = %s((const Pack<%s, %d>*)"
                         "(xp%d + col%d));" % (ct, V, k, c, c, k, c, "nt_load" if spec.nt else "*", ct, V, k, c))
        elif cls == "r":
            S.append("    const %s ro%d = ((const %s*)a.ptr[%d])[%s];" % (ct, k, ct, k, row_off(k)))

    def ext_expr(k, c, j):
        dt, cls = spec.ext[k]
        e = {"f": "x%d_%d.v[%d]" % (k, c, j), "r": "ro%d" % k, "c": "co%d_%d.v[%d]" % (k, c, j),
             "s": "sc%d" % k}[cls]
        return ("(%s != 0)" % e if dt == "bool" else e), dt

    outs = []      # per member: {(c, j): ([expr], [dtype])}
    rdt = {}       # member -> dtype of its row result
    for mi, m in enumerate(spec.members):
        red = m.get("reduce")
        if red:
            S.append("    %s acc%d = %s;" % (RTYPE[red["acc"]], mi, red_identity(red["op"], red["acc"])))
        mouts = {}

        def inputs_at(c, j):
            in_exprs, in_dts = [], []
            for r in m["ins"]:
                if r[0] == "e":
                    e, d = ext_expr(r[1], c, j)
                elif r[0] == "f":
                    es, ds = outs[r[1]][(c, j)]
                    e, d = es[r[2]], ds[r[2]]
                else:
                    e, d = "r%d" % r[1], rdt[r[1]]
                in_exprs.append(e)
                in_dts.append(d)
            return in_exprs, in_dts

        if m.get("rowlike"):
            # every input is per-row or scalar: one evaluation per row ([..., 1]-shaped values)
            in_exprs, in_dts = inputs_at(0, 0)
            lines, oe, od = emit_scalar_body(m["scalar"], in_exprs, in_dts, indent="    ",
                                             suffix="_m%d_r" % mi)
            S.extend(lines)
            for c in range(NCH):
                for j in range(V):
                    mouts[(c, j)] = (oe, od)
            outs.append(mouts)
            for oref, odt, slot in m.get("stores", []):
                S.append("    if (rv && sub == 0) ((%s*)a.ptr[%d])[%s] = %s;"
                         % (CTYPE[odt], slot, row_off(slot), _store_val(oe[oref], od[oref], odt)))
            continue
        for c in range(NCH):
            for j in range(V):
                in_exprs, in_dts = inputs_at(c, j)
                lines, oe, od = emit_scalar_body(m["scalar"], in_exprs, in_dts, indent="    ",
                                                 suffix="_m%d_%d_%d" % (mi, c, j))
                S.extend(lines)
                mouts[(c, j)] = (oe, od)
                if red:
                    v = _cast(oe[red["ref"]], od[red["ref"]], red["acc"])
                    S.append("    if (ok%d) acc%d = %s;" % (c, mi, red_combine(red["op"], red["acc"],
                                                                               "acc%d" % mi, v)))
        outs.append(mouts)
        if red:
            at_ = RTYPE[red["acc"]]
            msk = L_ // 2
            while msk >= 1:
                S.append("    acc%d = %s;" % (mi, red_combine(
                    red["op"], red["acc"], "acc%d" % mi, "shfl_xor_<%s>(acc%d, %d)" % (at_, mi, msk))))
                msk //= 2
            S.append("    const %s r%d = %s;" % (RTYPE[red["out"]], mi,
                                                  _cast("acc%d" % mi, red["acc"], red["out"])))
            rdt[mi] = red["out"]
            if red.get("slot") is not None:
                S.append("    if (rv && sub == 0) ((%s*)a.ptr[%d])[%s] = %s;"
                         % (CTYPE[red["out"]], red["slot"], row_off(red["slot"]),
                            _store_val("r%d" % mi, red["out"], red["out"])))
        for oref, odt, slot in m.get("stores", []):
            ct = CTYPE[odt]
            for c in range(NCH):
                S.append("    if (rv && ok%d) {" % c)
                S.append("      Pack<%s, %d> y;" % (ct, V))
                for j in range(V):
                    oe, od = mouts[(c, j)]
                    S.append("      y.v[%d] = %s;" % (j, _store_val(oe[oref], od[oref], odt)))
                S.append("      *(Pack<%s, %d>*)((%s*)a.ptr[%d] + %s + col%d) = y;"
                         % (ct, V, ct, slot, row_off(slot), c))
                S.append("    }")
    S.append("  }")
    S.append("}")
    return "\n".join(S) + "\n", (name,)


# ----------------------------------------------------------------------------------------
# small-M GEMM chain + Elemwise epilogue: f(A_0 @ B_0, A_1 @ B_1, ..., operands) in ONE kernel
# ----------------------------------------------------------------------------------------
GE_MAXDOTS = 3
GE_MAXOPS = 12

GE_PRELUDE = r"""
#define GE_MAXDOTS %d
#define GE_MAXOPS %d
struct GeArgs {
  i64 M; i64 N; i64 K[GE_MAXDOTS];
  const void* A[GE_MAXDOTS]; i64 a_rs[GE_MAXDOTS];
  const void* B[GE_MAXDOTS]; i64 b_rs[GE_MAXDOTS]; i64 b_cs[GE_MAXDOTS];
  void* ptr[GE_MAXOPS]; i64 rs[GE_MAXOPS]; i64 cs[GE_MAXOPS];
};
template <typename T> struct MfmaT;
template <> struct MfmaT<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void mma(acc_t& c, float a, float b) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int frag_row(int lane, int i) { return (lane >> 4) * 4 + i; }
};
template <> struct MfmaT<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static constexpr int VEC = 2;
  static __device__ __forceinline__ void mma(acc_t& c, double a, double b) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int frag_row(int lane, int i) { return (lane >> 4) + 4 * i; }
};

// One wavefront's share (a quarter of K) of a 16 x (16*NF) tile of A @ B; same schedule as
// gemm_skinny_kernel in csrc/gemm.hip (A k-contiguous, B n- or k-contiguous, vector loads in
// MFMA layout, two-group software pipeline, accumulators folded every 512 k).
template <typename T, int NF, bool BKC, int NW>
__device__ __forceinline__ void skinny_dot(const T* __restrict__ A, i64 a_rs, const T* __restrict__ B,
                                           i64 b_rs, i64 b_cs, i64 M, i64 N, i64 K, i64 m0, i64 n0,
                                           int lane, int wave, typename MfmaT<T>::acc_t (&res)[NF]) {
  typedef typename MfmaT<T>::acc_t acc_t;
  constexpr int VEC = MfmaT<T>::VEC;
  constexpr int G = 4 * VEC;
  struct alignas(sizeof(T) * VEC) KV { T v[VEC]; };
  struct alignas(sizeof(T) * NF) NV { T v[NF]; };
  struct Frag { KV a; KV bk[NF]; NV bn[VEC]; };
  const int r = lane & 15, kg = lane >> 4;
  const i64 kq = ((K + NW * G - 1) / (NW * G)) * G;   // K split over the NW wavefronts
  const i64 kbeg = wave * kq;
  const i64 kend = (kbeg + kq < K) ? kbeg + kq : K;
  const bool mok = m0 + r < M;
  const T* ap = A + (mok ? m0 + r : 0) * a_rs + VEC * kg;
  acc_t acc[NF], tot[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) { acc[f] = acc_t{0, 0, 0, 0}; tot[f] = acc_t{0, 0, 0, 0}; }
  auto load = [&](i64 k0, Frag& fr) {
    const bool kok = k0 + VEC * kg < kend;
    fr.a = (mok && kok) ? *reinterpret_cast<const KV*>(ap + k0) : KV{};
    if constexpr (BKC) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const i64 nn = n0 + 16 * f + r;
        fr.bk[f] = (nn < N && kok) ? *reinterpret_cast<const KV*>(B + nn * b_cs + k0 + VEC * kg) : KV{};
      }
    } else {
      const i64 nn = n0 + NF * r;
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        fr.bn[j] = (nn < N && kok)
            ? *reinterpret_cast<const NV*>(B + (k0 + VEC * kg + j) * b_rs + nn) : NV{};
    }
  };
  auto compute = [&](const Frag& fr) {
#pragma unroll
    for (int j = 0; j < VEC; ++j)
#pragma unroll
      for (int f = 0; f < NF; ++f)
        MfmaT<T>::mma(acc[f], fr.a.v[j], BKC ? fr.bk[f].v[j] : fr.bn[j].v[f]);
  };
  constexpr int DEPTH = 64 / G;
  if (kend - kbeg <= DEPTH * G) {
    // short K slice (a recurrent step: K / NW = 64): every load is issued before the first MFMA,
    // one memory round trip instead of one per pipeline stage (these kernels are latency-bound)
    Frag fr[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) load(kbeg + u * G, fr[u]);   // beyond kend: zeros, no access
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) compute(fr[u]);
#pragma unroll
    for (int f = 0; f < NF; ++f) res[f] = acc[f];
    return;
  }
  Frag f0, f1;
  i64 k0 = kbeg, next_fold = kbeg + 512;
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
      next_fold += 512;
    }
  }
#pragma unroll
  for (int f = 0; f < NF; ++f) res[f] = tot[f] + acc[f];
}
""" % (GE_MAXDOTS, GE_MAXOPS)


class GemmEpiSpec:
    """out[m, n] = f(dot_0[m, n], ..., dot_{D-1}[m, n], operands[m, n]) with dot_d = A_d @ B_d,
    for outputs too small to fill the chip with 128x128 tiles (a recurrent step with a batch of
    states, a small-batch MLP layer): the 16-row split-K MFMA schedule of csrc/gemm.hip with the
    Elemwise consumer evaluated on the accumulators.

    Replaces ``Gemm`` / ``Dot22`` nodes (tensor/blas.py:872 / :1659) followed by the ``Elemwise``
    that consumes them — one GRU gate ``sigmoid(h @ U + V_t) * h`` = 1 launch instead of 2-3.

    dtype : float32 | float64;  nf : column fragments per workgroup (1, 2, 4 <= VEC)
    bkc   : per dot, True when B is k-contiguous (x @ W.T), False when n-contiguous (x @ W)
    scalar: plan scalar expression; its first D inputs are the dot results
    """

    def __init__(self, dtype, nf, bkc, scalar, in_dtypes, out_dtypes, out_refs, waves=4):
        self.waves = waves   # wavefronts per workgroup = K slices (short K chains: these kernels
        #                      are bound by memory round trips per wavefront, not by MFMA rate)
        assert waves in (4, 8, 16)
        self.dtype, self.nf, self.bkc, self.scalar = dtype, nf, [bool(b) for b in bkc], scalar
        self.in_dtypes, self.out_dtypes, self.out_refs = list(in_dtypes), list(out_dtypes), list(out_refs)
        assert 1 <= len(self.bkc) <= GE_MAXDOTS
        assert all(self.bkc) or not any(self.bkc) or nf == 1   # one column map per kernel
        assert len(self.in_dtypes) + len(self.out_dtypes) <= GE_MAXOPS

    def key(self):
        fields = ["ge3", self.dtype, self.nf, self.bkc, self.in_dtypes, self.out_dtypes,
                  self.out_refs, self.waves]
        return _memo_key([self.scalar], fields, self._key)

    def _key(self):
        import json
        blob = json.dumps(["ge3", self.dtype, self.nf, self.bkc, self.scalar, self.in_dtypes,
                           self.out_dtypes, self.out_refs, self.waves], sort_keys=True)
        return hashlib.sha256(blob.encode()).hexdigest()[:24]


def generate_gemm_epilogue(spec: GemmEpiSpec):
    T = RTYPE[spec.dtype]
    NF, D = spec.nf, len(spec.bkc)
    nin, nout = len(spec.in_dtypes), len(spec.out_dtypes)
    name = "ge_" + spec.key()
    S = [PRELUDE, GE_PRELUDE]
    NW = spec.waves
    S.append('extern "C" __global__ __launch_bounds__(%d) void %s(GeArgs a) {' % (64 * NW, name))
    S.append("  typedef MfmaT<%s>::acc_t acc_t;" % T)
    S.append("  __shared__ %s part[%d][%d][%d];" % (T, NW, D, 256 * NF))
    S.append("  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15;")
    S.append("  const i64 m0 = (i64)blockIdx.y * 16, n0 = (i64)blockIdx.x * %d;" % (16 * NF))
    for d in range(D):
        S.append("  {")
        S.append("    acc_t res[%d];" % NF)
        S.append("    skinny_dot<%s, %d, %s, %d>((const %s*)a.A[%d], a.a_rs[%d], (const %s*)a.B[%d], "
                 "a.b_rs[%d], a.b_cs[%d], a.M, a.N, a.K[%d], m0, n0, lane, wave, res);"
                 % (T, NF, "true" if spec.bkc[d] else "false", NW, T, d, d, T, d, d, d, d))
        S.append("    for (int f = 0; f < %d; ++f)" % NF)
        S.append("      for (int i = 0; i < 4; ++i)")
        S.append("        part[wave][%d][(MfmaT<%s>::frag_row(lane, i) * 16 + r) * %d + f] = res[f][i];"
                 % (d, T, NF))
        S.append("  }")
    S.append("  __syncthreads();")
    S.append("  if (threadIdx.x >= 256) return;")
    S.append("  const int e = threadIdx.x, er = e >> 4, ec = e & 15;")
    S.append("  const i64 m = m0 + er;")
    S.append("  if (m >= a.M) return;")
    S.append("  for (int f = 0; f < %d; ++f) {" % NF)
    # column of fragment f: BKC dots use n0 + 16 f + ec, the others n0 + NF ec + f.  All dots of
    # one kernel must agree, which the executor guarantees by choosing NF = 1 for mixed layouts.
    if all(spec.bkc):
        S.append("    const i64 n = n0 + 16 * f + ec;")
    else:
        S.append("    const i64 n = n0 + %d * ec + f;" % NF)
    S.append("    if (n >= a.N) continue;")
    ins, in_dts = [], []
    for d in range(D):
        S.append("    %s d%d = part[0][%d][e * %d + f];" % (T, d, d, NF))
        S.append("    for (int w_ = 1; w_ < %d; ++w_) d%d += part[w_][%d][e * %d + f];" % (NW, d, d, NF))
        ins.append("d%d" % d)
        in_dts.append(spec.dtype)
    for k in range(nin):
        ct = CTYPE[spec.in_dtypes[k]]
        S.append("    const %s x%d = ((const %s*)a.ptr[%d])[m * a.rs[%d] + n * a.cs[%d]];"
                 % (ct, k, ct, k, k, k))
        ins.append("(x%d != 0)" % k if spec.in_dtypes[k] == "bool" else "x%d" % k)
        in_dts.append(spec.in_dtypes[k])
    lines, outs, odts = emit_scalar_body(spec.scalar, ins, in_dts, indent="    ")
    S.extend(lines)
    for k, ri in enumerate(spec.out_refs):
        S.append("    ((%s*)a.ptr[%d])[m * a.rs[%d] + n * a.cs[%d]] = %s;"
                 % (CTYPE[spec.out_dtypes[k]], nin + k, nin + k, nin + k,
                    _store_val(outs[ri], odts[ri], spec.out_dtypes[k])))
    S.append("  }")
    S.append("}")
    return "\n".join(S) + "\n", (name,)


def generate_rowchain_long(spec: RowChainSpec):
    """Row chains whose rows do not fit a wavefront's registers (K of tens of thousands: a
    vocabulary-sized softmax): ONE WORKGROUP PER ROW.  Every reduction of the chain is a stage
    that sweeps the row in 16-byte packs (block reduce through LDS, result broadcast to all
    threads); Elemwise members between the reductions are RE-EVALUATED in each later stage that
    needs them instead of being kept (exp(x - max) is computed in the sum stage and again in the
    scale stage).  The first sweep streams the row from HBM, the later sweeps re-read it from the
    L2 / memory-side cache (a row is a few hundred KB), so HBM traffic stays ~1 read + the stores.
    Same spec / argument block as generate_rowchain (spec.L and spec.nch are ignored)."""
    V = spec.V
    T_BLOCK = spec.block
    nw = T_BLOCK // 64
    members = spec.members
    name = "rcl_" + spec.key()
    S = [PRELUDE, RC_STRUCT]
    S.append('extern "C" __global__ __launch_bounds__(%d) void %s(RcArgs a) {' % (T_BLOCK, name))
    S.append("  __shared__ double red_sm[%d];" % nw)
    S.append("  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;")
    for k, (dt, cls) in enumerate(spec.ext):
        if cls == "s":
            S.append("  const %s sc%d = *(const %s*)a.ptr[%d];" % (CTYPE[dt], k, CTYPE[dt], k))
    S.append("  for (i64 row = blockIdx.x; row < a.N; row += gridDim.x) {")
    rem = "row"
    for d in range(spec.lnd - 1, 0, -1):
        S.append("    const i64 q%d = %s / a.lshape[%d];" % (d, rem, d))
        S.append("    const i64 lc%d = %s - q%d * a.lshape[%d];" % (d, rem, d, d))
        rem = "q%d" % d
    S.append("    const i64 lc0 = %s;" % rem)

    def row_off(k):
        return " + ".join("lc%d * a.ls[%d][%d]" % (d, k, d) for d in range(spec.lnd))

    for k, (dt, cls) in enumerate(spec.ext):
        ct = CTYPE[dt]
        if cls == "f":
            S.append("    const %s* __restrict__ xp%d = (const %s*)a.ptr[%d] + %s;" % (ct, k, ct, k, row_off(k)))
        elif cls == "c":
            S.append("    const %s* __restrict__ xp%d = (const %s*)a.ptr[%d];" % (ct, k, ct, k))
        elif cls == "r":
            S.append("    const %s ro%d = ((const %s*)a.ptr[%d])[%s];" % (ct, k, ct, k, row_off(k)))

    rowlike = [bool(m.get("rowlike")) for m in members]
    rdt = {}

    def closure(targets):
        """per-element members needed (in order) to evaluate `targets`"""
        need = set()

        def visit(mi):
            if mi in need or rowlike[mi]:
                return
            need.add(mi)
            for r in members[mi]["ins"]:
                if r[0] == "f":
                    visit(r[1])
        for t in targets:
            visit(t)
        return sorted(need)

    row_vals = {}     # rowlike member -> (out exprs, out dtypes), evaluated once per row

    def emit_rowlike_ready(upto_reduce_done):
        """evaluate row-like members whose inputs are all available now"""
        for mi, m in enumerate(members):
            if not rowlike[mi] or mi in row_vals:
                continue
            ok = all((r[0] == "e") or (r[0] == "r" and r[1] in rdt)
                     or (r[0] == "f" and r[1] in row_vals) for r in m["ins"])
            if not ok:
                continue
            ins, dts = [], []
            for r in m["ins"]:
                if r[0] == "e":
                    dt, cls = spec.ext[r[1]]
                    e = "ro%d" % r[1] if cls == "r" else "sc%d" % r[1]
                    ins.append("(%s != 0)" % e if dt == "bool" else e)
                    dts.append(dt)
                elif r[0] == "r":
                    ins.append("r%d" % r[1])
                    dts.append(rdt[r[1]])
                else:
                    es, ds = row_vals[r[1]]
                    ins.append(es[r[2]])
                    dts.append(ds[r[2]])
            lines, oe, od = emit_scalar_body(m["scalar"], ins, dts, indent="    ", suffix="_m%d_r" % mi)
            S.extend(lines)
            row_vals[mi] = (oe, od)
            for oref, odt, slot in m.get("stores", []):
                S.append("    if (threadIdx.x == 0) ((%s*)a.ptr[%d])[%s] = %s;"
                         % (CTYPE[odt], slot, row_off(slot), _store_val(oe[oref], od[oref], odt)))

    def emit_sweep(stage_id, needed, body_tail):
        """one pass over the row: loads, per-element evaluation of `needed`, then body_tail(outs)"""
        used_ext = sorted({r[1] for mi in needed for r in members[mi]["ins"] if r[0] == "e"
                           and spec.ext[r[1]][1] in "fc"})
        S.append("    for (i64 c0 = (i64)threadIdx.x * %d; c0 < a.K; c0 += %d) {" % (V, T_BLOCK * V))
        for k in used_ext:
            ct = CTYPE[spec.ext[k][0]]
            S.append("      const Pack<%s, %d> x%d = *(const Pack<%s, %d>*)(xp%d + c0);" % (ct, V, k, ct, V, k))
        for j in range(V):
            outs = {}
            for mi in needed:
                m = members[mi]
                ins, dts = [], []
                for r in m["ins"]:
                    if r[0] == "e":
                        dt, cls = spec.ext[r[1]]
                        e = {"f": "x%d.v[%d]" % (r[1], j), "c": "x%d.v[%d]" % (r[1], j),
                             "r": "ro%d" % r[1], "s": "sc%d" % r[1]}[cls]
                        ins.append("(%s != 0)" % e if dt == "bool" else e)
                        dts.append(dt)
                    elif r[0] == "r":
                        ins.append("r%d" % r[1])
                        dts.append(rdt[r[1]])
                    elif rowlike[r[1]]:
                        es, ds = row_vals[r[1]]
                        ins.append(es[r[2]])
                        dts.append(ds[r[2]])
                    else:
                        es, ds = outs[r[1]]
                        ins.append(es[r[2]])
                        dts.append(ds[r[2]])
                lines, oe, od = emit_scalar_body(m["scalar"], ins, dts, indent="      ",
                                                 suffix="_s%d_m%d_%d" % (stage_id, mi, j))
                S.extend(lines)
                outs[mi] = (oe, od)
            body_tail(j, outs)
        S.append("    }")

    def online_pair(mi):
        """max over a full operand immediately followed by sum(exp(x - max)) over the same
        operand (the head of every softmax / log-softmax): both come out of ONE sweep with the
        running-maximum rescaling  s <- s * exp(m_old - m_new) + sum exp(x - m_new)."""
        a_ = members[mi]
        if not (a_.get("reduce") and a_["reduce"]["op"] == "maximum" and not a_["scalar"]["nodes"]
                and a_["ins"] and a_["ins"][0][0] == "e" and spec.ext[a_["ins"][0][1]][1] == "f"
                and a_["scalar"]["out"][a_["reduce"]["ref"]] == ["i", 0]
                and spec.ext[a_["ins"][0][1]][0] in ("float32", "float64")):
            return None
        for bj in range(mi + 1, len(members)):
            b_ = members[bj]
            if rowlike[bj]:
                continue
            if not b_.get("reduce"):
                return None
            nodes = b_["scalar"]["nodes"]
            if (b_["reduce"]["op"] == "add" and len(nodes) == 2 and nodes[0]["op"] == "sub"
                    and nodes[1]["op"] == "exp" and nodes[1]["in"] == [["t", 0]]
                    and b_["scalar"]["out"][b_["reduce"]["ref"]] == ["t", 1]
                    and len(b_["ins"]) == 2 and b_["ins"][nodes[0]["in"][0][1]] == a_["ins"][0]
                    and b_["ins"][nodes[0]["in"][1][1]] == ["r", mi]
                    and nodes[0]["in"][0][0] == "i" and nodes[0]["in"][1][0] == "i"
                    and nodes[0]["dtype"] == nodes[1]["dtype"] == spec.ext[a_["ins"][0][1]][0]
                    and b_["reduce"]["acc"] in ("float32", "float64")):
                return bj
            return None
        return None

    stage = 0
    fused_done = set()
    for mi, m in enumerate(members):
        red = m.get("reduce")
        if not red or rowlike[mi] or mi in fused_done:
            continue
        emit_rowlike_ready(True)
        bj = online_pair(mi)
        if bj is not None:
            k = m["ins"][0][1]
            xt = CTYPE[spec.ext[k][0]]
            bt = RTYPE[members[bj]["reduce"]["acc"]]
            fexp = _fname("exp", spec.ext[k][0])
            S.append("    %s om = (%s)(-INFINITY); %s os = 0;" % (xt, xt, bt))
            S.append("    for (i64 c0 = (i64)threadIdx.x * %d; c0 < a.K; c0 += %d) {" % (V, T_BLOCK * V))
            S.append("      const Pack<%s, %d> xv = *(const Pack<%s, %d>*)(xp%d + c0);" % (xt, V, xt, V, k))
            S.append("      %s cm = xv.v[0];" % xt)
            for j in range(1, V):
                S.append("      cm = fmax_nan<%s>(cm, xv.v[%d]);" % (xt, j))
            S.append("      const %s nm = fmax_nan<%s>(om, cm);" % (xt, xt))
            S.append("      if (!(nm == om)) { os = os * (%s)exp((double)om - (double)nm); om = nm; }" % bt)
            for j in range(V):
                S.append("      os += (%s)%s(xv.v[%d] - om);" % (bt, fexp, j))
            S.append("    }")
            # block combine: global max, then rescaled sums in wave / lane order
            S.append("    %s gm = om;" % xt)
            S.append("    for (int s_ = 32; s_ > 0; s_ >>= 1) gm = fmax_nan<%s>(gm, shfl_xor_<%s>(gm, s_));" % (xt, xt))
            S.append("    __syncthreads();")
            S.append("    if (lane == 0) ((%s*)red_sm)[wave] = gm;" % xt)
            S.append("    __syncthreads();")
            S.append("    gm = ((%s*)red_sm)[0];" % xt)
            S.append("    for (int w_ = 1; w_ < %d; ++w_) gm = fmax_nan<%s>(gm, ((%s*)red_sm)[w_]);" % (nw, xt, xt))
            S.append("    %s gs = (om == gm || os == 0) ? os : os * (%s)exp((double)om - (double)gm);" % (bt, bt))
            S.append("    for (int s_ = 32; s_ > 0; s_ >>= 1) gs += shfl_xor_<%s>(gs, s_);" % bt)
            S.append("    __syncthreads();")
            S.append("    if (lane == 0) ((%s*)red_sm)[wave] = gs;" % bt)
            S.append("    __syncthreads();")
            S.append("    gs = ((%s*)red_sm)[0];" % bt)
            S.append("    for (int w_ = 1; w_ < %d; ++w_) gs += ((%s*)red_sm)[w_];" % (nw, bt))
            for idx, val, accd in ((mi, "gm", red["acc"]), (bj, "gs", members[bj]["reduce"]["acc"])):
                rr = members[idx]["reduce"]
                S.append("    const %s r%d = %s;" % (RTYPE[rr["out"]], idx, _cast(val, accd, rr["out"])))
                rdt[idx] = rr["out"]
                if rr.get("slot") is not None:
                    S.append("    if (threadIdx.x == 0) ((%s*)a.ptr[%d])[%s] = %s;"
                             % (CTYPE[rr["out"]], rr["slot"], row_off(rr["slot"]),
                                _store_val("r%d" % idx, rr["out"], rr["out"])))
            fused_done.add(bj)
            stage += 1
            continue
        acc_t = RTYPE[red["acc"]]
        S.append("    %s acc%d = %s;" % (acc_t, mi, red_identity(red["op"], red["acc"])))

        def tail(j, outs, mi=mi, red=red):
            oe, od = outs[mi]
            v = _cast(oe[red["ref"]], od[red["ref"]], red["acc"])
            S.append("      acc%d = %s;" % (mi, red_combine(red["op"], red["acc"], "acc%d" % mi, v)))
        emit_sweep(stage, closure([mi]), tail)
        stage += 1
        # block reduce, result to every thread (fixed order: deterministic)
        S.append("    for (int s_ = 32; s_ > 0; s_ >>= 1) acc%d = %s;"
                 % (mi, red_combine(red["op"], red["acc"], "acc%d" % mi,
                                    "shfl_xor_<%s>(acc%d, s_)" % (acc_t, mi))))
        S.append("    __syncthreads();")
        S.append("    if (lane == 0) ((%s*)red_sm)[wave] = acc%d;" % (acc_t, mi))
        S.append("    __syncthreads();")
        S.append("    %s tot%d = ((%s*)red_sm)[0];" % (acc_t, mi, acc_t))
        S.append("    for (int w_ = 1; w_ < %d; ++w_) tot%d = %s;"
                 % (nw, mi, red_combine(red["op"], red["acc"], "tot%d" % mi, "((%s*)red_sm)[w_]" % acc_t)))
        S.append("    const %s r%d = %s;" % (RTYPE[red["out"]], mi, _cast("tot%d" % mi, red["acc"], red["out"])))
        rdt[mi] = red["out"]
        if red.get("slot") is not None:
            S.append("    if (threadIdx.x == 0) ((%s*)a.ptr[%d])[%s] = %s;"
                     % (CTYPE[red["out"]], red["slot"], row_off(red["slot"]),
                        _store_val("r%d" % mi, red["out"], red["out"])))
    emit_rowlike_ready(True)
    # final sweep: members with full-size stores
    storing = [mi for mi, m in enumerate(members) if m.get("stores") and not rowlike[mi]]
    if storing:
        packs = {}
        for mi in storing:
            for oref, odt, slot in members[mi]["stores"]:
                packs[(mi, oref, slot)] = odt

        def tail(j, outs):
            for (mi, oref, slot), odt in packs.items():
                oe, od = outs[mi]
                if j == 0:
                    S.append("      Pack<%s, %d> y%d;" % (CTYPE[odt], V, slot))
                S.append("      y%d.v[%d] = %s;" % (slot, j, _store_val(oe[oref], od[oref], odt)))
                if j == V - 1:
                    S.append("      *(Pack<%s, %d>*)((%s*)a.ptr[%d] + %s + c0) = y%d;"
                             % (CTYPE[odt], V, CTYPE[odt], slot, row_off(slot), slot))
        emit_sweep(stage, closure(storing), tail)
    S.append("    __syncthreads();")
    S.append("  }")
    S.append("}")
    return "\n".join(S) + "\n", (name,)
