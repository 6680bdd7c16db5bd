// Shared helpers for the libaesara_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstddef>
#include <cstring>

#include "../../include/aesara_hip.h"

void ahip_set_error(const char* fmt, ...);

#define AHIP_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ahip_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                     __LINE__);                                                           \
      return AHIP_EHIP;                                                                   \
    }                                                                                     \
  } while (0)

#define AHIP_REQUIRE(cond, ...)     \
  do {                              \
    if (!(cond)) {                  \
      ahip_set_error(__VA_ARGS__);  \
      return AHIP_EINVAL;           \
    }                               \
  } while (0)

static inline int ahip_itemsize(int dt) {
  switch (dt) {
    case AHIP_BOOL: case AHIP_I8: case AHIP_U8: return 1;
    case AHIP_I16: case AHIP_U16: return 2;
    case AHIP_I32: case AHIP_U32: case AHIP_F32: return 4;
    case AHIP_I64: case AHIP_U64: case AHIP_F64: return 8;
    default: return 0;
  }
}

struct ahip_module_s { hipModule_t mod; };
struct ahip_func_s { hipFunction_t fn; };

int ahip_cu_count();  // cached after ahip_init / first use
void ahip_gemm_set_small_max_tiles(int64_t v);
void ahip_gemm_set_half_max_tiles(int64_t v);  // 64x64-tile window (ahip_set_param)
void ahip_gemm_set_half_min_tiles(int64_t v);
void ahip_gemm_set_half_ksplit(int64_t v);     // K groups inside a 64x64-tile workgroup (1 / 2 / 4)
void ahip_gemm_set_skinny_nf(int64_t v);  // gemm.hip tuning knob (ahip_set_param)
void ahip_copy_set_stream_bytes(int64_t v);     // copy.hip: streaming loads from this many bytes on (ahip_set_param)
void ahip_gemv_set_col_blocks_per_cu(int64_t v);  // gemv.hip tuning knobs (ahip_set_param)
void ahip_gemv_set_col_strip_lanes(int64_t v);
void ahip_index_set_argmax_max_slices(int64_t v);
void ahip_gemm_set_group(int64_t v);

// Every kernel launch of the library goes through these two helpers so that a launch list
// (ahip_list_*: the CVM analogue) can record the launches instead of executing them.
// `ptr_off` / `n_ptr`: byte offsets of the DEVICE POINTERS inside the argument block — what a
// recorded launch may have re-pointed at new buffers (ahip_list_bind_bases): declared by the
// owner of the struct, never guessed from the bytes.  n_ptr < 0 = not declared (such a launch
// makes its list non-rebindable).
int ahip_launch_static(const void* func, dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                       const void* arg, size_t arg_size, const uint16_t* ptr_off, int n_ptr);
int ahip_launch_module(hipFunction_t f, dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                       const void* arg, size_t arg_size, const uint16_t* ptr_off, int n_ptr,
                       int cooperative = 0);

#define AHIP_MAX_PTRS 192
// pointer map of an argument struct:  AHIP_PTRS_BEGIN(T) AHIP_PTR1(member) AHIP_PTRA(array, n) AHIP_PTRS_END
#define AHIP_PTRS_BEGIN(T)                                              \
  static inline int ahip_ptrs(const T*, uint16_t* o_) {                 \
    typedef T S_;                                                       \
    int n_ = 0;
#define AHIP_PTR1(m) o_[n_++] = (uint16_t)offsetof(S_, m);
#define AHIP_PTRA(m, cnt) \
  for (int i_ = 0; i_ < (int)(cnt); ++i_) o_[n_++] = (uint16_t)(offsetof(S_, m) + 8 * i_);
#define AHIP_PTRS_END \
    return n_;        \
  }

AHIP_PTRS_BEGIN(ahip_ew_args) AHIP_PTRA(ptr, AHIP_MAXOPS) AHIP_PTR1(ws) AHIP_PTR1(out) AHIP_PTRS_END
AHIP_PTRS_BEGIN(ahip_ewh_args) AHIP_PTRA(ptr, AHIP_HJOBS * AHIP_HOPS) AHIP_PTRA(out, AHIP_HJOBS) AHIP_PTR1(ws)
AHIP_PTRS_END
AHIP_PTRS_BEGIN(ahip_gv_args) AHIP_PTRA(A, AHIP_MAXDOTS) AHIP_PTRA(x, AHIP_MAXDOTS)
  AHIP_PTRA(ptr, AHIP_GV_MAXOPS) AHIP_PTRA(xin, AHIP_MAXDOTS * AHIP_GV_MAXXIN) AHIP_PTRA(xout, AHIP_MAXDOTS)
AHIP_PTRS_END
AHIP_PTRS_BEGIN(ahip_rp_args) AHIP_PTR1(X) AHIP_PTR1(w) AHIP_PTRA(ptr, AHIP_RP_MAXOPS) AHIP_PTR1(col_ws)
  AHIP_PTR1(red_ws) AHIP_PTRS_END
AHIP_PTRS_BEGIN(ahip_rc_args) AHIP_PTRA(ptr, AHIP_RC_MAXOPS) AHIP_PTRS_END
AHIP_PTRS_BEGIN(ahip_ge_args) AHIP_PTRA(A, AHIP_GE_MAXDOTS) AHIP_PTRA(B, AHIP_GE_MAXDOTS)
  AHIP_PTRA(ptr, AHIP_GE_MAXOPS) AHIP_PTRS_END

// kernel must take exactly one (struct) argument, passed by value; the struct needs a pointer map
#define AHIP_LAUNCH(kernel, grid, block, shmem, stream, arg)                                   \
  do {                                                                                         \
    uint16_t _po[AHIP_MAX_PTRS];                                                               \
    const int _np = ahip_ptrs(&(arg), _po);                                                    \
    int _rc = ahip_launch_static((const void*)(kernel), grid, block, shmem, stream, &(arg),    \
                                 sizeof(arg), _po, _np);                                       \
    if (_rc) return _rc;                                                                       \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// 8 XCDs on MI355X: workgroup b is dispatched to XCD b % 8 (speed hint only, never correctness).
#define AHIP_NUM_XCD 8
