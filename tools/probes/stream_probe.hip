// Read-only stream probe (round 5): what rate can ONE launch over a MALL-cold 128 MiB buffer reach
// with (V0) plain 16-byte loads, (V1) non-temporal loads, (V2) LDS-DMA (global_load_lds, 16 bytes
// per lane) by NL loader waves per workgroup with nothing consuming the data?  256 workgroups (one
// per CU), each streams its contiguous 1/256 of the buffer.  Prints us per launch and TB/s.
// build: hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NT, int U, int TH = 1024>
__global__ __launch_bounds__(TH) void k_plain(const d2* __restrict__ x, size_t vec_per_wg, double* out) {
  const d2* p = x + (size_t)blockIdx.x * vec_per_wg;
  double acc = 0;
  for (size_t i = threadIdx.x; i + (U - 1) * TH < vec_per_wg; i += U * TH) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * TH) : p[i + u * TH];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
  }
  // (no finalize: one partial per thread-0-of-wave is enough to keep the loads alive)
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = acc;
}

// NL loader waves; wave w streams fills w, w + NL, ... of the workgroup's chunk into ITS OWN two
// 16 KiB LDS slots (ping-pong), waits for each fill, and reads one element back (keeps the DMA live)
template <int NL>
__global__ __launch_bounds__(64 * NL) void k_dma(const char* __restrict__ x, size_t bytes_per_wg, double* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const char* p = x + (size_t)blockIdx.x * bytes_per_wg;
  const size_t nfill = bytes_per_wg / 16384;
  char* my = lds + wave * 32768;
  double acc = 0;
  int par = 0;
  for (size_t f = wave; f < nfill; f += NL, par ^= 1) {
    const char* src = p + f * 16384 + lane * 16;
    char* dst = my + par * 16384;
#pragma unroll
    for (int j = 0; j < 16; ++j)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + j * 1024),
                                       (void __attribute__((address_space(3)))*)(dst + j * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // the PREVIOUS fill of this wave has landed
    acc += *(const double*)(my + (par ^ 1) * 16384 + lane * 8);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  acc += *(const double*)(my + lane * 8);
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
  if (lane == 0) out[blockIdx.x * 16 + wave] = acc;
}

template <typename F> float time_it(F launch, int iters) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 16; ++i) launch(i);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) launch(i);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters * 1e3f;
}

int main() {
  const size_t N = 128ull << 20;          // bytes per buffer
  const int NB = 8, WG = 256;
  std::vector<char*> bufs(NB);
  for (auto& b : bufs) { CK(hipMalloc(&b, N)); CK(hipMemset(b, 0, N)); }
  double* out; CK(hipMalloc(&out, 4 * WG * 16 * 8));
  CK(hipFuncSetAttribute((const void*)k_dma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768));
  CK(hipFuncSetAttribute((const void*)k_dma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
  const size_t per = N / WG;
#define RUN(name, expr) { float us = time_it([&](int i) { const char* x = bufs[i % NB]; (void)x; expr; }, 400); \
    printf("%-44s %7.2f us  %5.2f TB/s\n", name, us, N / (us * 1e-6) / 1e12); }
  for (int rep = 0; rep < 2; ++rep) {
    RUN("plain 16B loads, 1024 thr, 2 in flight", (k_plain<0, 2><<<WG, 1024>>>((const d2*)x, per / 16, out)));
    RUN("plain 16B loads, 1024 thr, 4 in flight", (k_plain<0, 4><<<WG, 1024>>>((const d2*)x, per / 16, out)));
    RUN("plain 16B loads, 1024 thr, 8 in flight", (k_plain<0, 8><<<WG, 1024>>>((const d2*)x, per / 16, out)));
    RUN("nt    16B loads, 1024 thr, 4 in flight", (k_plain<1, 4><<<WG, 1024>>>((const d2*)x, per / 16, out)));
#define GRID(NT_, U_, TH_) RUN("nt=" #NT_ " 16B x " #U_ " in flight, " #TH_ " threads per CU", (k_plain<NT_, U_, TH_><<<WG, TH_>>>((const d2*)x, per / 16, out)))
    GRID(1, 1, 1024); GRID(1, 2, 1024); GRID(1, 3, 1024); GRID(1, 4, 1024);
    GRID(1, 1, 512); GRID(1, 2, 512); GRID(1, 3, 512); GRID(1, 4, 512); GRID(1, 8, 512);
    GRID(1, 2, 256); GRID(1, 4, 256); GRID(1, 8, 256); GRID(1, 16, 256);
    GRID(0, 1, 1024); GRID(0, 2, 512); GRID(0, 4, 512); GRID(0, 4, 256); GRID(0, 8, 256);
    RUN("LDS-DMA, 2 loader waves", (k_dma<2><<<WG, 128, 2 * 32768>>>(x, per, out)));
    RUN("LDS-DMA, 4 loader waves", (k_dma<4><<<WG, 256, 4 * 32768>>>(x, per, out)));
    RUN("LDS-DMA, 2 loader waves x 2 WG/CU", (k_dma<2><<<2 * WG, 128, 2 * 32768>>>(x, per / 2, out)));
  }
  CK(hipDeviceSynchronize());
  return 0;
}
