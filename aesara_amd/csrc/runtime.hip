// Runtime part of the C-ABI: init, errors, hiprtc compilation, module/kernel handles,
// hipGraph capture/replay, stream-side event timing.  See include/aesara_hip.h.
#include <hip/hiprtc.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[8192] = "";
static int g_cu_count = 0;

void ahip_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int ahip_cu_count() {
  if (g_cu_count == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess)
      g_cu_count = p.multiProcessorCount;
    else
      g_cu_count = 256;
  }
  return g_cu_count;
}

struct ahip_graph_s { hipGraph_t graph; hipGraphExec_t exec; };

// ---- launch lists --------------------------------------------------------------------------
struct ahip_comm_s;
enum RecKind : uint8_t { REC_STATIC = 0, REC_MODULE = 1, REC_MODULE_COOP = 2, REC_ALLREDUCE = 3 };
struct LaunchRec {
  const void* func;      // static kernel (hipLaunchKernel) or nullptr
  hipFunction_t mfunc;   // module kernel (hipModuleLaunchKernel) or nullptr
  dim3 grid, block;
  unsigned shmem;
  std::vector<char> arg; // by-value argument block (REC_ALLREDUCE: {send pointer, recv pointer})
  RecKind kind;
  bool has_ptrs;                 // the launch site declared where the device pointers are
  std::vector<uint16_t> ptrs;    // their byte offsets inside `arg`
  // REC_ALLREDUCE
  ahip_comm_s* comm; int dtype, op; int64_t count;
};
// relocation: 8-byte word `off` of launch `rec`'s argument block points into rebinding range `base`
struct Reloc { uint32_t rec, off, base; };
struct ahip_list_s {
  std::vector<LaunchRec> recs;
  std::vector<Reloc> relocs;
  std::vector<uint64_t> cur;   // the base address each rebinding range is currently patched to
};
static thread_local ahip_list_s* g_recording = nullptr;

int ahip_comm_issue(ahip_comm_s* c, int dtype, int op, const void* send, void* recv, int64_t count,
                    hipStream_t s);   // comm.hip

static int issue(const LaunchRec& r, hipStream_t s) {
  if (r.kind == REC_ALLREDUCE) {
    const void* send; void* recv;
    memcpy(&send, r.arg.data(), 8);
    memcpy(&recv, r.arg.data() + 8, 8);
    return ahip_comm_issue(r.comm, r.dtype, r.op, send, recv, r.count, s);
  }
  if (r.func) {
    void* args[] = {const_cast<char*>(r.arg.data())};
    AHIP_CHECK_HIP(hipLaunchKernel(r.func, r.grid, r.block, args, r.shmem, s));
  } else if (r.kind == REC_MODULE_COOP) {
    void* params[] = {const_cast<char*>(r.arg.data())};
    AHIP_CHECK_HIP(hipModuleLaunchCooperativeKernel(r.mfunc, r.grid.x, r.grid.y, r.grid.z, r.block.x,
                                                    r.block.y, r.block.z, r.shmem, s, params));
  } else {
    size_t sz = r.arg.size();
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<char*>(r.arg.data()),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    AHIP_CHECK_HIP(hipModuleLaunchKernel(r.mfunc, r.grid.x, r.grid.y, r.grid.z, r.block.x,
                                         r.block.y, r.block.z, r.shmem, s, nullptr, config));
  }
  return AHIP_OK;
}

static int launch_or_record(const void* func, hipFunction_t mfunc, dim3 grid, dim3 block,
                            size_t shmem, hipStream_t s, const void* arg, size_t arg_size,
                            const uint16_t* ptr_off, int n_ptr, int coop) {
  if (grid.x == 0 || grid.y == 0 || grid.z == 0) {
    ahip_set_error("empty launch grid");
    return AHIP_EINVAL;
  }
  for (int k = 0; k < n_ptr; ++k)
    if ((size_t)ptr_off[k] + 8 > arg_size || (ptr_off[k] & 7)) {
      ahip_set_error("pointer offset %d outside / misaligned in a %zu-byte argument block",
                     (int)ptr_off[k], arg_size);
      return AHIP_EINVAL;
    }
  if (g_recording) {
    LaunchRec r{};
    r.func = func; r.mfunc = mfunc; r.grid = grid; r.block = block; r.shmem = (unsigned)shmem;
    r.kind = func ? REC_STATIC : (coop ? REC_MODULE_COOP : REC_MODULE);
    r.arg.assign(static_cast<const char*>(arg), static_cast<const char*>(arg) + arg_size);
    r.has_ptrs = n_ptr >= 0;
    if (n_ptr > 0) r.ptrs.assign(ptr_off, ptr_off + n_ptr);
    g_recording->recs.push_back(std::move(r));
    return AHIP_OK;
  }
  if (func) {
    void* args[] = {const_cast<void*>(arg)};
    AHIP_CHECK_HIP(hipLaunchKernel(func, grid, block, args, shmem, s));
  } else if (coop) {
    void* params[] = {const_cast<void*>(arg)};
    AHIP_CHECK_HIP(hipModuleLaunchCooperativeKernel(mfunc, grid.x, grid.y, grid.z, block.x, block.y,
                                                    block.z, (unsigned)shmem, s, params));
  } else {
    size_t sz = arg_size;
    void* config[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, const_cast<void*>(arg),
                      HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
    AHIP_CHECK_HIP(hipModuleLaunchKernel(mfunc, grid.x, grid.y, grid.z, block.x, block.y, block.z,
                                         shmem, s, nullptr, config));
  }
  return AHIP_OK;
}

int ahip_launch_static(const void* func, dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                       const void* arg, size_t arg_size, const uint16_t* ptr_off, int n_ptr) {
  return launch_or_record(func, nullptr, grid, block, shmem, s, arg, arg_size, ptr_off, n_ptr, 0);
}
int ahip_launch_module(hipFunction_t f, dim3 grid, dim3 block, size_t shmem, hipStream_t s,
                       const void* arg, size_t arg_size, const uint16_t* ptr_off, int n_ptr,
                       int cooperative) {
  return launch_or_record(nullptr, f, grid, block, shmem, s, arg, arg_size, ptr_off, n_ptr,
                          cooperative);
}
// a collective as a launch-list entry (comm.hip calls this while a list is being recorded)
bool ahip_list_recording() { return g_recording != nullptr; }
int ahip_list_record_allreduce(ahip_comm_s* c, int dtype, int op, const void* send, void* recv,
                               int64_t count) {
  LaunchRec r{};
  r.kind = REC_ALLREDUCE; r.comm = c; r.dtype = dtype; r.op = op; r.count = count;
  r.grid = dim3(1); r.block = dim3(1);
  r.arg.resize(16);
  memcpy(r.arg.data(), &send, 8);
  memcpy(r.arg.data() + 8, &recv, 8);
  r.has_ptrs = true;
  r.ptrs = {0, 8};
  g_recording->recs.push_back(std::move(r));
  return AHIP_OK;
}
struct ahip_event_s { hipEvent_t ev; };

extern "C" {

int ahip_abi_version(void) { return AHIP_ABI_VERSION; }

const char* ahip_last_error(void) { return g_err; }

int ahip_init(int device_ordinal) {
  int n = 0;
  AHIP_CHECK_HIP(hipGetDeviceCount(&n));
  AHIP_REQUIRE(device_ordinal >= 0 && device_ordinal < n, "device %d out of range (%d devices)",
               device_ordinal, n);
  AHIP_CHECK_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t p;
  AHIP_CHECK_HIP(hipGetDeviceProperties(&p, device_ordinal));
  g_cu_count = p.multiProcessorCount;
  return AHIP_OK;
}

int ahip_get_device_info(ahip_device_info* out) {
  AHIP_REQUIRE(out != nullptr, "null out");
  int dev = 0;
  AHIP_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t p;
  AHIP_CHECK_HIP(hipGetDeviceProperties(&p, dev));
  memset(out, 0, sizeof(*out));
  out->device = dev;
  out->cu_count = p.multiProcessorCount;
  out->wavefront_size = p.warpSize;
  out->max_threads_per_block = p.maxThreadsPerBlock;
  out->total_mem = (int64_t)p.totalGlobalMem;
  out->lds_per_block = (int64_t)p.sharedMemPerBlock;
  out->clock_khz = p.clockRate;
  out->l2_bytes = p.l2CacheSize;
  snprintf(out->arch, sizeof(out->arch), "%s", p.gcnArchName);
  snprintf(out->name, sizeof(out->name), "%s", p.name);
  return AHIP_OK;
}

int ahip_stream_synchronize(void* stream) {
  AHIP_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));
  return AHIP_OK;
}

// ---- hiprtc ------------------------------------------------------------------------------
int ahip_compile(const char* source, const char* name, const char* const* options, int n_options,
                 void** code_out, size_t* size_out) {
  AHIP_REQUIRE(source && code_out && size_out, "null argument");
  hiprtcProgram prog;
  hiprtcResult r = hiprtcCreateProgram(&prog, source, name ? name : "aesara_hip_kernel.hip", 0,
                                       nullptr, nullptr);
  if (r != HIPRTC_SUCCESS) {
    ahip_set_error("hiprtcCreateProgram: %s", hiprtcGetErrorString(r));
    return AHIP_ECOMPILE;
  }
  std::vector<const char*> opts;
  bool has_arch = false;
  for (int i = 0; i < n_options; ++i) {
    opts.push_back(options[i]);
    if (strstr(options[i], "--offload-arch")) has_arch = true;
  }
  if (!has_arch) opts.push_back("--offload-arch=gfx950");
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls + 1, '\0');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    ahip_set_error("hiprtc compile failed (%s):\n%.7000s", hiprtcGetErrorString(r), log.c_str());
    hiprtcDestroyProgram(&prog);
    return AHIP_ECOMPILE;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  void* buf = malloc(cs);
  if (!buf) {
    hiprtcDestroyProgram(&prog);
    ahip_set_error("out of host memory");
    return AHIP_EINVAL;
  }
  hiprtcGetCode(prog, (char*)buf);
  hiprtcDestroyProgram(&prog);
  *code_out = buf;
  *size_out = cs;
  return AHIP_OK;
}

int ahip_free_code(void* code) {
  free(code);
  return AHIP_OK;
}

int ahip_module_load(const void* code, size_t size, ahip_module_t* out) {
  AHIP_REQUIRE(code && out && size > 0, "null argument");
  hipModule_t m;
  AHIP_CHECK_HIP(hipModuleLoadData(&m, code));
  *out = new ahip_module_s{m};
  return AHIP_OK;
}

int ahip_module_get_function(ahip_module_t m, const char* kernel_name, ahip_fn_t* out) {
  AHIP_REQUIRE(m && kernel_name && out, "null argument");
  hipFunction_t f;
  AHIP_CHECK_HIP(hipModuleGetFunction(&f, m->mod, kernel_name));
  *out = new ahip_func_s{f};
  return AHIP_OK;
}

int ahip_module_unload(ahip_module_t m) {
  if (!m) return AHIP_OK;
  AHIP_CHECK_HIP(hipModuleUnload(m->mod));
  delete m;
  return AHIP_OK;
}

int ahip_launch(ahip_fn_t f, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                uint32_t bz, uint32_t shmem_bytes, const void* kernarg, size_t kernarg_size,
                void* stream) {
  AHIP_REQUIRE(f != nullptr, "null kernel");
  AHIP_REQUIRE(gx > 0 && gy > 0 && gz > 0 && bx > 0, "empty launch");
  return ahip_launch_module(f->fn, dim3(gx, gy, gz), dim3(bx, by, bz), shmem_bytes,
                            as_stream(stream), kernarg, kernarg_size, nullptr, -1);
}

int ahip_launch_p(ahip_fn_t f, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx, uint32_t by,
                  uint32_t bz, uint32_t shmem_bytes, const void* kernarg, size_t kernarg_size,
                  const uint16_t* ptr_offsets, int n_ptrs, int cooperative, void* stream) {
  AHIP_REQUIRE(f != nullptr, "null kernel");
  AHIP_REQUIRE(gx > 0 && gy > 0 && gz > 0 && bx > 0, "empty launch");
  AHIP_REQUIRE(n_ptrs >= 0 && (n_ptrs == 0 || ptr_offsets), "bad pointer map");
  return ahip_launch_module(f->fn, dim3(gx, gy, gz), dim3(bx, by, bz), shmem_bytes,
                            as_stream(stream), kernarg, kernarg_size, ptr_offsets, n_ptrs,
                            cooperative);
}

// co-residency of a persistent (spinning) grid: how many workgroups of `f` fit on the device
int ahip_occupancy(ahip_fn_t f, int block_threads, size_t dyn_lds_bytes, int* blocks_per_cu,
                   int* cu_count) {
  AHIP_REQUIRE(f && blocks_per_cu && cu_count, "null argument");
  int nb = 0;
  AHIP_CHECK_HIP(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f->fn, block_threads,
                                                                    dyn_lds_bytes));
  *blocks_per_cu = nb;
  *cu_count = ahip_cu_count();
  return AHIP_OK;
}

// ---- hipGraph capture / replay -----------------------------------------------------------
int ahip_graph_begin(void* stream) {
  AHIP_CHECK_HIP(hipStreamBeginCapture(as_stream(stream), hipStreamCaptureModeThreadLocal));
  return AHIP_OK;
}

int ahip_graph_end(void* stream, ahip_graph_t* out) {
  AHIP_REQUIRE(out != nullptr, "null out");
  hipGraph_t g = nullptr;
  AHIP_CHECK_HIP(hipStreamEndCapture(as_stream(stream), &g));
  hipGraphExec_t e = nullptr;
  hipError_t err = hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
  if (err != hipSuccess) {
    (void)hipGraphDestroy(g);
    ahip_set_error("hipGraphInstantiate failed: %s", hipGetErrorString(err));
    return AHIP_EHIP;
  }
  *out = new ahip_graph_s{g, e};
  return AHIP_OK;
}

int ahip_graph_launch(ahip_graph_t g, void* stream) {
  AHIP_REQUIRE(g != nullptr, "null graph");
  AHIP_CHECK_HIP(hipGraphLaunch(g->exec, as_stream(stream)));
  return AHIP_OK;
}

int ahip_graph_destroy(ahip_graph_t g) {
  if (!g) return AHIP_OK;
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  delete g;
  return AHIP_OK;
}

// ---- launch lists: record once, replay with one host call (H1, the CVM analogue) --------
int ahip_list_begin(void) {
  AHIP_REQUIRE(g_recording == nullptr, "a launch list is already being recorded on this thread");
  g_recording = new ahip_list_s();
  return AHIP_OK;
}

int ahip_list_end(ahip_list_t* out) {
  AHIP_REQUIRE(g_recording != nullptr, "no launch list is being recorded");
  ahip_list_s* l = g_recording;
  g_recording = nullptr;
  if (!out) {
    delete l;
    ahip_set_error("null out");
    return AHIP_EINVAL;
  }
  *out = l;
  return AHIP_OK;
}

int ahip_list_length(ahip_list_t l) { return l ? (int)l->recs.size() : -1; }

int ahip_list_run(ahip_list_t l, void* stream) {
  AHIP_REQUIRE(l != nullptr, "null list");
  hipStream_t s = as_stream(stream);
  for (const LaunchRec& r : l->recs) {
    int rc = issue(r, s);
    if (rc) return rc;
  }
  return AHIP_OK;
}

// Zero-copy replay for fresh buffers (a training loop hands over a NEW batch tensor on every
// call): after recording, the caller names the address ranges of its rebindable buffers (plan
// inputs, `out=` targets); every DECLARED pointer word (the pointer map each launch site passes
// with its argument block) that points into one of them becomes a relocation.  ahip_list_run_rebased patches those words by the distance
// the buffers moved and re-issues the launches — one host call, no staging copy.
int ahip_list_bind_bases(ahip_list_t l, const uint64_t* lo, const uint64_t* hi, int n) {
  AHIP_REQUIRE(l != nullptr && (n == 0 || (lo && hi)), "null argument");
  l->relocs.clear();
  l->cur.assign(lo, lo + n);
  for (int a = 0; a < n; ++a)
    for (int b = 0; b < a; ++b)
      AHIP_REQUIRE(hi[a] <= lo[b] || hi[b] <= lo[a] || lo[a] == hi[a] || lo[b] == hi[b],
                   "rebinding ranges %d and %d overlap", a, b);
  // only words a launch site DECLARED as device pointers are candidates: a scalar whose bit
  // pattern happens to fall into a range is never touched, a pointer is never missed
  for (size_t r = 0; r < l->recs.size(); ++r) {
    const LaunchRec& rec = l->recs[r];
    if (!rec.has_ptrs) {
      ahip_set_error("launch %zu of the list was recorded without a pointer map", r);
      return -2;
    }
    for (uint16_t off : rec.ptrs) {
      uint64_t v;
      memcpy(&v, rec.arg.data() + off, 8);
      for (int k = 0; k < n; ++k)
        if (v >= lo[k] && v < hi[k]) {
          l->relocs.push_back(Reloc{(uint32_t)r, (uint32_t)off, (uint32_t)k});
          break;
        }
    }
  }
  return (int)l->relocs.size();
}

int ahip_list_run_rebased(ahip_list_t l, const uint64_t* bases, int n, void* stream) {
  AHIP_REQUIRE(l != nullptr && (size_t)n == l->cur.size(), "rebinding count mismatch");
  bool moved = false;
  for (int k = 0; k < n; ++k) moved |= (bases[k] != l->cur[k]);
  if (moved) {
    for (const Reloc& q : l->relocs) {
      const uint64_t delta = bases[q.base] - l->cur[q.base];
      if (!delta) continue;
      uint64_t v;
      char* p = l->recs[q.rec].arg.data() + q.off;
      memcpy(&v, p, 8);
      v += delta;
      memcpy(p, &v, 8);
    }
    l->cur.assign(bases, bases + n);
  }
  return ahip_list_run(l, stream);
}

int ahip_list_destroy(ahip_list_t l) {
  delete l;
  return AHIP_OK;
}

// ---- events ------------------------------------------------------------------------------
int ahip_event_create(ahip_event_t* out) {
  AHIP_REQUIRE(out != nullptr, "null out");
  hipEvent_t e;
  AHIP_CHECK_HIP(hipEventCreate(&e));
  *out = new ahip_event_s{e};
  return AHIP_OK;
}

int ahip_event_record(ahip_event_t e, void* stream) {
  AHIP_REQUIRE(e != nullptr, "null event");
  AHIP_CHECK_HIP(hipEventRecord(e->ev, as_stream(stream)));
  return AHIP_OK;
}

int ahip_event_elapsed_ms(ahip_event_t start, ahip_event_t stop, float* ms) {
  AHIP_REQUIRE(start && stop && ms, "null argument");
  AHIP_CHECK_HIP(hipEventSynchronize(stop->ev));
  AHIP_CHECK_HIP(hipEventElapsedTime(ms, start->ev, stop->ev));
  return AHIP_OK;
}

int ahip_event_query(ahip_event_t e) {
  AHIP_REQUIRE(e != nullptr, "null event");
  hipError_t r = hipEventQuery(e->ev);
  if (r == hipSuccess) return 0;
  if (r == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
  ahip_set_error("hipEventQuery: %s", hipGetErrorString(r));
  return AHIP_EHIP;
}

int ahip_event_destroy(ahip_event_t e) {
  if (!e) return AHIP_OK;
  (void)hipEventDestroy(e->ev);
  delete e;
  return AHIP_OK;
}

}  // extern "C"
