// Gradient exchange behind the C ABI (include/ocr_hip.h "Data-parallel exchange"): the all-reduce that
// replaces `average_gradients` (multigpu_train.py:70-85) / `sum_gradients` (train_pixellink.py:179-194),
// issued on a caller-supplied stream through a caller-owned RCCL communicator, plus the two stream-ordering
// primitives the step needs around it (event record / stream wait).  With these the whole training step —
// exchange included — is a flat list of C-ABI calls.
//
// RCCL is bound at first use with dlsym/dlopen, not at link time: a process that imported PyTorch-ROCm already
// holds ONE librccl (torch's); binding to that copy keeps a single RCCL runtime in the process, and the
// library still loads (and every non-exchange entry point works) on a box without RCCL.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include "common.h"

namespace {

// the subset of rccl.h this file needs (ABI-stable NCCL 2.x definitions; /opt/rocm/include/rccl/rccl.h:40-43,
// 187,220,260,339,448-468,611)
struct NcclUniqueId { char internal[OCR_COMM_ID_BYTES]; };
typedef void* NcclComm;
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*CommDestroyFn)(NcclComm);
typedef int (*CommCountFn)(NcclComm, int*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef const char* (*ErrStrFn)(int);

struct Rccl {
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  CommCountFn comm_count = nullptr;
  AllReduceFn all_reduce = nullptr;
  ErrStrFn err_str = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
char g_last_error[256] = "";

void* find(void* handle, const char* name) {
  void* p = dlsym(RTLD_DEFAULT, name);          // the copy already in the process (torch's librccl)
  if (!p && handle) p = dlsym(handle, name);
  return p;
}

void bind() {
  void* h = nullptr;
  if (!dlsym(RTLD_DEFAULT, "ncclAllReduce")) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
    if (!h) {
      snprintf(g_last_error, sizeof(g_last_error), "librccl not found: %s", dlerror());
      return;
    }
  }
  g_rccl.get_unique_id = (GetUniqueIdFn)find(h, "ncclGetUniqueId");
  g_rccl.comm_init_rank = (CommInitRankFn)find(h, "ncclCommInitRank");
  g_rccl.comm_destroy = (CommDestroyFn)find(h, "ncclCommDestroy");
  g_rccl.comm_count = (CommCountFn)find(h, "ncclCommCount");
  g_rccl.all_reduce = (AllReduceFn)find(h, "ncclAllReduce");
  g_rccl.err_str = (ErrStrFn)find(h, "ncclGetErrorString");
  g_rccl.ok = g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce;
  if (!g_rccl.ok) snprintf(g_last_error, sizeof(g_last_error), "librccl lacks a required symbol");
}

bool rccl() {
  std::call_once(g_once, bind);
  return g_rccl.ok;
}

int fail(int rc, const char* what) {
  snprintf(g_last_error, sizeof(g_last_error), "%s: %s (%d)", what,
           g_rccl.err_str ? g_rccl.err_str(rc) : "rccl error", rc);
  return OCR_ERR_RCCL;
}

}  // namespace

extern "C" const char* ocr_comm_last_error(void) { return g_last_error; }

extern "C" int ocr_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int ocr_comm_unique_id(void* id_out) {
  if (!id_out) return OCR_ERR_INVALID_ARG;
  if (!rccl()) return OCR_ERR_RCCL;
  NcclUniqueId id;
  int rc = g_rccl.get_unique_id(&id);
  if (rc != 0) return fail(rc, "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return OCR_OK;
}

extern "C" int ocr_comm_init_rank(void** comm_out, int nranks, const void* id, int rank) {
  if (!comm_out || !id || nranks < 1 || rank < 0 || rank >= nranks) return OCR_ERR_INVALID_ARG;
  if (!rccl()) return OCR_ERR_RCCL;
  NcclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  NcclComm c = nullptr;
  int rc = g_rccl.comm_init_rank(&c, nranks, uid, rank);
  if (rc != 0) return fail(rc, "ncclCommInitRank");
  *comm_out = c;
  return OCR_OK;
}

extern "C" int ocr_comm_size(void* comm) {
  if (!comm || !rccl() || !g_rccl.comm_count) return OCR_ERR_INVALID_ARG;
  int n = 0;
  int rc = g_rccl.comm_count((NcclComm)comm, &n);
  if (rc != 0) return fail(rc, "ncclCommCount");
  return n;
}

extern "C" int ocr_comm_destroy(void* comm) {
  if (!comm) return OCR_OK;
  if (!rccl()) return OCR_ERR_RCCL;
  int rc = g_rccl.comm_destroy((NcclComm)comm);
  return rc == 0 ? OCR_OK : fail(rc, "ncclCommDestroy");
}

extern "C" int ocr_allreduce_bucket(void* comm, void* buf, size_t count, int dtype, int op, void* stream) {
  if (!comm || (!buf && count)) return OCR_ERR_INVALID_ARG;
  int nt, no;
  switch (dtype) {
    case OCR_DT_F32: nt = 7; break;       // ncclFloat32
    case OCR_DT_F16: nt = 6; break;       // ncclFloat16
    case OCR_DT_BF16: nt = 9; break;      // ncclBfloat16
    case OCR_DT_I32: nt = 2; break;       // ncclInt32
    default: return OCR_ERR_INVALID_ARG;
  }
  switch (op) {
    case OCR_RED_SUM: no = 0; break;      // ncclSum
    case OCR_RED_MAX: no = 2; break;      // ncclMax
    default: return OCR_ERR_INVALID_ARG;
  }
  if (count == 0) return OCR_OK;
  if (!rccl()) return OCR_ERR_RCCL;
  int rc = g_rccl.all_reduce(buf, buf, count, nt, no, (NcclComm)comm, (hipStream_t)stream);
  return rc == 0 ? OCR_OK : fail(rc, "ncclAllReduce");
}

// --- stream ordering --------------------------------------------------------------------------------------
extern "C" int ocr_event_create(void** event_out) {
  if (!event_out) return OCR_ERR_INVALID_ARG;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return OCR_ERR_HIP;
  *event_out = e;
  return OCR_OK;
}

extern "C" int ocr_event_destroy(void* event) {
  if (!event) return OCR_OK;
  return hipEventDestroy((hipEvent_t)event) == hipSuccess ? OCR_OK : OCR_ERR_HIP;
}

extern "C" int ocr_event_record(void* event, void* stream) {
  if (!event) return OCR_ERR_INVALID_ARG;
  return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? OCR_OK : OCR_ERR_HIP;
}

extern "C" int ocr_stream_wait_event(void* stream, void* event) {
  if (!event) return OCR_ERR_INVALID_ARG;
  return hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) == hipSuccess ? OCR_OK : OCR_ERR_HIP;
}

// --- one-GPU stand-in for the device side of a multi-rank ring all-reduce --------------------
// No 8-GPU node is available to this build, so how RCCL's kernels co-schedule with the step's conv workgroups (one
// 512-VGPR wave per SIMD: a CU that holds one has no registers left for anything else) has never been observed.  This
// kernel has the SHAPE of RCCL's device code on the comm stream — a persistent grid of a few dozen 256-thread
// workgroups with a small register footprint, walking the bucket in 256 KB chunks, every byte read once and written
// once (2 x bucket bytes through HBM, the values unchanged), chunk k released no earlier than k * (2 * chunk bytes /
// link rate) after the workgroup started: the pace one xGMI link (~150 GB/s) sets for a ring — and is launched where
// the recorded step launches ocr_allreduce_bucket, with the same event ordering.  stats (8 x u64, device, set by
// the caller once to {~0, 0, ...}): [3] accumulates the busy time of every launch (last workgroup's end - first workgroup's start,
// 100 MHz ticks), [4] counts launches; [0..2] are per-launch scratch that resets itself.
// Round 6: the stand-in's own footprint decides what it can share a CU with — 18 registers sit beside
// anything that leaves 24 free, RCCL's kernels do not.  FAT = the same walk with ~128 VGPRs and 64 KB of LDS per
// workgroup (the order of RCCL's generic all-reduce kernels: unrolled 16-byte loads in flight per lane, a staging
// buffer per channel), so the two arms BRACKET what a real ring's device side costs the step (bench.py: exchange.proxy /
// exchange.proxy_fat).  ocr_comm_proxy_set_footprint selects the variant (and optionally the workgroup count) for the
// launches that follow — the recorded step's ocr_comm_proxy calls are replayed unchanged.
namespace {
constexpr size_t kProxyChunk = 256 << 10;
int g_proxy_fat = 0, g_proxy_wgs = 0;
template <bool FAT>
__global__ __launch_bounds__(256) void comm_proxy_kernel(uint4* __restrict__ buf, size_t bytes, unsigned long long ticks_per_chunk,
                                                         unsigned long long* __restrict__ stats) {
  if (FAT) {
    extern __shared__ unsigned fat_lds[];
    asm volatile("v_mov_b32 v127, 0" ::: "v127");            // the allocation reaches 128 VGPRs
    fat_lds[threadIdx.x] = threadIdx.x;                       // (the 64 KB are requested at launch; keep the segment live)
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t nchunks = (bytes + kProxyChunk - 1) / kProxyChunk;
  for (size_t k = blockIdx.x; k < nchunks; k += gridDim.x) {
    const unsigned long long due = t0 + k * ticks_per_chunk;
    while (__builtin_amdgcn_s_memrealtime() < due) __builtin_amdgcn_s_sleep(32);
    const size_t b0 = k * kProxyChunk, b1 = b0 + kProxyChunk < bytes ? b0 + kProxyChunk : bytes;
    for (size_t o = b0 + (size_t)threadIdx.x * 16; o + 16 <= b1; o += 256 * 16) {
      uint4 v = buf[o / 16];
      asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));       // (keep the round trip: same values back)
      buf[o / 16] = v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    atomicMin(stats + 0, t0);
    atomicMax(stats + 1, t1);
    __threadfence();
    if (atomicAdd(stats + 2, 1ull) == (unsigned long long)gridDim.x - 1) {       // last workgroup out
      __threadfence();
      const unsigned long long a = atomicMin(stats + 0, ~0ull), b = atomicMax(stats + 1, 0ull);
      stats[3] += b - a;
      stats[4] += 1;
      stats[0] = ~0ull;
      stats[1] = 0ull;
      stats[2] = 0ull;
    }
  }
}
}  // namespace

extern "C" int ocr_comm_proxy(void* buf, size_t bytes, int workgroups, float link_gbps, void* stats_u64x8, void* stream) {
  if (!buf || bytes < 16 || workgroups < 1 || workgroups > 256 || !(link_gbps > 0.f) || ((uintptr_t)buf & 15)) return OCR_ERR_INVALID_ARG;
  const double sec_per_chunk = 2.0 * (double)kProxyChunk / ((double)link_gbps * 1e9);
  const unsigned long long ticks = (unsigned long long)(sec_per_chunk * 1e8 + 0.5);              // s_memrealtime: 100 MHz
  if (g_proxy_wgs > 0) workgroups = g_proxy_wgs;
  if (g_proxy_fat)
    hipLaunchKernelGGL(comm_proxy_kernel<true>, dim3(workgroups), dim3(256), 64 * 1024, static_cast<hipStream_t>(stream),
                       static_cast<uint4*>(buf), bytes, ticks, static_cast<unsigned long long*>(stats_u64x8));
  else
    hipLaunchKernelGGL(comm_proxy_kernel<false>, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint4*>(buf), bytes, ticks, static_cast<unsigned long long*>(stats_u64x8));
  return ocr_launch_status();
}

extern "C" int ocr_comm_proxy_set_footprint(int fat, int workgroups) {
  if (workgroups < 0 || workgroups > 256) return OCR_ERR_INVALID_ARG;
  static bool configured = false;
  if (fat && !configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(comm_proxy_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            64 * 1024) != hipSuccess)
      return OCR_ERR_HIP;
    configured = true;
  }
  g_proxy_fat = fat ? 1 : 0;
  g_proxy_wgs = workgroups;
  return OCR_OK;
}
