// Co-residency probe: can a small "guest" wave share a CU with a matrix-core "host" wave that
// owns most of the register file, and what does each cost the other?  Standalone (no library): synthetic host
// kernels with the resource footprint of wgrad3_kernel<9,128> (198 VGPR + 256 AGPR, 139 KB LDS, one wave per SIMD,
// grid = one workgroup per CU) and of conv3x3_w4_kernel (256 + 256), guests that spin or stream HBM.  Every
// workgroup of both records (XCC, SE/SH/CU from HW_ID, start, end in 100 MHz ticks): co-residency is read off the
// records, not inferred from durations.
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/coresidency_probe scripts/coresidency_probe.hip
//   gpurun_out/coresidency_probe > gpurun_out/coresidency_probe.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg(4 | (31 << 11)); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15; }

// host: `iters` rounds of 64 MFMAs (16 independent accumulator quads x 4) per wave, optional LDS reads in between
#define HOST_KERNEL(NAME, VREG, AREG)                                                                   \
  __global__ __launch_bounds__(256) void NAME(int iters, int lds_reads, Rec* rec, float* sink) {        \
    extern __shared__ __attribute__((aligned(16))) char smem[];                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                      \
    asm volatile("v_mov_b32 " #VREG ", 0\n\tv_accvgpr_write_b32 " #AREG ", 0" ::: #VREG, #AREG);          \
    f32x4 acc[16];                                                                                       \
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};                                     \
    half8_t a, b;                                                                                        \
    for (int k = 0; k < 8; ++k) {                                                                        \
      a[k] = (_Float16)(0.001f * (float)((threadIdx.x * 7 + k * 3) % 97) - 0.05f);                       \
      b[k] = (_Float16)(0.002f * (float)((threadIdx.x * 5 + k * 11) % 89) - 0.09f);                      \
    }                                                                                                    \
    const uint4* lp = reinterpret_cast<const uint4*>(smem) + threadIdx.x;                                \
    uint4 junk = make_uint4(0, 0, 0, 0);                                                                 \
    for (int i = 0; i < iters; ++i) {                                                                    \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                    \
        _Pragma("unroll") for (int j = 0; j < 16; ++j)                                                   \
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[j], 0, 0, 0);                        \
        if (lds_reads) {                                                                                 \
          uint4 v = lp[(i * 4 + r) & 63];                                                                \
          junk.x ^= v.x;                                                                                 \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
    float s = (float)junk.x;                                                                             \
    for (int j = 0; j < 16; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];                     \
    if (s == 12345.678f) sink[threadIdx.x] = s;                                                          \
    __syncthreads();                                                                                     \
    if (threadIdx.x == 0) {                                                                              \
      Rec r;                                                                                             \
      r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw_id(); r.xcc = xcc_id();              \
      rec[blockIdx.x] = r;                                                                               \
    }                                                                                                    \
  }

HOST_KERNEL(host_456, v197, a255)     // wgrad3_kernel<9,128>'s footprint: 200 + 256
HOST_KERNEL(host_512, v255, a255)     // conv3x3_w4_kernel's: the whole file
HOST_KERNEL(host_448, v191, a255)     // VERDICT item 1c's target for conv3x3_w4
HOST_KERNEL(host_384, v127, a255)
HOST_KERNEL(host_256, v127, a127)

// guest A: hold a CU slot for `ticks` (100 MHz), no memory traffic
__global__ __launch_bounds__(256) void guest_spin(unsigned long long ticks, Rec* rec) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  __syncthreads();
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw_id(); r.xcc = xcc_id();
    rec[blockIdx.x] = r;
  }
}

// guest B: HBM stream (read + write, the shape of a BN-backward apply pass), U 16-byte loads in flight per lane.
// Persistent grid: workgroup g walks chunks g, g + grid, ...
template <int U, int THREADS>
__global__ __launch_bounds__(THREADS) void guest_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, Rec* rec) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const size_t chunk = (size_t)THREADS * U;
  for (size_t base = (size_t)blockIdx.x * chunk; base < n16; base += (size_t)gridDim.x * chunk) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * THREADS + threadIdx.x;
      v[u] = src[i < n16 ? i : n16 - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t i = base + (size_t)u * THREADS + threadIdx.x;
      v[u].x += 1u;
      if (i < n16) dst[i] = v[u];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Rec r;
    r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw_id(); r.xcc = xcc_id();
    rec[blockIdx.x] = r;
  }
}

struct Launch {
  const char* name;
  void (*fn)(int, int, Rec*, float*);
};

static std::vector<Rec> fetch(Rec* d, int n) {
  std::vector<Rec> h(n);
  CK(hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost));
  return h;
}
static unsigned cu_key(const Rec& r) { return (r.xcc << 8) | ((r.hw >> 8) & 0xff); }   // XCC | SE/SH/CU bits of HW_ID

// how many host workgroups ran on a CU WHILE a guest workgroup was resident there (intervals intersect by > 10 % of the host's)
static void coresidency(const std::vector<Rec>& host, const std::vector<Rec>& guest, int* shared, int* host_cus, int* guest_cus, int* both_cus) {
  std::map<unsigned, std::vector<const Rec*>> g;
  for (const Rec& r : guest) g[cu_key(r)].push_back(&r);
  std::map<unsigned, int> hc;
  int sh = 0;
  for (const Rec& h : host) {
    hc[cu_key(h)]++;
    auto it = g.find(cu_key(h));
    if (it == g.end()) continue;
    for (const Rec* q : it->second) {
      const unsigned long long lo = std::max(h.t0, q->t0), hi = std::min(h.t1, q->t1);
      if (hi > lo && (hi - lo) * 10 > (h.t1 - h.t0)) { ++sh; break; }
    }
  }
  int both = 0;
  for (auto& kv : hc) both += g.count(kv.first) ? 1 : 0;
  *shared = sh; *host_cus = (int)hc.size(); *guest_cus = (int)g.size(); *both_cus = both;
}

static double span_ms(const std::vector<Rec>& v) {
  unsigned long long lo = ~0ull, hi = 0;
  for (const Rec& r : v) { lo = std::min(lo, r.t0); hi = std::max(hi, r.t1); }
  return (double)(hi - lo) * 1e-5;
}
static double mean_wg_ms(const std::vector<Rec>& v) {
  double s = 0;
  for (const Rec& r : v) s += (double)(r.t1 - r.t0);
  return s / v.size() * 1e-5;
}

int main(int argc, char** argv) {
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  fprintf(stderr, "device %s, %d CUs\n", prop.name, cus);
  hipStream_t sa, sb, sb_hi;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  int lo_p, hi_p;
  CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
  CK(hipStreamCreateWithPriority(&sb_hi, hipStreamNonBlocking, hi_p));
  const int MAXWG = 8192;
  Rec *rec_h, *rec_g;
  CK(hipMalloc(&rec_h, sizeof(Rec) * MAXWG * 16));
  CK(hipMalloc(&rec_g, sizeof(Rec) * MAXWG));
  float* sink;
  CK(hipMalloc(&sink, 4096));
  const size_t copy_bytes = 1ull << 30;
  uint4 *src, *dst;
  CK(hipMalloc(&src, copy_bytes));
  CK(hipMalloc(&dst, copy_bytes));
  CK(hipMemset(src, 1, copy_bytes));
  CK(hipMemset(dst, 0, copy_bytes));
  const size_t n16 = copy_bytes / 16;

  const Launch hosts[] = {{"host_456", host_456}, {"host_512", host_512}, {"host_448", host_448}, {"host_384", host_384}, {"host_256", host_256}};
  for (const Launch& h : hosts)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(h.fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  hipEvent_t e0, e1, g0, g1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
  const int NL = 8;              // host launches back to back
  // calibrate iters so that one host launch takes ~300 us (wgrad3's launch)
  int iters = 2000;
  {
    host_456<<<cus, 256, 139 * 1024, sa>>>(iters, 0, rec_h, sink);
    CK(hipStreamSynchronize(sa));
    CK(hipEventRecord(e0, sa));
    host_456<<<cus, 256, 139 * 1024, sa>>>(iters, 0, rec_h, sink);
    CK(hipEventRecord(e1, sa));
    CK(hipStreamSynchronize(sa));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    iters = (int)(iters * 0.3f / ms);
    fprintf(stderr, "calibrated iters = %d (%.3f ms at 2000)\n", iters, ms);
  }

  printf("{\"device\": \"%s\", \"cus\": %d, \"host_iters\": %d, \"cases\": [\n", prop.name, cus, iters);
  bool first = true;

  auto run_case = [&](const char* label, const Launch& h, int host_grid, size_t host_lds, int lds_reads,
                      const char* guest_kind, int guest_grid, hipStream_t gs, bool guest_first) {
    // host alone
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, sa));
      for (int l = 0; l < NL; ++l) h.fn<<<host_grid, 256, host_lds, sa>>>(iters, lds_reads, rec_h + (size_t)l * MAXWG, sink);
      CK(hipEventRecord(e1, sa));
      CK(hipStreamSynchronize(sa));
    }
    float host_alone;
    CK(hipEventElapsedTime(&host_alone, e0, e1));
    std::vector<Rec> ha = fetch(rec_h, host_grid);
    const double host_wg_alone = mean_wg_ms(ha);
    auto launch_guest = [&](hipStream_t s) {
      if (!strcmp(guest_kind, "spin")) guest_spin<<<guest_grid, 256, 0, s>>>(100000ull /* 1 ms */, rec_g);
      else if (!strcmp(guest_kind, "copy_u2")) guest_copy<2, 256><<<guest_grid, 256, 0, s>>>(src, dst, n16, rec_g);
      else if (!strcmp(guest_kind, "copy_u4")) guest_copy<4, 256><<<guest_grid, 256, 0, s>>>(src, dst, n16, rec_g);
      else if (!strcmp(guest_kind, "copy_u8")) guest_copy<8, 256><<<guest_grid, 256, 0, s>>>(src, dst, n16, rec_g);
      else if (!strcmp(guest_kind, "copy_u4_w1")) guest_copy<4, 64><<<guest_grid, 64, 0, s>>>(src, dst, n16, rec_g);
      else if (!strcmp(guest_kind, "copy_u8_w1")) guest_copy<8, 64><<<guest_grid, 64, 0, s>>>(src, dst, n16, rec_g);
    };
    // guest alone
    float guest_alone = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(g0, gs));
      launch_guest(gs);
      CK(hipEventRecord(g1, gs));
      CK(hipStreamSynchronize(gs));
      CK(hipEventElapsedTime(&guest_alone, g0, g1));
    }
    // together
    CK(hipDeviceSynchronize());
    if (guest_first) {
      CK(hipEventRecord(g0, gs));
      launch_guest(gs);
      CK(hipEventRecord(g1, gs));
    }
    CK(hipEventRecord(e0, sa));
    for (int l = 0; l < NL; ++l) {
      h.fn<<<host_grid, 256, host_lds, sa>>>(iters, lds_reads, rec_h + (size_t)l * MAXWG, sink);
      if (!guest_first && l == 0) {
        CK(hipEventRecord(g0, gs));
        launch_guest(gs);
        CK(hipEventRecord(g1, gs));
      }
    }
    CK(hipEventRecord(e1, sa));
    CK(hipDeviceSynchronize());
    float host_with, guest_with;
    CK(hipEventElapsedTime(&host_with, e0, e1));
    CK(hipEventElapsedTime(&guest_with, g0, g1));
    std::vector<Rec> g = fetch(rec_g, guest_grid);
    int shared_total = 0, hc = 0, gc = 0, bc = 0, overlapped_launches = 0;
    double host_wg_with = 0;
    int host_wg_n = 0;
    unsigned long long glo = ~0ull, ghi = 0;
    for (const Rec& r : g) { glo = std::min(glo, r.t0); ghi = std::max(ghi, r.t1); }
    std::string spans = "[";
    unsigned long long first_lo = 0;
    for (int l = 0; l < NL; ++l) {
      std::vector<Rec> hl = fetch(rec_h + (size_t)l * MAXWG, host_grid);
      unsigned long long lo = ~0ull, hi = 0;
      for (const Rec& r : hl) { lo = std::min(lo, r.t0); hi = std::max(hi, r.t1); }
      if (l == 0) first_lo = lo;
      char tmp[96];
      snprintf(tmp, sizeof tmp, "%s[%.3f, %.3f]", l ? ", " : "", (double)((long long)lo - (long long)first_lo) * 1e-5, (double)(hi - lo) * 1e-5);
      spans += tmp;
      if (hi <= glo || lo >= ghi) continue;          // this launch did not overlap the guest in time
      ++overlapped_launches;
      int sh;
      coresidency(hl, g, &sh, &hc, &gc, &bc);
      shared_total += sh;
      host_wg_with += mean_wg_ms(hl) * hl.size();
      host_wg_n += (int)hl.size();
    }
    printf("%s {\"case\": \"%s\", \"host\": \"%s\", \"host_grid\": %d, \"host_lds\": %zu, \"lds_reads\": %d, \"guest\": \"%s\", \"guest_grid\": %d, "
           "\"guest_first\": %s, \"host_alone_ms\": %.3f, \"host_with_ms\": %.3f, \"guest_alone_ms\": %.3f, \"guest_with_ms\": %.3f, "
           "\"host_wg_alone_ms\": %.4f, \"host_wg_with_ms\": %.4f, \"launches_overlapping_guest\": %d, "
           "\"host_wgs_sharing_a_cu_with_a_resident_guest\": %d, \"host_wgs_in_those_launches\": %d, \"guest_cus\": %d, \"cus_used_by_both\": %d, "
           "\"guest_start_end_ms\": [%.3f, %.3f], \"host_launch_start_span_ms\": %s]}",
           first ? " " : ",\n", label, h.name, host_grid, host_lds, lds_reads, guest_kind, guest_grid, guest_first ? "true" : "false",
           host_alone, host_with, guest_alone, guest_with, host_wg_alone, host_wg_n ? host_wg_with / host_wg_n : 0.0,
           overlapped_launches, shared_total, host_wg_n, gc, bc,
           (double)((long long)glo - (long long)first_lo) * 1e-5, (double)((long long)ghi - (long long)first_lo) * 1e-5, spans.c_str());
    first = false;
    fflush(stdout);
  };

  const size_t L139 = 139 * 1024;
  // 1. the question as VERDICT asks it: wgrad3's footprint + a small spinning guest
  run_case("spin24_guest_first", hosts[0], cus, L139, 0, "spin", 24, sb, true);
  run_case("spin24_host_first", hosts[0], cus, L139, 0, "spin", 24, sb, false);
  run_case("spin256_guest_first", hosts[0], cus, L139, 0, "spin", cus, sb, true);
  run_case("spin24_hi_prio", hosts[0], cus, L139, 0, "spin", 24, sb_hi, true);
  // 2. the whole file: no room by construction
  run_case("spin24_vs_512", hosts[1], cus, L139, 0, "spin", 24, sb, true);
  run_case("spin24_vs_448", hosts[2], cus, L139, 0, "spin", 24, sb, true);
  // 3. a grid of several rounds (conv3x3_w4: 1024 tiles)
  run_case("spin24_vs_456_4rounds", hosts[0], 4 * cus, L139, 0, "spin", 24, sb, true);
  run_case("spin24_vs_456_4rounds_host_first", hosts[0], 4 * cus, L139, 0, "spin", 24, sb, false);
  run_case("spin256_vs_456_4rounds", hosts[0], 4 * cus, L139, 0, "spin", cus, sb, true);
  run_case("copy_u4_256wg_vs_456_4rounds", hosts[0], 4 * cus, L139, 0, "copy_u4", cus, sb, false);
  run_case("spin24_vs_512_4rounds", hosts[1], 4 * cus, L139, 0, "spin", 24, sb, true);
  // 4. HBM-streaming guests beside the 456-register host: rate kept by each
  run_case("copy_u4_256wg", hosts[0], cus, L139, 0, "copy_u4", cus, sb, false);
  run_case("copy_u8_256wg", hosts[0], cus, L139, 0, "copy_u8", cus, sb, false);
  run_case("copy_u2_512wg", hosts[0], cus, L139, 0, "copy_u2", 2 * cus, sb, false);
  run_case("copy_u4_512wg", hosts[0], cus, L139, 0, "copy_u4", 2 * cus, sb, false);
  run_case("copy_u4_w1_1024wg", hosts[0], cus, L139, 0, "copy_u4_w1", 4 * cus, sb, false);
  run_case("copy_u8_w1_1024wg", hosts[0], cus, L139, 0, "copy_u8_w1", 4 * cus, sb, false);
  run_case("copy_u4_256wg_guest_first", hosts[0], cus, L139, 0, "copy_u4", cus, sb, true);
  run_case("copy_u4_2048wg_guest_first", hosts[0], cus, L139, 0, "copy_u4", 8 * cus, sb, true);
  run_case("copy_u4_256wg_lds_reads", hosts[0], cus, L139, 1, "copy_u4", cus, sb, false);
  // 5. smaller hosts: more room for guests
  run_case("copy_u8_512wg_vs_384", hosts[3], cus, L139, 0, "copy_u8", 2 * cus, sb, false);
  run_case("copy_u8_1024wg_vs_256regs_64k", hosts[4], cus, 64 * 1024, 0, "copy_u8", 4 * cus, sb, false);
  // 6. the whole file again, with a streaming guest: what time-slicing looks like
  run_case("copy_u4_256wg_vs_512", hosts[1], cus, L139, 0, "copy_u4", cus, sb, false);
  printf("\n]}\n");
  return 0;
}
