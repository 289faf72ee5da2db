// Round 6, VERDICT r5 item 1 (a): stand-alone probe of a FUSED Winograd F(2x2, 3x3) convolution on conv4_2's forward shape
// (32 x 64 x 64, 512 -> 512 channels, f16, f32 accumulate; the direct kernel conv3x3_w4 takes 412 us there).  Keep-going
// criterion: <= 300 us.
//
// What decides the question before any tuning: the 16 transform positions quarter the register tile.  A CU holds 65 536
// f32 accumulators (4 waves x 256 registers); the direct kernel spends them on 256 pixels x 256 couts, the Winograd form on
// 16 positions x T tiles x C couts with T * C = 4096, so every staged operand byte feeds a quarter of the outputs, and the
// operand stream L2 -> LDS grows from 2.75 GB (direct) to 4.7-5.7 GB per launch while the MFMA count falls 2.25 x.  The
// input transform (2 packed adds per transformed value, once per tile and channel, amortised over C couts only) costs a
// wave T * 4 issue cycles per 16 channels next to T * C / 8 MFMA cycles: C = 32 makes it equal to the MFMA time, so C >= 64,
// T <= 64, 5.65 GB.
//
// This file MEASURES that bound instead of arguing it: a skeleton with the data movement, the instruction mix and the
// resource footprint of the fused kernel in its best-case form — every LDS access lane-linear (conflict-free), every DMA a
// full 1 KiB wave piece, weights pre-packed [k16][position][cout][16] so that each slab is contiguous — in stages:
//   mode 0  operand stream only: halo chunks + transformed-weight slabs by LDS-DMA through the ring, barriers as in the loop
//   mode 1  + the fragment reads (8 raw + ... per tile block, B slabs) and the 16 MFMAs per wave and 16 channels
//   mode 2  + the input transform (32 v_pk_add_f16 per tile block) on the values read
//   mode 3  + the output transform through LDS and the f16 stores (the whole skeleton)
// Results are data-dependent on random inputs (no dead code) but are NOT a convolution: the skeleton's addressing is the
// pattern, not the map.  A skeleton that misses 300 us ends the experiment; one that makes it would be the scaffold.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/_bin/wino_probe scripts/winograd/wino_probe.hip
//   scripts/_bin/wino_probe > gpurun_out/wino_probe.json
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                         \
    }                                                                                  \
  } while (0)

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int N = 32, H = 64, W = 64, CIN = 512, COUT = 512;

// ---- geometry of one workgroup: TY x TX tiles of 2 x 2 outputs, C couts, all 16 positions; wave w = transform row xi = w
template <int TY, int TX, int C>
struct Geo {
  static constexpr int T = TY * TX;                      // tiles
  static constexpr int HY = 2 * TY + 2, HX = 2 * TX + 2;  // halo
  static constexpr int HSLOTS = HY * HX * 4;             // 16-byte slots of one 32-channel halo chunk
  static constexpr int HR = (HSLOTS + 255) / 256;        // DMA rounds
  static constexpr int HBYTES = HR * 256 * 16;
  static constexpr int USLAB = 16 * C * 32;              // bytes of one k16 slab of transformed weights
  static constexpr int UR = USLAB / 4096;                // DMA rounds
  static constexpr int RING = 3;
  static constexpr int LDS = 2 * HBYTES + RING * USLAB;
  static constexpr int TB = T / 32, CB = C / 32;         // 32-tile blocks, 32-cout blocks
  static_assert(TB * CB * 4 * 16 == 256, "a wave's accumulators fill 256 registers");
};

template <int TY, int TX, int C, int MODE>
__global__ __launch_bounds__(256) void wino_skel(const half_t* __restrict__ x, const half_t* __restrict__ u,
                                                 half_t* __restrict__ y, unsigned long long* __restrict__ stamps) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass only needs the launch stub; hipcc drops it when the body holds device builtins it cannot type)
  typedef Geo<TY, TX, C> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const ubuf = smem + 2 * G::HBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (image, pixel tile, cout tile): cout tiles of one pixel tile adjacent (one XCD's L2 serves the halo)
  constexpr int PX = (H / (2 * TY)) * (W / (2 * TX));
  constexpr int NT = COUT / C;
  int bid = blockIdx.x;
  bid = (bid & 7) * (int)(gridDim.x >> 3) + (bid >> 3);
  const int nt = bid % NT;
  const int mt = bid / NT;
  const int img = mt / PX, pt = mt % PX;
  const int ty0 = (pt / (W / (2 * TX))) * 2 * TY - 1, tx0 = (pt % (W / (2 * TX))) * 2 * TX - 1;
  const int co0 = nt * C;

  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<half_t*>(x) + (size_t)img * H * W * CIN, 0, H * W * CIN * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(u), 0, 16 * COUT * CIN * 2, 0x00020000);
  unsigned hvo[G::HR];
#pragma unroll
  for (int r = 0; r < G::HR; ++r) {
    const int idx = r * 256 + tid, hp = idx >> 2, sl = idx & 3;
    const int hy = hp / G::HX, hx = hp - hy * G::HX;
    const int iy = ty0 + hy, ix = tx0 + hx;
    hvo[r] = (hp < G::HY * G::HX && iy >= 0 && iy < H && ix >= 0 && ix < W) ? (unsigned)(((iy * W + ix) * CIN + sl * 8) * 2) : OOB;
  }
  // weights [k16][pos][cout][16]: one DMA round = 4 KiB = 4096 / (C * 32) positions' slabs of this cout tile
  constexpr int PPR = 4096 / (C * 32) > 0 ? 4096 / (C * 32) : 1;      // positions per round (C = 64: 2, C = 32: 4)
  const unsigned uvo = (unsigned)((tid / (256 / PPR)) * COUT * 32 + (tid % (256 / PPR)) * 16);

  auto dma_halo = [&](int step, int hb) {
#pragma unroll
    for (int r = 0; r < G::HR; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          xrs, (__attribute__((address_space(3))) void*)(smem + hb * G::HBYTES + (r * 256 + wave * 64) * 16), 16, hvo[r],
          step * 64, 0, 0);
  };
  auto dma_u = [&](int kk, int ring) {
#pragma unroll
    for (int r = 0; r < G::UR; ++r)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          urs, (__attribute__((address_space(3))) void*)(ubuf + ring * G::USLAB + (r * 256 + wave * 64) * 16), 16, uvo,
          ((kk * 16 + r * PPR) * COUT + co0) * 32, 0, 0);
  };

  f32x16 acc[4][G::TB][G::CB];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int a = 0; a < G::TB; ++a)
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][a][b][e] = 0.f;

  constexpr int NSTEP = CIN / 32;      // k32 steps: one halo chunk, two weight slabs
  unsigned long long t0 = 0, c0 = 0;
  if (stamps && tid == 0) { t0 = __builtin_amdgcn_s_memrealtime(); c0 = __builtin_amdgcn_s_memtime(); }
  dma_halo(0, 0);
  dma_u(0, 0);
  dma_u(1, 1);
  for (int s = 0; s < NSTEP; ++s) {
    // stage s has landed for every wave
    __builtin_amdgcn_s_waitcnt(0x0f70);      // vmcnt(0) (expcnt / lgkmcnt untouched)
    __syncthreads();
    if (s + 1 < NSTEP) dma_halo(s + 1, (s + 1) & 1);
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int kk = 2 * s + hf;
      if (kk + 2 < 2 * NSTEP) dma_u(kk + 2, (kk + 2) % G::RING);
      if (hf == 1) {                                           // the second slab of this step was issued one half-step ago
        __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();
      }
      if (MODE >= 1) {
        const char* hb = smem + (s & 1) * G::HBYTES;
        const char* ub = ubuf + (kk % G::RING) * G::USLAB;
        half8_t bf[4][G::CB];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int b = 0; b < G::CB; ++b)
            bf[p][b] = *reinterpret_cast<const half8_t*>(ub + (((wave * 4 + p) * G::CB + b) * 64 + lane) * 16);
#pragma unroll
        for (int a = 0; a < G::TB; ++a) {
          half8_t raw[8];
#pragma unroll
          for (int q = 0; q < 8; ++q)        // 2 halo rows x 4 columns of this lane's tile, 8 channels: lane-linear stand-in
            raw[q] = *reinterpret_cast<const half8_t*>(hb + ((((a * 8 + q) * 2 + hf) * 64 + lane) * 16) % G::HBYTES);
          half8_t v[4];
          if (MODE >= 2) {
            // row xi of B^T d (one add per column), then the four nu combinations: 32 packed adds
            half8_t t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = (wave == 1) ? raw[j] + raw[4 + j] : raw[j] - raw[4 + j];
            v[0] = t[0] - t[2];
            v[1] = t[1] + t[2];
            v[2] = t[2] - t[1];
            v[3] = t[1] - t[3];
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              // no transform: keep all eight reads alive with a bitwise fold the compiler cannot drop (2 ops per register
              // would be the transform's cost; v_xor of whole registers is what is left here: 4 per fragment)
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              u32x4 z = __builtin_bit_cast(u32x4, raw[j]) ^ __builtin_bit_cast(u32x4, raw[4 + j]);
              v[j] = __builtin_bit_cast(half8_t, z);
            }
          }
#pragma unroll
          for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int b = 0; b < G::CB; ++b)
              acc[p][a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v[p], bf[p][b], acc[p][a][b], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);      // one tile block's reads in flight at a time (register room)
        }
      }
    }
  }
  if (stamps && tid == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime() - c0;
  }
  __syncthreads();
  if (MODE >= 3) {
    // output transform: wave xi's share Z[q] = sum_nu M[xi][nu] A[nu][q] (2 values per tile and cout) -> LDS (f32), summed
    // over xi into the 2 x 2 outputs, stored as f16
    float* zl = reinterpret_cast<float*>(smem);        // [wave][q][TB][CB][16 regs][64 lanes]
#pragma unroll
    for (int a = 0; a < G::TB; ++a)
#pragma unroll
      for (int b = 0; b < G::CB; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float z0 = acc[0][a][b][e] + acc[1][a][b][e] + acc[2][a][b][e];
          const float z1 = acc[1][a][b][e] - acc[2][a][b][e] - acc[3][a][b][e];
          zl[((((wave * 2 + 0) * G::TB + a) * G::CB + b) * 16 + e) * 64 + lane] = z0;
          zl[((((wave * 2 + 1) * G::TB + a) * G::CB + b) * 16 + e) * 64 + lane] = z1;
        }
    __syncthreads();
    // 4 T C outputs = 16384 values, 64 per thread: thread -> (pixel, 8 couts) pieces of 16 bytes
    constexpr int PER = G::TB * G::CB * 16 * 64;       // values per (wave, q)
    half_t* const yo = y + ((size_t)img * H * W) * COUT;
    for (int i = tid; i < G::T * 4 * C / 8; i += 256) {
      const int c8 = i % (C / 8), px = i / (C / 8);
      half8_t o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = (px * C + c8 * 8 + e) % PER;
        const int q = px & 1, pr = (px >> 1) & 1;
        const float s0 = zl[(0 * 2 + q) * PER + k], s1 = zl[(1 * 2 + q) * PER + k], s2 = zl[(2 * 2 + q) * PER + k],
                    s3 = zl[(3 * 2 + q) * PER + k];
        o[e] = (half_t)(pr ? (s1 - s2 - s3) : (s0 + s1 + s2));
      }
      const int ty = px / (2 * TX) % (2 * TY), tx = px % (2 * TX);
      const int oy = ty0 + 1 + ty, ox = tx0 + 1 + tx;
      *reinterpret_cast<half8_t*>(yo + ((size_t)(oy * W + ox)) * COUT + co0 + c8 * 8) = o;
    }
  } else if (MODE >= 1) {
    float sink = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int a = 0; a < G::TB; ++a)
#pragma unroll
        for (int b = 0; b < G::CB; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) sink += acc[p][a][b][e];
    if (sink == 12345.678f) y[blockIdx.x] = (half_t)sink;
  } else {
    // mode 0: the landed bytes must be observable
    const unsigned v = *reinterpret_cast<const unsigned*>(smem + tid * 16) ^ *reinterpret_cast<const unsigned*>(ubuf + tid * 16);
    if (v == 0x12345678u) y[blockIdx.x] = (half_t)1.f;
  }
#endif
}

template <int TY, int TX, int C, int MODE>
static void run(const char* name, const half_t* x, const half_t* u, half_t* y, unsigned long long* stamps, bool last) {
  typedef Geo<TY, TX, C> G;
  void (*kern)(const half_t*, const half_t*, half_t*, unsigned long long*) = wino_skel<TY, TX, C, MODE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int grid = N * (H / (2 * TY)) * (W / (2 * TX)) * (COUT / C);
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // >= 2 s of back-to-back launches first (the clock settles under load), then 50 timed launches
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)G::LDS, 0, x, u, y, (unsigned long long*)nullptr);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  float ms = 0.f;
  int warm = 0;
  do {
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)G::LDS, 0, x, u, y, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    warm += 200;
  } while (ms < 2000.f && warm < 20000);
  CK(hipEventRecord(e0));
  const int reps = 50;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)G::LDS, 0, x, u, y, (unsigned long long*)nullptr);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1000.0 / reps;
  // one stamped launch right behind them: in-kernel clock = shader cycles / 100 MHz ticks
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), (size_t)G::LDS, 0, x, u, y, stamps);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> st(2 * grid);
  CK(hipMemcpy(st.data(), stamps, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  for (int i = 0; i < grid; ++i)
    if (st[2 * i]) ghz.push_back((double)st[2 * i + 1] / (double)st[2 * i] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  const double halo = (double)G::HY * G::HX * CIN * 2 * grid, wts = 16.0 * C * CIN * 2 * grid;
  const double mfma_flop = 2.0 * 16 * (double)G::T * C * CIN * grid;
  printf("  {\"name\": \"%s\", \"tiles\": \"%dx%d\", \"couts\": %d, \"mode\": %d, \"grid\": %d, \"lds_bytes\": %d, \"vgpr\": %d, "
         "\"us\": %.1f, \"operand_stream_GB\": %.3f, \"operand_stream_TBps\": %.2f, \"mfma_TFLOPs_issued\": %.1f, "
         "\"direct_equivalent_frac_of_2.5PF\": %.3f, \"in_kernel_GHz_median\": %.3f}%s\n",
         name, TY, TX, C, MODE, grid, G::LDS, fa.numRegs, us, (halo + wts) / 1e9, (halo + wts) / us / 1e6,
         MODE >= 1 ? mfma_flop / us / 1e6 : 0.0, 2.0 * N * H * W * 9.0 * CIN * COUT / (us * 1e-6) / 2.5e15,
         ghz.empty() ? 0.0 : ghz[ghz.size() / 2], last ? "" : ",");
  fflush(stdout);
}

int main() {
  const size_t nx = (size_t)N * H * W * CIN, nu = (size_t)16 * COUT * CIN, ny = (size_t)N * H * W * COUT;
  std::vector<half_t> hx(nx), hu(nu);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) / 16777216.f) * 2.f - 1.f; };
  for (auto& v : hx) v = (half_t)(rnd() * 1.7f);
  for (auto& v : hu) v = (half_t)(rnd() * 0.05f);
  half_t *x, *u, *y;
  unsigned long long* stamps;
  CK(hipMalloc(&x, nx * 2));
  CK(hipMalloc(&u, nu * 2));
  CK(hipMalloc(&y, ny * 2));
  CK(hipMalloc(&stamps, sizeof(unsigned long long) * 2 * 8192));
  CK(hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(u, hu.data(), nu * 2, hipMemcpyHostToDevice));
  CK(hipMemset(y, 0, ny * 2));
  printf("{\"shape\": \"conv4_2 forward: 32 x 64 x 64, 512 -> 512, 3x3\", \"direct_kernel_us\": 412, \"keep_going_if_us_le\": 300,\n \"rows\": [\n");
  run<8, 8, 64, 0>("T64_C64_stream_only", x, u, y, stamps, false);
  run<8, 8, 64, 1>("T64_C64_stream_reads_mfma", x, u, y, stamps, false);
  run<8, 8, 64, 2>("T64_C64_plus_input_transform", x, u, y, stamps, false);
  run<8, 8, 64, 3>("T64_C64_whole_skeleton", x, u, y, stamps, false);
  run<8, 16, 32, 0>("T128_C32_stream_only", x, u, y, stamps, false);
  run<8, 16, 32, 1>("T128_C32_stream_reads_mfma", x, u, y, stamps, false);
  run<8, 16, 32, 2>("T128_C32_plus_input_transform", x, u, y, stamps, false);
  run<8, 16, 32, 3>("T128_C32_whole_skeleton", x, u, y, stamps, true);
  printf(" ]}\n");
  return 0;
}
