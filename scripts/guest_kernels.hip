// Guest kernels for scripts/guest_probe.py (measurement aid, not part of the product library): small-footprint waves that
// hold a CU slot (spin) or stream HBM (copy) on a second stream while the library's own kernels run on the first.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o scripts/_bin/libguest.so scripts/guest_kernels.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg(4 | (31 << 11)); }
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15; }

__global__ __launch_bounds__(256) void guest_spin_kernel(unsigned long long ticks, Rec* rec) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  __syncthreads();
  if (threadIdx.x == 0 && rec) {
    Rec r;
    r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw_id(); r.xcc = xcc_id();
    rec[blockIdx.x] = r;
  }
}

template <int U, int THREADS>
__global__ __launch_bounds__(THREADS) void guest_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16, Rec* rec) {
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
  if (threadIdx.x == 0 && rec) {
    Rec r;
    r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw_id(); r.xcc = xcc_id();
    rec[blockIdx.x] = r;
  }
}

extern "C" int guest_spin(unsigned long long ticks, int grid, void* rec, void* stream) {
  guest_spin_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(ticks, (Rec*)rec);
  return (int)hipGetLastError();
}
extern "C" int guest_copy(const void* src, void* dst, size_t bytes, int grid, int u, int threads, void* rec, void* stream) {
  const size_t n16 = bytes / 16;
  hipStream_t s = (hipStream_t)stream;
  if (threads == 256) {
    if (u == 2) guest_copy_kernel<2, 256><<<grid, 256, 0, s>>>((const uint4*)src, (uint4*)dst, n16, (Rec*)rec);
    else if (u == 4) guest_copy_kernel<4, 256><<<grid, 256, 0, s>>>((const uint4*)src, (uint4*)dst, n16, (Rec*)rec);
    else guest_copy_kernel<8, 256><<<grid, 256, 0, s>>>((const uint4*)src, (uint4*)dst, n16, (Rec*)rec);
  } else {
    if (u == 4) guest_copy_kernel<4, 64><<<grid, 64, 0, s>>>((const uint4*)src, (uint4*)dst, n16, (Rec*)rec);
    else guest_copy_kernel<8, 64><<<grid, 64, 0, s>>>((const uint4*)src, (uint4*)dst, n16, (Rec*)rec);
  }
  return (int)hipGetLastError();
}
