// Round 6: what is the ~6 us between a large kernel's end and its dependent successor's start on one HIP stream
// (profiles/r05_guest_pairs_trace.txt: conv -> BnFin 6 us, BnFin -> bn_relu 0 us)?  Pairs (A, tiny B) launched back to
// back on one stream, A varied: tiny | streams 256 MiB of plain / non-temporal stores | reads 256 MiB | holds the CU with
// 155 KB of LDS and a compute loop, no memory traffic | writes through 32 MiB only.  Read the gaps off
//   rocprofv3 --kernel-trace -d gpurun_out/gap -- scripts/_bin/gap_probe
// with scripts/gap_probe/gaps.py.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void b_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void a_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[1] += 1.f; }
__global__ __launch_bounds__(256) void a_store_plain(f4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = f4{1.f, 2.f, 3.f, (float)i};
}
__global__ __launch_bounds__(256) void a_store_nt(f4* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    __builtin_nontemporal_store(f4{1.f, 2.f, 3.f, (float)i}, p + i);
}
__global__ __launch_bounds__(256) void a_load(const f4* p, size_t n, float* out) {
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
  if (s[0] + s[1] + s[2] + s[3] == 12345.f) out[2] = 1.f;
}
__global__ __launch_bounds__(256) void a_compute_lds(float* out, int iters) {
  extern __shared__ float sm[];
  float v = threadIdx.x * 0.001f;
  for (int i = 0; i < iters; ++i) { sm[(threadIdx.x + i) & 1023] = v; v = v * 1.0001f + sm[(threadIdx.x * 7 + i) & 1023]; }
  if (v == 12345.f) out[3] = v;
}
int main() {
  const size_t bytes = 256u << 20, n = bytes / 16;
  f4* buf; float* small;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&small, 64)); CK(hipMemset(small, 0, 64)); CK(hipMemset(buf, 0, bytes));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(a_compute_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int R = 40;
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_tiny, dim3(1), dim3(64), 0, 0, small); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_store_plain, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_store_nt, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_load, dim3(2048), dim3(256), 0, 0, buf, n, small); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_compute_lds, dim3(1024), dim3(256), 155 * 1024, 0, small, 20000); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  // a small store (8 MiB: fits every L2) and a 32 MiB one
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_store_plain, dim3(512), dim3(256), 0, 0, buf, (size_t)(8u << 20) / 16); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_store_plain, dim3(2048), dim3(256), 0, 0, buf, (size_t)(32u << 20) / 16); hipLaunchKernelGGL(b_tiny, dim3(1), dim3(64), 0, 0, small); }
  // B large after A large (both directions of the conv <-> bn_relu pattern)
  for (int r = 0; r < R; ++r) { hipLaunchKernelGGL(a_store_nt, dim3(2048), dim3(256), 0, 0, buf, n); hipLaunchKernelGGL(a_compute_lds, dim3(1024), dim3(256), 155 * 1024, 0, small, 20000); }
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
