// Loaded latency of dependent wave-wide loads from per-wavefront regions of a large buffer (diagnostic; hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(64) chase(const double2* buf, size_t stride_d2, int iters, int span_d2, unsigned long long* out, double* sink, int lds_pad) {
  extern __shared__ char lds[];
  const double2* p = buf + (size_t)blockIdx.x * stride_d2;
  double acc = 0;
  unsigned idx = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
    const double2 v = p[(idx + (unsigned)(i * 4099)) % (unsigned)span_d2];  // a 1-KB row somewhere in the wavefront's region
    acc += v.x + v.y;
    idx = threadIdx.x + ((unsigned)__double2loint(acc) & 0u);  // dependency
  }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 12345.678) sink[0] = acc + lds[lds_pad];
}
int main(int argc, char** argv) {
  const int iters = 2000;
  for (int per_cu : {1, 4, 8, 11, 16}) {
    for (size_t region_kb : {110, 2900}) {
      const int waves = 256 * per_cu;
      const size_t stride_d2 = region_kb * 1024 / 16;
      double2* buf; unsigned long long* out; double* sink;
      hipMalloc(&buf, (size_t)waves * stride_d2 * 16);
      hipMemset(buf, 0, (size_t)waves * stride_d2 * 16);
      hipMalloc(&out, waves * 8); hipMalloc(&sink, 8);
      const int lds = 160 * 1024 / per_cu - 512;   // forces per_cu workgroups per CU
      hipFuncSetAttribute((const void*)chase, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL(chase, dim3(waves), dim3(64), lds, 0, buf, stride_d2, iters, (int)stride_d2 - 64, out, sink, 0);
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(waves);
      hipMemcpy(h.data(), out, waves * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += (double)v;
      printf("waves/CU %2d region %5zu KB per wave (%6.1f MB total): %.0f ns per dependent 1-KB load\n", per_cu, region_kb, waves * region_kb / 1024.0, s / waves / iters * 10.0);
      hipFree(buf); hipFree(out); hipFree(sink);
    }
  }
  return 0;
}
