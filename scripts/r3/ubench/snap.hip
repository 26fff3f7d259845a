// Write a few KB with plain stores, compute for a while, read them back with several loads in flight: the pattern of a node snapshot
// (diagnostic; hipcc --offload-arch=gfx950 -O3).  Reports cycles of the read-back per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) snap(double2* buf, size_t stride_d2, int iters, int nload, int work, unsigned long long* out, double* sink) {
  extern __shared__ char lds[];
  double2* p = buf + (size_t)blockIdx.x * stride_d2;
  double acc = threadIdx.x;
  unsigned long long total = 0;
  for (int it = 0; it < iters; it++) {
    double2* q = p + (size_t)(it % 4) * 704;  // level slot: 11264 B apart
    for (int j = 0; j < nload; j++) q[j * 64 + threadIdx.x] = make_double2(acc + j, acc - j);
    for (int w = 0; w < work; w++) acc = acc * 1.0000001 + 0.5;  // dependent FP64 chain
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    const unsigned long long t0 = clock64();
    double2 v[8];
    for (int j = 0; j < 8; j++) v[j] = q[(j < nload ? j : 0) * 64 + threadIdx.x];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    total += clock64() - t0;
    double s = 0;
    for (int j = 0; j < 8; j++) s += v[j].x + v[j].y;
    acc += s * 1e-30;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = total;
  if (acc == 12345.678) sink[0] = acc + lds[0];
}
int main() {
  const int iters = 200;
  for (int per_cu : {1, 8, 11}) for (int nload : {1, 5}) for (int work : {200, 20000}) for (size_t stride : {112640ul, 112640ul + 4352ul}) {
    const int waves = 256 * per_cu;
    const size_t stride_d2 = stride / 16;
    double2* buf; unsigned long long* out; double* sink;
    (void)hipMalloc(&buf, (size_t)waves * stride); (void)hipMemset(buf, 0, (size_t)waves * stride);
    (void)hipMalloc(&out, waves * 8); (void)hipMalloc(&sink, 8);
    const int lds = 160 * 1024 / per_cu - 512;
    (void)hipFuncSetAttribute((const void*)snap, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(snap, dim3(waves), dim3(64), lds, 0, buf, stride_d2, iters, nload, work, out, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), out, waves * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("waves/CU %2d, %d KB written, %5d flops between, stride %6zu: read-back %.0f cycles\n", per_cu, nload, work, stride, s / waves / iters);
    (void)hipFree(buf); (void)hipFree(out); (void)hipFree(sink);
  }
  return 0;
}
