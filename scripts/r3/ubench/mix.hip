// Global-load latency of a wavefront while the CU's other wavefronts (and itself, between loads) run LDS sweeps / FP64 chains / both
// (diagnostic).  mode 0: nothing between loads, 1: LDS read sweeps, 2: FP64 chain, 3: atomics on one global counter
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) mix(const double2* buf, size_t stride_d2, int iters, int mode, int work, unsigned long long* out, double* sink, unsigned* counter) {
  extern __shared__ double lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = i;
  const double2* p = buf + (size_t)blockIdx.x * stride_d2;
  double acc = threadIdx.x;
  double priv[64];
  for (int i = 0; i < 64; i++) priv[i] = i + acc;
  double2* wr = const_cast<double2*>(buf) + (size_t)gridDim.x * stride_d2 + (size_t)blockIdx.x * 65536;  // 1 MB per wave of streamed writes
  unsigned long long total = 0;
  for (int it = 0; it < iters; it++) {
    if (mode == 1 || mode == 4) for (int w = 0; w < work; w++) acc += lds[(threadIdx.x * 25 + w * 7 + ((unsigned)__double2loint(acc) & 1u)) & 1023];
    if (mode == 2 || mode == 4) for (int w = 0; w < work; w++) acc = acc * 1.0000001 + 0.5;
    if (mode == 3 && threadIdx.x == 0 && (it & 63) == 0) acc += atomicAdd(counter, 1u) & 1u;
    if (mode == 5) { for (int w = 0; w < 64; w++) priv[(w * 7 + (int)acc) & 63] += acc; acc += priv[it & 63]; }
    if (mode == 6) { for (int w = 0; w < 16; w++) wr[(size_t)((it * 16 + w) & 1023) * 64 + threadIdx.x] = make_double2(acc, acc); }
    const unsigned long long t0 = clock64();
    const double2 v = p[(threadIdx.x + (unsigned)(it * 4099)) % (unsigned)(stride_d2 - 64)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    total += clock64() - t0;
    acc += (v.x + v.y) * 1e-30;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = total;
  if (acc == 12345.678) sink[0] = acc;
}
int main() {
  const int iters = 300;
  const char* names[] = {"nothing between loads", "LDS sweeps between loads", "FP64 chain between loads", "one atomic on a shared counter per 64 loads", "LDS sweeps + FP64 chain", "private array with dynamic index (scratch) between loads", "16 KB of streamed stores between loads"};
  for (int per_cu : {1, 11}) for (int mode : {0, 3, 5, 6}) {
    const int waves = 256 * per_cu;
    const size_t stride = 112640 * 4, stride_d2 = stride / 16;
    double2* buf; unsigned long long* out; double* sink; unsigned* counter;
    (void)hipMalloc(&buf, (size_t)waves * (stride + (1 << 20))); (void)hipMemset(buf, 0, (size_t)waves * stride);
    (void)hipMalloc(&out, waves * 8); (void)hipMalloc(&sink, 8); (void)hipMalloc(&counter, 4); (void)hipMemset(counter, 0, 4);
    const int lds = 160 * 1024 / per_cu - 512;
    (void)hipFuncSetAttribute((const void*)mix, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(mix, dim3(waves), dim3(64), lds, 0, buf, stride_d2, iters, mode, 2000, out, sink, counter);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), out, waves * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("waves/CU %2d, %-46s: %.0f cycles per 1-KB load\n", per_cu, names[mode], s / waves / iters);
    (void)hipFree(buf); (void)hipFree(out); (void)hipFree(sink); (void)hipFree(counter);
  }
  return 0;
}
