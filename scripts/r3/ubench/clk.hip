#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64(), r0 = __builtin_readcyclecounter();
  double acc = threadIdx.x;
  for (int w = 0; w < 200000; w++) acc = acc * 1.0000001 + 0.5;
  const unsigned long long c1 = clock64(), w1 = wall_clock64(), r1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = r1 - r0; out[3] = (unsigned long long)acc; }
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 32);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); (void)hipDeviceSynchronize();
  unsigned long long h[4]; (void)hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("200000 dependent FP64 FMAs: clock64 %llu, wall_clock64 (100 MHz) %llu = %.1f us, readcyclecounter %llu => clock64 runs at %.0f MHz, %.1f ticks per FMA\n", h[0], h[1], h[1] / 100.0, h[2], h[0] / (h[1] / 100.0), (double)h[0] / 200000);
  return 0;
}
