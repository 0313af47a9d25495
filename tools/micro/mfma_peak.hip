// Effective fp32-MFMA ceiling of the box: registers-only v_mfma_f32_32x32x2_f32 loop.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* d; hipMalloc(&d, 4096 * 256 * 4);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int wpc : {1, 2, 3}) {
    int blocks = 256 * wpc, iters = 4000;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 100, 0.5f, 0.25f);
    hipDeviceSynchronize();
    hipEventRecord(s);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * 4096.0;
    printf("blocks/CU %d: %.3f ms, %.1f TFLOP/s\n", wpc, ms, flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}
