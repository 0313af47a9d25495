// Probe of __builtin_amdgcn_global_load_lds (16-byte form) on gfx950: per-lane source, wave-linear LDS destination.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, const float* __restrict__ zero, float* __restrict__ out, int rows_valid) {
  __shared__ __attribute__((aligned(16))) float tile[128 * 32];  // 128 rows x 32 floats, linear
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // each wave instruction: 8 rows x 128 B.  4 passes x 4 waves = 128 rows... (pass p, wave w) -> rows (p*4+w)*8 .. +7
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = (p * 4 + wave) * 8 + (lane >> 3);
    const int chunk = lane & 7;
    const int src_chunk = chunk ^ ((row >> 1) & 7);  // XOR swizzle on the SOURCE address
    const float* g = row < rows_valid ? src + (long long)row * 32 + src_chunk * 4 : zero;
    float* l = tile + (p * 4 + wave) * 8 * 32;  // wave-uniform base
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_ptr_t)l, 16, 0, 0);
  }
  __syncthreads();
  // read back un-swizzled
  for (int e = tid; e < 128 * 8; e += 256) {
    const int row = e >> 3, chunk = e & 7;
    const float4 v = *reinterpret_cast<const float4*>(&tile[row * 32 + (chunk ^ ((row >> 1) & 7)) * 4]);
    *reinterpret_cast<float4*>(&out[row * 32 + chunk * 4]) = v;
  }
}
int main() {
  std::vector<float> h(128 * 32), o(128 * 32, -1.f);
  for (int i = 0; i < 128 * 32; ++i) h[i] = (float)i;
  float *d, *z, *dout;
  hipMalloc(&d, h.size() * 4); hipMalloc(&z, 256); hipMalloc(&dout, h.size() * 4);
  hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemset(z, 0, 256);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, z, dout, 100);
  hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int r = 0; r < 128; ++r) for (int c = 0; c < 32; ++c) {
    const float want = r < 100 ? (float)(r * 32 + c) : 0.f;
    if (o[r * 32 + c] != want) { if (bad < 5) printf("mismatch r=%d c=%d got %f want %f\n", r, c, o[r * 32 + c], want); ++bad; }
  }
  printf("glds test: %s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
  return bad != 0;
}
