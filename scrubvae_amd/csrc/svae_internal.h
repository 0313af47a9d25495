// Internal helpers shared by the HIP translation units of libscrubvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include "../../include/scrubvae_hip.h"

namespace svae {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SVAE_ERR_LAUNCH;
  }
  return SVAE_OK;
}

// Per-device "done once" flag for settings that belong to the current device's copy of a kernel (hipFuncSetAttribute):
// bit d = done on device d.  The guarded action is idempotent, so two threads racing through the first call are harmless.
struct DeviceOnce {
  std::atomic<unsigned long long> bits{0};
  bool need(int* dev_out) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    *dev_out = dev;
    return dev >= 64 || !(bits.load(std::memory_order_acquire) & (1ull << dev));
  }
  void done(int dev) { if (dev < 64) bits.fetch_or(1ull << dev, std::memory_order_release); }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

#define SVAE_REQUIRE(cond, code, ...)      \
  do {                                     \
    if (!(cond)) {                         \
      svae::set_error(__VA_ARGS__);        \
      return (code);                       \
    }                                      \
  } while (0)

// ---- device helpers ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// x2 linear upsample (align_corners = false): 0.75 near + 0.25 far, ONE rounding order everywhere it is evaluated (the standalone
// kernel, the kernels that blend while staging an operand, the by-product they leave for the weight gradient): bit-identical rows
__device__ __forceinline__ float up2_blend(float near_, float far_) { return __builtin_fmaf(0.25f, far_, 0.75f * near_); }
__device__ __forceinline__ float4 up2_blend4(float4 p, float4 q) {
  return make_float4(up2_blend(p.x, q.x), up2_blend(p.y, q.y), up2_blend(p.z, q.z), up2_blend(p.w, q.w));
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x == 256 (4 waves); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem4[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = smem4[0] + smem4[1] + smem4[2] + smem4[3];
  __syncthreads();
  return r;
}

}  // namespace svae
