// Per-CU global -> LDS fetch rate of LDS-DMA (global_load_lds_dwordx4) on gfx950, as the halo GEMM kernels use it: 12-wave
// workgroups (1 per CU: 150 KB of LDS), 4 loader waves issue `per_wave` 1-KiB wave-instructions per stage with a counted
// s_waitcnt vmcnt that leaves DEPTH-1 stages in flight, one s_barrier per stage shared with 8 other waves that either idle
// or (BUSY) run the consumers' load: ds_read_b128 + MFMA.  Prints cycles per stage and bytes per cycle per CU.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_rate fetch_rate.hip && ./fetch_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int MAXN>
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
  if (n == N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  else if constexpr (N < MAXN) wait_vmcnt_rt<N + 1, MAXN>(n);
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXN) : "memory");
}

struct Args {
  const char* base;
  unsigned long long footprint_mask;  // region offsets are taken modulo footprint (power of two)
  unsigned long long wg_stride;       // byte offset between the streams of consecutive "sharing groups"
  int share;                          // workgroups b and b' read the same stream when b / share == b' / share ... see main
  int share_mode;                     // 0: groups of consecutive ids; 1: ids equal modulo (grid / share)
  int iters, per_wave, depth, rows, loaders, busy, regstage, prio, selfdma;
  unsigned long long* out;
};

__global__ __launch_bounds__(768) void fetch_kernel(Args a) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[144 * 1024];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int grp = a.share_mode == 0 ? b / a.share : b % (gridDim.x / a.share);
  const unsigned long long start = (unsigned long long)grp * a.wg_stride;
  if (wave >= 8) {
    const int pw = wave - 8;
    if (pw >= a.loaders) {
      for (int s = 0; s < a.iters; ++s) __builtin_amdgcn_s_barrier();
      return;
    }
    unsigned long long t0 = 0, t1 = 0;
    if (a.prio) __builtin_amdgcn_s_setprio(3);
    const int stage_instr = a.loaders * a.per_wave;
    const int slots = a.depth + 1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    if (a.regstage) {
      // register staging: global_load_dwordx4 one stage ahead, ds_write_b128 into the ring (4 wave-instructions per wave and stage)
      constexpr int PER = 4;
      uint4 ra[PER], rb[PER];
      auto src_of = [&](int s, int i) {
        const unsigned long long k = (unsigned long long)s * stage_instr + pw * PER + i;
        return reinterpret_cast<const uint4*>(a.base + ((start + k * 1024 + lane * 16) & a.footprint_mask));
      };
      auto dst_of = [&](int s, int i) { return reinterpret_cast<uint4*>(smem + ((s % slots) * stage_instr + pw * PER + i) * 1024 + lane * 16); };
#pragma unroll
      for (int i = 0; i < PER; ++i) ra[i] = *src_of(0, i);
      for (int s = 0; s < a.iters; s += 2) {
#pragma unroll
        for (int i = 0; i < PER; ++i) rb[i] = *src_of(s + 1, i);
#pragma unroll
        for (int i = 0; i < PER; ++i) *dst_of(s, i) = ra[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < PER; ++i) ra[i] = *src_of(s + 2, i);
#pragma unroll
        for (int i = 0; i < PER; ++i) *dst_of(s + 1, i) = rb[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
      if (lane == 0 && pw == 0) a.out[b] = t1 - t0;
      if (ra[0].x == 0x12345u) a.out[1] = 2;
      return;
    }
    for (int s = 0; s < a.iters; ++s) {
      for (int i = 0; i < a.per_wave; ++i) {
        const unsigned long long k = (unsigned long long)s * stage_instr + pw * a.per_wave + i;  // running wave-instruction number
        unsigned long long off;
        if (a.rows) {
          // 8 rows x 128 bytes at a 2 KiB stride: 33 such instructions make an "image", the next 32-channel block is 128 bytes on
          const unsigned long long img = k / 33, blk = img & 15, tile = img >> 4;
          off = (tile * 264 + (k % 33) * 8 + (lane >> 3)) * 2048ull + blk * 128 + (lane & 7) * 16;
        } else {
          off = k * 1024 + lane * 16;
        }
        const char* src = a.base + ((start + off) & a.footprint_mask);
        unsigned char* dst = smem + ((s % slots) * stage_instr + pw * a.per_wave + i) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
      }
      wait_vmcnt_rt<0, 48>(a.per_wave * (a.depth - 1));
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (lane == 0 && pw == 0) a.out[b] = t1 - t0;
    return;
  }
  // the other 8 waves
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int rrow = lane & 31;
  const unsigned char* rd = smem + rrow * 64 + (((lane >> 5) ^ ((rrow >> 2) & 3)) << 4) + wave * 2048;
  unsigned long long ct0 = 0, ct1 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct0));
  for (int s = 0; s < a.iters; ++s) {
    if (a.selfdma) {  // the computing waves stage the tiles themselves: selfdma wave-instructions each, ring of depth + 1 slots above 64 KiB
      for (int i = 0; i < a.selfdma; ++i) {
        const unsigned long long k = (unsigned long long)s * 8 * a.selfdma + wave * a.selfdma + i;
        const char* src = a.base + ((start + k * 1024 + lane * 16) & a.footprint_mask);
        unsigned char* dst = smem + 65536 + ((s % (a.depth + 1)) * 8 * a.selfdma + wave * a.selfdma + i) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
      }
    }
    if (a.busy) {
      uint4 v[8];
      if (a.busy & 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const uint4*>(rd + (i & 3) * 16384);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = make_uint4(s + i, lane, i, 3);
      }
      if (a.busy & 2) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i)
            acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v[i]), __builtin_bit_cast(bf16x8, v[(i + 1) & 7]), acc[i & 3], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i & 3][0] += __builtin_bit_cast(float, v[i].x ^ v[i].w);
      }
    }
    if (a.selfdma) wait_vmcnt_rt<0, 48>(a.selfdma * (a.depth - 1));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ct1));
  if (a.selfdma && tid == 0) a.out[b] = ct1 - ct0;
  float sum = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) sum += acc[i][r];
  if (sum == 123.456f) a.out[0] = 1;  // keep the work alive
}

int main() {
  const size_t bytes = 4ull << 30;
  char* buf;
  if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  unsigned long long* out;
  const int grid = 512;
  hipMalloc(&out, grid * sizeof(unsigned long long));
  struct Case { const char* name; unsigned long long footprint, wg_stride; int share, share_mode, per_wave, depth, rows, loaders, busy, regstage, prio, selfdma; };
  std::vector<Case> cases = {
      // name, footprint, stride between streams, share, mode, per_wave, depth, rows, loaders, busy, regstage, prio, selfdma
      {"reads+mfma, no DMA", 2ull << 20, 4ull << 10, 1, 0, 0, 3, 0, 4, 3, 0, 0, 0},
      {"mfma only, no DMA", 2ull << 20, 4ull << 10, 1, 0, 0, 3, 0, 4, 2, 0, 0, 0},
      {"reads only, no DMA", 2ull << 20, 4ull << 10, 1, 0, 0, 3, 0, 4, 1, 0, 0, 0},
      {"DMA 4x4 only", 16ull << 20, 2ull << 20, 8, 1, 4, 3, 0, 4, 0, 0, 0, 0},
      {"DMA 4x4 + reads+mfma", 16ull << 20, 2ull << 20, 8, 1, 4, 3, 0, 4, 3, 0, 0, 0},
      {"DMA 4x4 + mfma only", 16ull << 20, 2ull << 20, 8, 1, 4, 3, 0, 4, 2, 0, 0, 0},
      {"DMA 4x4 + reads only", 16ull << 20, 2ull << 20, 8, 1, 4, 3, 0, 4, 1, 0, 0, 0},
      {"DMA 4x1 + mfma only", 16ull << 20, 2ull << 20, 8, 1, 1, 3, 0, 4, 2, 0, 0, 0},
      {"DMA 4x1 + reads only", 16ull << 20, 2ull << 20, 8, 1, 1, 3, 0, 4, 1, 0, 0, 0},
  };
  const int iters = 400;
  for (const Case& c : cases) {
    Args a;
    a.base = buf; a.footprint_mask = c.footprint - 1; a.wg_stride = c.wg_stride; a.share = c.share; a.share_mode = c.share_mode;
    a.iters = iters; a.per_wave = c.per_wave; a.depth = c.depth; a.rows = c.rows; a.loaders = c.loaders; a.busy = c.busy; a.regstage = c.regstage; a.prio = c.prio; a.selfdma = c.selfdma; a.out = out;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fetch_kernel, dim3(grid), dim3(768), 0, 0, a);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), out, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[grid / 2] / iters;
    const double kib = c.selfdma ? 8.0 * c.selfdma : (double)c.loaders * c.per_wave;
    printf("%-58s %7.0f cycles/stage  %5.1f KiB/stage  %5.1f B/clk/CU  (%.0f us, %.2f TB/s chip)\n", c.name, cyc, kib, kib * 1024 / cyc, ms * 1e3,
           (double)grid * iters * kib * 1024 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
