// fp32 implicit-GEMM convolution kernels for gfx950 (MI355X).
//
// Every nn.Conv1d / nn.ConvTranspose1d / nn.Linear of the SC-VAE trunk
// (reference: src/scrubvae/model/residual.py:79-109,137-170,198,219-222,264,286) and their
// backward passes run through the three kernels in this file:
//
//   gather_gemm<BN,B_KC>  C[m][n] (+)= bias[n] + sum_t sum_c A[row(m,t)][c] * W_t[c][n]
//        forward conv (B_KC=false: weights read as [k][n]) and data-gradient
//        (B_KC=true: the SAME weight tensor read as [n][k]); rows are gathered per tap, so
//        im2col never exists in memory; stride-2 transposed convs are split by output
//        parity so no MFMA work is spent on structural zeros.
//   wgrad_gemm<BN>        dW_t[c][n] = sum_r X[xrow(r,t)][c] * dY[yrow(r,t)][n], split over
//        the reduction rows into slabs that are summed in a fixed order (reproducible).
//
// Data layout: activations channels-last [B*L][ld]; weights w[tap][c_in][c_out].
// Math: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
// Tile: BM x BN (64/128 each) per workgroup of 4 waves (2x2); the gather kernel stages 32-deep
// K tiles (each A row segment is one full 128-byte line), the weight-gradient kernel 16-deep
// ones.  K-contiguous operands sit in LDS as [row][K+4] and are fetched with one
// ds_read_b128 per 4 k-steps: the k order inside an 8-chunk is permuted (lane half h holds
// k = 8q+4h+j) which is legal because both operands use the same permutation.
// Row-contiguous operands sit as [k][cols] and are fetched with conflict-free ds_read_b32.
// Two LDS stages + register prefetch (global loads of tile k+1 are issued before the MFMAs
// of tile k, written to LDS after them): one barrier per K tile.
#include "gemm_common.h"
#include <type_traits>

namespace svae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int LDK = BK + 4;  // padded row of a K-contiguous LDS tile (conflict-free b128)


__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

constexpr int GBK = 32;         // K depth of one LDS stage of the gather kernel (two 16-deep halves)
constexpr int GLDK = GBK + 4;   // padded row of a K-contiguous tile: conflict-free ds_read_b128

template <int BM, int BN, bool B_KC>
__global__ __launch_bounds__(256) void gather_gemm_kernel(const GatherArgs g) {
  constexpr int WM = BM / 2;   // wave tile rows
  constexpr int MT = WM / 32;  // MFMA tiles down
  constexpr int WN = BN / 2;   // wave tile columns
  constexpr int NT = WN / 32;  // MFMA tiles across
  constexpr int APASS = BM / 32;                      // float4 per thread for the A tile (8 per row)
  constexpr int B_ELEMS = B_KC ? BN * GLDK : GBK * BN;
  constexpr int F4_PER_ROW = BN / 4;                  // row-contiguous B: float4 per k-row
  constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
  constexpr int BPASS = B_KC ? BN / 32 : GBK / ROWS_PER_PASS;
  __shared__ __attribute__((aligned(16))) float As[2][BM * GLDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][B_ELEMS];
  __shared__ long long rowoff[BM];

  const int tid = threadIdx.x;
  int bx = blockIdx.x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = blockIdx.y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  const int* __restrict__ tap_base = g.base[phase];
  const int* __restrict__ tap_w = g.widx[phase];

  // ---- per-thread A rows: 8 lanes cover one 128-byte row segment (a full cache line)
  const int akq = tid & 7;
  long long a_off[APASS];
  int a_j[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const long long m = m0 + (tid >> 3) + 32 * i;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      a_off[i] = b * (long long)g.Lin * g.ldA;
      a_j[i] = j * g.sj;
    } else {
      a_off[i] = 0;
      a_j[i] = -(1 << 28);
    }
  }
  if (tid < BM) {
    const long long m = m0 + tid;
    long long off = -1;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      off = (b * g.Lout + (phase + g.n_phase * j)) * (long long)g.ldC;
    }
    rowoff[tid] = off;
  }

  const int tiles_per_tap = (g.Kc + GBK - 1) / GBK;  // the last tile of a tap may be half full
  const int nk_all = ntaps * tiles_per_tap;
  // split-K: this block's share of the K tiles
  const int ks = g.ksplit > 1 ? g.ksplit : 1;
  const int kt_begin = (int)((long long)nk_all * blockIdx.z / ks);
  const int nk = (int)((long long)nk_all * (blockIdx.z + 1) / ks) - kt_begin;

  float4 ra[APASS], rb[BPASS];
  bool ra_ok[APASS], rb_ok[BPASS];  // zero-fill is applied when the tile is written to LDS
  // tap bookkeeping is advanced incrementally and one tile AHEAD of its use, so that the
  // scalar loads of the tap table never sit in front of the global loads / MFMAs
  int nx_ti = kt_begin / tiles_per_tap;
  int nx_c0 = (kt_begin - nx_ti * tiles_per_tap) * GBK;
  int nx_tb = ntaps > 0 ? tap_base[nx_ti < ntaps ? nx_ti : ntaps - 1] : 0;
  long long nx_woff = ntaps > 0 ? (long long)tap_w[nx_ti < ntaps ? nx_ti : ntaps - 1] * g.w_tap_stride : 0;
  auto load_tile = [&]() {
    const int c0 = nx_c0;
    const int tb = nx_tb;
    const float* wt = g.W + nx_woff;
    {  // branch-free advance (uniform selects + one scalar table load per tile)
      nx_c0 += GBK;
      const bool wrap = nx_c0 >= g.Kc;
      nx_c0 = wrap ? 0 : nx_c0;
      nx_ti += wrap ? 1 : 0;
      const int tic = nx_ti < ntaps ? nx_ti : ntaps - 1;
      nx_tb = tap_base[tic];
      nx_woff = (long long)tap_w[tic] * g.w_tap_stride;
    }
    // branch-free loads: clamped address now, zero-select when the registers go to LDS
    const bool kq_ok = c0 + akq * 4 < g.Kc;
    const int cq = kq_ok ? c0 + akq * 4 : 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int li = a_j[i] + tb;
      const bool ok = li >= 0 && li < g.Lin;
      const int lic = ok ? li : 0;
      ra[i] = ld4(g.A + a_off[i] + (long long)lic * g.ldA + cq);
      ra_ok[i] = ok && kq_ok;
    }
    if constexpr (B_KC) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        const bool ok = n < g.N;
        rb[i] = ld4(wt + (long long)(ok ? n : 0) * g.ldW + cq);
        rb_ok[i] = ok && kq_ok;
      }
    } else {
      const int n = n0 + (tid % F4_PER_ROW) * 4;
      const bool ok = n < g.N;
      const int nc = ok ? n : 0;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int k = tid / F4_PER_ROW + ROWS_PER_PASS * i;
        const bool kok = c0 + k < g.Kc;
        rb[i] = ld4(wt + (long long)(kok ? c0 + k : 0) * g.ldW + nc);
        rb_ok[i] = ok && kok;
      }
    }
  };
  auto store_tile = [&](int buf) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < APASS; ++i) st4(&As[buf][((tid >> 3) + 32 * i) * GLDK + akq * 4], ra_ok[i] ? ra[i] : zero4);
    if constexpr (B_KC) {
#pragma unroll
      for (int i = 0; i < BPASS; ++i) st4(&Bs[buf][((tid >> 3) + 32 * i) * GLDK + akq * 4], rb_ok[i] ? rb[i] : zero4);
    } else {
#pragma unroll
      for (int i = 0; i < BPASS; ++i)
        st4(&Bs[buf][(tid / F4_PER_ROW + ROWS_PER_PASS * i) * BN + (tid % F4_PER_ROW) * 4], rb_ok[i] ? rb[i] : zero4);
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, h = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // software pipeline: tile k is in LDS, tile k+1 is in flight in registers.  Early in tile k
  // the registers are written to the other LDS stage and immediately re-issued for tile k+2:
  // global-load latency gets a whole tile of MFMAs, the ds_writes hide under them, and only a
  // bare barrier is left at the end of the iteration.
  if (nk > 0) {
    load_tile();
    store_tile(0);
    if (nk > 1) load_tile();
  }
  __syncthreads();

  // One K tile (32 deep = two 16-deep halves).  do_store/do_load are compile-time so that the
  // steady-state loop body is a single basic block the compiler can interleave freely.
  auto k_tile = [&](int kt, auto do_store, auto do_load) {
    const int buf = kt & 1;
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 av[2][MT], bv[2][NT];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int q = half * 2 + q2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[q2][mt] = ld4(&as[(wr * WM + mt * 32 + lr) * GLDK + q * 8 + h * 4]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (B_KC) {
            bv[q2][nt] = ld4(&bs[(wc * WN + nt * 32 + lr) * GLDK + q * 8 + h * 4]);
          } else {
            const float* p = &bs[(q * 8 + h * 4) * BN + wc * WN + nt * 32 + lr];
            bv[q2][nt] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
          }
        }
      }
      if (half == 0) {
        if constexpr (decltype(do_store)::value) store_tile(buf ^ 1);
        if constexpr (decltype(do_load)::value) load_tile();
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = jj == 0 ? av[q2][mt].x : jj == 1 ? av[q2][mt].y : jj == 2 ? av[q2][mt].z : av[q2][mt].w;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float b = jj == 0 ? bv[q2][nt].x : jj == 1 ? bv[q2][nt].y : jj == 2 ? bv[q2][nt].z : bv[q2][nt].w;
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt][nt], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  };
  {
    int kt = 0;
    for (; kt + 2 < nk; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { k_tile(kt, std::true_type{}, std::false_type{}); ++kt; }
    if (kt < nk) k_tile(kt, std::false_type{}, std::false_type{});
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5)
  if (g.ksplit > 1) {  // partial tile -> slab[z][m][N] (bias / accumulate are applied by splitk_reduce_kernel)
    float* slab = g.slab + (long long)blockIdx.z * Mp * g.N;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wc * WN + nt * 32 + lr;
      if (col >= g.N) continue;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long long m = m0 + wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < Mp) slab[m * g.N + col] = acc[mt][nt][r];
        }
    }
    return;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + wc * WN + nt * 32 + lr;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long off = rowoff[row];
        if (off >= 0) {
          float* dst = g.C + off + col;
          float v = acc[mt][nt][r] + bv;
          if (g.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ LDS-DMA variant
// Same math and tiles as gather_gemm_kernel, but both operands go global -> LDS directly
// (global_load_lds_dwordx4): no staging VGPRs, no ds_write, no zero-select VALU.  The LDS image
// of a K-contiguous tile is linear [row][32] (one 128-byte row = 8 chunks of 16 B; a wave
// instruction fills 8 rows); bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle applied on the per-lane SOURCE address (chunk' = chunk ^ ((row>>1)&7)) and again on
// the read.  Rows that fall into conv padding / past M / past Kc read from a zero page.
__device__ __attribute__((aligned(16))) float g_zero_page[64];

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;

template <int BM, int BN, bool B_KC>
__global__ __launch_bounds__(256) void gather_gemm_dma_kernel(const GatherArgs g) {
  constexpr int WM = BM / 2, MT = WM / 32, WN = BN / 2, NT = WN / 32;
  constexpr int A_ELEMS = BM * GBK;
  constexpr int B_ELEMS = BN * GBK;
  constexpr int APASS = BM / 32;                 // wave instructions per wave for A (8 rows each)
  constexpr int BPASS = B_KC ? BN / 32 : BN / 32;  // KC: 8 rows of 128 B; KN: 1 KiB = 256/BN k-rows
  // ONE shared array (a second __shared__ object next to an LDS-DMA target makes hipcc drain vmcnt early)
  __shared__ __attribute__((aligned(16))) float smem[2 * A_ELEMS + 2 * B_ELEMS + 2 * BM];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * A_ELEMS;
  long long* const rowoff = reinterpret_cast<long long*>(smem + 2 * A_ELEMS + 2 * B_ELEMS);

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  int bx = blockIdx.x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = blockIdx.y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  const int* __restrict__ tap_base = g.base[phase];
  const int* __restrict__ tap_w = g.widx[phase];

  // ---- A: lane -> (row, LDS chunk position); the logical chunk it fetches is swizzled
  const int a_rl = lane >> 3, a_pos = lane & 7;
  long long a_off[APASS];
  int a_j[APASS], a_chunk[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int row = (i * 4 + wave) * 8 + a_rl;
    const long long m = m0 + row;
    a_chunk[i] = a_pos ^ ((row >> 1) & 7);
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      a_off[i] = b * (long long)g.Lin * g.ldA;
      a_j[i] = j * g.sj;
    } else {
      a_off[i] = 0;
      a_j[i] = -(1 << 28);
    }
  }
  if (tid < BM) {
    const long long m = m0 + tid;
    long long off = -1;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      off = (b * g.Lout + (phase + g.n_phase * j)) * (long long)g.ldC;
    }
    rowoff[tid] = off;
  }
  // ---- B lane mapping
  int b_row[BPASS], b_chunk[BPASS];  // KC: (n row, logical k chunk); KN: (k row, n float4 index)
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    if constexpr (B_KC) {
      const int row = (i * 4 + wave) * 8 + (lane >> 3);
      b_row[i] = row;
      b_chunk[i] = (lane & 7) ^ ((row >> 1) & 7);
    } else {
      constexpr int F4 = BN / 4;            // float4 per k-row
      constexpr int RPI = 64 / F4;          // k-rows per wave instruction (2 for BN=128, 4 for BN=64)
      b_row[i] = (i * 4 + wave) * RPI + lane / F4;
      b_chunk[i] = lane % F4;
    }
  }

  const int tiles_per_tap = (g.Kc + GBK - 1) / GBK;
  const int nk = ntaps * tiles_per_tap;
  int nx_c0 = 0, nx_ti = 0;
  int nx_tb = ntaps > 0 ? tap_base[0] : 0;
  long long nx_woff = ntaps > 0 ? (long long)tap_w[0] * g.w_tap_stride : 0;
  const float* const zp = g_zero_page;

  auto issue_tile = [&](int buf) {
    const int c0 = nx_c0;
    const int tb = nx_tb;
    const float* wt = g.W + nx_woff;
    {
      nx_c0 += GBK;
      const bool wrap = nx_c0 >= g.Kc;
      nx_c0 = wrap ? 0 : nx_c0;
      nx_ti += wrap ? 1 : 0;
      const int tic = nx_ti < ntaps ? nx_ti : ntaps - 1;
      nx_tb = tap_base[tic];
      nx_woff = (long long)tap_w[tic] * g.w_tap_stride;
    }
    float* as = As0 + buf * A_ELEMS;
    float* bs = Bs0 + buf * B_ELEMS;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int li = a_j[i] + tb;
      const int c = c0 + a_chunk[i] * 4;
      const bool ok = li >= 0 && li < g.Lin && c < g.Kc;
      const float* src = ok ? g.A + a_off[i] + (long long)li * g.ldA + c : zp;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(as + (i * 4 + wave) * 8 * GBK), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      if constexpr (B_KC) {
        const int n = n0 + b_row[i];
        const int c = c0 + b_chunk[i] * 4;
        const bool ok = n < g.N && c < g.Kc;
        const float* src = ok ? wt + (long long)n * g.ldW + c : zp;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(bs + (i * 4 + wave) * 8 * GBK), 16, 0, 0);
      } else {
        const int k = c0 + b_row[i];
        const int n = n0 + b_chunk[i] * 4;
        const bool ok = k < g.Kc && n < g.N;
        const float* src = ok ? wt + (long long)k * g.ldW + n : zp;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(bs + (i * 4 + wave) * 256), 16, 0, 0);
      }
    }
  };

  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, h = lane >> 5;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nk > 0) issue_tile(0);
  __syncthreads();  // (emits vmcnt(0) for the DMA in flight)

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue_tile(buf ^ 1);  // the other stage is free since the last barrier
    const float* as = As0 + buf * A_ELEMS;
    const float* bs = Bs0 + buf * B_ELEMS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 av[2][MT], bv[2][NT];
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
        const int q = half * 2 + q2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int row = wr * WM + mt * 32 + lr;
          av[q2][mt] = ld4(&as[row * GBK + (((q * 2 + h) ^ ((row >> 1) & 7)) << 2)]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (B_KC) {
            const int row = wc * WN + nt * 32 + lr;
            bv[q2][nt] = ld4(&bs[row * GBK + (((q * 2 + h) ^ ((row >> 1) & 7)) << 2)]);
          } else {
            const float* p = &bs[(q * 8 + h * 4) * BN + wc * WN + nt * 32 + lr];
            bv[q2][nt] = make_float4(p[0], p[BN], p[2 * BN], p[3 * BN]);
          }
        }
      }
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const float a = jj == 0 ? av[q2][mt].x : jj == 1 ? av[q2][mt].y : jj == 2 ? av[q2][mt].z : av[q2][mt].w;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const float b = jj == 0 ? bv[q2][nt].x : jj == 1 ? bv[q2][nt].y : jj == 2 ? bv[q2][nt].z : bv[q2][nt].w;
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt][nt], 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + wc * WN + nt * 32 + lr;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long off = rowoff[row];
        if (off >= 0) {
          float* dst = g.C + off + col;
          float v = acc[mt][nt][r] + bv;
          if (g.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------- weight grad

constexpr int WBK = 16;  // reduction rows per LDS stage of the weight-gradient kernel (32 measured 4 % slower)

template <int BM, int BN>
__global__ __launch_bounds__(256) void wgrad_gemm_kernel(const WgradArgs g) {
  constexpr int WM = BM / 2;
  constexpr int MT = WM / 32;
  constexpr int APASS = BM * WBK / 1024;
  constexpr int AF4_PER_ROW = BM / 4;
  constexpr int AROWS_PER_PASS = 256 / AF4_PER_ROW;
  constexpr int WN = BN / 2;
  constexpr int NT = WN / 32;
  constexpr int BPASS = BN * WBK / 1024;
  __shared__ __attribute__((aligned(16))) float As[2][WBK * BM];
  __shared__ __attribute__((aligned(16))) float Bs[2][WBK * BN];

  const int tid = threadIdx.x;
  const int ti = blockIdx.x / g.ctiles;
  const int c0 = (blockIdx.x - ti * g.ctiles) * BM;
  const int n0 = blockIdx.y * BN;
  const long long r_begin = (long long)blockIdx.z * g.rows_per_split;
  long long r_end = r_begin + g.rows_per_split;
  if (r_end > g.R) r_end = g.R;
  const int tbx = g.bx[ti], tby = g.by[ti];

  const int a_c = c0 + (tid % AF4_PER_ROW) * 4;
  const int a_r = tid / AF4_PER_ROW;
  constexpr int F4_PER_ROW = BN / 4;
  constexpr int ROWS_PER_PASS = 256 / F4_PER_ROW;
  const int b_n = n0 + (tid % F4_PER_ROW) * 4;
  const int b_r = tid / F4_PER_ROW;

  float4 ra[APASS], rb[BPASS];
  // Reduction rows advance by BK per tile.  Everything a staged row needs -- its X / dY row
  // pointers and the two in-range tests -- is advanced with adds only (no division, no 64-bit
  // multiply in the loop: the weight-gradient kernel was VALU-issue heavy, ~9 VALU per MFMA).
  const int q16 = WBK / g.nj, r16 = WBK % g.nj;
  const long long x_step = ((long long)q16 * g.Lx + (long long)r16 * g.sx) * g.ldX;   // +WBK rows, no wrap
  const long long x_wrap = ((long long)g.Lx - (long long)g.nj * g.sx) * g.ldX;         // j wrapped into next sample
  const long long y_step = ((long long)q16 * g.Ly + (long long)r16 * g.sy) * g.ldY;
  const long long y_wrap = ((long long)g.Ly - (long long)g.nj * g.sy) * g.ldY;
  const float* a_ptr[APASS];
  const float* b_ptr[BPASS];
  int a_jj[APASS], b_jj[BPASS];
  long long a_left[APASS], b_left[BPASS];  // rows left before r_end (row valid iff > 0)
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const long long r = r_begin + a_r + AROWS_PER_PASS * i;
    const long long b = r / g.nj;
    a_jj[i] = (int)(r - b * g.nj);
    a_ptr[i] = g.X + (b * g.Lx + (long long)a_jj[i] * g.sx + tbx) * (long long)g.ldX + (a_c < g.Kc ? a_c : 0);
    a_left[i] = r_end - r;
  }
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const long long r = r_begin + b_r + ROWS_PER_PASS * i;
    const long long b = r / g.nj;
    b_jj[i] = (int)(r - b * g.nj);
    b_ptr[i] = g.dY + (b * g.Ly + (long long)b_jj[i] * g.sy + tby) * (long long)g.ldY + (b_n < g.N ? b_n : 0);
    b_left[i] = r_end - r;
  }
  const bool a_col_ok = a_c < g.Kc, b_col_ok = b_n < g.N;
  bool ra_ok[APASS], rb_ok[BPASS];
  // branch-free loads (clamped address now, zero-select at the LDS store): conditional loads
  // make hipcc wait vmcnt(0) after each one, serialising the memory latencies of a tile
  auto load_tile = [&](long long) {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int xr = a_jj[i] * g.sx + tbx, yr = a_jj[i] * g.sy + tby;
      const bool ok = a_left[i] > 0 && a_col_ok && xr >= 0 && xr < g.Lx && yr >= 0 && yr < g.Ly;
      ra[i] = ld4(ok ? a_ptr[i] : g.X);
      ra_ok[i] = ok;
      a_jj[i] += r16;
      a_ptr[i] += x_step;
      a_left[i] -= WBK;
      if (a_jj[i] >= g.nj) { a_jj[i] -= g.nj; a_ptr[i] += x_wrap; }
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int xr = b_jj[i] * g.sx + tbx, yr = b_jj[i] * g.sy + tby;
      const bool ok = b_left[i] > 0 && b_col_ok && xr >= 0 && xr < g.Lx && yr >= 0 && yr < g.Ly;
      rb[i] = ld4(ok ? b_ptr[i] : g.dY);
      rb_ok[i] = ok;
      b_jj[i] += r16;
      b_ptr[i] += y_step;
      b_left[i] -= WBK;
      if (b_jj[i] >= g.nj) { b_jj[i] -= g.nj; b_ptr[i] += y_wrap; }
    }
  };
  auto store_tile = [&](int buf) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < APASS; ++i)
      st4(&As[buf][(a_r + AROWS_PER_PASS * i) * BM + (tid % AF4_PER_ROW) * 4], ra_ok[i] ? ra[i] : zero4);
#pragma unroll
    for (int i = 0; i < BPASS; ++i)
      st4(&Bs[buf][(b_r + ROWS_PER_PASS * i) * BN + (tid % F4_PER_ROW) * 4], rb_ok[i] ? rb[i] : zero4);
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, h = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)((r_end - r_begin + WBK - 1) / WBK);
  if (nk > 0) {
    load_tile(r_begin);
    store_tile(0);
    if (nk > 1) load_tile(r_begin + WBK);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    const float* as = As[buf];
    const float* bs = Bs[buf];
#pragma unroll
    for (int half = 0; half < WBK / 16; ++half) {
      float av[2][MT][4], bv[2][NT][4];
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          const int kk = half * 16 + q * 8 + h * 4 + jj;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) av[q][mt][jj] = as[kk * BM + wr * WM + mt * 32 + lr];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) bv[q][nt][jj] = bs[kk * BN + wc * WN + nt * 32 + lr];
        }
      if (half == 0) {  // mid-tile: registers -> other LDS stage, then re-issue them for tile k+2
        if (kt + 1 < nk) store_tile(buf ^ 1);
        if (kt + 2 < nk) load_tile(r_begin + (long long)(kt + 2) * WBK);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][mt][jj], bv[q][nt][jj], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }

  float* out = g.out + (long long)blockIdx.z * g.slab_stride + (long long)ti * g.Kc * g.ldW;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + wc * WN + nt * 32 + lr;
    if (col >= g.N) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (c < g.Kc) {
          float* dst = out + (long long)c * g.ldW + col;
          float v = acc[mt][nt][r];
          if (g.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------- tap-fused weight gradient
// Small-weight layers (few output tiles, very long reductions): one workgroup owns a 64x64
// (c_in x c_out) tile for up to TG taps at once.  The operand whose rows do not depend on the
// tap ("shared": dY for a conv, X for a transposed conv) is staged once per 16 reduction rows,
// the other one once per tap; every tap has its own 32x32 accumulator per wave.  Compared with
// wgrad_gemm this does TG x the MFMA work per staged shared tile and produces TG x fewer slabs.
struct WgradFusedArgs {
  const float* S;   // shared operand   [rows = b*Ls + j][ldS]
  const float* P;   // per-tap operand  [rows = b*Lp + j*sp + bp[t]][ldP]
  float* out;       // slab base [nsplit][T][Kc][ldW] (or dw itself when nsplit == 1)
  long long R, rows_per_split, slab_stride;
  int nj, Ls, Lp, sp;
  int bp[SVAE_MAX_TAPS];
  int T, Kc, N, ldS, ldP, ldW;
  int ctiles, ntiles, accumulate;
};

template <int TG, bool P_IS_A>
__global__ __launch_bounds__(256) void wgrad_fused_kernel(const WgradFusedArgs g) {
  constexpr int TB = 64;  // tile edge (both ways)
  __shared__ __attribute__((aligned(16))) float Ps[2][TG][BK * TB];
  __shared__ __attribute__((aligned(16))) float Ss[2][BK * TB];
  const int tid = threadIdx.x;
  const int ct = blockIdx.x / g.ntiles, nt_ = blockIdx.x - ct * g.ntiles;
  const int c0 = ct * TB, n0 = nt_ * TB;
  const int t0 = blockIdx.y * TG;
  const int ntap = (g.T - t0) < TG ? (g.T - t0) : TG;
  const long long r_begin = (long long)blockIdx.z * g.rows_per_split;
  long long r_end = r_begin + g.rows_per_split;
  if (r_end > g.R) r_end = g.R;
  // the A operand has c_in columns, the B operand c_out columns
  const int s_col0 = P_IS_A ? n0 : c0, p_col0 = P_IS_A ? c0 : n0;
  const int s_cols = P_IS_A ? g.N : g.Kc, p_cols = P_IS_A ? g.Kc : g.N;

  const int lrow = tid >> 4, lc4 = (tid & 15) * 4;  // one float4 of a 16 x 64 tile per thread
  const bool s_ok_col = s_col0 + lc4 < s_cols, p_ok_col = p_col0 + lc4 < p_cols;
  const int q16 = BK / g.nj, r16 = BK % g.nj;
  long long rb;
  int rj;
  {
    const long long r = r_begin + lrow;
    rb = r / g.nj;
    rj = (int)(r - rb * g.nj);
  }
  float4 rs, rp[TG];
  bool rs_ok, rp_ok[TG];
  long long next_r = r_begin + lrow;
  auto load_tile = [&]() {
    const bool row_ok = next_r < r_end;
    rs_ok = row_ok && s_ok_col;
    rs = ld4(g.S + ((rs_ok ? rb : 0) * g.Ls + (rs_ok ? rj : 0)) * (long long)g.ldS + (s_ok_col ? s_col0 + lc4 : 0));
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < ntap) {
        const int pr = rj * g.sp + g.bp[t0 + t];
        const bool ok = row_ok && p_ok_col && pr >= 0 && pr < g.Lp;
        rp_ok[t] = ok;
        rp[t] = ld4(g.P + ((ok ? rb : 0) * g.Lp + (ok ? pr : 0)) * (long long)g.ldP + (p_ok_col ? p_col0 + lc4 : 0));
      }
    }
    next_r += BK;
    rj += r16; rb += q16;
    if (rj >= g.nj) { rj -= g.nj; ++rb; }
  };
  auto store_tile = [&](int buf) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    st4(&Ss[buf][lrow * TB + lc4], rs_ok ? rs : zero4);
#pragma unroll
    for (int t = 0; t < TG; ++t)
      if (t < ntap) st4(&Ps[buf][t][lrow * TB + lc4], rp_ok[t] ? rp[t] : zero4);
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, h = lane >> 5;
  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int nk = (int)((r_end - r_begin + BK - 1) / BK);
  if (nk > 0) {
    load_tile();
    store_tile(0);
    if (nk > 1) load_tile();
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    // shared-operand fragment (8 k values per lane), reused by every tap
    float sv[8];
    const int s_lane_col = (P_IS_A ? wc : wr) * 32 + lr;
    const int p_lane_col = (P_IS_A ? wr : wc) * 32 + lr;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) sv[q * 4 + jj] = Ss[buf][(q * 8 + h * 4 + jj) * TB + s_lane_col];
    if (kt + 1 < nk) store_tile(buf ^ 1);
    if (kt + 2 < nk) load_tile();
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < ntap) {
        float pv[8];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) pv[q * 4 + jj] = Ps[buf][t][(q * 8 + h * 4 + jj) * TB + p_lane_col];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if constexpr (P_IS_A) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(pv[e], sv[e], acc[t], 0, 0, 0);
          else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sv[e], pv[e], acc[t], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  const int col = n0 + wc * 32 + lr;
  if (col < g.N) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      if (t < ntap) {
        float* out = g.out + (long long)blockIdx.z * g.slab_stride + (long long)(t0 + t) * g.Kc * g.ldW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = c0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (c < g.Kc) {
            float* dst = out + (long long)c * g.ldW + col;
            float v = acc[t][r];
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
}

// dst[i] (+)= sum_s slab[s][i], fixed order
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ dst,
                                                            long long n4, long long stride, int nsplit, int accumulate) {
  // 16 slab loads in flight per thread (the slabs are `stride` apart: every load is its own
  // latency), summed in slab order -> same bits every run
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 s = accumulate ? ld4(dst + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 16 <= nsplit; k += 16) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = ld4(slab + (long long)(k + u) * stride + i * 4);
#pragma unroll
      for (int u = 0; u < 16; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < nsplit; ++k) {
      const float4 v = ld4(slab + (long long)k * stride + i * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    st4(dst + i * 4, s);
  }
}

// column sums of a [rows][ld] matrix: part[chunk][C]
constexpr int COLSUM_ROWS = 512;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, long long rows, int C, int ld,
                                                              float* __restrict__ part) {
  __shared__ float red[16][65];
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + c4 * 4;
  const long long r0 = (long long)blockIdx.x * COLSUM_ROWS;
  long long r1 = r0 + COLSUM_ROWS;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += 16) {
      const float4 v = ld4(x + r * ld + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  red[rl][c4 * 4 + 0] = s.x; red[rl][c4 * 4 + 1] = s.y; red[rl][c4 * 4 + 2] = s.z; red[rl][c4 * 4 + 3] = s.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x];
    const int cc = blockIdx.y * 64 + threadIdx.x;
    if (cc < C) part[(long long)blockIdx.x * C + cc] = t;
  }
}

__global__ void colsum_final_kernel(const float* __restrict__ part, int chunks, int C, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int k = 0; k < chunks; ++k) s += (double)part[(long long)k * C + c];
  out[c] = (accumulate ? out[c] : 0.f) + (float)s;
}

// C[m][col] (+)= bias[col] + sum_z slab[z][m][col]   (fixed z order; single-phase plans: row m of C is m*ldC)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias,
                                                            float* __restrict__ C, long long M, int N, int ldC, int ksplit, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= M * N) return;
  const long long m = i / N;
  const int col = (int)(i - m * N);
  float v = bias ? bias[col] : 0.f;
  float s = 0.f;
  for (int z = 0; z < ksplit; ++z) s += slab[(long long)z * M * N + i];
  v += s;
  float* dst = C + m * ldC + col;
  if (accumulate) v += *dst;
  *dst = v;
}

// split-K factor for skinny problems (few output tiles, long K): 0 = do not split
static int splitk_factor(const GatherArgs& g, int bm, int bn) {
  if (g.n_phase != 1 || g.nj[0] != g.Lout) return 0;
  const long long tiles = ((g.M[0] + bm - 1) / bm) * ((g.N + bn - 1) / bn);
  const int nk = g.ntaps[0] * ((g.Kc + GBK - 1) / GBK);
  if (tiles >= 128 || nk < 16) return 0;
  long long ks = 512 / tiles;
  if (ks > nk / 4) ks = nk / 4;
  if (ks > 64) ks = 64;
  return ks >= 2 ? (int)ks : 0;
}

// -------------------------------------------------------------------------- host side

template <bool B_KC>
static int launch_gather_auto(GatherArgs& g, hipStream_t st, int override_code, void* ws = nullptr, size_t ws_bytes = 0) {
  Tile t;
  if (!decode_tile(override_code, t)) t = pick_tile(g.M[0], g.M[1], g.N);
  for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + t.bm - 1) / t.bm);
  const int bm = g.blocks_m[0] + g.blocks_m[1];
  if (bm == 0) return SVAE_OK;
  dim3 grid(bm, (g.N + t.bn - 1) / t.bn);
  if (ws != nullptr) {  // split-K for skinny problems when the caller provides the slab workspace
    const int ks = splitk_factor(g, 64, 64);
    if (ks >= 2 && ws_bytes >= (size_t)ks * g.M[0] * g.N * sizeof(float)) {
      g.ksplit = ks;
      g.slab = (float*)ws;
      g.blocks_m[0] = (int)((g.M[0] + 63) / 64);
      dim3 sgrid(g.blocks_m[0], (g.N + 63) / 64, ks);
      hipLaunchKernelGGL((gather_gemm_kernel<64, 64, B_KC>), sgrid, dim3(256), 0, st, g);
      if (int e = check_launch("gather_gemm(split-K)")) return e;
      const long long n = g.M[0] * g.N;
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g.slab, g.bias, g.C, g.M[0], g.N,
                         g.ldC, ks, g.accumulate);
      return check_launch("splitk_reduce");
    }
  }
  if (t.dma) {
    if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((gather_gemm_dma_kernel<128, 128, B_KC>), grid, dim3(256), 0, st, g);
    else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((gather_gemm_dma_kernel<128, 64, B_KC>), grid, dim3(256), 0, st, g);
    else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((gather_gemm_dma_kernel<64, 128, B_KC>), grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gather_gemm_dma_kernel<64, 64, B_KC>), grid, dim3(256), 0, st, g);
    return check_launch("gather_gemm_dma");
  }
  if (t.bm == 128 && t.bn == 128) hipLaunchKernelGGL((gather_gemm_kernel<128, 128, B_KC>), grid, dim3(256), 0, st, g);
  else if (t.bm == 128 && t.bn == 64) hipLaunchKernelGGL((gather_gemm_kernel<128, 64, B_KC>), grid, dim3(256), 0, st, g);
  else if (t.bm == 64 && t.bn == 128) hipLaunchKernelGGL((gather_gemm_kernel<64, 128, B_KC>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gather_gemm_kernel<64, 64, B_KC>), grid, dim3(256), 0, st, g);
  return check_launch("gather_gemm");
}

}  // namespace svae

using namespace svae;

extern "C" int svae_conv_fwd(const svae_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                             int accumulate, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(!d->up2, SVAE_ERR_SHAPE, "conv_fwd: the fused x2 upsample of the input (up2) exists in the split-precision halo kernels only");
  SVAE_REQUIRE(x && w && y, SVAE_ERR_ARG, "conv_fwd: null pointer");
  SVAE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), SVAE_ERR_ALIGN, "conv_fwd: pointers must be 16-byte aligned");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.A = x; g.W = w; g.bias = bias; g.C = y;
  g.Kc = d->c_in; g.ldA = d->ld_in; g.ldC = d->ld_out; g.ldW = d->c_out;
  g.w_tap_stride = (long long)d->c_in * d->c_out;
  g.N = d->c_out;
  g.accumulate = accumulate;
  build_plan(g, d, /*strided=*/!d->transposed, d->l_out, d->l_in);
  return launch_gather_auto<false>(g, (hipStream_t)stream, d->tile[0]);
}

// ---- the same two calls with an optional split-K workspace: problems with few output tiles and a long reduction
// (the Linear heads: M = batch rows, N = z) are split over K into `svae_conv_splitk_workspace` bytes of partial
// tiles that are summed in a fixed order.  ws == NULL or too small: identical to svae_conv_fwd / svae_conv_dgrad.
extern "C" size_t svae_conv_splitk_workspace(const svae_conv_desc* d, int kind) {
  if (validate(d) || (kind != 0 && kind != 1)) return 0;
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  if (kind == 0) { g.Kc = d->c_in; g.N = d->c_out; build_plan(g, d, !d->transposed, d->l_out, d->l_in); }
  else { g.Kc = d->c_out; g.N = d->c_in; build_plan(g, d, d->transposed != 0, d->l_in, d->l_out); }
  const int ks = splitk_factor(g, 64, 64);
  return ks >= 2 ? (size_t)ks * g.M[0] * g.N * sizeof(float) : 0;
}

extern "C" int svae_conv_fwd_ws(const svae_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                                int accumulate, void* ws, size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(!d->up2, SVAE_ERR_SHAPE, "conv_fwd: the fused x2 upsample of the input (up2) exists in the split-precision halo kernels only");
  SVAE_REQUIRE(x && w && y, SVAE_ERR_ARG, "conv_fwd: null pointer");
  SVAE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(ws), SVAE_ERR_ALIGN, "conv_fwd: pointers must be 16-byte aligned");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.A = x; g.W = w; g.bias = bias; g.C = y;
  g.Kc = d->c_in; g.ldA = d->ld_in; g.ldC = d->ld_out; g.ldW = d->c_out;
  g.w_tap_stride = (long long)d->c_in * d->c_out;
  g.N = d->c_out;
  g.accumulate = accumulate;
  build_plan(g, d, /*strided=*/!d->transposed, d->l_out, d->l_in);
  return launch_gather_auto<false>(g, (hipStream_t)stream, d->tile[0], ws, ws_bytes);
}

extern "C" int svae_conv_dgrad_ws(const svae_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(dy && w && dx, SVAE_ERR_ARG, "conv_dgrad: null pointer");
  SVAE_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx) && aligned16(ws), SVAE_ERR_ALIGN, "conv_dgrad: pointers must be 16-byte aligned");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dy; g.W = w; g.bias = nullptr; g.C = dx;
  g.Kc = d->c_out; g.ldA = d->ld_out; g.ldC = d->ld_in; g.ldW = d->c_out;
  g.w_tap_stride = (long long)d->c_in * d->c_out;
  g.N = d->c_in;
  g.accumulate = accumulate;
  build_plan(g, d, /*strided=*/d->transposed != 0, d->l_in, d->l_out);
  return launch_gather_auto<true>(g, (hipStream_t)stream, d->tile[1], ws, ws_bytes);
}

extern "C" int svae_conv_dgrad(const svae_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                               void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(dy && w && dx, SVAE_ERR_ARG, "conv_dgrad: null pointer");
  SVAE_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx), SVAE_ERR_ALIGN, "conv_dgrad: pointers must be 16-byte aligned");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dy; g.W = w; g.bias = nullptr; g.C = dx;
  g.Kc = d->c_out; g.ldA = d->ld_out; g.ldC = d->ld_in; g.ldW = d->c_out;
  g.w_tap_stride = (long long)d->c_in * d->c_out;
  g.N = d->c_in;
  g.accumulate = accumulate;
  // conv: lo = (li + pad - t*dil)/stride (fractional); convT: lo = li*stride + t*dil - pad (strided)
  build_plan(g, d, /*strided=*/d->transposed != 0, d->l_in, d->l_out);
  return launch_gather_auto<true>(g, (hipStream_t)stream, d->tile[1]);
}

namespace svae {
struct WgradGeo { int bm, bn, nsplit, nj, ctiles; long long rps, R; int fused, tg, tgroups; };

// up2 convs exist on the all-taps kernels only: that is also what tile[2] = 0 (the built-in choice) means for them
static inline svae_conv_desc wgrad_desc(const svae_conv_desc* d) {
  svae_conv_desc e = *d;
  if (e.up2 && e.tile[2] == 0) e.tile[2] = 4064128;
  return e;
}

static WgradGeo wgrad_geometry(const svae_conv_desc* d) {
  WgradGeo w;
  w.nj = d->transposed ? d->l_in : d->l_out;
  w.R = (long long)d->batch * w.nj;
  w.fused = 0; w.tg = 0; w.tgroups = 0;
  if (d->tile[2] == 1) {  // tap-fused small-weight kernel: 64x64 tiles, up to TG taps per workgroup
    w.fused = 1;
    w.bm = w.bn = 64;
    w.tg = d->kernel <= 5 ? 5 : 8;
    w.tgroups = (d->kernel + w.tg - 1) / w.tg;
    w.ctiles = (d->c_in + 63) / 64;
    const long long tiles = (long long)w.ctiles * ((d->c_out + 63) / 64) * w.tgroups;
    const long long slots = 256 * 3;
    long long want = tiles >= slots ? 1 : slots / tiles;
    const long long maxs = (w.R + 127) / 128;
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    w.rps = (w.R + want - 1) / want;
    w.rps = ((w.rps + BK - 1) / BK) * BK;
    w.nsplit = (int)((w.R + w.rps - 1) / w.rps);
    return w;
  }
  if ((d->tile[2] / 1000000) & 4) {  // all-taps split-bf16 kernel (wgrad_bf16s.hip): one workgroup per tile of ALL taps
    Tile tv;
    if (!decode_tile(d->tile[2] % 1000000, tv)) { tv.bm = 128; tv.bn = 128; }
    w.bm = tv.bm; w.bn = tv.bn;
    const int mdim = d->transposed ? d->c_out : d->c_in, ndim = d->transposed ? d->c_in : d->c_out;
    w.ctiles = (mdim + w.bm - 1) / w.bm;
    const long long tiles = (long long)w.ctiles * ((ndim + w.bn - 1) / w.bn);
    long long want = tiles >= 256 ? 1 : 256 / tiles;  // one 8-wave workgroup per CU
    const long long maxs = (w.R + 255) / 256;           // at least 256 reduction rows per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 512) want = 512;
    w.rps = (w.R + want - 1) / want;
    w.rps = ((w.rps + 31) / 32) * 32;
    w.nsplit = (int)((w.R + w.rps - 1) / w.rps);
    return w;
  }
  // tile rows run over the input channels of ONE tap: 64-row tiles when c_in has no 128 multiple
  w.bm = (d->c_in % 128 == 0) ? 128 : 64;
  const int w128 = ((d->c_out + 127) / 128) * 128, w64 = ((d->c_out + 63) / 64) * 64;
  w.bn = (w64 < w128) ? 64 : 128;
  w.ctiles = (d->c_in + w.bm - 1) / w.bm;
  long long tiles = (long long)d->kernel * w.ctiles * ((d->c_out + w.bn - 1) / w.bn);
  Tile ov;
  const int tcode = d->tile[2] > 0 ? d->tile[2] % 1000000 : 0;
  const bool flat = d->tile[2] > 0 && ((d->tile[2] / 1000000) & 16) && !((d->tile[2] / 1000000) & 4);  // taps folded into the dY columns
  const bool big = tcode == 256256 || tcode == 256128 || tcode == 128256;  // 8-wave split-bf16 tiles (wgrad_bf16s.hip)
  if (big) {
    w.bm = tcode / 1000; w.bn = tcode % 1000;
    w.ctiles = (d->c_in + w.bm - 1) / w.bm;
    tiles = (long long)d->kernel * w.ctiles * ((d->c_out + w.bn - 1) / w.bn);
  } else if (decode_tile(d->tile[2], ov)) {
    w.bm = ov.bm; w.bn = ov.bn;
    w.ctiles = (d->c_in + w.bm - 1) / w.bm;
    tiles = (long long)d->kernel * w.ctiles * ((d->c_out + w.bn - 1) / w.bn);
  } else if (w.bm == 128 && w.bn == 128 && tiles < 256 && w.R < 16384) {  // few, short tiles: go finer
    w.bm = 64;
    w.ctiles = (d->c_in + 63) / 64;
    tiles = (long long)d->kernel * w.ctiles * ((d->c_out + w.bn - 1) / w.bn);
  }
  if (flat && d->transposed) tiles = (long long)w.ctiles * (((long long)d->kernel * d->c_out + w.bn - 1) / w.bn);
  if (flat && !d->transposed) {  // taps folded into the X channel rows
    w.ctiles = (int)(((long long)d->kernel * ((d->c_in + 3) / 4 * 4) + w.bm - 1) / w.bm);
    tiles = (long long)w.ctiles * ((d->c_out + w.bn - 1) / w.bn);
  }
  // resident blocks: 3 per CU for the 128x128 tile (144 VGPR+AGPR), 4 for the smaller ones
  const long long slots = big ? 256 : 256 * ((w.bm == 128 && w.bn == 128) ? 3 : 4);  // 8-wave tiles: one workgroup per CU
  long long want = tiles >= slots ? 1 : slots / tiles;
  const long long maxs = (w.R + 127) / 128;  // at least 128 reduction rows per split
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  if (want > 512) want = 512;
  w.rps = (w.R + want - 1) / want;
  w.rps = ((w.rps + WBK - 1) / WBK) * WBK;
  w.nsplit = (int)((w.R + w.rps - 1) / w.rps);
  return w;
}
}  // namespace svae

extern "C" size_t svae_conv_wgrad_workspace(const svae_conv_desc* d) {
  if (validate(d)) return 0;
  const svae_conv_desc de = wgrad_desc(d);
  d = &de;
  const WgradGeo wg = wgrad_geometry(d);
  const int nsplit = wg.nsplit;
  const long long rows = (long long)d->batch * d->l_out;
  const long long chunks = (rows + COLSUM_ROWS - 1) / COLSUM_ROWS;
  size_t slab = nsplit > 1 ? (size_t)nsplit * d->kernel * d->c_in * d->c_out : 0;
  return (slab + (size_t)chunks * d->c_out) * sizeof(float) + 256;
}

static int conv_wgrad_impl(const svae_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                           size_t ws_bytes, int accumulate, void* stream, int pieces) {
  if (int e = validate(d)) return e;
  const svae_conv_desc de = wgrad_desc(d);
  d = &de;
  SVAE_REQUIRE(x && dy && dw, SVAE_ERR_ARG, "conv_wgrad: null pointer");
  SVAE_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dw) && aligned16(ws), SVAE_ERR_ALIGN,
               "conv_wgrad: pointers must be 16-byte aligned");
  SVAE_REQUIRE(ws_bytes >= svae_conv_wgrad_workspace(d), SVAE_ERR_WORKSPACE, "conv_wgrad: workspace %zu < %zu", ws_bytes,
               svae_conv_wgrad_workspace(d));
  hipStream_t st = (hipStream_t)stream;
  const WgradGeo wg = wgrad_geometry(d);
  const int nsplit = wg.nsplit;
  WgradArgs g;
  memset(&g, 0, sizeof(g));
  g.X = x; g.dY = dy;
  g.R = wg.R; g.rows_per_split = wg.rps; g.nj = wg.nj;
  g.Lx = d->l_in; g.Ly = d->l_out;
  g.T = d->kernel; g.Kc = d->c_in; g.N = d->c_out;
  g.ldX = d->ld_in; g.ldY = d->ld_out; g.ldW = d->c_out;
  g.ctiles = wg.ctiles;
  g.bm = wg.bm;
  for (int t = 0; t < d->kernel; ++t) {
    if (!d->transposed) { g.bx[t] = t * d->dilation - d->padding; g.by[t] = 0; }
    else { g.bx[t] = 0; g.by[t] = t * d->dilation - d->padding; }
  }
  g.sx = d->transposed ? 1 : d->stride;
  g.sy = d->transposed ? d->stride : 1;
  const long long wsize = (long long)d->kernel * d->c_in * d->c_out;
  float* slab = (float*)ws;
  if (nsplit > 1) { g.out = slab; g.slab_stride = wsize; g.accumulate = 0; }
  else { g.out = dw; g.slab_stride = 0; g.accumulate = accumulate; }
  SVAE_REQUIRE(pieces > 0 || (wg.bm <= 128 && wg.bn <= 128), SVAE_ERR_SHAPE, "conv_wgrad: tile %dx%d exists for the split-bf16 kernel only",
               wg.bm, wg.bn);
  const int wvariant = d->tile[2] / 1000000;
  SVAE_REQUIRE(!d->up2 || (pieces > 0 && (wvariant & 4)), SVAE_ERR_SHAPE,
               "conv_wgrad: the fused x2 upsample of x exists in the all-taps split kernels only (tile codes 4 / 6 / 12 / 14 BBBNNN)");
  if (pieces > 0 && (wvariant & 4)) {
    const int Ls = d->transposed ? d->l_out : d->l_in;
    // Ls = nj stride + e: e = 0 contiguous convs, +1 the (k+1)-tap skip convs (2L -> 2L - 1), -1 odd-length stride-2 (transposed) convs.
    // The stage image (one contiguous run of S rows) grows by e per sample boundary a 32-row stage can cross; it must fit the rows
    // the kernel allocates (whole staging passes of 512 / (bm / 4) rows).
    const int e_rows = Ls - wg.nj * d->stride;
    const int sr_rows = 31 * d->stride + d->kernel, pass_rows = 512 / (wg.bm / 4);
    const int sr_alloc = (sr_rows + pass_rows - 1) / pass_rows * pass_rows;
    const int crossings = (wg.nj - 1 + 31) / wg.nj;
    const int srows = sr_rows + (e_rows > 0 ? e_rows * crossings : 0);
    SVAE_REQUIRE(pieces == 2 && d->dilation == 1 && (d->kernel == 5 || d->kernel == 6) && srows <= sr_alloc &&
                 (e_rows >= 0 || (e_rows == -1 && d->stride == 2)), SVAE_ERR_SHAPE,
                 "conv_wgrad: the all-taps kernel needs 2 pieces, dilation 1, 5 or 6 taps and l = nj * stride + e with a stage image that fits "
                 "(e = %d, %d of %d rows)", e_rows, srows, sr_alloc);
    WgradTapsArgs a;
    memset(&a, 0, sizeof(a));
    a.e = e_rows; a.srows = srows; a.up = d->up2;
    if (d->up2) a.up_inv = ((1u << 20) + (unsigned)Ls - 1) / (unsigned)Ls;
    a.S = d->transposed ? dy : x; a.F = d->transposed ? x : dy;
    a.R = wg.R; a.rows_per_split = wg.rps; a.nj = wg.nj;
    a.Ls = Ls; a.ss = d->stride; a.dil = d->dilation; a.pad = d->padding; a.T = d->kernel;
    a.rowsS = (long long)d->batch * Ls;
    a.Cs = d->transposed ? d->c_out : d->c_in; a.Cf = d->transposed ? d->c_in : d->c_out;
    a.ldS = d->transposed ? d->ld_out : d->ld_in; a.ldF = d->transposed ? d->ld_in : d->ld_out; a.ldW = d->c_out;
    a.xmap = (wvariant >> 1) & 1;
    if (nsplit > 1) { a.out = slab; a.slab_stride = wsize; a.accumulate = 0; }
    else { a.out = dw; a.slab_stride = 0; a.accumulate = accumulate; }
    dim3 grid(wg.ctiles, (a.Cf + wg.bn - 1) / wg.bn, nsplit);
    if (int e = launch_wgrad_taps(a, grid, st, wg.bm, wg.bn, d->transposed ? 1 : 0, (wvariant >> 3) & 1)) return e;
  } else if (pieces > 0 && (wvariant & 16)) {
    SVAE_REQUIRE(d->kernel >= 2 && (d->transposed ? d->c_out % 4 == 0 : (((d->c_in + 3) / 4 * 4) <= d->ld_in)), SVAE_ERR_SHAPE,
                 "conv_wgrad: the flat-tap variant needs >= 2 taps and a tap's channels in whole quads");
    if (d->transposed) {  // x rows tap-independent: taps fold into the dY columns
      g.flat_np = d->c_out;
      g.N = d->kernel * d->c_out;
      dim3 grid(g.ctiles, (g.N + wg.bn - 1) / wg.bn, nsplit);
      if (int e = launch_wgrad_split(g, grid, st, wg.bm, wg.bn, pieces, wvariant & 3)) return e;
    } else {              // dY rows tap-independent: taps fold into the X channel rows (wg.ctiles counts tiles of the flat rows)
      g.flat_mp = (d->c_in + 3) / 4 * 4;
      dim3 grid(g.ctiles, (d->c_out + wg.bn - 1) / wg.bn, nsplit);
      if (int e = launch_wgrad_split(g, grid, st, wg.bm, wg.bn, pieces, wvariant & 3)) return e;
    }
  } else if (pieces > 0) {
    dim3 grid(d->kernel * g.ctiles, (d->c_out + wg.bn - 1) / wg.bn, nsplit);
    if (int e = launch_wgrad_split(g, grid, st, wg.bm, wg.bn, pieces, d->tile[2] / 1000000)) return e;
  } else if (wg.fused) {
    WgradFusedArgs f;
    memset(&f, 0, sizeof(f));
    f.R = wg.R; f.rows_per_split = wg.rps; f.nj = wg.nj;
    f.T = d->kernel; f.Kc = d->c_in; f.N = d->c_out; f.ldW = d->c_out;
    f.ctiles = wg.ctiles; f.ntiles = (d->c_out + 63) / 64;
    for (int t = 0; t < d->kernel; ++t) f.bp[t] = t * d->dilation - d->padding;
    f.sp = d->stride;
    if (!d->transposed) { f.S = dy; f.Ls = d->l_out; f.ldS = d->ld_out; f.P = x; f.Lp = d->l_in; f.ldP = d->ld_in; }
    else { f.S = x; f.Ls = d->l_in; f.ldS = d->ld_in; f.P = dy; f.Lp = d->l_out; f.ldP = d->ld_out; }
    if (nsplit > 1) { f.out = slab; f.slab_stride = wsize; f.accumulate = 0; }
    else { f.out = dw; f.slab_stride = 0; f.accumulate = accumulate; }
    dim3 fgrid(f.ctiles * f.ntiles, wg.tgroups, nsplit);
    if (!d->transposed) {
      if (wg.tg == 5) hipLaunchKernelGGL((wgrad_fused_kernel<5, true>), fgrid, dim3(256), 0, st, f);
      else hipLaunchKernelGGL((wgrad_fused_kernel<8, true>), fgrid, dim3(256), 0, st, f);
    } else {
      if (wg.tg == 5) hipLaunchKernelGGL((wgrad_fused_kernel<5, false>), fgrid, dim3(256), 0, st, f);
      else hipLaunchKernelGGL((wgrad_fused_kernel<8, false>), fgrid, dim3(256), 0, st, f);
    }
    if (int e = check_launch("wgrad_fused")) return e;
  } else {
  dim3 grid(d->kernel * g.ctiles, (d->c_out + wg.bn - 1) / wg.bn, nsplit);
  if (wg.bm == 128 && wg.bn == 128) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 128>), grid, dim3(256), 0, st, g);
  else if (wg.bm == 128 && wg.bn == 64) hipLaunchKernelGGL((wgrad_gemm_kernel<128, 64>), grid, dim3(256), 0, st, g);
  else if (wg.bm == 64 && wg.bn == 128) hipLaunchKernelGGL((wgrad_gemm_kernel<64, 128>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((wgrad_gemm_kernel<64, 64>), grid, dim3(256), 0, st, g);
  if (int e = check_launch("wgrad_gemm")) return e;
  }
  if (nsplit > 1) {
    const long long n4 = wsize / 4;
    int blocks = (int)((n4 + 63) / 64);  // 64-thread blocks: small weights still spread over many CUs
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(64), 0, st, slab, dw, n4, wsize, nsplit, accumulate);
    if (int e = check_launch("reduce_slabs")) return e;
  }
  if (db) {
    float* part = slab + (nsplit > 1 ? (size_t)nsplit * wsize : 0);
    const long long rows = (long long)d->batch * d->l_out;
    const int chunks = (int)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(chunks, (d->c_out + 63) / 64), dim3(256), 0, st, dy, rows, d->c_out,
                       d->ld_out, part);
    if (int e = check_launch("colsum_partial")) return e;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((d->c_out + 127) / 128), dim3(128), 0, st, part, chunks, d->c_out, db,
                       accumulate);
    if (int e = check_launch("colsum_final")) return e;
  }
  return SVAE_OK;
}

extern "C" int svae_conv_wgrad(const svae_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                               size_t ws_bytes, int accumulate, void* stream) {
  return conv_wgrad_impl(d, x, dy, dw, db, ws, ws_bytes, accumulate, stream, 0);
}

// same contract, products on the bf16 matrix cores with `pieces` bf16 pieces per operand (gemm_bf16s.hip);
// d->tile[2] must not select the tap-fused variant
extern "C" int svae_conv_wgrad_split(const svae_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                                     size_t ws_bytes, int accumulate, int pieces, void* stream) {
  SVAE_REQUIRE(pieces >= 1 && pieces <= 3, SVAE_ERR_ARG, "conv_wgrad_split: pieces %d not in 1..3", pieces);
  SVAE_REQUIRE(d == nullptr || d->tile[2] != 1, SVAE_ERR_ARG, "conv_wgrad_split: the tap-fused tile code is fp32 only");
  return conv_wgrad_impl(d, x, dy, dw, db, ws, ws_bytes, accumulate, stream, pieces);
}

// which tile the dispatcher picks for this problem (bench.py names the kernel template with it)
extern "C" int svae_conv_tile(const svae_conv_desc* d, int kind, int* bm, int* bn) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(bm && bn && kind >= 0 && kind <= 2, SVAE_ERR_ARG, "conv_tile: bad args");
  if (kind == 2) {
    const svae_conv_desc de = wgrad_desc(d);
    const WgradGeo wg = wgrad_geometry(&de);
    *bm = wg.fused ? -wg.tg : wg.bm;  // negative: tap-fused kernel with TG = -bm
    *bn = wg.bn;
    return SVAE_OK;
  }
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  if (kind == 0) { g.N = d->c_out; build_plan(g, d, !d->transposed, d->l_out, d->l_in); }
  else { g.N = d->c_in; build_plan(g, d, d->transposed != 0, d->l_in, d->l_out); }
  Tile t;
  if (!decode_tile(d->tile[kind], t)) t = pick_tile(g.M[0], g.M[1], g.N);
  *bm = t.bm + 1000 * t.dma;  // + 1000 * kernel variant (fp32: 1 = LDS-DMA staging)
  *bn = t.bn;
  return SVAE_OK;
}
