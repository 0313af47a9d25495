// Shared by the forward / data-gradient translation units of the split-precision implicit GEMM (gemm_bf16s.hip: per-tap gather
// kernels, the 8-wave halo kernel, dispatch and C entry points; halo_ws_bf16s.hip: the wave-specialised halo kernels): launch
// arguments, the epilogue (bias, accumulate, fused BatchNorm sums), the halo-image row bound.
#pragma once
#include "split_common.h"

namespace svae {

constexpr int SBK = 32;  // K depth of one LDS stage: a 128-byte line of every gathered fp32 row

struct SplitGatherArgs {
  GatherArgs g;               // g.W / g.ldW / g.w_tap_stride unused
  const unsigned short* Wp;   // weight pieces, pre-tiled [piece][tap][k/32][n][32] (zero padded in k)
  long long w_piece_stride;   // elements between piece planes
  long long rowsA;            // rows of the gathered operand (batch * Lin): bound of the halo image
  int KB;                     // 32-deep k blocks per tap
  float* up_out;              // g.up: where the upsampled operand rows go as a by-product ([rowsA][ldA], NULL: nowhere)
};

// ---- epilogue shared by the gather kernels: C (+)= acc + bias, and -- when g.stats != NULL -- the BatchNorm batch statistics of
// the values just written (reference: nn.BatchNorm1d in train mode right behind the conv, residual.py:88,112,146,173): per-column
// (sum v, sum v^2) over the tile's valid rows go to stats[blockIdx.x][2][N], the layout bn_stats_partial writes per 128-row chunk,
// so the finalize kernel sums row tiles instead of chunks and the separate statistics pass over the conv output disappears.
// Fixed summation order (lane rows, the two half-waves, then the WR row-waves): bit-reproducible.
template <int MT, int NT, int WM, int WN, int WR, int BN>
__device__ __forceinline__ void tile_epilogue(const GatherArgs& g, f32x16 (&acc)[MT][NT], const long long* rowoff, int n0, int wr, int wc,
                                              int lr, int h, float* red, int tid, int nth, float oscale = 1.f, int tile_x = -1,
                                              int tile_y = 0) {
  if (tile_x < 0) { tile_x = blockIdx.x; tile_y = blockIdx.y; }  // (row tile, column tile) of this workgroup
  float cs[NT], cq[NT];
  double da = 0.0;  // the PReLU slope's partial: a sum of ~1e6 cancelling terms over the launch -- fp64 products and sums
  const bool bwd = g.bn_x != nullptr;        // uniform
  const bool th = g.bn_alpha == nullptr;     // tanh instead of PReLU
  const float slope = (bwd && !th) ? g.bn_alpha[0] : 0.f;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int col = n0 + wc * WN + nt * 32 + lr;
    cs[nt] = 0.f;
    cq[nt] = 0.f;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
    float sc = 1.f, sh = 0.f, mu = 0.f, rs = 0.f;
    if (bwd) {
      if (g.bn_scale) { sc = g.bn_scale[col]; sh = g.bn_shift[col]; }
      if (g.bn_mean) { mu = g.bn_mean[col]; rs = g.bn_rstd[col]; }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const long long off = rowoff[row];
        if (off >= 0) {
          float* dst = g.C + off + col;
          float v = acc[mt][nt][r] * oscale + bv;
          if (g.accumulate) v += *dst;
          *dst = v;
          if (bwd) {
            const float x = g.bn_x[off + col];
            const float u = x * sc + sh;
            float du;
            if (th) { const float t = tanhf(u); du = v * (1.f - t * t); }
            else { du = u > 0.f ? v : slope * v; if (!(u > 0.f)) da += (double)v * (double)u; }
            cs[nt] += du;
            cq[nt] += du * (x - mu) * rs;
          } else {
            cs[nt] += v;
            cq[nt] += v * v;
          }
        }
      }
    }
  }
  if (g.stats == nullptr) return;  // uniform
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    cs[nt] += __shfl_xor(cs[nt], 32, 64);
    cq[nt] += __shfl_xor(cq[nt], 32, 64);
  }
  __syncthreads();  // every wave is past its last LDS operand read: the staging buffers are free
  if (h == 0) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c = wc * WN + nt * 32 + lr;
      red[(0 * WR + wr) * BN + c] = cs[nt];
      red[(1 * WR + wr) * BN + c] = cq[nt];
    }
  }
  float* dred = red + 2 * WR * BN;  // one (hi, lo) slot per wave for the slope partial
  if (bwd && g.bn_dalpha) {
    da = wave_sum_d(da);
    if ((tid & 63) == 0) { const float hi = (float)da; dred[2 * (tid >> 6)] = hi; dred[2 * (tid >> 6) + 1] = (float)(da - (double)hi); }
  }
  __syncthreads();
  for (int i = tid; i < 2 * BN; i += nth) {
    const int k = i / BN, c = i - k * BN;
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WR; ++w) t += red[(k * WR + w) * BN + c];
    if (n0 + c < g.N) g.stats[((long long)tile_x * 2 + k) * g.N + n0 + c] = t;
  }
  if (bwd && g.bn_dalpha && tid == 0) {  // the tile's partial leaves as a (hi, lo) float pair: the reduction kernels sum partials in fp64
    double t = 0.0;
    for (int w = 0; w < nth / 64; ++w) t += (double)dred[2 * w] + (double)dred[2 * w + 1];
    const float hi = (float)t;
    float* o = g.bn_dalpha + 2 * ((long long)tile_x * gridDim.y + tile_y);
    o[0] = hi;
    o[1] = (float)(t - (double)hi);
  }
}

// upper bound of the halo image rows over all BM-row tiles of both phases
inline int halo_rows(const GatherArgs& g, int bm) {
  int worst = 0;
  for (int p = 0; p < 2; ++p) {
    if (g.M[p] <= 0 || g.ntaps[p] <= 0) continue;
    int bmin = 1 << 30, bmax = -(1 << 30);
    for (int t = 0; t < g.ntaps[p]; ++t) {
      bmin = g.base[p][t] < bmin ? g.base[p][t] : bmin;
      bmax = g.base[p][t] > bmax ? g.base[p][t] : bmax;
    }
    const int nj = g.nj[p];
    int extra = g.Lin - nj * g.sj;  // additional anchor step at a sample boundary
    if (extra < 0) extra = 0;
    const int crossings = bm >= 2 ? (bm - 2) / nj + 1 : 0;
    const int span = (bm - 1) * g.sj + crossings * extra + (bmax - bmin) + 1;
    worst = span > worst ? span : worst;
  }
  return worst;
}

// wave-specialised halo kernels (halo_ws_bf16s.hip): tile codes V = 10 .. 18 (and the diagnostic variants of the ablation build).
// Returns SVAE_OK / an error; *handled = false when the code is none of theirs.
int launch_split_halo_ws(SplitGatherArgs& sa, hipStream_t st, const Tile& t, int code, int pieces, bool* handled);

}  // namespace svae
