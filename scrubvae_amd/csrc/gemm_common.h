// Shared by the fp32 (gemm_f32.hip) and split-bf16 (gemm_bf16s.hip) implicit-GEMM kernels:
// the gather plan of a convolution and the workgroup-tile selection.
#pragma once
#include "svae_internal.h"

namespace svae {

struct GatherArgs {
  const float* A;
  const float* W;
  const float* bias;
  float* C;
  long long M[2];  // rows per phase = batch * nj[p]
  int nj[2];
  int ntaps[2];
  int base[2][SVAE_MAX_TAPS];  // li = j*sj + base
  int widx[2][SVAE_MAX_TAPS];  // weight tap of that entry
  // the same two tables in closed form (they are arithmetic progressions: base[p][i] = base0[p] + i * bstep[p], widx likewise):
  // kernels that walk the taps in order compute them instead of fetching them with a scalar load every stage
  int base0[2], bstep[2], w0[2], wstep[2];
  int blocks_m[2];
  int Lin, Lout, sj, n_phase;
  int Kc;  // reduction channels per tap, multiple of 16
  int up;  // 1 (forward of a conv behind nn.Upsample(x2, linear)): A points to the HALF-length tensor [batch][Lin / 2][ldA]; the gathered operand is its upsample (Lin rows per sample)
  int ldA, ldC, ldW;
  long long w_tap_stride;
  int N;  // padded output channels
  int accumulate;
  // split-K (fp32 gather kernel, single-phase plans with few output tiles): blockIdx.z handles K tiles
  // [z*nk/ksplit, (z+1)*nk/ksplit) and writes its partial tile to slab[z][m][N]; splitk_reduce adds them in z order
  int ksplit;
  float* slab;
  // split-bf16 gather kernels: per-row-tile BatchNorm sums of the written values, [row tiles][2][N] (NULL: off).
  // bn_x == NULL: forward statistics (sum v, sum v^2).  bn_x != NULL: the launch writes the gradient dy with respect to the
  // OUTPUT of a BatchNorm + PReLU / tanh stage whose saved input is bn_x (same row layout as C): the sums are the backward's
  // (sum du, sum du * xhat), du = dy * act'(bn_x * scale + shift), xhat = (bn_x - mean) * rstd; scale / shift / mean may be NULL
  // (bare activation); bn_alpha == NULL = tanh; bn_dalpha[blockIdx.x * gridDim.y + blockIdx.y] = sum over the tile of dy * u on
  // the PReLU's negative side (the slope's gradient partial).
  float* stats;
  const float* bn_x;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_rstd;
  const float* bn_alpha;
  float* bn_dalpha;
};

inline int validate(const svae_conv_desc* d) {
  SVAE_REQUIRE(d != nullptr, SVAE_ERR_ARG, "conv: null descriptor");
  SVAE_REQUIRE(d->batch > 0 && d->l_in > 0 && d->l_out > 0, SVAE_ERR_SHAPE, "conv: non-positive batch/length");
  SVAE_REQUIRE(d->c_in > 0 && d->c_out > 0 && d->c_in % 16 == 0 && d->c_out % 16 == 0, SVAE_ERR_SHAPE,
               "conv: padded channel counts must be positive multiples of 16 (got %d,%d)", d->c_in, d->c_out);
  SVAE_REQUIRE(d->ld_in >= d->c_in && d->ld_out >= d->c_out && d->ld_in % 4 == 0 && d->ld_out % 4 == 0, SVAE_ERR_ALIGN,
               "conv: leading dimensions must be >= channels and multiples of 4");
  SVAE_REQUIRE(d->kernel >= 1 && d->kernel <= SVAE_MAX_TAPS, SVAE_ERR_SHAPE, "conv: kernel %d not in [1,%d]", d->kernel,
               SVAE_MAX_TAPS);
  SVAE_REQUIRE(d->stride == 1 || d->stride == 2, SVAE_ERR_SHAPE, "conv: stride %d unsupported", d->stride);
  SVAE_REQUIRE(d->dilation >= 1 && d->padding >= 0, SVAE_ERR_SHAPE, "conv: bad dilation/padding");
  int expect;
  if (!d->transposed)
    expect = (d->l_in + 2 * d->padding - d->dilation * (d->kernel - 1) - 1) / d->stride + 1;
  else
    expect = (d->l_in - 1) * d->stride - 2 * d->padding + d->dilation * (d->kernel - 1) + 1;
  SVAE_REQUIRE(expect == d->l_out, SVAE_ERR_SHAPE, "conv: l_out %d != formula %d", d->l_out, expect);
  SVAE_REQUIRE(d->up2 == 0 || (d->up2 == 1 && !d->transposed && d->stride == 1 && d->l_in % 2 == 0), SVAE_ERR_SHAPE,
               "conv: up2 (input = x2 linear upsample of a half-length tensor) needs a stride-1 nn.Conv1d geometry with an even l_in");
  return SVAE_OK;
}

// Build the gather plan.  strided=true: src = j*stride + t*dil - pad (one phase);
// strided=false: src = (dst + pad - t*dil)/stride, split by dst parity.
inline void build_plan(GatherArgs& g, const svae_conv_desc* d, bool strided, int Ldst, int Lsrc) {
  g.Lin = Lsrc;
  g.Lout = Ldst;
  for (int p = 0; p < 2; ++p) { g.nj[p] = 0; g.ntaps[p] = 0; g.M[p] = 0; g.blocks_m[p] = 0; }
  if (strided) {
    g.n_phase = 1;
    g.sj = d->stride;
    g.nj[0] = Ldst;
    g.ntaps[0] = d->kernel;
    for (int t = 0; t < d->kernel; ++t) { g.base[0][t] = t * d->dilation - d->padding; g.widx[0][t] = t; }
  } else {
    const int s = d->stride;
    g.n_phase = s;
    g.sj = 1;
    for (int p = 0; p < s; ++p) {
      g.nj[p] = (Ldst - p + s - 1) / s;
      if (g.nj[p] < 0) g.nj[p] = 0;
      int n = 0;
      for (int t = 0; t < d->kernel; ++t) {
        const int num = p + d->padding - t * d->dilation;
        if (((num % s) + s) % s != 0) continue;
        g.base[p][n] = num / s;  // exact division
        g.widx[p][n] = t;
        ++n;
      }
      g.ntaps[p] = n;
    }
  }
  for (int p = 0; p < g.n_phase; ++p) g.M[p] = (long long)d->batch * g.nj[p];
  for (int p = 0; p < 2; ++p) {
    g.base0[p] = g.ntaps[p] > 0 ? g.base[p][0] : 0;
    g.w0[p] = g.ntaps[p] > 0 ? g.widx[p][0] : 0;
    g.bstep[p] = g.ntaps[p] > 1 ? g.base[p][1] - g.base[p][0] : 0;
    g.wstep[p] = g.ntaps[p] > 1 ? g.widx[p][1] - g.widx[p][0] : 0;
  }
}

// true when the tap tables are the arithmetic progressions base0 + i * bstep / w0 + i * wstep (always, by construction: checked)
inline bool plan_is_affine(const GatherArgs& g) {
  for (int p = 0; p < g.n_phase; ++p)
    for (int i = 0; i < g.ntaps[p]; ++i)
      if (g.base[p][i] != g.base0[p] + i * g.bstep[p] || g.widx[p][i] != g.w0[p] + i * g.wstep[p]) return false;
  return true;
}

// ---- tile selection.  Per-block work is MFMA-bound and co-resident blocks hide each other's
// barrier / LDS-fill stalls, so prefer the largest tile that still gives >= 2 blocks per CU;
// below that, more (smaller) blocks win.  Scores are relative throughput estimates.
struct Tile { int bm, bn; int dma = 0; };

inline double tile_score(long long M0, long long M1, int N, int bm, int bn) {
  const long long bmk = (M0 + bm - 1) / bm + (M1 + bm - 1) / bm;
  const long long bnk = (N + bn - 1) / bn;
  const long long blocks = bmk * bnk;
  if (blocks == 0) return 0.0;
  const double useful = (double)(M0 + M1) * N / ((double)bmk * bm * bnk * bn);
  const double resident = 256.0 * (bm * bn >= 128 * 128 ? 2.0 : 3.0);  // blocks the chip holds at once
  const double waves = (double)((blocks + (long long)resident - 1) / (long long)resident);
  const double fill = (double)blocks / (waves * resident);
  const double occ = blocks >= 512 ? 1.0 : (blocks >= 256 ? 0.8 : 0.8 * blocks / 256.0);
  const double teff = (bm * bn >= 128 * 128) ? 1.0 : (bm * bn >= 64 * 128 ? 0.93 : 0.85);
  return useful * teff * occ * (0.5 + 0.5 * fill);
}

inline Tile pick_tile(long long M0, long long M1, int N) {
  const Tile cand[4] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
  Tile best = cand[0];
  double bs = -1.0;
  for (const Tile& t : cand) {
    const double sc = tile_score(M0, M1, N, t.bm, t.bn);
    if (sc > bs * 1.001) { bs = sc; best = t; }
  }
  return best;
}

inline bool decode_tile(int code, Tile& t) {
  if (code <= 0) return false;
  t.dma = code / 1000000;  // VBBBNNN: kernel variant (fp32 gather: 1 = LDS-DMA staging; split-bf16: see gemm_bf16s.hip)
  code %= 1000000;
  t.bm = code / 1000;
  t.bn = code % 1000;
  return (t.bm == 64 || t.bm == 128) && (t.bn == 64 || t.bn == 128);
}

struct WgradArgs {
  const float* X;
  const float* dY;
  float* out;  // slab base [nsplit][T][Kc][ldW] or dw itself when nsplit == 1
  long long R;  // reduction rows = batch * nj
  long long rows_per_split;
  long long slab_stride;
  int nj, Lx, Ly, sx, sy;
  int bx[SVAE_MAX_TAPS], by[SVAE_MAX_TAPS];
  int T, Kc, N, ldX, ldY, ldW, ctiles, bm;
  int accumulate;
  int xmap;  // split-bf16 kernel: XCD-aware workgroup order (tile code variants 2 / 3)
  int flat_mp;  // split-bf16 kernel, > 0: the taps are folded into the X channels (M = T * flat_mp: convs, dY rows tap-independent; variant bit 16)
  int flat_np;  // split-bf16 kernel, > 0: the taps are folded into the dY columns (N = T * flat_np, variant bit 16; see wgrad_bf16s.hip)
};

// all-taps split-bf16 weight-gradient kernel (wgrad_bf16s.hip): grid (row tiles, column tiles, splits)
struct WgradTapsArgs {
  const float* S;       // shifted operand (x of a conv, dY of a transposed conv): row b Ls + j ss + t dil - pad
  const float* F;       // fixed operand (dY of a conv, x of a transposed conv): row b nj + j
  float* out;           // slab base [split][T][c_in][ldW] or dw itself
  long long R, rows_per_split, slab_stride, rowsS;
  int nj, Ls, ss, dil, pad, T;
  int Cs, Cf, ldS, ldF, ldW;
  int accumulate, xmap;
  int e;       // Ls - nj ss: rows a sample of S has beyond the contiguous geometry (0, +1: skip convs, -1: odd-length stride-2 convs)
  int srows;   // image rows of a stage: (32 - 1) ss + T + max(e, 0) * (sample boundaries a stage can cross)
  unsigned up_inv;  // up: ceil(2^20 / Ls) (row -> (sample, position) by multiply-shift)
  int up;      // 1: S points to the HALF-length tensor [batch][Ls / 2][Cs]; the kernel's shifted operand is its x2 linear upsample (Ls rows per sample)
};
int launch_wgrad_taps(const WgradTapsArgs& g, dim3 grid, hipStream_t st, int bm, int bn, int trans_out, int m16);
// split-bf16 weight-gradient main kernel (wgrad_bf16s.hip); same grid / slabs as wgrad_gemm_kernel
int launch_wgrad_split(const WgradArgs& g, dim3 grid, hipStream_t st, int bm, int bn, int pieces, int variant);

}  // namespace svae
