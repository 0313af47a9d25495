// Fused MLP-ensemble kernels for gfx950 (MI355X): the scrubber heads of the SC-VAE.
//
// Reference: MLPEnsemble (src/scrubvae/model/disentangle.py:583-632) = four small MLPs on the same input,
// (in,in,in,out), (in,in,out), (in,in,in/2,out), (in,2in,2in,out) with ReLU between; used by GRScrubber (:635-660) on mu and by
// AdvNetScrubber (:663-684) on cat([mu;mu], [v;v_shuffle]).  ~22 kFLOP per window: launch- and latency-bound, not MFMA
// material.  One launch runs the whole ensemble forward, one the whole backward (recompute + data-gradients + weight
// gradients), where the GEMM-per-Linear path needed 11 GEMM + 7 ReLU launches forward and ~40 backward.
//
// Work decomposition: workgroup = (tile of 64 rows, ensemble member); lane = row.  Activations of the tile live in LDS
// column-major ([feature][row], 65-float columns: lane-contiguous for the per-row passes and conflict-free for the
// weight-gradient pass whose lanes index features).  A wave computes 8 output features of its 64 rows at a time: per input
// feature one ds_read_b32 (its row's activation) and 8 v_fma with the weights as SGPR operands -- W[k][n0..n0+7] is the same
// for every lane, so it is fetched by s_load_dwordx8 through the scalar cache, not through the vector memory path or LDS.
// fp32 FMA throughout (the same arithmetic as the reference's fp32 Linear layers, summation order k ascending).
//
// Inputs are assembled in-kernel: columns [0,n0) from src0[b], [n0,n0+n1) from src1[b]; with halves = 2 the rows b and B + b are
// lanes l and 32 + l of the same workgroup and the second copy reads column shuf_col of src1 from row perm[b]
// (AdvNetScrubber.shuffle, disentangle.py:678-684).  The backward writes per-member input gradients and per-tile weight-
// gradient partials to a workspace; ens_finish_kernel sums them in a fixed order (bit-reproducible, no atomics) into
// the parameter gradients and into d_src0 (+= coef * sum over members and halves: gradient reversal = coef -alpha).
#include "svae_internal.h"

namespace svae {

constexpr int ENS_LD = 65;      // floats per LDS column: 64 rows + 1 pad
constexpr int ENS_ROWS = 64;    // rows per workgroup
constexpr int ENS_THREADS = 512;
constexpr int ENS_WAVES = ENS_THREADS / 64;
constexpr int ENS_CW = 8;       // output features per wave chunk

struct EnsLayerDev {
  const float* w;   // [K][N]
  const float* b;   // [N]
  int K, N;
  int in_col, out_col, g_col;  // LDS columns: input activations, output activations (hidden layers), gradient wrt the output
  long long ws_off;            // floats: per-tile weight-gradient partials [tiles][K*N + N] in the workspace
};

struct EnsMemberDev {
  EnsLayerDev l[SVAE_ENS_MAX_LAYERS];
  int nl;
  float* out;
  const float* d_out;
  int param_grads;
};

struct EnsArgs {
  EnsMemberDev m[SVAE_ENS_MEMBERS];
  const float* src0; const float* src1;
  const long long* perm;
  const float* shuf_vals;
  int ld0, n0, ld1, n1, shuf_col;
  int B, halves, in_p;
  float* ws;            // backward workspace
  long long gx_off;     // floats: input gradients [member][rows][in_p]
  int tiles;
};

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// Weights and biases are read-only for the whole launch and indexed by wave-uniform values: viewed through the constant
// address space their loads become s_load_dwordx8 (scalar cache, SGPR operands of the FMAs) instead of 64-lane vector
// loads of one address.
typedef const float __attribute__((address_space(4))) cfloat;
__device__ __forceinline__ cfloat* as_const(const float* p) { return (cfloat*)(unsigned long long)p; }

// this lane's row: base sample index b (or -1) and the row of the [rows = B * halves] output tensors
__device__ __forceinline__ void lane_row(const EnsArgs& a, int lane, int& b, int& half, long long& row) {
  const int rt = ENS_ROWS / a.halves;
  half = lane / rt;
  b = blockIdx.x * rt + (lane - half * rt);
  if (b >= a.B) b = -1;
  row = b < 0 ? -1 : (long long)half * a.B + b;
}

// stage the assembled input of the tile into LDS columns [0, in_p)
__device__ __forceinline__ void stage_input(const EnsArgs& a, float* smem, int wave, int lane, int b, int half) {
  for (int c = wave; c < a.in_p; c += ENS_WAVES) {
    float v = 0.f;
    if (b >= 0) {
      if (c < a.n0) v = a.src0[(long long)b * a.ld0 + c];
      else if (c < a.n0 + a.n1) {
        const int cc = c - a.n0;
        if (half == 1 && cc == a.shuf_col) {
          v = a.shuf_vals ? a.shuf_vals[b] : a.src1[a.perm[b] * a.ld1 + cc];
        } else {
          v = a.src1[(long long)b * a.ld1 + cc];
        }
      }
    }
    smem[c * ENS_LD + lane] = v;
  }
}

// one Linear (+ ReLU) of the member for the tile: out[n][row] = b[n] + sum_k in[k][row] * W[k][n]
template <bool FINAL_TO_GLOBAL>
__device__ __forceinline__ void layer_fwd(const EnsLayerDev& L, float* smem, int wave, int lane, bool last, float* out, long long row) {
  const int nch = L.N / ENS_CW;
  for (int ch = wave; ch < nch; ch += ENS_WAVES) {
    const int n0 = uniform(ch * ENS_CW);
    float acc[ENS_CW];
#pragma unroll
    for (int j = 0; j < ENS_CW; ++j) acc[j] = as_const(L.b)[n0 + j];
    cfloat* wk = as_const(L.w) + n0;
    const float* __restrict__ in = smem + L.in_col * ENS_LD + lane;
#pragma unroll 4
    for (int k = 0; k < L.K; ++k) {
      const float x = in[k * ENS_LD];
#pragma unroll
      for (int j = 0; j < ENS_CW; ++j) acc[j] = fmaf(x, wk[(long long)k * L.N + j], acc[j]);
    }
    if (last) {
      if (FINAL_TO_GLOBAL && row >= 0) {
        float4* o = reinterpret_cast<float4*>(out + row * L.N + n0);
        o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < ENS_CW; ++j) smem[(L.out_col + n0 + j) * ENS_LD + lane] = fmaxf(acc[j], 0.f);
    }
  }
}

__global__ __launch_bounds__(ENS_THREADS) void ens_fwd_kernel(const EnsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = uniform(threadIdx.x >> 6);
  const EnsMemberDev& m = a.m[blockIdx.y];
  int b, half;
  long long row;
  lane_row(a, lane, b, half, row);
  stage_input(a, smem, wave, lane, b, half);
  __syncthreads();
  for (int li = 0; li < m.nl; ++li) {
    layer_fwd<true>(m.l[li], smem, wave, lane, li == m.nl - 1, m.out, row);
    __syncthreads();
  }
}

// backward of one member for one tile: recompute the hidden activations, then per layer (last to first) the weight /
// bias gradient partials of the tile and the gradient wrt the layer input (ReLU mask from the recomputed activation).
__global__ __launch_bounds__(ENS_THREADS) void ens_bwd_kernel(const EnsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = uniform(tid >> 6);
  const int mi = blockIdx.y;
  const EnsMemberDev& m = a.m[mi];
  int b, half;
  long long row;
  lane_row(a, lane, b, half, row);
  stage_input(a, smem, wave, lane, b, half);
  {  // gradient wrt the member's output: zero for rows past the batch (their recomputed activations then contribute nothing)
    const EnsLayerDev& L = m.l[m.nl - 1];
    for (int n = wave; n < L.N; n += ENS_WAVES) smem[(L.g_col + n) * ENS_LD + lane] = row >= 0 ? m.d_out[row * L.N + n] : 0.f;
  }
  __syncthreads();
  for (int li = 0; li + 1 < m.nl; ++li) {
    layer_fwd<false>(m.l[li], smem, wave, lane, false, nullptr, row);
    __syncthreads();
  }
  const long long rows = (long long)a.B * a.halves;
  for (int li = m.nl - 1; li >= 0; --li) {
    const EnsLayerDev& L = m.l[li];
    const float* __restrict__ g = smem + L.g_col * ENS_LD;
    const float* __restrict__ x = smem + L.in_col * ENS_LD;
    if (m.param_grads) {
      // dW[k][n] = sum_rows x[k][row] * g[n][row]: a thread owns 4 x 4 entries and walks the 64 rows
      float* __restrict__ pw = a.ws + L.ws_off + (long long)blockIdx.x * ((long long)L.K * L.N + L.N);
      const int nb4 = L.N / 4, blocks = (L.K / 4) * nb4;
      for (int blk = tid; blk < blocks; blk += ENS_THREADS) {
        const int kb = blk / nb4, nb = blk - kb * nb4;
        const float* xr = x + kb * 4 * ENS_LD;
        const float* gr = g + nb * 4 * ENS_LD;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
#pragma unroll 4
        for (int r = 0; r < ENS_ROWS; ++r) {
          float xv[4], gv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) { xv[i] = xr[i * ENS_LD + r]; gv[i] = gr[i * ENS_LD + r]; }
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(xv[i], gv[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(pw + (long long)(kb * 4 + i) * L.N + nb * 4) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
      }
      for (int n = tid; n < L.N; n += ENS_THREADS) {  // bias gradient partial
        float s = 0.f;
        for (int r = 0; r < ENS_ROWS; ++r) s += g[n * ENS_LD + r];
        pw[(long long)L.K * L.N + n] = s;
      }
    }
    // gradient wrt the layer input: gin[k][row] = sum_n g[n][row] * W[k][n]
    const int kch = L.K / ENS_CW;
    for (int ch = wave; ch < kch; ch += ENS_WAVES) {
      const int k0 = uniform(ch * ENS_CW);
      float acc[ENS_CW];
#pragma unroll
      for (int j = 0; j < ENS_CW; ++j) acc[j] = 0.f;
      cfloat* wk = as_const(L.w) + (long long)k0 * L.N;
      for (int n = 0; n < L.N; n += 4) {
        float gv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) gv[i] = g[(n + i) * ENS_LD + lane];
#pragma unroll
        for (int j = 0; j < ENS_CW; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[j] = fmaf(gv[i], wk[(long long)j * L.N + n + i], acc[j]);
      }
      if (li > 0) {
        const EnsLayerDev& P = m.l[li - 1];
#pragma unroll
        for (int j = 0; j < ENS_CW; ++j) {
          const float act = smem[(L.in_col + k0 + j) * ENS_LD + lane];
          smem[(P.g_col + k0 + j) * ENS_LD + lane] = act > 0.f ? acc[j] : 0.f;
        }
      } else if (row >= 0) {
        float4* o = reinterpret_cast<float4*>(a.ws + a.gx_off + ((long long)mi * rows + row) * a.in_p + k0);
        o[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        o[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      }
    }
    __syncthreads();
  }
}

struct EnsFinishArgs {
  // parameter gradients: segment s = one (member, layer): count = K*N + N partial floats per tile
  const float* ws;
  long long seg_ws[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS];
  float* seg_dw[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS];
  float* seg_db[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS];
  int seg_kn[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS], seg_n[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS];
  int seg_first[SVAE_ENS_MEMBERS * SVAE_ENS_MAX_LAYERS + 1];  // first 256-thread block of each segment
  int n_seg, tiles, accumulate;
  // input gradient
  long long gx_off;
  int n_members, B, halves, in_p, n0;
  float* d_src0; int ld_d; float coef;
  float* gx_raw;       // [B * halves][in_p] or NULL
  int gx_first;        // first block of the input-gradient part
};

__global__ __launch_bounds__(256) void ens_finish_kernel(const EnsFinishArgs f) {
  const int bx = blockIdx.x;
  if (bx < f.gx_first) {
    int s = 0;
    while (s + 1 < f.n_seg && bx >= f.seg_first[s + 1]) ++s;
    const int e = (bx - f.seg_first[s]) * 256 + threadIdx.x;
    const int cnt = f.seg_kn[s] + f.seg_n[s];
    if (e >= cnt) return;
    const float* p = f.ws + f.seg_ws[s] + e;
    float sum = 0.f;
    for (int t = 0; t < f.tiles; ++t) sum += p[(long long)t * cnt];
    float* dst = e < f.seg_kn[s] ? f.seg_dw[s] + e : f.seg_db[s] + (e - f.seg_kn[s]);
    *dst = (f.accumulate ? *dst : 0.f) + sum;
    return;
  }
  const long long i = (long long)(bx - f.gx_first) * 256 + threadIdx.x;
  const long long rows = (long long)f.B * f.halves;
  if (f.gx_raw) {
    if (i < rows * f.in_p) {
      float s = 0.f;
      for (int m = 0; m < f.n_members; ++m) s += f.ws[f.gx_off + (long long)m * rows * f.in_p + i];
      f.gx_raw[i] = s;
    }
  }
  if (f.d_src0 && i < (long long)f.B * f.n0) {
    const long long b = i / f.n0;
    const int k = (int)(i - b * f.n0);
    float s = 0.f;
    for (int h = 0; h < f.halves; ++h)
      for (int m = 0; m < f.n_members; ++m) s += f.ws[f.gx_off + ((long long)m * rows + (long long)h * f.B + b) * f.in_p + k];
    f.d_src0[b * f.ld_d + k] += f.coef * s;
  }
}

// ---- losses of all four members in one launch (losses.py:267-309)
// kind 0: sum of squared errors vs target [rows][ld_t] (first C columns); 1: cross entropy vs int labels; 2: AdvNet: CrossEntropy
// applied to the softmax output against class = (row >= rows/2) (double softmax quirk, disentangle.py:675 + losses.py:304-307).
// part[m * nblocks + block] = lw[m] * (sum over the block's rows); dpred[m] = gs[m] * d loss / d out.
struct EnsLossArgs {
  const float* out[SVAE_ENS_MEMBERS];
  float* dpred[SVAE_ENS_MEMBERS];
  float lw[SVAE_ENS_MEMBERS], gs[SVAE_ENS_MEMBERS];
  const float* target; const int* labels;
  float* part;
  int rows, C, ld, ld_t, kind, n_members;
};

__global__ __launch_bounds__(256) void ens_loss_kernel(const EnsLossArgs a) {
  __shared__ float red4[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  for (int m = 0; m < a.n_members; ++m) {
    float loss = 0.f;
    if (r < a.rows) {
      const float* x = a.out[m] + (long long)r * a.ld;
      float* d = a.dpred[m] ? a.dpred[m] + (long long)r * a.ld : nullptr;
      const float gs = a.gs[m];
      if (a.kind == 0) {
        for (int c = 0; c < a.C; ++c) {
          const float e = x[c] - a.target[(long long)r * a.ld_t + c];
          loss += e * e;
          if (d) d[c] = 2.f * gs * e;
        }
      } else if (a.kind == 1) {
        float mx = x[0];
        for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, x[c]);
        float se = 0.f;
        for (int c = 0; c < a.C; ++c) se += expf(x[c] - mx);
        const float lse = mx + logf(se);
        const int y = a.labels[r];
        loss = lse - x[y];
        if (d)
          for (int c = 0; c < a.C; ++c) d[c] = gs * (expf(x[c] - lse) - (c == y ? 1.f : 0.f));
      } else {
        const float x0 = x[0], x1 = x[1];
        const float mx = fmaxf(x0, x1);
        const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
        const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
        const float m2 = fmaxf(p0, p1);
        const float lse = m2 + logf(expf(p0 - m2) + expf(p1 - m2));
        const int cls = r >= a.rows / 2 ? 1 : 0;
        loss = lse - (cls ? p1 : p0);
        if (d) {
          const float q0 = expf(p0 - lse), q1 = expf(p1 - lse);
          const float g0 = q0 - (cls == 0 ? 1.f : 0.f), g1 = q1 - (cls == 1 ? 1.f : 0.f);
          const float dot = g0 * p0 + g1 * p1;
          d[0] = gs * p0 * (g0 - dot);
          d[1] = gs * p1 * (g1 - dot);
        }
      }
    }
    const float t = block_sum_256(loss, red4);
    if (threadIdx.x == 0) a.part[m * gridDim.x + blockIdx.x] = a.lw[m] * t;
  }
}

// ------------------------------------------------------------------------------ host side
static int build(const svae_ens_desc* d, EnsArgs& a, bool bwd, size_t* smem_bytes) {
  SVAE_REQUIRE(d, SVAE_ERR_ARG, "ens: null descriptor");
  SVAE_REQUIRE(d->n_members >= 1 && d->n_members <= SVAE_ENS_MEMBERS, SVAE_ERR_ARG, "ens: 1..%d members", SVAE_ENS_MEMBERS);
  SVAE_REQUIRE(d->batch > 0 && (d->halves == 1 || d->halves == 2), SVAE_ERR_ARG, "ens: batch > 0, halves in {1,2}");
  SVAE_REQUIRE(d->src0 && d->n0 > 0 && d->ld0 >= d->n0 && d->n1 >= 0 && (d->n1 == 0 || (d->src1 && d->ld1 >= d->n1)), SVAE_ERR_ARG,
               "ens: bad input segments");
  SVAE_REQUIRE(d->halves == 1 || ((d->perm || d->shuf_vals) && d->shuf_col >= 0 && d->shuf_col < d->n1), SVAE_ERR_ARG,
               "ens: halves = 2 needs a permutation (or the shuffled values) and a shuffled column of src1");
  memset(&a, 0, sizeof(a));
  a.src0 = d->src0; a.src1 = d->src1; a.perm = d->perm; a.shuf_vals = d->shuf_vals;
  a.ld0 = d->ld0; a.n0 = d->n0; a.ld1 = d->ld1; a.n1 = d->n1; a.shuf_col = d->halves == 2 ? d->shuf_col : -1;
  a.B = d->batch; a.halves = d->halves;
  const int rt = ENS_ROWS / d->halves;
  a.tiles = (d->batch + rt - 1) / rt;
  const int in_p = d->member[0].layer[0].K;
  SVAE_REQUIRE(in_p >= d->n0 + d->n1 && in_p % 16 == 0, SVAE_ERR_SHAPE, "ens: padded input width %d < %d columns", in_p, d->n0 + d->n1);
  a.in_p = in_p;
  int worst = 0;
  long long ws = 0;
  const long long rows = (long long)d->batch * d->halves;
  for (int mi = 0; mi < d->n_members; ++mi) {
    const svae_ens_member& M = d->member[mi];
    SVAE_REQUIRE(M.n_layers >= 1 && M.n_layers <= SVAE_ENS_MAX_LAYERS, SVAE_ERR_ARG, "ens: member %d has %d layers", mi, M.n_layers);
    EnsMemberDev& o = a.m[mi];
    o.nl = M.n_layers;
    o.out = M.out; o.d_out = M.d_out;
    int col = in_p, prev = 0, prev_n = in_p;
    bool grads = true;
    for (int li = 0; li < M.n_layers; ++li) {
      const svae_ens_layer& L = M.layer[li];
      SVAE_REQUIRE(L.w && L.b && L.K > 0 && L.N > 0 && L.K % 16 == 0 && L.N % 16 == 0 && L.K == prev_n, SVAE_ERR_SHAPE,
                   "ens: member %d layer %d: K=%d N=%d must be multiples of 16 and chain", mi, li, L.K, L.N);
      SVAE_REQUIRE(aligned16(L.w) && aligned16(L.b), SVAE_ERR_ALIGN, "ens: weights must be 16-byte aligned");
      EnsLayerDev& q = o.l[li];
      q.w = L.w; q.b = L.b; q.K = L.K; q.N = L.N;
      q.in_col = prev;
      q.out_col = li + 1 < M.n_layers ? col : -1;
      if (li + 1 < M.n_layers) { prev = col; col += L.N; }
      prev_n = L.N;
      grads = grads && L.dw && L.db;
    }
    if (bwd) {
      for (int li = 0; li < M.n_layers; ++li) { o.l[li].g_col = col; col += o.l[li].N; }
      o.param_grads = grads ? 1 : 0;
      if (grads)
        for (int li = 0; li < M.n_layers; ++li) {
          o.l[li].ws_off = ws;
          ws += (long long)a.tiles * ((long long)o.l[li].K * o.l[li].N + o.l[li].N);
        }
      SVAE_REQUIRE(M.d_out, SVAE_ERR_ARG, "ens_bwd: member %d has no output gradient", mi);
    } else {
      SVAE_REQUIRE(M.out && aligned16(M.out), SVAE_ERR_ARG, "ens_fwd: member %d has no (aligned) output", mi);
    }
    worst = col > worst ? col : worst;
  }
  a.gx_off = (ws + 3) / 4 * 4;
  ws = a.gx_off + (long long)d->n_members * rows * in_p;
  *smem_bytes = (size_t)worst * ENS_LD * sizeof(float);
  SVAE_REQUIRE(*smem_bytes <= 160 * 1024, SVAE_ERR_SHAPE, "ens: %zu B of LDS per workgroup exceed 160 KiB (layers too wide for the fused kernel)",
               *smem_bytes);
  a.ws = nullptr;
  // the caller reads the workspace size back through gx_off + ...: stash it in tiles? no: recomputed by the callers below
  return SVAE_OK;
}

static long long ws_floats(const EnsArgs& a, int n_members) {
  return a.gx_off + (long long)n_members * a.B * a.halves * a.in_p;
}

template <typename K>
static int set_lds(K kernel, DeviceOnce& once, const char* what) {
  int dev;
  if (once.need(&dev)) {  // per device: the attribute belongs to the current device's copy of the kernel
    const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SVAE_REQUIRE(e == hipSuccess, SVAE_ERR_LAUNCH, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", what, hipGetErrorString(e));
    once.done(dev);
  }
  return SVAE_OK;
}

}  // namespace svae

using namespace svae;

extern "C" int svae_ens_fwd(const svae_ens_desc* d, void* stream) {
  EnsArgs a;
  size_t smem = 0;
  if (int e = build(d, a, false, &smem)) return e;
  static DeviceOnce attr;
  if (int e = set_lds(ens_fwd_kernel, attr, "ens_fwd")) return e;
  hipLaunchKernelGGL(ens_fwd_kernel, dim3(a.tiles, d->n_members), dim3(ENS_THREADS), smem, (hipStream_t)stream, a);
  return check_launch("ens_fwd");
}

extern "C" size_t svae_ens_bwd_workspace(const svae_ens_desc* d) {
  EnsArgs a;
  size_t smem = 0;
  if (build(d, a, true, &smem)) return 0;
  return (size_t)ws_floats(a, d->n_members) * sizeof(float) + 256;
}

extern "C" int svae_ens_bwd(const svae_ens_desc* d, float* d_src0, int ld_d, float coef, float* gx_raw, void* ws, size_t ws_bytes,
                            int accumulate_param_grads, void* stream) {
  EnsArgs a;
  size_t smem = 0;
  if (int e = build(d, a, true, &smem)) return e;
  SVAE_REQUIRE(ws && aligned16(ws) && ws_bytes >= (size_t)ws_floats(a, d->n_members) * sizeof(float), SVAE_ERR_WORKSPACE,
               "ens_bwd: workspace too small (need %zu B)", (size_t)ws_floats(a, d->n_members) * sizeof(float));
  SVAE_REQUIRE(!d_src0 || ld_d >= d->n0, SVAE_ERR_ARG, "ens_bwd: ld of d_src0 %d < %d", ld_d, d->n0);
  a.ws = (float*)ws;
  static DeviceOnce attr;
  if (int e = set_lds(ens_bwd_kernel, attr, "ens_bwd")) return e;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ens_bwd_kernel, dim3(a.tiles, d->n_members), dim3(ENS_THREADS), smem, st, a);
  if (int e = check_launch("ens_bwd")) return e;
  EnsFinishArgs f;
  memset(&f, 0, sizeof(f));
  f.ws = a.ws; f.tiles = a.tiles; f.accumulate = accumulate_param_grads;
  int blk = 0, ns = 0;
  for (int mi = 0; mi < d->n_members; ++mi) {
    if (!a.m[mi].param_grads) continue;
    for (int li = 0; li < a.m[mi].nl; ++li) {
      const EnsLayerDev& L = a.m[mi].l[li];
      f.seg_ws[ns] = L.ws_off; f.seg_dw[ns] = d->member[mi].layer[li].dw; f.seg_db[ns] = d->member[mi].layer[li].db;
      f.seg_kn[ns] = L.K * L.N; f.seg_n[ns] = L.N;
      f.seg_first[ns] = blk;
      blk += (L.K * L.N + L.N + 255) / 256;
      ++ns;
    }
  }
  f.seg_first[ns] = blk;
  f.n_seg = ns;
  f.gx_first = blk;
  f.gx_off = a.gx_off; f.n_members = d->n_members; f.B = a.B; f.halves = a.halves; f.in_p = a.in_p; f.n0 = a.n0;
  f.d_src0 = d_src0; f.ld_d = ld_d; f.coef = coef; f.gx_raw = gx_raw;
  if (d_src0 || gx_raw) {
    const long long n = gx_raw ? (long long)a.B * a.halves * a.in_p : (long long)a.B * a.n0;
    blk += (int)((n + 255) / 256);
  }
  if (blk == 0) return SVAE_OK;
  hipLaunchKernelGGL(ens_finish_kernel, dim3(blk), dim3(256), 0, st, f);
  return check_launch("ens_finish");
}

extern "C" int svae_ens_loss(int kind, const float* const* outs, float* const* dpred, const float* loss_w, const float* grad_s,
                             int n_members, const float* target, int ld_t, const int* labels, int rows, int C, int ld, float* part,
                             void* stream) {
  SVAE_REQUIRE(outs && dpred && loss_w && grad_s && part && n_members >= 1 && n_members <= SVAE_ENS_MEMBERS && rows > 0 && C > 0 && ld >= C,
               SVAE_ERR_ARG, "ens_loss: bad args");
  SVAE_REQUIRE((kind == 0 && target && ld_t >= C) || (kind == 1 && labels) || (kind == 2 && C == 2 && rows % 2 == 0), SVAE_ERR_ARG,
               "ens_loss: kind %d with inconsistent operands", kind);
  EnsLossArgs a;
  memset(&a, 0, sizeof(a));
  for (int m = 0; m < n_members; ++m) {
    SVAE_REQUIRE(outs[m], SVAE_ERR_ARG, "ens_loss: member %d has no output", m);
    a.out[m] = outs[m]; a.dpred[m] = dpred[m]; a.lw[m] = loss_w[m]; a.gs[m] = grad_s[m];
  }
  a.target = target; a.labels = labels; a.part = part;
  a.rows = rows; a.C = C; a.ld = ld; a.ld_t = ld_t; a.kind = kind; a.n_members = n_members;
  hipLaunchKernelGGL(ens_loss_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("ens_loss");
}
