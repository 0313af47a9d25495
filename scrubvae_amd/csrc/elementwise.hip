// HBM-bound row/channel kernels of the SC-VAE step (gfx950): input pack, train-mode
// BatchNorm statistics, fused BN-affine + PReLU forward/backward, linear x2 upsample,
// latent heads (softplus / reparameterisation / KL), small loss reductions, fused Adam.
// All activations are channels-last [rows][ld] fp32; every kernel moves 16 bytes per lane.
#include "svae_internal.h"

namespace svae {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

static inline int grid_for(long long work_items, int per_block = 256, int cap = 4096) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

// ------------------------------------------------------------------------ pack input
struct Arena { float a0[3], a1[3]; int valid; };
// one thread per (row, channel quad): 32-bit index arithmetic, one 16-byte store (the per-element version spent its time in a
// 64-bit division per value: 2.9 TB/s)
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x6d, const float* __restrict__ root,
                                                          const Arena arena, float* __restrict__ out,
                                                          long long rows, int c6, int ld) {
  const int ld4 = ld >> 2;
  const long long nq = rows * ld4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nq; i += (long long)gridDim.x * 256) {
    const long long r = i / ld4;
    const int c = (int)(i - r * ld4) * 4;
    float v[4];
    const float* xr = x6d + r * c6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cc = c + k;
      float t = 0.f;
      if (cc < c6) {
        t = xr[cc];
      } else if (arena.valid && cc < c6 + 3) {
        const int a = cc - c6;
        const float a0 = arena.a0[a], a1 = arena.a1[a];
        t = 2.f * (root[r * 3 + a] - a0) / (a1 - a0) - 1.f;
      }
      v[k] = t;
    }
    *reinterpret_cast<float4*>(out + r * ld + c) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// --------------------------------------------------------------------------- BN stats
constexpr int STAT_ROWS = 128;  // rows per block of the BatchNorm partial-sum kernels (512 gave only 128 blocks per layer: 1.5 TB/s)

__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ x, long long rows, int C, int ld,
                                                                float* __restrict__ part) {
  __shared__ float red[2][16][65];
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + c4 * 4;
  const long long r0 = (long long)blockIdx.x * STAT_ROWS;
  long long r1 = r0 + STAT_ROWS;
  if (r1 > rows) r1 = rows;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f), q = s;
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += 16) {
      const float4 v = ld4(x + r * ld + c);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      q.x += v.x * v.x; q.y += v.y * v.y; q.z += v.z * v.z; q.w += v.w * v.w;
    }
  red[0][rl][c4 * 4 + 0] = s.x; red[0][rl][c4 * 4 + 1] = s.y; red[0][rl][c4 * 4 + 2] = s.z; red[0][rl][c4 * 4 + 3] = s.w;
  red[1][rl][c4 * 4 + 0] = q.x; red[1][rl][c4 * 4 + 1] = q.y; red[1][rl][c4 * 4 + 2] = q.z; red[1][rl][c4 * 4 + 3] = q.w;
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6, cc = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[k][i][cc];
    const int col = blockIdx.y * 64 + cc;
    if (col < C) part[((long long)blockIdx.x * 2 + k) * C + col] = t;
  }
}

// sums[k][c] = sum over chunks (fp64, fixed order)
__global__ void reduce_partials_kernel(const float* __restrict__ part, int chunks, int K, int C, float* __restrict__ sums) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * C) return;
  const int k = i / C, c = i - k * C;
  double s = 0.0;
  for (int ch = 0; ch < chunks; ++ch) s += (double)part[((long long)ch * K + k) * C + c];
  sums[i] = (float)s;
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, double count, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_o, float* rstd_o, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = (double)sums[c] / count;
  double var = (double)sums[C + c] / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[c], b = beta[c];
  mean_o[c] = (float)mean;
  rstd_o[c] = rstd;
  scale[c] = g * rstd;
  shift[c] = b - (float)mean * g * rstd;
  if (running_mean != nullptr) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// reduce the chunk partials AND finalize in one launch (single-rank path: no all-reduce between)
// 256 threads = cpb channels x (256 / cpb) chunk-lanes: lane s of a channel sums chunks s, s + nsub, ... in fp64, the lane sums are
// combined in lane order (deterministic); adjacent threads read adjacent channels.  cpb shrinks with the channel count so that
// the shallow layers -- few channels, thousands of row-tile partials -- still spread over >= 64 workgroups (a block per 8
// channels left 8 workgroups walking 2,048 partials each: 39 us per BatchNorm on the critical path).
constexpr int FIN_THREADS = 256;
static inline int fin_cpb(int C) { return C >= 512 ? 8 : (C >= 256 ? 4 : (C >= 128 ? 2 : 1)); }

__device__ __forceinline__ bool chunk_sums(const float* __restrict__ part, int chunks, int C, int cpb, double& s0, double& s1, int& c,
                                           double (*red)[FIN_THREADS]) {
  const int nsub = FIN_THREADS / cpb;
  const int sub = threadIdx.x / cpb, cl = threadIdx.x % cpb;
  c = blockIdx.x * cpb + cl;
  double a0 = 0.0, a1 = 0.0;
  if (c < C)
    for (int ch = sub; ch < chunks; ch += nsub) {
      a0 += (double)part[((long long)ch * 2) * C + c];
      a1 += (double)part[((long long)ch * 2 + 1) * C + c];
    }
  red[0][threadIdx.x] = a0;
  red[1][threadIdx.x] = a1;
  __syncthreads();
  if (sub != 0 || c >= C) return false;
  s0 = 0.0; s1 = 0.0;
  for (int i = 0; i < nsub; ++i) { s0 += red[0][i * cpb + cl]; s1 += red[1][i * cpb + cl]; }
  return true;
}

__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ part, int chunks, double count, int C, int cpb,
                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                         float* running_mean, float* running_var, long long* num_batches_tracked, float* mean_o,
                                         float* rstd_o, float* scale, float* shift) {
  __shared__ double red[2][FIN_THREADS];
  if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked != nullptr) *num_batches_tracked += 1;
  double s0, s1;
  int c;
  if (!chunk_sums(part, chunks, C, cpb, s0, s1, c, red)) return;
  const double mean = s0 / count;
  double var = s1 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma[c], b = beta[c];
  mean_o[c] = (float)mean;
  rstd_o[c] = rstd;
  scale[c] = g * rstd;
  shift[c] = b - (float)mean * g * rstd;
  if (running_mean != nullptr) {
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

// backward: reduce the chunk partials to sums[2][C] and emit the parameter gradients
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ part, int chunks, int C, int cpb, float* __restrict__ sums,
                                     float* dgamma, float* dbeta, float* dalpha, const float* __restrict__ dalpha_part, int n_parts,
                                     int accumulate) {
  __shared__ double red[2][FIN_THREADS];
  __shared__ double dsum[256];
  if (blockIdx.x == 0 && dalpha != nullptr) {  // PReLU slope: all partials, strided fp64 sums + fixed-order tree
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_parts; i += 256) acc += (double)dalpha_part[i];
    dsum[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) dsum[threadIdx.x] += dsum[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) dalpha[0] = (accumulate ? dalpha[0] : 0.f) + (float)dsum[0];
  }
  double s0, s1;
  int c;
  if (!chunk_sums(part, chunks, C, cpb, s0, s1, c, red)) return;
  sums[c] = (float)s0;
  sums[C + c] = (float)s1;
  if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s0;
  if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)s1;
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      const float* __restrict__ rm, const float* __restrict__ rv, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float rstd = 1.f / sqrtf(rv[c] + eps);
  scale[c] = gamma[c] * rstd;
  shift[c] = beta[c] - rm[c] * gamma[c] * rstd;
}

// ------------------------------------------------------------- affine + PReLU (alpha != NULL) / tanh (alpha == NULL)
// model.activation = "tanh" (residual.py:89,113,147,174,199) replaces every PReLU by nn.Tanh: same kernels, the slope
// pointer is NULL, act(u) = tanh(u), act'(u) = 1 - tanh(u)^2 recomputed from the saved pre-activation input
__device__ __forceinline__ float act_fwd(float u, float a, bool th) { return th ? tanhf(u) : (u > 0.f ? u : a * u); }
__device__ __forceinline__ float act_grad(float u, float a, bool th) {
  if (th) { const float t = tanhf(u); return 1.f - t * t; }
  return u > 0.f ? 1.f : a;
}

__global__ __launch_bounds__(256) void affine_prelu_fwd_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, const float* __restrict__ alpha,
                                                                float* __restrict__ y, long long rows, int C4, int ld) {
  const bool th = alpha == nullptr;
  const float a = th ? 0.f : alpha[0];
  const long long total = rows * C4;
  if ((C4 & (C4 - 1)) == 0 && C4 <= 256) {
    // power-of-two channel quads (every BatchNorm of the model): a thread keeps ONE channel quad for all its rows -- its coefficients
    // are loaded once and the loop has no 64-bit division (the generic loop below spends more on `i / C4` than on the memory access)
    const int sh = __ffs(C4) - 1;
    const int c = (threadIdx.x & (C4 - 1)) * 4;
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (scale != nullptr) { s4 = ld4(scale + c); t4 = ld4(shift + c); }
    const long long rstep = ((long long)gridDim.x * 256) >> sh;
    for (long long r = ((long long)blockIdx.x * 256 + threadIdx.x) >> sh; r < rows; r += rstep) {
      float4 v = ld4(x + r * ld + c);
      v.x = act_fwd(v.x * s4.x + t4.x, a, th); v.y = act_fwd(v.y * s4.y + t4.y, a, th);
      v.z = act_fwd(v.z * s4.z + t4.z, a, th); v.w = act_fwd(v.w * s4.w + t4.w, a, th);
      st4(y + r * ld + c, v);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    float4 v = ld4(x + r * ld + c);
    if (scale != nullptr) {
      const float4 s = ld4(scale + c), t = ld4(shift + c);
      v.x = v.x * s.x + t.x; v.y = v.y * s.y + t.y; v.z = v.z * s.z + t.z; v.w = v.w * s.w + t.w;
    }
    v.x = act_fwd(v.x, a, th); v.y = act_fwd(v.y, a, th);
    v.z = act_fwd(v.z, a, th); v.w = act_fwd(v.w, a, th);
    st4(y + r * ld + c, v);
  }
}

// backward pass 1: part[chunk][2][C] = (sum du, sum du*xhat); dalpha_part[2 * (chunk*gridDim.y + by)] = (hi, lo) of the block's
// slope partial, formed from fp64 products and sums (the slope's gradient is a sum of ~1e6 cancelling terms)
__global__ __launch_bounds__(256) void affine_prelu_bwd_partial_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ alpha, long long rows, int C, int ld,
    float* __restrict__ part, float* __restrict__ dalpha_part) {
  __shared__ float red[2][16][65];
  __shared__ double red4[4];
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + c4 * 4;
  const long long r0 = (long long)blockIdx.x * STAT_ROWS;
  long long r1 = r0 + STAT_ROWS;
  if (r1 > rows) r1 = rows;
  const bool th = alpha == nullptr;
  const float a = th ? 0.f : alpha[0];
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  double da = 0.0;
  if (c < C) {
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    if (scale != nullptr) {
      const float4 t0 = ld4(scale + c), t1 = ld4(shift + c);
      sc[0] = t0.x; sc[1] = t0.y; sc[2] = t0.z; sc[3] = t0.w;
      sh[0] = t1.x; sh[1] = t1.y; sh[2] = t1.z; sh[3] = t1.w;
    }
    if (mean != nullptr) {
      const float4 t0 = ld4(mean + c), t1 = ld4(rstd + c);
      mu[0] = t0.x; mu[1] = t0.y; mu[2] = t0.z; mu[3] = t0.w;
      rs[0] = t1.x; rs[1] = t1.y; rs[2] = t1.z; rs[3] = t1.w;
    }
    for (long long r = r0 + rl; r < r1; r += 16) {
      const float4 xv4 = ld4(x + r * ld + c), dv4 = ld4(dy + r * ld + c);
      const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = xv[k] * sc[k] + sh[k];
        const float du = dv[k] * act_grad(u, a, th);
        if (!th && !(u > 0.f)) da += (double)dv[k] * (double)u;
        s0[k] += du;
        s1[k] += du * (xv[k] - mu[k]) * rs[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { red[0][rl][c4 * 4 + k] = s0[k]; red[1][rl][c4 * 4 + k] = s1[k]; }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int k = threadIdx.x >> 6, cc = threadIdx.x & 63;
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += red[k][i][cc];
    const int col = blockIdx.y * 64 + cc;
    if (col < C) part[((long long)blockIdx.x * 2 + k) * C + col] = t;
  }
  da = wave_sum_d(da);
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = da;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double tot = (red4[0] + red4[1]) + (red4[2] + red4[3]);
    const float hi = (float)tot;
    float* o = dalpha_part + 2 * ((long long)blockIdx.x * gridDim.y + blockIdx.y);
    o[0] = hi;
    o[1] = (float)(tot - (double)hi);
  }
}

// backward pass 2
__global__ __launch_bounds__(256) void affine_prelu_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ alpha, const float* __restrict__ sums, float inv_count, float* __restrict__ dx, long long rows,
    int C, int ld, float* __restrict__ colsum_part) {
  const bool th = alpha == nullptr;
  const float a = th ? 0.f : alpha[0];
  const int C4 = C / 4;
  const long long total = rows * C4;
  if ((C4 & (C4 - 1)) == 0 && C4 <= 256) {  // one channel quad per thread: coefficients hoisted, no 64-bit division (see affine_prelu_fwd_kernel)
    const int sh = __ffs(C4) - 1;
    const int c = (threadIdx.x & (C4 - 1)) * 4;
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, shf[4] = {0.f, 0.f, 0.f, 0.f};
    if (scale != nullptr) {
      const float4 a4 = ld4(scale + c), b4 = ld4(shift + c);
      sc[0] = a4.x; sc[1] = a4.y; sc[2] = a4.z; sc[3] = a4.w;
      shf[0] = b4.x; shf[1] = b4.y; shf[2] = b4.z; shf[3] = b4.w;
    }
    // (same expression and evaluation order as the generic loop below: bit-identical results)
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f},
          s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (sums != nullptr) {
      const float4 m4 = ld4(mean + c), r4 = ld4(rstd + c), g4 = ld4(gamma + c), p4 = ld4(sums + c), q4 = ld4(sums + C + c);
      mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
      rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
      gm[0] = g4.x; gm[1] = g4.y; gm[2] = g4.z; gm[3] = g4.w;
      s0[0] = p4.x; s0[1] = p4.y; s0[2] = p4.z; s0[3] = p4.w;
      s1[0] = q4.x; s1[1] = q4.y; s1[2] = q4.z; s1[3] = q4.w;
    }
    const long long rstep = ((long long)gridDim.x * 256) >> sh;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long r = ((long long)blockIdx.x * 256 + threadIdx.x) >> sh; r < rows; r += rstep) {
      const float4 xv4 = ld4(x + r * ld + c), dv4 = ld4(dy + r * ld + c);
      const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float u = xv[k] * sc[k] + shf[k];
        const float du = dv[k] * act_grad(u, a, th);
        if (sums != nullptr) {
          const float xh = (xv[k] - mu[k]) * rs[k];
          o[k] = gm[k] * rs[k] * (du - s0[k] * inv_count - xh * s1[k] * inv_count);
        } else {
          o[k] = du * sc[k];
        }
      }
      st4(dx + r * ld + c, make_float4(o[0], o[1], o[2], o[3]));
#pragma unroll
      for (int k = 0; k < 4; ++k) cs[k] += o[k];
    }
    // column sums of dx = the bias gradient of the conv(s) in front of this stage (autograd of `bias=True`): the values are in
    // registers here, so the separate pass over dx is saved.  colsum_part[gridDim.x][C]: one row per workgroup, summed in row
    // order by svae_colsum_from_partials (deterministic).
    if (colsum_part != nullptr) {  // uniform
      __shared__ float red[256][5];
#pragma unroll
      for (int k = 0; k < 4; ++k) red[threadIdx.x][k] = cs[k];
      __syncthreads();
      if ((int)threadIdx.x < C4) {
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = threadIdx.x; j < 256; j += C4)
#pragma unroll
          for (int k = 0; k < 4; ++k) t[k] += red[j][k];
        st4(colsum_part + (long long)blockIdx.x * C + c, make_float4(t[0], t[1], t[2], t[3]));
      }
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const float4 xv4 = ld4(x + r * ld + c), dv4 = ld4(dy + r * ld + c);
    const float xv[4] = {xv4.x, xv4.y, xv4.z, xv4.w}, dv[4] = {dv4.x, dv4.y, dv4.z, dv4.w};
    // per-channel coefficients as float4 loads (C and c are multiples of 4): 7 vector loads instead of 28 scalar ones
    float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
    if (scale != nullptr) {
      const float4 a4 = ld4(scale + c), b4 = ld4(shift + c);
      sc[0] = a4.x; sc[1] = a4.y; sc[2] = a4.z; sc[3] = a4.w;
      sh[0] = b4.x; sh[1] = b4.y; sh[2] = b4.z; sh[3] = b4.w;
    }
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, gm[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f},
          s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (sums != nullptr) {
      const float4 m4 = ld4(mean + c), r4 = ld4(rstd + c), g4 = ld4(gamma + c), p4 = ld4(sums + c), q4 = ld4(sums + C + c);
      mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
      rs[0] = r4.x; rs[1] = r4.y; rs[2] = r4.z; rs[3] = r4.w;
      gm[0] = g4.x; gm[1] = g4.y; gm[2] = g4.z; gm[3] = g4.w;
      s0[0] = p4.x; s0[1] = p4.y; s0[2] = p4.z; s0[3] = p4.w;
      s1[0] = q4.x; s1[1] = q4.y; s1[2] = q4.z; s1[3] = q4.w;
    }
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float u = xv[k] * sc[k] + sh[k];
      const float du = dv[k] * act_grad(u, a, th);
      if (sums != nullptr) {
        const float xh = (xv[k] - mu[k]) * rs[k];
        o[k] = gm[k] * rs[k] * (du - s0[k] * inv_count - xh * s1[k] * inv_count);
      } else {
        o[k] = du * sc[k];
      }
    }
    st4(dx + r * ld + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}

__global__ __launch_bounds__(128) void bn_param_grads_kernel(const float* __restrict__ sums, int C, float* dgamma, float* dbeta, float* dalpha,
                                      const float* __restrict__ dalpha_part, int n_parts, int accumulate) {
  __shared__ double dsum[128];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (sums != nullptr && c < C) {
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sums[c];
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sums[C + c];
  }
  if (blockIdx.x == 0 && dalpha != nullptr) {  // strided fp64 sums + fixed-order tree (deterministic)
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_parts; i += 128) acc += (double)dalpha_part[i];
    dsum[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 64; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) dsum[threadIdx.x] += dsum[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) dalpha[0] = (accumulate ? dalpha[0] : 0.f) + (float)dsum[0];
  }
}

// ------------------------------------------------------------------------- upsample x2
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int batch, int L,
                                                             int C4, int ld) {
  const long long total = (long long)batch * 2 * L * C4;
  if ((C4 & (C4 - 1)) == 0 && C4 <= 256 && (long long)batch * 2 * L < 0x7fffffffLL) {
    // one channel quad per thread, 32-bit row arithmetic (the generic loop's two 64-bit divisions cost more than its memory accesses)
    const int sh = __ffs(C4) - 1;
    const int c = (threadIdx.x & (C4 - 1)) * 4;
    const unsigned nrow = (unsigned)batch * 2u * (unsigned)L, rstep = (unsigned)(((long long)gridDim.x * 256) >> sh), L2 = 2u * (unsigned)L;
    for (unsigned row = (unsigned)(((long long)blockIdx.x * 256 + threadIdx.x) >> sh); row < nrow; row += rstep) {
      const unsigned b = row / L2;
      const int ro = (int)(row - b * L2);
      const int ii = ro >> 1;
      const int i2 = (ro & 1) ? (ii + 1 < L ? ii + 1 : L - 1) : (ii > 0 ? ii - 1 : 0);
      const float4 p = ld4(x + ((long long)b * L + ii) * ld + c), q = ld4(x + ((long long)b * L + i2) * ld + c);
      st4(y + (long long)row * ld + c, up2_blend4(p, q));
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / C4;
    const int c = (int)(i - row * C4) * 4;
    const long long b = row / (2 * L);
    const int ro = (int)(row - b * 2 * L);
    const int ii = ro >> 1;
    const int i2 = (ro & 1) ? (ii + 1 < L ? ii + 1 : L - 1) : (ii > 0 ? ii - 1 : 0);
    const float4 p = ld4(x + (b * L + ii) * ld + c), q = ld4(x + (b * L + i2) * ld + c);
    st4(y + row * ld + c, up2_blend4(p, q));
  }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int batch, int L,
                                                             int C4, int ld, int accumulate) {
  const long long total = (long long)batch * L * C4;
  if ((C4 & (C4 - 1)) == 0 && C4 <= 256 && (long long)batch * L < 0x7fffffffLL) {  // as in upsample2_fwd_kernel
    const int sh = __ffs(C4) - 1;
    const int c = (threadIdx.x & (C4 - 1)) * 4;
    const unsigned nrow = (unsigned)batch * (unsigned)L, rstep = (unsigned)(((long long)gridDim.x * 256) >> sh);
    for (unsigned row = (unsigned)(((long long)blockIdx.x * 256 + threadIdx.x) >> sh); row < nrow; row += rstep) {
      const unsigned b = row / (unsigned)L;
      const int ii = (int)(row - b * (unsigned)L);
      const float* base = dy + (long long)b * 2 * L * ld + c;
      const float4 e = ld4(base + (long long)(2 * ii) * ld), o = ld4(base + (long long)(2 * ii + 1) * ld);
      const float4 nx = ld4(base + (long long)(ii + 1 < L ? 2 * ii + 2 : 2 * L - 1) * ld);
      const float4 pv = ld4(base + (long long)(ii >= 1 ? 2 * ii - 1 : 0) * ld);
      float4 r = make_float4(0.75f * (e.x + o.x) + 0.25f * (nx.x + pv.x), 0.75f * (e.y + o.y) + 0.25f * (nx.y + pv.y),
                             0.75f * (e.z + o.z) + 0.25f * (nx.z + pv.z), 0.75f * (e.w + o.w) + 0.25f * (nx.w + pv.w));
      if (accumulate) {
        const float4 old = ld4(dx + (long long)row * ld + c);
        r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
      }
      st4(dx + (long long)row * ld + c, r);
    }
    return;
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long row = i / C4;
    const int c = (int)(i - row * C4) * 4;
    const long long b = row / L;
    const int ii = (int)(row - b * L);
    const float* base = dy + b * 2 * L * ld + c;
    const float4 e = ld4(base + (long long)(2 * ii) * ld), o = ld4(base + (long long)(2 * ii + 1) * ld);
    const float4 nx = ld4(base + (long long)(ii + 1 < L ? 2 * ii + 2 : 2 * L - 1) * ld);
    const float4 pv = ld4(base + (long long)(ii >= 1 ? 2 * ii - 1 : 0) * ld);
    float4 r = make_float4(0.75f * (e.x + o.x) + 0.25f * (nx.x + pv.x), 0.75f * (e.y + o.y) + 0.25f * (nx.y + pv.y),
                           0.75f * (e.z + o.z) + 0.25f * (nx.z + pv.z), 0.75f * (e.w + o.w) + 0.25f * (nx.w + pv.w));
    if (accumulate) {
      const float4 old = ld4(dx + row * ld + c);
      r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
    }
    st4(dx + row * ld + c, r);
  }
}

// ---------------------------------------------------------------- batched column sums
// bias gradients of every conv/linear of a step in two launches: task t sums the rows of its
// dY matrix per column.  part layout: task -> [chunks_t][C_t] starting at part_off[t].
constexpr int CS_ROWS = 512;
struct ColsumTasks {
  const float* x[SVAE_MAX_COLSUM_TASKS];
  float* out[SVAE_MAX_COLSUM_TASKS];
  long long rows[SVAE_MAX_COLSUM_TASKS];
  long long part_off[SVAE_MAX_COLSUM_TASKS];
  int C[SVAE_MAX_COLSUM_TASKS], ld[SVAE_MAX_COLSUM_TASKS];
  int blk_begin[SVAE_MAX_COLSUM_TASKS + 1];  // first block id of each task (stage 1)
  int col_begin[SVAE_MAX_COLSUM_TASKS + 1];  // first global column of each task (stage 2)
  int n;
};

__global__ __launch_bounds__(256) void colsum_batched_partial_kernel(const ColsumTasks T, float* __restrict__ part) {
  __shared__ float red[16][65];
  int t = 0;
  while (t + 1 < T.n && (int)blockIdx.x >= T.blk_begin[t + 1]) ++t;
  const int local = blockIdx.x - T.blk_begin[t];
  const int C = T.C[t], ld = T.ld[t];
  const int colblocks = (C + 63) / 64;
  const int chunk = local / colblocks, cb = local - chunk * colblocks;
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = cb * 64 + c4 * 4;
  const long long r0 = (long long)chunk * CS_ROWS;
  long long r1 = r0 + CS_ROWS;
  if (r1 > T.rows[t]) r1 = T.rows[t];
  const float* x = T.x[t];
  float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C)
    for (long long r = r0 + rl; r < r1; r += 16) {
      const float4 v = ld4(x + r * ld + c);
      sv.x += v.x; sv.y += v.y; sv.z += v.z; sv.w += v.w;
    }
  red[rl][c4 * 4 + 0] = sv.x; red[rl][c4 * 4 + 1] = sv.y; red[rl][c4 * 4 + 2] = sv.z; red[rl][c4 * 4 + 3] = sv.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    float tt = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) tt += red[i][threadIdx.x];
    const int cc = cb * 64 + threadIdx.x;
    if (cc < C) part[T.part_off[t] + (long long)chunk * C + cc] = tt;
  }
}

// 256 threads = 8 columns x 32 chunk-lanes: lane s of a column sums chunks s, s+32, ... in fp64, the lane sums are
// combined in lane order (deterministic)
__global__ __launch_bounds__(256) void colsum_batched_final_kernel(const ColsumTasks T, const float* __restrict__ part, int accumulate) {
  __shared__ double red[32][8];
  const int cl = threadIdx.x & 7, sub = threadIdx.x >> 3;
  const int gc = blockIdx.x * 8 + cl;
  const bool ok = gc < T.col_begin[T.n];
  int t = 0, c = 0, C = 1, chunks = 0;
  if (ok) {
    while (t + 1 < T.n && gc >= T.col_begin[t + 1]) ++t;
    c = gc - T.col_begin[t];
    C = T.C[t];
    chunks = (int)((T.rows[t] + CS_ROWS - 1) / CS_ROWS);
  }
  double s = 0.0;
  for (int k = sub; k < chunks; k += 32) s += (double)part[T.part_off[t] + (long long)k * C + c];
  red[sub][cl] = s;
  __syncthreads();
  if (sub == 0 && ok) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += red[i][cl];
    float* o = T.out[t];
    o[c] = (accumulate ? o[c] : 0.f) + (float)tot;
  }
}

// ------------------------------------------------------------------------ latent heads
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(256) void heads_diag_fwd_kernel(const float* __restrict__ h, int ld, const float* __restrict__ eps,
                                                              float* __restrict__ mu, float* __restrict__ sigma,
                                                              float* __restrict__ z, int ldz, float* __restrict__ kl_part, int batch,
                                                              int zd, int raw_off, int ldm) {
  __shared__ float red4[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float kl = 0.f;
  if (i < (long long)batch * zd) {
    const int b = (int)(i / zd), k = (int)(i - (long long)b * zd);
    const float m = h[(long long)b * ld + k];
    const float s = softplus_f(h[(long long)b * ld + raw_off + k]);
    mu[(long long)b * ldm + k] = m;
    sigma[(long long)b * ldm + k] = s;
    z[(long long)b * ldz + k] = eps ? m + s * eps[i] : m;
    kl = -0.5f * (1.f + 2.f * logf(s) - m * m - s * s);
  }
  const float t = block_sum_256(kl, red4);
  if (threadIdx.x == 0) kl_part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void heads_diag_bwd_kernel(const float* __restrict__ h, int ld, const float* __restrict__ eps,
                                                              const float* __restrict__ sigma, const float* __restrict__ dz, int lddz,
                                                              const float* __restrict__ dmu, const float* __restrict__ dsigma,
                                                              float kl_scale, float* __restrict__ dh, int batch, int zd, int raw_off, int ldm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)batch * zd) return;
  const int b = (int)(i / zd), k = (int)(i - (long long)b * zd);
  const float m = h[(long long)b * ld + k], raw = h[(long long)b * ld + raw_off + k];
  const float s = sigma[(long long)b * ldm + k];
  const float gz = dz ? dz[(long long)b * lddz + k] : 0.f;
  float gm = gz + kl_scale * m;
  if (dmu) gm += dmu[(long long)b * ldm + k];
  float gs = kl_scale * (s - 1.f / s);
  if (eps) gs += gz * eps[i];
  if (dsigma) gs += dsigma[(long long)b * ldm + k];
  const float sig = 1.f / (1.f + expf(-raw));
  dh[(long long)b * ld + k] = gm;
  dh[(long long)b * ld + raw_off + k] = gs * sig;
}

// -------------------------------------------------------------------------- optimizer
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float wd, float step_size, float inv_bc2_sqrt, int decoupled, float gscale) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pv = ld4(p + i * 4), gv = ld4(g + i * 4), mv = ld4(m + i * 4), vv = ld4(v + i * 4);
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w},
          vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = gg[k] * gscale;
      if (decoupled) pp[k] *= (1.f - lr * wd);
      else if (wd != 0.f) gk += wd * pp[k];
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vq[k] = b2 * vq[k] + (1.f - b2) * gk * gk;
      const float denom = sqrtf(vq[k]) * inv_bc2_sqrt + eps;
      pp[k] -= step_size * mm[k] / denom;
    }
    st4(p + i * 4, make_float4(pp[0], pp[1], pp[2], pp[3]));
    st4(m + i * 4, make_float4(mm[0], mm[1], mm[2], mm[3]));
    st4(v + i * 4, make_float4(vq[0], vq[1], vq[2], vq[3]));
  }
}

// hipGraph-safe variant: the step-dependent scalars (lr, lr/bias_correction1, 1/sqrt(bias_correction2))
// live in device memory and are refreshed by the host before each replay
__global__ __launch_bounds__(256) void adam_kernel_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long long n, const float* __restrict__ hyper,
                                                        float b1, float b2, float eps, float wd, int decoupled, float gscale) {
  const float lr = hyper[0], step_size = hyper[1], inv_bc2_sqrt = hyper[2];
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 pv = ld4(p + i * 4), gv = ld4(g + i * 4), mv = ld4(m + i * 4), vv = ld4(v + i * 4);
    float pp[4] = {pv.x, pv.y, pv.z, pv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w}, mm[4] = {mv.x, mv.y, mv.z, mv.w},
          vq[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = gg[k] * gscale;
      if (decoupled) pp[k] *= (1.f - lr * wd);
      else if (wd != 0.f) gk += wd * pp[k];
      mm[k] = b1 * mm[k] + (1.f - b1) * gk;
      vq[k] = b2 * vq[k] + (1.f - b2) * gk * gk;
      const float denom = sqrtf(vq[k]) * inv_bc2_sqrt + eps;
      pp[k] -= step_size * mm[k] / denom;
    }
    st4(p + i * 4, make_float4(pp[0], pp[1], pp[2], pp[3]));
    st4(m + i * 4, make_float4(mm[0], mm[1], mm[2], mm[3]));
    st4(v + i * 4, make_float4(vq[0], vq[1], vq[2], vq[3]));
  }
}

// Device-side step counter for captured (hipGraph) optimizer steps: hyper = {lr, lr / bias_correction1, 1 / sqrt(bias_correction2),
// step}.  Advancing it inside the graph means no host buffer is re-written while earlier replays are still queued.
__global__ void adam_advance_kernel(float* __restrict__ hyper, float b1, float b2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const double t = (double)hyper[3] + 1.0;
  hyper[3] = (float)t;
  hyper[1] = (float)((double)hyper[0] / (1.0 - pow((double)b1, t)));
  hyper[2] = (float)(1.0 / sqrt(1.0 - pow((double)b2, t)));
}

// torch.nn.utils.clip_grad_norm_: g *= min(1, max_norm / (norm + 1e-6)), norm = sqrt(*sumsq).  Every thread reads the scalar; when
// the clip does not bite (the reference's max_norm = 1e6) the kernel returns without touching the gradients.
__global__ __launch_bounds__(256) void clip_grads_kernel(float* __restrict__ g, long long n, const float* __restrict__ sumsq,
                                                          float max_norm, float* __restrict__ norm_out) {
  const float norm = sqrtf(sumsq[0]);
  if (norm_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) norm_out[0] = norm;  // the total norm clip_grad_norm_ returns
  const float coef = max_norm / (norm + 1e-6f);
  if (coef >= 1.f) return;  // a NaN norm falls through and poisons every gradient, as torch's clamp(NaN, max=1) * g does
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    float4 v = ld4(g + i * 4);
    st4(g + i * 4, make_float4(v.x * coef, v.y * coef, v.z * coef, v.w * coef));
  }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float red4[4];
  float s = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = ld4(x + i * 4);
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  const float t = block_sum_256(s, red4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// one 256-thread block per column: fp64 partials, fixed-shape tree => reproducible
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int rows, int k, float scale, float* out,
                                                           int accumulate) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(long long)r * k + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (accumulate ? out[c] : 0.f) + (float)(red[0] * (double)scale);
}

// out[c] = scale[c] * sum over rows of part[rows][k] (k <= 8): reduce_rows_kernel with one scale per column, so that loss terms
// normalised differently (jpe: 1 / (B 3 J), root: 1 / B; losses.py:171,216-219) leave ONE launch
struct ColScales { float s[8]; };
__global__ __launch_bounds__(256) void reduce_rows_scaled_kernel(const float* __restrict__ part, int rows, int k, ColScales sc, float* out) {
  __shared__ double red[256];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(long long)r * k + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)(red[0] * (double)sc.s[c]);
}

// total = sum_i w[i] * terms[i] in index order, fp32 like the reference's running `total += scale * loss` (losses.py:320-322)
struct LossWeights { float w[SVAE_MAX_LOSS_TERMS]; };
__global__ void loss_total_kernel(const float* __restrict__ terms, LossWeights lw, int n, float* out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  float t = 0.f;
  for (int i = 0; i < n; ++i)
    if (lw.w[i] != 0.f) t += lw.w[i] * terms[i];
  out[0] = t;
}

// ------------------------------------------------------------------ small elementwise
__global__ void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = x[i] > 0.f ? x[i] : 0.f;
}
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}
__global__ void axpy_kernel(float a, const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}
__global__ void fill_kernel(float* __restrict__ x, float v, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) x[i] = v;
}

// row-wise losses: one thread per row, 256 rows per block
__global__ __launch_bounds__(256) void mse_sum_kernel(const float* __restrict__ pred, int ld, const float* __restrict__ target,
                                                       int ld_t, int rows, int C, float scale, float* __restrict__ part,
                                                       float* __restrict__ dpred) {
  __shared__ float red4[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (r < rows)
    for (int c = 0; c < C; ++c) {
      const float d = pred[(long long)r * ld + c] - target[(long long)r * ld_t + c];
      s += d * d;
      if (dpred) dpred[(long long)r * ld + c] = 2.f * scale * d;
    }
  const float t = block_sum_256(s, red4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void ce_sum_kernel(const float* __restrict__ logits, int ld, const int* __restrict__ labels,
                                                      int rows, int C, float scale, float* __restrict__ part,
                                                      float* __restrict__ dlogits) {
  __shared__ float red4[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  if (r < rows) {
    const float* x = logits + (long long)r * ld;
    float mx = x[0];
    for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(x[c] - mx);
    const float lse = mx + logf(se);
    const int y = labels[r];
    loss = lse - x[y];
    if (dlogits)
      for (int c = 0; c < C; ++c) dlogits[(long long)r * ld + c] = scale * (expf(x[c] - lse) - (c == y ? 1.f : 0.f));
  }
  const float t = block_sum_256(loss, red4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// two classes: p = softmax(x); loss = -log_softmax(p)[cls]; cls = row >= rows/2
__global__ __launch_bounds__(256) void double_softmax_ce_kernel(const float* __restrict__ logits, int ld, int rows, float scale,
                                                                 float* __restrict__ part, float* __restrict__ dlogits) {
  __shared__ float red4[4];
  const int r = blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  if (r < rows) {
    const float x0 = logits[(long long)r * ld], x1 = logits[(long long)r * ld + 1];
    const float mx = fmaxf(x0, x1);
    const float e0 = expf(x0 - mx), e1 = expf(x1 - mx);
    const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
    const float m2 = fmaxf(p0, p1);
    const float lse = m2 + logf(expf(p0 - m2) + expf(p1 - m2));
    const int cls = r >= rows / 2 ? 1 : 0;
    loss = lse - (cls ? p1 : p0);
    if (dlogits) {
      // dL/dp_c = softmax(p)_c - [c==cls]; then through p = softmax(x)
      const float q0 = expf(p0 - lse), q1 = expf(p1 - lse);
      const float g0 = q0 - (cls == 0 ? 1.f : 0.f), g1 = q1 - (cls == 1 ? 1.f : 0.f);
      const float dot = g0 * p0 + g1 * p1;
      dlogits[(long long)r * ld] = scale * p0 * (g0 - dot);
      dlogits[(long long)r * ld + 1] = scale * p1 * (g1 - dot);
    }
  }
  const float t = block_sum_256(loss, red4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

}  // namespace svae

using namespace svae;
#define ST(s) ((hipStream_t)(s))

extern "C" int svae_pack_input(const float* x6d, const float* root, const float* arena, float* x_in, long long rows,
                               int n_joints, int ld, void* stream) {
  SVAE_REQUIRE(x6d && x_in && rows > 0, SVAE_ERR_ARG, "pack_input: null pointer / no rows");
  const int c6 = 6 * n_joints;
  SVAE_REQUIRE(ld >= c6 + (arena ? 3 : 0), SVAE_ERR_SHAPE, "pack_input: ld %d too small", ld);
  SVAE_REQUIRE(!arena || root, SVAE_ERR_ARG, "pack_input: arena given without root");
  SVAE_REQUIRE(ld % 4 == 0 && aligned16(x_in), SVAE_ERR_ALIGN, "pack_input: ld must be a multiple of 4 and x_in 16-byte aligned");
  Arena ar;
  memset(&ar, 0, sizeof(ar));
  if (arena) { for (int k = 0; k < 3; ++k) { ar.a0[k] = arena[k]; ar.a1[k] = arena[3 + k]; } ar.valid = 1; }
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid_for(rows * (ld / 4))), dim3(256), 0, ST(stream), x6d, root, ar, x_in, rows, c6, ld);
  return check_launch("pack_input");
}

extern "C" int svae_bn_chunks(long long rows) { return (int)((rows + STAT_ROWS - 1) / STAT_ROWS); }

extern "C" int svae_bn_stats_partial(const float* x, long long rows, int C, int ld, float* part, void* stream) {
  SVAE_REQUIRE(x && part && rows > 0, SVAE_ERR_ARG, "bn_stats: null pointer");
  SVAE_REQUIRE(C % 4 == 0 && ld % 4 == 0 && aligned16(x), SVAE_ERR_ALIGN, "bn_stats: C, ld must be multiples of 4");
  hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(svae_bn_chunks(rows), (C + 63) / 64), dim3(256), 0, ST(stream), x, rows, C, ld, part);
  return check_launch("bn_stats_partial");
}

extern "C" int svae_bn_reduce_partials(const float* part, int n_chunks, int C, float* sums, void* stream) {
  SVAE_REQUIRE(part && sums && n_chunks > 0, SVAE_ERR_ARG, "bn_reduce_partials: bad args");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, ST(stream), part, n_chunks, 2, C, sums);
  return check_launch("bn_reduce_partials");
}

extern "C" int svae_bn_finalize(const float* sums, double count, int C, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, float* mean, float* rstd,
                                float* scale, float* shift, void* stream) {
  SVAE_REQUIRE(sums && gamma && beta && mean && rstd && scale && shift && count > 0, SVAE_ERR_ARG, "bn_finalize: bad args");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), sums, count, C, gamma, beta, eps,
                     momentum, running_mean, running_var, mean, rstd, scale, shift);
  return check_launch("bn_finalize");
}

extern "C" int svae_bn_eval_coeffs(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                                   const float* running_var, float* scale, float* shift, void* stream) {
  SVAE_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, SVAE_ERR_ARG, "bn_eval_coeffs: null pointer");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), C, gamma, beta, eps, running_mean,
                     running_var, scale, shift);
  return check_launch("bn_eval_coeffs");
}

extern "C" int svae_affine_prelu_fwd(const float* x, const float* scale, const float* shift, const float* alpha, float* y,
                                     long long rows, int C, int ld, void* stream) {
  SVAE_REQUIRE(x && y && rows > 0, SVAE_ERR_ARG, "affine_prelu_fwd: null pointer");
  SVAE_REQUIRE(C % 4 == 0 && ld % 4 == 0, SVAE_ERR_ALIGN, "affine_prelu_fwd: C, ld must be multiples of 4");
  hipLaunchKernelGGL(affine_prelu_fwd_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, ST(stream), x, scale, shift, alpha, y,
                     rows, C / 4, ld);
  return check_launch("affine_prelu_fwd");
}

extern "C" int svae_affine_prelu_bwd_partial(const float* dy, const float* x, const float* scale, const float* shift,
                                             const float* mean, const float* rstd, const float* alpha, long long rows, int C,
                                             int ld, float* part, float* dalpha_part, void* stream) {
  SVAE_REQUIRE(dy && x && part && dalpha_part && rows > 0, SVAE_ERR_ARG, "affine_prelu_bwd_partial: null pointer");
  SVAE_REQUIRE(C % 4 == 0 && ld % 4 == 0, SVAE_ERR_ALIGN, "affine_prelu_bwd_partial: C, ld must be multiples of 4");
  hipLaunchKernelGGL(affine_prelu_bwd_partial_kernel, dim3(svae_bn_chunks(rows), (C + 63) / 64), dim3(256), 0, ST(stream), dy, x,
                     scale, shift, mean, rstd, alpha, rows, C, ld, part, dalpha_part);
  return check_launch("affine_prelu_bwd_partial");
}

/* rows of colsum_part the apply pass writes (= its grid: at most 1024 workgroups, so that the partials stay small) */
extern "C" int svae_affine_prelu_colsum_rows(long long rows, int C) { return grid_for(rows * (C / 4), 256, 1024); }

extern "C" int svae_affine_prelu_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift,
                                           const float* mean, const float* rstd, const float* gamma, const float* alpha,
                                           const float* sums, double count, float* dx, long long rows, int C, int ld,
                                           float* dgamma, float* dbeta, float* dalpha, const float* dalpha_part, int n_parts,
                                           int accumulate_param_grads, void* stream) {
  return svae_affine_prelu_bwd_apply_colsum(dy, x, scale, shift, mean, rstd, gamma, alpha, sums, count, dx, rows, C, ld, dgamma, dbeta,
                                            dalpha, dalpha_part, n_parts, accumulate_param_grads, nullptr, stream);
}

extern "C" int svae_affine_prelu_bwd_apply_colsum(const float* dy, const float* x, const float* scale, const float* shift,
                                                  const float* mean, const float* rstd, const float* gamma, const float* alpha,
                                                  const float* sums, double count, float* dx, long long rows, int C, int ld,
                                                  float* dgamma, float* dbeta, float* dalpha, const float* dalpha_part, int n_parts,
                                                  int accumulate_param_grads, float* colsum_part, void* stream) {
  SVAE_REQUIRE(dy && x && dx && rows > 0, SVAE_ERR_ARG, "affine_prelu_bwd_apply: null pointer");
  SVAE_REQUIRE(!sums || (mean && rstd && gamma && count > 0), SVAE_ERR_ARG, "affine_prelu_bwd_apply: BN tensors missing");
  if (colsum_part) {
    const int C4 = C / 4;
    SVAE_REQUIRE(C % 4 == 0 && (C4 & (C4 - 1)) == 0 && C4 <= 256 && aligned16(colsum_part), SVAE_ERR_SHAPE,
                 "affine_prelu_bwd_apply: column sums need C / 4 a power of two <= 256");
  }
  const int grid = colsum_part ? svae_affine_prelu_colsum_rows(rows, C) : grid_for(rows * (C / 4));
  hipLaunchKernelGGL(affine_prelu_bwd_apply_kernel, dim3(grid), dim3(256), 0, ST(stream), dy, x, scale, shift,
                     mean, rstd, gamma, alpha, sums, sums ? (float)(1.0 / count) : 0.f, dx, rows, C, ld, colsum_part);
  if (int e = check_launch("affine_prelu_bwd_apply")) return e;
  if (dgamma || dbeta || dalpha) {
    hipLaunchKernelGGL(bn_param_grads_kernel, dim3((C + 127) / 128), dim3(128), 0, ST(stream), sums, C, dgamma, dbeta, dalpha,
                       dalpha_part, n_parts, accumulate_param_grads);
    return check_launch("bn_param_grads");
  }
  return SVAE_OK;
}

extern "C" int svae_upsample2_fwd(const float* x, float* y, int batch, int l_in, int C, int ld, void* stream) {
  SVAE_REQUIRE(x && y && batch > 0 && l_in > 0 && C % 4 == 0 && ld % 4 == 0, SVAE_ERR_ARG, "upsample2_fwd: bad args");
  hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(grid_for((long long)batch * 2 * l_in * (C / 4))), dim3(256), 0, ST(stream), x, y,
                     batch, l_in, C / 4, ld);
  return check_launch("upsample2_fwd");
}

extern "C" int svae_upsample2_bwd(const float* dy, float* dx, int batch, int l_in, int C, int ld, int accumulate, void* stream) {
  SVAE_REQUIRE(dy && dx && batch > 0 && l_in > 0 && C % 4 == 0 && ld % 4 == 0, SVAE_ERR_ARG, "upsample2_bwd: bad args");
  hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(grid_for((long long)batch * l_in * (C / 4))), dim3(256), 0, ST(stream), dy, dx,
                     batch, l_in, C / 4, ld, accumulate);
  return check_launch("upsample2_bwd");
}

extern "C" int svae_heads_blocks(int batch, int zdim) { return (int)(((long long)batch * zdim + 255) / 256); }

extern "C" int svae_heads_diag_fwd(const float* h, int ld, const float* eps, float* mu, float* sigma, float* z, int ldz,
                                   float* kl_part, int batch, int zdim, int raw_off, int ldm, void* stream) {
  SVAE_REQUIRE(h && mu && sigma && z && kl_part && batch > 0 && zdim > 0 && raw_off >= zdim && ld >= raw_off + zdim && ldz >= zdim, SVAE_ERR_ARG,
               "heads_diag_fwd: bad args");
  hipLaunchKernelGGL(heads_diag_fwd_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, ST(stream), h, ld, eps, mu, sigma,
                     z, ldz, kl_part, batch, zdim, raw_off, ldm);
  return check_launch("heads_diag_fwd");
}

extern "C" int svae_heads_diag_bwd(const float* h, int ld, const float* eps, const float* sigma, const float* dz, int lddz,
                                   const float* dmu, const float* dsigma, float kl_scale, float* dh, int batch, int zdim,
                                   int raw_off, int ldm, void* stream) {
  SVAE_REQUIRE(h && sigma && dh && batch > 0 && zdim > 0, SVAE_ERR_ARG, "heads_diag_bwd: bad args");
  hipLaunchKernelGGL(heads_diag_bwd_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, ST(stream), h, ld, eps, sigma, dz,
                     lddz, dmu, dsigma, kl_scale, dh, batch, zdim, raw_off, ldm);
  return check_launch("heads_diag_bwd");
}

extern "C" int svae_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step_t, int decoupled, float grad_scale, void* stream) {
  SVAE_REQUIRE(p && g && m && v && n > 0 && n % 4 == 0 && step_t >= 1, SVAE_ERR_ARG, "adam_step: bad args (n must be a multiple of 4)");
  const double bc1 = 1.0 - pow((double)beta1, step_t), bc2 = 1.0 - pow((double)beta2, step_t);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)), decoupled, grad_scale);
  return check_launch("adam_step");
}

extern "C" int svae_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper, float beta1,
                                  float beta2, float eps, float weight_decay, int decoupled, float grad_scale, void* stream) {
  SVAE_REQUIRE(p && g && m && v && hyper && n > 0 && n % 4 == 0, SVAE_ERR_ARG, "adam_step_dev: bad args");
  hipLaunchKernelGGL(adam_kernel_dev, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, ST(stream), p, g, m, v, n, hyper, beta1, beta2,
                     eps, weight_decay, decoupled, grad_scale);
  return check_launch("adam_step_dev");
}

extern "C" int svae_adam_advance(float* hyper, float beta1, float beta2, void* stream) {
  SVAE_REQUIRE(hyper, SVAE_ERR_ARG, "adam_advance: null pointer");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, ST(stream), hyper, beta1, beta2);
  return check_launch("adam_advance");
}

extern "C" int svae_clip_grads(float* g, long long n, const float* sumsq, float max_norm, float* norm_out, void* stream) {
  SVAE_REQUIRE(g && sumsq && n > 0 && n % 4 == 0 && max_norm > 0.f, SVAE_ERR_ARG, "clip_grads: bad args");
  hipLaunchKernelGGL(clip_grads_kernel, dim3(grid_for(n / 4, 256, 2048)), dim3(256), 0, ST(stream), g, n, sumsq, max_norm, norm_out);
  return check_launch("clip_grads");
}

extern "C" int svae_sumsq_blocks(long long n) { return grid_for(n / 4, 256, 1024); }

extern "C" int svae_sumsq_partial(const float* x, long long n, float* part, void* stream) {
  SVAE_REQUIRE(x && part && n > 0 && n % 4 == 0, SVAE_ERR_ARG, "sumsq_partial: bad args");
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(svae_sumsq_blocks(n)), dim3(256), 0, ST(stream), x, n, part);
  return check_launch("sumsq_partial");
}

extern "C" int svae_reduce_rows(const float* part, int rows, int k, float scale, float* out, int accumulate, void* stream) {
  SVAE_REQUIRE(part && out && rows > 0 && k > 0, SVAE_ERR_ARG, "reduce_rows: bad args");
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(k), dim3(256), 0, ST(stream), part, rows, k, scale, out, accumulate);
  return check_launch("reduce_rows");
}

extern "C" int svae_reduce_rows_scaled(const float* part, int rows, int k, const float* scales, float* out, void* stream) {
  SVAE_REQUIRE(part && out && scales && rows > 0 && k > 0 && k <= 8, SVAE_ERR_ARG, "reduce_rows_scaled: bad args (k <= 8)");
  ColScales sc;
  for (int i = 0; i < 8; ++i) sc.s[i] = i < k ? scales[i] : 0.f;
  hipLaunchKernelGGL(reduce_rows_scaled_kernel, dim3(k), dim3(256), 0, ST(stream), part, rows, k, sc, out);
  return check_launch("reduce_rows_scaled");
}

extern "C" int svae_loss_total(const float* terms, const float* weights, int n, float* out, void* stream) {
  SVAE_REQUIRE(terms && weights && out && n > 0 && n <= SVAE_MAX_LOSS_TERMS, SVAE_ERR_ARG, "loss_total: bad args (n <= %d)",
               SVAE_MAX_LOSS_TERMS);
  LossWeights lw;
  for (int i = 0; i < SVAE_MAX_LOSS_TERMS; ++i) lw.w[i] = i < n ? weights[i] : 0.f;
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, ST(stream), terms, lw, n, out);
  return check_launch("loss_total");
}

extern "C" int svae_relu_fwd(const float* x, float* y, long long n, void* stream) {
  SVAE_REQUIRE(x && y && n > 0, SVAE_ERR_ARG, "relu_fwd: bad args");
  hipLaunchKernelGGL(relu_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, y, n);
  return check_launch("relu_fwd");
}
extern "C" int svae_relu_bwd(const float* dy, const float* y, float* dx, long long n, void* stream) {
  SVAE_REQUIRE(dy && y && dx && n > 0, SVAE_ERR_ARG, "relu_bwd: bad args");
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), dy, y, dx, n);
  return check_launch("relu_bwd");
}
extern "C" int svae_axpy(float a, const float* x, float* y, long long n, void* stream) {
  SVAE_REQUIRE(x && y && n > 0, SVAE_ERR_ARG, "axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), a, x, y, n);
  return check_launch("axpy");
}
extern "C" int svae_fill(float* x, float v, long long n, void* stream) {
  SVAE_REQUIRE(x && n > 0, SVAE_ERR_ARG, "fill: bad args");
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, ST(stream), x, v, n);
  return check_launch("fill");
}

extern "C" int svae_rowloss_blocks(int rows) { return (rows + 255) / 256; }

extern "C" int svae_mse_sum(const float* pred, int ld, const float* target, int ld_t, int rows, int C, float scale, float* part,
                            float* dpred, void* stream) {
  SVAE_REQUIRE(pred && target && part && rows > 0 && C > 0, SVAE_ERR_ARG, "mse_sum: bad args");
  hipLaunchKernelGGL(mse_sum_kernel, dim3(svae_rowloss_blocks(rows)), dim3(256), 0, ST(stream), pred, ld, target, ld_t, rows, C,
                     scale, part, dpred);
  return check_launch("mse_sum");
}
extern "C" int svae_ce_sum(const float* logits, int ld, const int* labels, int rows, int C, float scale, float* part,
                           float* dlogits, void* stream) {
  SVAE_REQUIRE(logits && labels && part && rows > 0 && C > 0, SVAE_ERR_ARG, "ce_sum: bad args");
  hipLaunchKernelGGL(ce_sum_kernel, dim3(svae_rowloss_blocks(rows)), dim3(256), 0, ST(stream), logits, ld, labels, rows, C, scale,
                     part, dlogits);
  return check_launch("ce_sum");
}
extern "C" int svae_double_softmax_ce_sum(const float* logits, int ld, int rows, float scale, float* part, float* dlogits,
                                          void* stream) {
  SVAE_REQUIRE(logits && part && rows > 0 && rows % 2 == 0, SVAE_ERR_ARG, "double_softmax_ce_sum: bad args");
  hipLaunchKernelGGL(double_softmax_ce_kernel, dim3(svae_rowloss_blocks(rows)), dim3(256), 0, ST(stream), logits, ld, rows, scale,
                     part, dlogits);
  return check_launch("double_softmax_ce_sum");
}

extern "C" int svae_bn_stats_finalize(const float* part, int n_chunks, double count, int C, const float* gamma,
                                      const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                      long long* num_batches_tracked, float* mean, float* rstd, float* scale, float* shift,
                                      void* stream) {
  SVAE_REQUIRE(part && gamma && beta && mean && rstd && scale && shift && count > 0 && n_chunks > 0, SVAE_ERR_ARG,
               "bn_stats_finalize: bad args");
  const int cpb = fin_cpb(C);
  hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3((C + cpb - 1) / cpb), dim3(256), 0, ST(stream), part, n_chunks, count, C, cpb, gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean, rstd, scale, shift);
  return check_launch("bn_stats_finalize");
}

extern "C" int svae_bn_bwd_reduce(const float* part, int n_chunks, int C, float* sums, float* dgamma, float* dbeta,
                                  float* dalpha, const float* dalpha_part, int n_parts, int accumulate, void* stream) {
  SVAE_REQUIRE(part && sums && n_chunks > 0 && (!dalpha || dalpha_part), SVAE_ERR_ARG, "bn_bwd_reduce: bad args");
  const int cpb = fin_cpb(C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((C + cpb - 1) / cpb), dim3(256), 0, ST(stream), part, n_chunks, C, cpb, sums, dgamma, dbeta,
                     dalpha, dalpha_part, n_parts, accumulate);
  return check_launch("bn_bwd_reduce");
}

extern "C" size_t svae_colsum_batched_workspace(const svae_colsum_task* tasks, int n) {
  if (!tasks || n <= 0 || n > SVAE_MAX_COLSUM_TASKS) return 0;
  size_t f = 0;
  for (int t = 0; t < n; ++t) f += (size_t)((tasks[t].rows + CS_ROWS - 1) / CS_ROWS) * tasks[t].C;
  return f * sizeof(float) + 256;
}

extern "C" int svae_colsum_batched(const svae_colsum_task* tasks, int n, void* ws, size_t ws_bytes, int accumulate,
                                   void* stream) {
  SVAE_REQUIRE(tasks && n > 0 && n <= SVAE_MAX_COLSUM_TASKS && ws, SVAE_ERR_ARG, "colsum_batched: bad args (n=%d)", n);
  SVAE_REQUIRE(ws_bytes >= svae_colsum_batched_workspace(tasks, n), SVAE_ERR_WORKSPACE, "colsum_batched: workspace too small");
  ColsumTasks T;
  memset(&T, 0, sizeof(T));
  T.n = n;
  long long off = 0;
  int blk = 0, col = 0;
  for (int t = 0; t < n; ++t) {
    SVAE_REQUIRE(tasks[t].x && tasks[t].out && tasks[t].rows > 0 && tasks[t].C > 0 && tasks[t].C % 4 == 0 && tasks[t].ld % 4 == 0,
                 SVAE_ERR_ARG, "colsum_batched: bad task %d", t);
    T.x[t] = tasks[t].x; T.out[t] = tasks[t].out; T.rows[t] = tasks[t].rows; T.C[t] = tasks[t].C; T.ld[t] = tasks[t].ld;
    const int chunks = (int)((tasks[t].rows + CS_ROWS - 1) / CS_ROWS);
    // the same matrix queued twice (the second conv and the skip conv of a residual block share their dY): the later task reads
    // the earlier one's partials and owns no stage-1 blocks
    int same = -1;
    for (int u = 0; u < t && same < 0; ++u)
      if (tasks[u].x == tasks[t].x && tasks[u].rows == tasks[t].rows && tasks[u].C == tasks[t].C && tasks[u].ld == tasks[t].ld) same = u;
    T.blk_begin[t] = blk;
    if (same >= 0) {
      T.part_off[t] = T.part_off[same];
    } else {
      T.part_off[t] = off;
      off += (long long)chunks * tasks[t].C;
      blk += chunks * ((tasks[t].C + 63) / 64);
    }
    T.col_begin[t] = col;
    col += tasks[t].C;
  }
  T.blk_begin[n] = blk;
  T.col_begin[n] = col;
  hipLaunchKernelGGL(colsum_batched_partial_kernel, dim3(blk), dim3(256), 0, ST(stream), T, (float*)ws);
  if (int e = check_launch("colsum_batched_partial")) return e;
  hipLaunchKernelGGL(colsum_batched_final_kernel, dim3((col + 7) / 8), dim3(256), 0, ST(stream), T, (const float*)ws, accumulate);
  return check_launch("colsum_batched_final");
}

// column sums from partials other kernels left behind (affine_prelu_bwd_apply_kernel): task t: out[c] (+)= sum over rows of part[rows][C],
// 32 row-lanes per column in fp64, combined in lane order (deterministic)
struct ColsumPartTasks {
  const float* part[SVAE_MAX_COLSUM_TASKS];
  float* out[SVAE_MAX_COLSUM_TASKS];
  int rows[SVAE_MAX_COLSUM_TASKS], C[SVAE_MAX_COLSUM_TASKS];
  int col_begin[SVAE_MAX_COLSUM_TASKS + 1];
  int n;
};

__global__ __launch_bounds__(256) void colsum_from_partials_kernel(const ColsumPartTasks T, int accumulate) {
  __shared__ double red[32][8];
  const int cl = threadIdx.x & 7, sub = threadIdx.x >> 3;
  const int gc = blockIdx.x * 8 + cl;
  const bool ok = gc < T.col_begin[T.n];
  int t = 0, c = 0, C = 1, rows = 0;
  if (ok) {
    while (t + 1 < T.n && gc >= T.col_begin[t + 1]) ++t;
    c = gc - T.col_begin[t];
    C = T.C[t];
    rows = T.rows[t];
  }
  double s = 0.0;
  for (int k = sub; k < rows; k += 32) s += (double)T.part[t][(long long)k * C + c];
  red[sub][cl] = s;
  __syncthreads();
  if (sub == 0 && ok) {
    double tot = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) tot += red[i][cl];
    float* o = T.out[t];
    o[c] = (accumulate ? o[c] : 0.f) + (float)tot;
  }
}

extern "C" int svae_colsum_from_partials(const svae_colsum_part_task* tasks, int n, int accumulate, void* stream) {
  SVAE_REQUIRE(tasks && n > 0 && n <= SVAE_MAX_COLSUM_TASKS, SVAE_ERR_ARG, "colsum_from_partials: bad args (n=%d)", n);
  ColsumPartTasks T;
  memset(&T, 0, sizeof(T));
  T.n = n;
  int col = 0;
  for (int t = 0; t < n; ++t) {
    SVAE_REQUIRE(tasks[t].part && tasks[t].out && tasks[t].rows > 0 && tasks[t].C > 0, SVAE_ERR_ARG, "colsum_from_partials: bad task %d", t);
    T.part[t] = tasks[t].part; T.out[t] = tasks[t].out; T.rows[t] = tasks[t].rows; T.C[t] = tasks[t].C;
    T.col_begin[t] = col;
    col += tasks[t].C;
  }
  T.col_begin[n] = col;
  hipLaunchKernelGGL(colsum_from_partials_kernel, dim3((col + 7) / 8), dim3(256), 0, ST(stream), T, accumulate);
  return check_launch("colsum_from_partials");
}
