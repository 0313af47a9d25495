// Full-Cholesky latent heads (model.diag = False) for gfx950.
//   CholeskyL.forward (reference residual.py:39-68): raw [B, z(z+1)/2] in torch.tril_indices
//   order (row-major lower triangle) -> L with softplus on the diagonal;
//   VAE.sampling (residual.py:305-316): z = L eps + mu;
//   prior_loss (losses.py:138-146): -0.5 * sum(1 + 2 log diag(L) - mu^2 - diag(L L^T)) -- the
//   z^3 bmm of the reference is replaced by the row sums of squares it equals.
// One thread per (sample, latent row); HBM-bound and tiny next to the trunk.
#include "svae_internal.h"

namespace svae {

__device__ __forceinline__ float softplus_l(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ __launch_bounds__(256) void heads_tril_fwd_kernel(const float* __restrict__ h, int ld, const float* __restrict__ eps,
                                                              float* __restrict__ mu, int ldm, float* __restrict__ L,
                                                              float* __restrict__ z, int ldz, float* __restrict__ kl_part, int batch,
                                                              int zd, int raw_off) {
  __shared__ float red4[4];
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  float kl = 0.f;
  if (t < (long long)batch * zd) {
    const int b = (int)(t / zd), i = (int)(t - (long long)b * zd);
    const float* hb = h + (long long)b * ld;
    const float* raw = hb + raw_off + (long long)i * (i + 1) / 2;
    float* Lr = L + ((long long)b * zd + i) * zd;
    const float m = hb[i];
    float acc = 0.f, ss = 0.f;
    for (int j = 0; j < i; ++j) {
      const float v = raw[j];
      Lr[j] = v;
      ss += v * v;
      if (eps) acc += v * eps[(long long)b * zd + j];
    }
    const float d = softplus_l(raw[i]);
    Lr[i] = d;
    ss += d * d;
    if (eps) acc += d * eps[(long long)b * zd + i];
    for (int j = i + 1; j < zd; ++j) Lr[j] = 0.f;
    mu[(long long)b * ldm + i] = m;
    z[(long long)b * ldz + i] = m + acc;
    kl = -0.5f * (1.f + 2.f * logf(d) - m * m - ss);
  }
  const float tot = block_sum_256(kl, red4);
  if (threadIdx.x == 0) kl_part[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void heads_tril_bwd_kernel(const float* __restrict__ h, int ld, const float* __restrict__ eps,
                                                              const float* __restrict__ L, const float* __restrict__ dz, int lddz,
                                                              const float* __restrict__ dmu, int ldm, float kl_scale,
                                                              float* __restrict__ dh, int batch, int zd, int raw_off) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)batch * zd) return;
  const int b = (int)(t / zd), i = (int)(t - (long long)b * zd);
  const float* hb = h + (long long)b * ld;
  float* dhb = dh + (long long)b * ld;
  const float* Lr = L + ((long long)b * zd + i) * zd;
  const long long ro = raw_off + (long long)i * (i + 1) / 2;
  const float gz = dz ? dz[(long long)b * lddz + i] : 0.f;
  float gm = gz + kl_scale * hb[i];
  if (dmu) gm += dmu[(long long)b * ldm + i];
  dhb[i] = gm;
  for (int j = 0; j < i; ++j) dhb[ro + j] = (eps ? gz * eps[(long long)b * zd + j] : 0.f) + kl_scale * Lr[j];
  const float d = Lr[i];
  const float gd = (eps ? gz * eps[(long long)b * zd + i] : 0.f) + kl_scale * (d - 1.f / d);
  dhb[ro + i] = gd / (1.f + expf(-hb[ro + i]));
}

}  // namespace svae

using namespace svae;

extern "C" int svae_heads_tril_fwd(const float* h, int ld, const float* eps, float* mu, int ldm, float* L, float* z, int ldz,
                                   float* kl_part, int batch, int zdim, int raw_off, void* stream) {
  SVAE_REQUIRE(h && mu && L && z && kl_part && batch > 0 && zdim > 0 && raw_off >= zdim &&
                   ld >= raw_off + zdim * (zdim + 1) / 2 && ldz >= zdim && ldm >= zdim,
               SVAE_ERR_ARG, "heads_tril_fwd: bad args");
  hipLaunchKernelGGL(heads_tril_fwd_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, (hipStream_t)stream, h, ld, eps, mu,
                     ldm, L, z, ldz, kl_part, batch, zdim, raw_off);
  return check_launch("heads_tril_fwd");
}

extern "C" int svae_heads_tril_bwd(const float* h, int ld, const float* eps, const float* L, const float* dz, int lddz,
                                   const float* dmu, int ldm, float kl_scale, float* dh, int batch, int zdim, int raw_off,
                                   void* stream) {
  SVAE_REQUIRE(h && L && dh && batch > 0 && zdim > 0, SVAE_ERR_ARG, "heads_tril_bwd: bad args");
  hipLaunchKernelGGL(heads_tril_bwd_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, (hipStream_t)stream, h, ld, eps, L,
                     dz, lddz, dmu, ldm, kl_scale, dh, batch, zdim, raw_off);
  return check_launch("heads_tril_bwd");
}
