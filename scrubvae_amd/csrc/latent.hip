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
                                                              const float* __restrict__ dlv,
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
  // lv_i = log(sum_j L_ij^2): an upstream d/d lv_i reaches L_ij as dlv_i * 2 L_ij / rowss_i
  float lvk = 0.f;
  if (dlv != nullptr) {
    float ssq = 0.f;
    for (int j = 0; j <= i; ++j) ssq += Lr[j] * Lr[j];
    lvk = 2.f * dlv[(long long)b * zd + i] / ssq;
  }
  for (int j = 0; j < i; ++j) dhb[ro + j] = (eps ? gz * eps[(long long)b * zd + j] : 0.f) + (kl_scale + lvk) * Lr[j];
  const float d = Lr[i];
  const float gd = (eps ? gz * eps[(long long)b * zd + i] : 0.f) + kl_scale * (d - 1.f / d) + lvk * d;
  dhb[ro + i] = gd / (1.f + expf(-hb[ro + i]));
}

// ------------------------------------------------------------------ total correlation
// beta-TCVAE minibatch estimator (reference losses.py:41-101):
//   lq[j,i,l] = -0.5*(exp(-lv[i,l])*(z[j,l]-mu[i,l])^2 + lv[i,l] + log(2 pi)),  z detached
//   loss_j = logsumexp_i(sum_l lq[j,i,l]) - sum_l logsumexp_i(lq[j,i,l]);  TC = mean_j loss_j
// One workgroup per sample j; 8 groups of 32 lanes: lane = latent dim (chunks of 32), group =
// stripe of the i loop.  Online logsumexp per lane (over i) and per group (over i of the
// lane-summed a_i); O(B^2 z) work, no [B,B,z] tensor is ever materialised.
constexpr float LN2PI_F = 1.8378770664093453f;
constexpr int TC_MAXCH = 4;  // z_dim <= 128

__device__ __forceinline__ void lse_push(float& m, float& s, float x) {
  if (x > m) { s = s * expf(m - x) + 1.f; m = x; }
  else s += expf(x - m);
}
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  if (m2 > m) { s = s * expf(m - m2) + s2; m = m2; }
  else s += s2 * expf(m2 - m);
}
__device__ __forceinline__ float half_sum32(float v) {  // sum over the 32 lanes of a half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void tc_prep_kernel(const float* __restrict__ sigma, int lds, const float* __restrict__ L,
                                                       float* __restrict__ lv, int batch, int zd) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)batch * zd) return;
  const int b = (int)(t / zd), i = (int)(t - (long long)b * zd);
  if (L == nullptr) {
    lv[t] = 2.f * logf(sigma[(long long)b * lds + i]);
  } else {
    const float* r = L + ((long long)b * zd + i) * zd;
    float ss = 0.f;
    for (int j = 0; j <= i; ++j) ss += r[j] * r[j];
    lv[t] = logf(ss);
  }
}

__global__ __launch_bounds__(256) void tc_fwd_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ mu, int ldm,
                                                      const float* __restrict__ lv, int batch, int zd, float* __restrict__ lse_l,
                                                      float* __restrict__ lse_a, float* __restrict__ loss) {
  __shared__ float sm[8][TC_MAXCH * 32], ss[8][TC_MAXCH * 32], sM[8], sT[8], red4[4];
  const int j = blockIdx.x, lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int nch = (zd + 31) / 32;
  float zj[TC_MAXCH], m[TC_MAXCH], s[TC_MAXCH];
#pragma unroll
  for (int c = 0; c < TC_MAXCH; ++c) {
    const int l = c * 32 + lane;
    zj[c] = (c < nch && l < zd) ? z[(long long)j * ldz + l] : 0.f;
    m[c] = -3.0e38f;
    s[c] = 0.f;
  }
  float M = -3.0e38f, T = 0.f;
  for (int i = grp; i < batch; i += 8) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < TC_MAXCH; ++c) {
      const int l = c * 32 + lane;
      if (c < nch && l < zd) {
        const float v = lv[(long long)i * zd + l], d = zj[c] - mu[(long long)i * ldm + l];
        const float q = -0.5f * (expf(-v) * d * d + v + LN2PI_F);
        lse_push(m[c], s[c], q);
        a += q;
      }
    }
    a = half_sum32(a);
    lse_push(M, T, a);
  }
#pragma unroll
  for (int c = 0; c < TC_MAXCH; ++c) { sm[grp][c * 32 + lane] = m[c]; ss[grp][c * 32 + lane] = s[c]; }
  if (lane == 0) { sM[grp] = M; sT[grp] = T; }
  __syncthreads();
  float prod = 0.f;
  if (threadIdx.x < zd) {
    float mm = sm[0][threadIdx.x], s2 = ss[0][threadIdx.x];
    for (int g2 = 1; g2 < 8; ++g2) lse_merge(mm, s2, sm[g2][threadIdx.x], ss[g2][threadIdx.x]);
    const float l = mm + logf(s2);
    lse_l[(long long)j * zd + threadIdx.x] = l;
    prod = l;
  }
  const float tot = block_sum_256(prod, red4);
  if (threadIdx.x == 0) {
    float mm = sM[0], s2 = sT[0];
    for (int g2 = 1; g2 < 8; ++g2) lse_merge(mm, s2, sM[g2], sT[g2]);
    const float la = mm + logf(s2);
    lse_a[j] = la;
    loss[j] = la - tot;
  }
}

// gradient w.r.t. mu[i,:] and lv[i,:] (z is detached): one workgroup per i, loop over j
__global__ __launch_bounds__(256) void tc_bwd_kernel(const float* __restrict__ z, int ldz, const float* __restrict__ mu, int ldm,
                                                      const float* __restrict__ lv, int batch, int zd,
                                                      const float* __restrict__ lse_l, const float* __restrict__ lse_a, float w,
                                                      float* __restrict__ d_mu, int ldd, float* __restrict__ d_lv, int ldv,
                                                      const float* __restrict__ sigma, int lds) {
  __shared__ float g1[8][TC_MAXCH * 32], g2s[8][TC_MAXCH * 32];
  const int i = blockIdx.x, lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int nch = (zd + 31) / 32;
  float mi[TC_MAXCH], vi[TC_MAXCH], ei[TC_MAXCH], gm[TC_MAXCH], gv[TC_MAXCH];
#pragma unroll
  for (int c = 0; c < TC_MAXCH; ++c) {
    const int l = c * 32 + lane;
    const bool ok = c < nch && l < zd;
    mi[c] = ok ? mu[(long long)i * ldm + l] : 0.f;
    vi[c] = ok ? lv[(long long)i * zd + l] : 0.f;
    ei[c] = expf(-vi[c]);
    gm[c] = gv[c] = 0.f;
  }
  for (int j = grp; j < batch; j += 8) {
    float q[TC_MAXCH], d[TC_MAXCH];
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < TC_MAXCH; ++c) {
      const int l = c * 32 + lane;
      q[c] = 0.f; d[c] = 0.f;
      if (c < nch && l < zd) {
        d[c] = z[(long long)j * ldz + l] - mi[c];
        q[c] = -0.5f * (ei[c] * d[c] * d[c] + vi[c] + LN2PI_F);
        a += q[c];
      }
    }
    a = half_sum32(a);
    const float p = expf(a - lse_a[j]);
#pragma unroll
    for (int c = 0; c < TC_MAXCH; ++c) {
      const int l = c * 32 + lane;
      if (c < nch && l < zd) {
        const float g = w * (p - expf(q[c] - lse_l[(long long)j * zd + l]));
        gm[c] += g * ei[c] * d[c];
        gv[c] += g * 0.5f * (ei[c] * d[c] * d[c] - 1.f);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < TC_MAXCH; ++c) { g1[grp][c * 32 + lane] = gm[c]; g2s[grp][c * 32 + lane] = gv[c]; }
  __syncthreads();
  if (threadIdx.x < zd) {
    float a1 = 0.f, a2 = 0.f;
    for (int g = 0; g < 8; ++g) { a1 += g1[g][threadIdx.x]; a2 += g2s[g][threadIdx.x]; }
    d_mu[(long long)i * ldd + threadIdx.x] += a1;
    // diagonal posterior: lv = 2 log sigma, so hand back d/d sigma = d/d lv * 2 / sigma
    if (sigma != nullptr) a2 *= 2.f / sigma[(long long)i * lds + threadIdx.x];
    d_lv[(long long)i * ldv + threadIdx.x] = a2;
  }
}


// ------------------------------------------------------------------------------------------ Beta posterior (model.prior = "beta")
// Reference: ResidualEncoder.forward residual.py:235-239 (alpha = softplus(fc_alpha) + 1, beta likewise), ResVAE.encode :453-456
// (mu = the mode rescaled to (-1, 1)), VAE.forward :328-331 (z = Beta(alpha, beta).rsample() * 2 - 1), losses.py:198-206
// (KL(Beta(alpha, beta) || Beta(1, 1)).sum() / B).  The arithmetic behind `rsample` and `kl_divergence` lives in the reference's
// dependency PyTorch (environment.yml: pytorch=1.13.1; same algorithm in 2.10): the draw goes through Dirichlet([alpha, beta]) and
// is differentiated IMPLICITLY -- torch/distributions/dirichlet.py: grad = dirichlet_grad(x, conc, total) * (g - sum(x g)) -- with
// the scaled reparameterised gradient  D(x; a, total) = -(d/da cdf(x; a, total - a)) / pdf(x) / (1 - x)  evaluated by the published
// piecewise approximation of ATen/native/Distributions.h (dirichlet_grad_one): a Taylor series for x near 0, the mirrored one for x
// near 1, a Rice saddle-point expansion when both shapes exceed 6, else a fitted rational correction (its 72 coefficients are part of
// the algorithm) to the analytic approximation x (psi(total) - psi(a)) / b.  Restated here in fp64 per element -- [B, z] elements, nothing
// next to the trunk -- so that gradients agree with the CPU reference's fp64-accumulator evaluation.  The draw x itself comes from the
// caller (torch's sampler as RNG plumbing, like eps of the Gaussian heads; injected in parity tests).
__device__ inline double digamma_d(double x) {  // Cephes-style: recurrence up to x >= 10, then the asymptotic series
  double r = 0.0;
  while (x < 10.0) { r -= 1.0 / x; x += 1.0; }
  const double z = 1.0 / (x * x);
  const double y = z * (8.33333333333333333333e-2 + z * (-8.33333333333333333333e-3 + z * (3.96825396825396825397e-3 +
                   z * (-4.16666666666666666667e-3 + z * (7.57575757575757575758e-3 + z * (-2.10927960927960927961e-2 +
                   z * 8.33333333333333333333e-2))))));
  return r + log(x) - 0.5 / x - y;
}
__device__ inline double trigamma_d(double x) {
  double r = 0.0;
  while (x < 10.0) { r += 1.0 / (x * x); x += 1.0; }
  const double z = 1.0 / (x * x);
  return r + 1.0 / x + 0.5 * z + (1.0 / x) * z * (1.0 / 6.0 + z * (-1.0 / 30.0 + z * (1.0 / 42.0 + z * (-1.0 / 30.0 + z * (5.0 / 66.0)))));
}
__device__ inline double beta_grad_alpha_small(double x, double a, double b) {  // x near 0: Taylor series in x
  const double factor = digamma_d(a) - digamma_d(a + b) - log(x);
  double numer = 1.0, series = numer / a * (factor + 1.0 / a);
  for (int i = 1; i <= 10; ++i) {
    numer *= ((double)i - b) * x / (double)i;
    const double denom = a + (double)i;
    series += numer / denom * (factor + 1.0 / denom);
  }
  const double r = x * pow(1.0 - x, -b) * series;
  return r != r ? 0.0 : r;
}
__device__ inline double beta_grad_beta_small(double x, double a, double b) {
  const double factor = digamma_d(a + b) - digamma_d(b);
  double numer = 1.0, betas = 1.0, dbetas = 0.0, series = factor / a;
  for (int i = 1; i <= 8; ++i) {
    numer *= -x / (double)i;
    dbetas = dbetas * (b - (double)i) + betas;
    betas = betas * (b - (double)i);
    series += numer / (a + (double)i) * (dbetas + factor * betas);
  }
  const double r = -pow(1.0 - x, 1.0 - b) * series;
  return r != r ? 0.0 : r;
}
__device__ inline double beta_grad_alpha_mid(double x, double a, double b) {  // both shapes large: Rice saddle-point expansion
  const double total = a + b, mean = a / total, sd = sqrt(a * b / (total + 1.0)) / total;
  if (mean - 0.1 * sd <= x && x <= mean + 0.1 * sd) {  // the singularity at x = mean
    const double poly = 47.0 * x * (b * b) * (b * b) + a * ((43.0 + 20.0 * (16.0 + 27.0 * b) * x) * (b * b) * b + a * (
                        3.0 * (59.0 + 180.0 * b - 90.0 * x) * (b * b) + a * ((453.0 + 1620.0 * b * (1.0 - x) - 455.0 * x) * b + a * (
                        8.0 * (1.0 - x) * (135.0 * b - 11.0)))));
    const double pn = (1.0 + 12.0 * a) * (1.0 + 12.0 * b) / (total * total);
    const double pd = 12960.0 * a * a * a * b * b * (1.0 + 12.0 * total);
    return pn / (1.0 - x) * poly / pd;
  }
  const double prefactor = -x / sqrt(2.0 * a * b / total);
  const double stirling = (1.0 + 1.0 / (12.0 * a) + 1.0 / (288.0 * a * a)) * (1.0 + 1.0 / (12.0 * b) + 1.0 / (288.0 * b * b)) /
                          (1.0 + 1.0 / (12.0 * total) + 1.0 / (288.0 * total * total));
  const double t1n = 2.0 * (a * a) * (x - 1.0) + a * b * (x - 1.0) - x * (b * b);
  const double axbx = a * (x - 1.0) + b * x;
  const double t1 = t1n / (sqrt(2.0 * a / b) * pow(total, 1.5) * axbx * axbx);
  const double t2 = 0.5 * log(a / (total * x));
  const double t3 = sqrt(8.0 * a * b / total) / (b * x + a * (x - 1.0));
  const double t4 = pow(b * log(b / (total * (1.0 - x))) + a * log(a / (total * x)), -1.5);
  return stirling * prefactor * (t1 + t2 * (t3 + (x < mean ? t4 : -t4)));
}
__device__ const double kDirichletC[2][3][3][4] = {  // fitted coefficients of ATen's rational correction (part of the published algorithm)
    {{{1.003668233, -0.01061107488, -0.0657888334, 0.01201642863},
      {0.6336835991, -0.3557432599, 0.05486251648, -0.001465281033},
      {-0.03276231906, 0.004474107445, 0.002429354597, -0.0001557569013}},
     {{0.221950385, -0.3187676331, 0.01799915743, 0.01074823814},
      {-0.2951249643, 0.06219954479, 0.01535556598, 0.001550077057},
      {0.02155310298, 0.004170831599, 0.001292462449, 6.976601077e-05}},
     {{-0.05980841433, 0.008441916499, 0.01085618172, 0.002319392565},
      {0.02911413504, 0.01400243777, -0.002721828457, 0.000751041181},
      {0.005900514878, -0.001936558688, -9.495446725e-06, 5.385558597e-05}}},
    {{{1, -0.02924021934, -0.04438342661, 0.007285809825},
      {0.6357567472, -0.3473456711, 0.05454656494, -0.002407477521},
      {-0.03301322327, 0.004845219414, 0.00231480583, -0.0002307248149}},
     {{0.5925320577, -0.1757678135, 0.01505928619, 0.000564515273},
      {0.1014815858, -0.06589186703, 0.01272886114, -0.0007316646956},
      {-0.007258481865, 0.001096195486, 0.0003934994223, -4.12701925e-05}},
     {{0.06469649321, -0.0236701437, 0.002902096474, -5.896963079e-05},
      {0.001925008108, -0.002869809258, 0.0008000589141, -6.063713228e-05},
      {-0.0003477407336, 6.959756487e-05, 1.097287507e-05, -1.650964693e-06}}}};
__device__ inline double dirichlet_grad_d(float xf, float af, float totalf) {
  // the branch decisions are taken in the input precision (float), the evaluation in double -- as the CPU kernel does for float tensors
  const float bf = totalf - af, boundary = totalf * xf * (1.f - xf);
  const double x = xf, a = af, total = totalf, b = total - a;
  if (xf <= 0.5f && boundary < 2.5f) return (double)(float)beta_grad_alpha_small((float)x, (float)a, (float)bf);
  if (xf >= 0.5f && boundary < 0.75f) return -(double)(float)beta_grad_beta_small((double)(1.f - xf), (double)bf, (double)af);
  if (af > 6.f && bf > 6.f) return beta_grad_alpha_mid(x, a, b);
  const double u = log(x), la = log(a) - u, lb = log(total) - la;
  const double pu[3] = {1.0, u, u * u}, pa[3] = {1.0, la, la * la};
  double p = 0.0, q = 0.0;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double ua = pu[i] * pa[j];
      p += ua * (kDirichletC[0][i][j][0] + lb * (kDirichletC[0][i][j][1] + lb * (kDirichletC[0][i][j][2] + lb * kDirichletC[0][i][j][3])));
      q += ua * (kDirichletC[1][i][j][0] + lb * (kDirichletC[1][i][j][1] + lb * (kDirichletC[1][i][j][2] + lb * kDirichletC[1][i][j][3])));
    }
  return p / q * (x * (digamma_d(total) - digamma_d(a)) / b);
}

// alpha / beta / mu [B, ldm], kl_part[blocks] = partial of sum KL(Beta(alpha, beta) || Beta(1, 1)) (before the / B)
__global__ __launch_bounds__(256) void heads_beta_fwd_kernel(const float* __restrict__ h, int ld, float* __restrict__ alpha,
                                                              float* __restrict__ beta, float* __restrict__ mu, int ldm,
                                                              float* __restrict__ kl_part, int batch, int zd, int raw_off) {
  __shared__ float red4[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float kl = 0.f;
  if (i < (long long)batch * zd) {
    const int b = (int)(i / zd), k = (int)(i - (long long)b * zd);
    const float a = softplus_l(h[(long long)b * ld + k]) + 1.f, bt = softplus_l(h[(long long)b * ld + raw_off + k]) + 1.f;
    alpha[(long long)b * ldm + k] = a;
    beta[(long long)b * ldm + k] = bt;
    mu[(long long)b * ldm + k] = (a - 1.f + 1e-8f) / (a + bt - 2.f + 2e-8f) * 2.f - 1.f;
    const double ad = a, bd = bt, sd = (double)a + (double)bt;
    kl = (float)(lgamma(sd) - lgamma(ad) - lgamma(bd) + (ad - 1.0) * digamma_d(ad) + (bd - 1.0) * digamma_d(bd) + (2.0 - sd) * digamma_d(sd));
  }
  const float t = block_sum_256(kl, red4);
  if (threadIdx.x == 0) kl_part[blockIdx.x] = t;
}

// dh = [d raw_alpha | d raw_beta] from: dz (gradient wrt z = 2 x - 1 through the implicit reparameterisation of the draw x), dmu (seed on
// the rescaled mode: the scrubbing heads read mu), kl_scale * d KL
__global__ __launch_bounds__(256) void heads_beta_bwd_kernel(const float* __restrict__ h, int ld, const float* __restrict__ x,
                                                              const float* __restrict__ alpha, const float* __restrict__ beta, int ldm,
                                                              const float* __restrict__ dz, int lddz, const float* __restrict__ dmu,
                                                              float kl_scale, float* __restrict__ dh, int batch, int zd, int raw_off) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)batch * zd) return;
  const int b = (int)(i / zd), k = (int)(i - (long long)b * zd);
  const float a = alpha[(long long)b * ldm + k], bt = beta[(long long)b * ldm + k];
  const float total = a + bt;
  double ga = 0.0, gb = 0.0;
  if (dz) {
    const float xv = x[i];
    const double g = 2.0 * (double)dz[(long long)b * lddz + k];
    ga += dirichlet_grad_d(xv, a, total) * (1.0 - (double)xv) * g;
    gb -= dirichlet_grad_d(1.f - xv, bt, total) * (double)xv * g;
  }
  if (kl_scale != 0.f) {
    const double tg = (2.0 - (double)total) * trigamma_d((double)total);
    ga += (double)kl_scale * (((double)a - 1.0) * trigamma_d((double)a) + tg);
    gb += (double)kl_scale * (((double)bt - 1.0) * trigamma_d((double)bt) + tg);
  }
  if (dmu) {
    const double gm = dmu[(long long)b * ldm + k], S = (double)a + (double)bt - 2.0 + 2e-8, N = (double)a - 1.0 + 1e-8;
    ga += gm * 2.0 * (S - N) / (S * S);
    gb -= gm * 2.0 * N / (S * S);
  }
  const float ra = h[(long long)b * ld + k], rb = h[(long long)b * ld + raw_off + k];
  dh[(long long)b * ld + k] = (float)(ga / (1.0 + exp(-(double)ra)));
  dh[(long long)b * ld + raw_off + k] = (float)(gb / (1.0 + exp(-(double)rb)));
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
                                   const float* dmu, int ldm, float kl_scale, const float* dlv, float* dh, int batch, int zdim,
                                   int raw_off, void* stream) {
  SVAE_REQUIRE(h && L && dh && batch > 0 && zdim > 0, SVAE_ERR_ARG, "heads_tril_bwd: bad args");
  hipLaunchKernelGGL(heads_tril_bwd_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, (hipStream_t)stream, h, ld, eps, L,
                     dz, lddz, dmu, ldm, kl_scale, dlv, dh, batch, zdim, raw_off);
  return check_launch("heads_tril_bwd");
}


extern "C" int svae_heads_beta_fwd(const float* h, int ld, float* alpha, float* beta, float* mu, int ldm, float* kl_part, int batch,
                                   int zdim, int raw_off, void* stream) {
  SVAE_REQUIRE(h && alpha && beta && mu && kl_part && batch > 0 && zdim > 0 && raw_off >= zdim && ld >= raw_off + zdim && ldm >= zdim,
               SVAE_ERR_ARG, "heads_beta_fwd: bad args");
  const int blocks = (int)(((long long)batch * zdim + 255) / 256);
  hipLaunchKernelGGL(heads_beta_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h, ld, alpha, beta, mu, ldm, kl_part, batch,
                     zdim, raw_off);
  return check_launch("heads_beta_fwd");
}

extern "C" int svae_heads_beta_bwd(const float* h, int ld, const float* x, const float* alpha, const float* beta, int ldm, const float* dz,
                                   int lddz, const float* dmu, float kl_scale, float* dh, int batch, int zdim, int raw_off, void* stream) {
  SVAE_REQUIRE(h && alpha && beta && dh && batch > 0 && zdim > 0 && (!dz || x), SVAE_ERR_ARG, "heads_beta_bwd: bad args");
  const int blocks = (int)(((long long)batch * zdim + 255) / 256);
  hipLaunchKernelGGL(heads_beta_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, h, ld, x, alpha, beta, ldm, dz, lddz, dmu,
                     kl_scale, dh, batch, zdim, raw_off);
  return check_launch("heads_beta_bwd");
}

extern "C" int svae_tc_logvar(const float* sigma, int lds, const float* L, float* lv, int batch, int zdim, void* stream) {
  SVAE_REQUIRE((sigma || L) && lv && batch > 0 && zdim > 0, SVAE_ERR_ARG, "tc_logvar: bad args");
  hipLaunchKernelGGL(tc_prep_kernel, dim3(svae_heads_blocks(batch, zdim)), dim3(256), 0, (hipStream_t)stream, sigma, lds, L, lv,
                     batch, zdim);
  return check_launch("tc_logvar");
}

extern "C" int svae_tc_fwd(const float* z, int ldz, const float* mu, int ldm, const float* lv, int batch, int zdim,
                           float* lse_l, float* lse_a, float* loss, void* stream) {
  SVAE_REQUIRE(z && mu && lv && lse_l && lse_a && loss && batch > 0 && zdim > 0 && zdim <= 32 * TC_MAXCH, SVAE_ERR_ARG,
               "tc_fwd: bad args (z_dim must be <= %d)", 32 * TC_MAXCH);
  hipLaunchKernelGGL(tc_fwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, z, ldz, mu, ldm, lv, batch, zdim, lse_l, lse_a,
                     loss);
  return check_launch("tc_fwd");
}

extern "C" int svae_tc_bwd(const float* z, int ldz, const float* mu, int ldm, const float* lv, int batch, int zdim,
                           const float* lse_l, const float* lse_a, float weight, float* d_mu, int ldd, float* d_lv, int ldv,
                           const float* sigma, int lds, void* stream) {
  SVAE_REQUIRE(z && mu && lv && lse_l && lse_a && d_mu && d_lv && batch > 0 && zdim > 0 && zdim <= 32 * TC_MAXCH, SVAE_ERR_ARG,
               "tc_bwd: bad args");
  hipLaunchKernelGGL(tc_bwd_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, z, ldz, mu, ldm, lv, batch, zdim, lse_l, lse_a,
                     weight, d_mu, ldd, d_lv, ldv, sigma, lds);
  return check_launch("tc_bwd");
}

// ---------------------------------------------------------------- batched small dense solves
// The streaming scrubbers solve z x z (plus a bias column) normal equations every step -- MovingAvgLeastSquares.forward
// (reference model/disentangle.py:466-486: two systems, W = (Sxx + l2 I)^-1 Sxy) and direct_lsq_loss (train/losses.py:173-179) --
// which torch.linalg.solve turns into a handful of solver-library launches per system.  One workgroup per system here: LU with partial
// pivoting on the augmented matrix [A + diag | B] held in LDS (n <= 64 rows = lanes of wave 0 for the pivot search, the row
// updates spread over 256 threads), the same elimination order as the reference's LAPACK getrf / getrs without blocking, fp32.
// A singular pivot (exactly zero) yields inf / nan like the library does.
constexpr int SOLVE_MAXN = 64, SOLVE_MAXRHS = 64;
__global__ __launch_bounds__(256) void small_solve_kernel(const float* __restrict__ A, long long strideA, const float* __restrict__ diag,
                                                           const float* __restrict__ Bm, long long strideB, float* __restrict__ X,
                                                           long long strideX, int n, int nrhs) {
  __shared__ float M[SOLVE_MAXN][SOLVE_MAXN + SOLVE_MAXRHS + 1];
  __shared__ int piv;
  const int tid = threadIdx.x, w = n + nrhs;
  const float* a = A + blockIdx.x * strideA;
  const float* b = Bm + blockIdx.x * strideB;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    M[i][j] = a[e] + ((diag && i == j) ? diag[i] : 0.f);
  }
  for (int e = tid; e < n * nrhs; e += 256) {
    const int i = e / nrhs, j = e - i * nrhs;
    M[i][n + j] = b[e];
  }
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    if (tid < 64) {  // pivot: the first row of maximal |M[i][k]|, i >= k
      float v = (tid >= k && tid < n) ? fabsf(M[tid][k]) : -1.f;
      int idx = tid;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (tid == 0) piv = idx;
    }
    __syncthreads();
    const int p = piv;
    if (p != k)
      for (int j = tid; j < w; j += 256) { const float t = M[k][j]; M[k][j] = M[p][j]; M[p][j] = t; }
    __syncthreads();
    const float inv = 1.f / M[k][k];
    // rows i > k: l = M[i][k] / pivot, M[i][j] -= l M[k][j] for j > k (the multiplier column is not kept)
    const int rows = n - k - 1, cols = w - k - 1;
    for (int e = tid; e < rows * cols; e += 256) {
      const int i = k + 1 + e / cols, j = k + 1 + e % cols;
      M[i][j] -= (M[i][k] * inv) * M[k][j];
    }
    __syncthreads();
  }
  // back substitution, one right-hand side per thread
  if (tid < nrhs) {
    const int c = n + tid;
    for (int i = n - 1; i >= 0; --i) {
      float sacc = M[i][c];
      for (int j = i + 1; j < n; ++j) sacc -= M[i][j] * M[j][c];
      M[i][c] = sacc / M[i][i];
    }
  }
  __syncthreads();
  float* x = X + blockIdx.x * strideX;
  for (int e = tid; e < n * nrhs; e += 256) {
    const int i = e / nrhs, j = e - i * nrhs;
    x[e] = M[i][n + j];
  }
}

extern "C" int svae_small_solve(const float* A, long long strideA, const float* diag, const float* B, long long strideB, float* X,
                                long long strideX, int n, int nrhs, int batch, void* stream) {
  SVAE_REQUIRE(A && B && X && n >= 1 && n <= SOLVE_MAXN && nrhs >= 1 && nrhs <= SOLVE_MAXRHS && batch >= 1, SVAE_ERR_ARG,
               "small_solve: n <= %d, nrhs <= %d", SOLVE_MAXN, SOLVE_MAXRHS);
  hipLaunchKernelGGL(small_solve_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream, A, strideA, diag, B, strideB, X, strideX, n, nrhs);
  return check_launch("small_solve");
}

// ------------------------------------------------------------------------------------------------------------------
// Gaussian log-likelihoods of the streaming quadratic discriminants (reference: QuadraticDiscriminantFilter.cgll,
// disentangle.py:129-134: -0.5 (logdet S + r^T S^-1 r), r = x - m, evaluated 4 x per class and step through
// torch.linalg.solve / torch.logdet).  One launch for all (mean, covariance) pairs: grid (sample tiles, pairs); every workgroup
// inverts ITS pair's S in LDS ([S | I], Gauss-Jordan with partial pivoting: n <= 64, a few thousand operations) and its 256 threads
// then own one sample each: y1 = S^-1 r, y2 = S^-T r, ll = -0.5 (logdet + r . y1), and -- what the backward pass needs --
// the gradient field d ll / d x = -0.5 (y1 + y2).  logdet follows torch.logdet: NaN for a negative determinant, -inf for a
// singular matrix.
template <int NT>
__global__ __launch_bounds__(256) void gauss_ll_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                        const float* __restrict__ S, float* __restrict__ ll, float* __restrict__ grad,
                                                        int batch, int n) {
  __shared__ float M[SOLVE_MAXN][2 * SOLVE_MAXN + 1];
  __shared__ int piv;
  __shared__ float logdet_s;
  const int tid = threadIdx.x, w = 2 * n, pair = blockIdx.y;
  const float* a = S + (long long)pair * n * n;
  for (int e = tid; e < n * n; e += 256) {
    const int i = e / n, j = e - i * n;
    M[i][j] = a[e];
    M[i][n + j] = i == j ? 1.f : 0.f;
  }
  float logdet = 0.f;
  int neg = 0;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    if (tid < 64) {
      float v = (tid >= k && tid < n) ? fabsf(M[tid][k]) : -1.f;
      int idx = tid;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
      }
      if (tid == 0) piv = idx;
    }
    __syncthreads();
    const int p = piv;
    if (p != k) {
      neg ^= 1;
      for (int j = tid; j < w; j += 256) { const float t = M[k][j]; M[k][j] = M[p][j]; M[p][j] = t; }
    }
    __syncthreads();
    const float pv = M[k][k];
    logdet += logf(fabsf(pv));  // log(0) = -inf, as torch.logdet of a singular matrix
    neg ^= pv < 0.f;
    const float inv = 1.f / pv;
    // Gauss-Jordan: every row i != k loses its column-k entry (columns > k only: the others are final or unused)
    const int cols = w - k - 1;
    for (int e = tid; e < (n - 1) * cols; e += 256) {
      int i = e / cols;
      i += i >= k;
      const int j = k + 1 + e % cols;
      M[i][j] -= (M[i][k] * inv) * M[k][j];
    }
    __syncthreads();
    for (int j = k + 1 + tid; j < w; j += 256) M[k][j] *= inv;  // row k normalised; its rows-i updates above used the old row through `inv`
    __syncthreads();
  }
  if (tid == 0) logdet_s = neg ? __builtin_nanf("") : logdet;
  __syncthreads();
  // M[:, n:] = S^-1.  One sample per thread, its residual in registers (NT = 32 or 64 >= n, loops unrolled over NT)
  const int b = blockIdx.x * 256 + tid;
  const bool live = b < batch;
  const float* xb = x + (long long)(live ? b : 0) * ldx;
  const float* mp = mean + (long long)pair * n;
  float r[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) r[j] = j < n ? xb[j] - mp[j] : 0.f;
  float quad = 0.f;
  for (int i = 0; i < n; ++i) {  // y1[i] = sum_j Sinv[i][j] r[j];  y2[i] = sum_j Sinv[j][i] r[j]  (LDS broadcasts)
    float y1 = 0.f, y2 = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      if (j < n) {
        y1 += M[i][n + j] * r[j];
        y2 += M[j][n + i] * r[j];
      }
    }
    quad += (xb[i] - mp[i]) * y1;
    if (grad && live) grad[((long long)pair * batch + b) * n + i] = -0.5f * (y1 + y2);
  }
  if (live) ll[(long long)pair * batch + b] = -0.5f * (logdet_s + quad);
}

extern "C" int svae_gauss_ll(const float* x, int ldx, const float* mean, const float* S, float* ll, float* grad, int batch, int n,
                             int pairs, void* stream) {
  SVAE_REQUIRE(x && mean && S && ll && batch >= 1 && pairs >= 1 && n >= 1 && n <= SOLVE_MAXN && ldx >= n, SVAE_ERR_ARG,
               "gauss_ll: n <= %d", SOLVE_MAXN);
  const dim3 grid((batch + 255) / 256, pairs);
  if (n <= 32) hipLaunchKernelGGL(gauss_ll_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, mean, S, ll, grad, batch, n);
  else hipLaunchKernelGGL(gauss_ll_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, mean, S, ll, grad, batch, n);
  return check_launch("gauss_ll");
}

// ------------------------------------------------------------------------------------------------------------------
// Kernel-density mutual information between the latent means and the conditioning variables (reference: MutInfoEstimator.forward,
// disentangle.py:278-317; loss key `mcmi`, losses.py:221-225).  Per sample b and mixture centre s:
//   a_s = -0.5 (logA_x[s] + logA_y + |x_b - x_s|^2 / var_s + |y_b - y_s|^2 / gamma),  b_s = -0.5 (logA_x[s] + |x_b - x_s|^2 / var_s),
//   c_s = -0.5 (logA_y + |y_b - y_s|^2 / gamma);   val_b = lse_s a_s - lse_s b_s - lse_s c_s   (the reference's un-normalised sums)
// and the gradient the encoder is seeded with, d val_b / d x_b = sum_s (softmax(a)_s - softmax(b)_s) (x_s - x_b) / var_s.
// The reference materialises [B, S, z] difference tensors; here a workgroup owns 16 samples, 16 lanes per sample share the centres
// of a 64-centre LDS tile, pass 1 keeps running (max, sum) pairs of the three log-sum-exps, pass 2 re-evaluates the exponents against
// the final values and accumulates the gradient in registers.  var: one value ("sphere") or [S][zx] ("diagonal").
constexpr int KDE_TS = 64;
template <int ZT>
__global__ __launch_bounds__(256) void kde_mi_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y, int ldy,
                                                      const float* __restrict__ xs, const float* __restrict__ ys,
                                                      const float* __restrict__ var, int var_per_centre, const float* __restrict__ logAx,
                                                      float logAy, float gamma, float* __restrict__ val, float* __restrict__ grad,
                                                      int batch, int S, int zx, int dy) {
  __shared__ float xs_t[KDE_TS][ZT + 1], iv_t[KDE_TS][ZT + 1], ys_t[KDE_TS][SOLVE_MAXN + 1], la_t[KDE_TS];
  __shared__ float ysam[16][SOLVE_MAXN + 1];
  const int tid = threadIdx.x, sb = tid >> 4, l = tid & 15;
  const int b = blockIdx.x * 16 + sb;
  const bool live = b < batch;
  const float* xb = x + (long long)(live ? b : 0) * ldx;
  float xr[ZT];
#pragma unroll
  for (int d = 0; d < ZT; ++d) xr[d] = d < zx ? xb[d] : 0.f;
  for (int e = tid; e < 16 * dy; e += 256) {
    const int i = e / dy, j = e - i * dy;
    const int bb = blockIdx.x * 16 + i;
    ysam[i][j] = bb < batch ? y[(long long)bb * ldy + j] : 0.f;
  }
  const float ig = 1.f / gamma;
  auto load_tile = [&](int s0) {
    __syncthreads();  // the previous tile is consumed
    for (int e = tid; e < KDE_TS * zx; e += 256) {
      const int i = e / zx, j = e - i * zx;
      const int s = s0 + i;
      xs_t[i][j] = s < S ? xs[(long long)s * zx + j] : 0.f;
      iv_t[i][j] = 1.f / (var_per_centre ? (s < S ? var[(long long)s * zx + j] : 1.f) : var[0]);
    }
    for (int e = tid; e < KDE_TS * dy; e += 256) {
      const int i = e / dy, j = e - i * dy;
      ys_t[i][j] = s0 + i < S ? ys[(long long)(s0 + i) * dy + j] : 0.f;
    }
    if (tid < KDE_TS) la_t[tid] = var_per_centre ? (s0 + tid < S ? logAx[s0 + tid] : 0.f) : logAx[0];
    __syncthreads();
  };
  auto exponents = [&](int i, float& sdx, float& sdy) {
    sdx = 0.f;
#pragma unroll
    for (int d = 0; d < ZT; ++d)
      if (d < zx) { const float df = xr[d] - xs_t[i][d]; sdx += df * iv_t[i][d] * df; }
    sdy = 0.f;
    for (int e = 0; e < dy; ++e) { const float df = ysam[sb][e] - ys_t[i][e]; sdy += df * ig * df; }
  };
  // pass 1: running (max, sum) of the three log-sum-exps over this lane's centres
  float ma = -INFINITY, mb = -INFINITY, mc = -INFINITY, sa = 0.f, sbb = 0.f, sc = 0.f;
  auto push = [](float v, float& m, float& sum) {
    if (v > m) { sum = sum * expf(m - v) + 1.f; m = v; }
    else sum += expf(v - m);
  };
  for (int s0 = 0; s0 < S; s0 += KDE_TS) {
    load_tile(s0);
#pragma unroll
    for (int q = 0; q < KDE_TS / 16; ++q) {
      const int i = l + 16 * q;
      if (s0 + i < S) {
        float sdx, sdy;
        exponents(i, sdx, sdy);
        push(-0.5f * (la_t[i] + logAy + sdx + sdy), ma, sa);
        push(-0.5f * (la_t[i] + sdx), mb, sbb);
        push(-0.5f * (logAy + sdy), mc, sc);
      }
    }
  }
  auto combine = [](float m, float sum) {  // over the 16 lanes of a sample
    float mt = m;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) mt = fmaxf(mt, __shfl_xor(mt, o, 64));
    float st = sum > 0.f ? sum * expf(m - mt) : 0.f;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) st += __shfl_xor(st, o, 64);
    return mt + logf(st);
  };
  const float pxy = combine(ma, sa), px = combine(mb, sbb), py = combine(mc, sc);
  if (live && l == 0) val[b] = pxy - px - py;
  if (grad == nullptr) return;  // uniform
  // pass 2: d val / d x = sum_s (exp(a_s - pxy) - exp(b_s - px)) (x_s - x) / var_s
  float g[ZT];
#pragma unroll
  for (int d = 0; d < ZT; ++d) g[d] = 0.f;
  for (int s0 = 0; s0 < S; s0 += KDE_TS) {
    load_tile(s0);
#pragma unroll
    for (int q = 0; q < KDE_TS / 16; ++q) {
      const int i = l + 16 * q;
      if (s0 + i < S) {
        float sdx, sdy;
        exponents(i, sdx, sdy);
        const float cw = expf(-0.5f * (la_t[i] + logAy + sdx + sdy) - pxy) - expf(-0.5f * (la_t[i] + sdx) - px);
#pragma unroll
        for (int d = 0; d < ZT; ++d)
          if (d < zx) g[d] += cw * (xs_t[i][d] - xr[d]) * iv_t[i][d];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < ZT; ++d) {
    if (d < zx) {
      float t = g[d];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o, 64);
      if (live && l == 0) grad[(long long)b * zx + d] = t;
    }
  }
}

extern "C" int svae_kde_mi(const float* x, int ldx, const float* y, int ldy, const float* xs, const float* ys, const float* var,
                           int var_per_centre, const float* logAx, float logAy, float gamma, float* val, float* grad, int batch,
                           int centres, int zx, int dy, void* stream) {
  SVAE_REQUIRE(x && y && xs && ys && var && logAx && val && batch >= 1 && centres >= 1 && zx >= 1 && zx <= SOLVE_MAXN && dy >= 1 &&
                   dy <= SOLVE_MAXN && ldx >= zx && ldy >= dy && gamma > 0.f,
               SVAE_ERR_ARG, "kde_mi: 1 <= zx, dy <= %d", SOLVE_MAXN);
  const dim3 grid((batch + 15) / 16);
  if (zx <= 32)
    hipLaunchKernelGGL(kde_mi_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, xs, ys, var, var_per_centre, logAx,
                       logAy, gamma, val, grad, batch, centres, zx, dy);
  else
    hipLaunchKernelGGL(kde_mi_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, xs, ys, var, var_per_centre, logAx,
                       logAy, gamma, val, grad, batch, centres, zx, dy);
  return check_launch("kde_mi");
}
