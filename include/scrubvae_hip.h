/*
 * scrubvae_hip.h -- C ABI of libscrubvae_hip.so (gfx950 / MI355X).
 *
 * The reference (tdunnlab/scrubvae) is pure Python and has NO FFI/plugin boundary
 * (SURVEY.md 8b); this ABI is new and sits underneath the reference's Python API.  Each
 * entry point states which reference op (file:line under /root/reference) it replaces.
 * The Python binding is scrubvae_amd/_lib.py (ctypes); INTEGRATION.md shows the stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *  - plain C, no C++/torch types; every pointer is a DEVICE pointer unless noted;
 *  - the caller owns every buffer (inputs, outputs, workspaces); the callee never
 *    allocates, frees, synchronises or keeps a pointer past return;
 *  - all launches go to the hipStream_t passed as `void* stream`;
 *  - return value: 0 = ok, negative = svae_status error; svae_last_error() gives text;
 *  - activations are channels-last ("NLC"): row r = b*L + l, `ld` floats per row, the
 *    first C entries valid, entries C..Cp-1 (Cp = channels padded to a multiple of 16)
 *    are zero.  This makes the reference's [B,W,C] <-> [B,C,W] moveaxis copies
 *    (residual.py:450,479) free;
 *  - conv / linear weights are "TIO": w[tap][c_in_pad][c_out_pad], pads zero.
 *    Conv1d weight [Cout,Cin,k]      -> w[t][ci][co] = W[co][ci][t]
 *    ConvTranspose1d weight [Cin,Cout,k] -> w[t][ci][co] = W[ci][co][t]
 *    Linear weight [out,in]          -> w[0][in][out].
 */
#ifndef SCRUBVAE_HIP_H
#define SCRUBVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  SVAE_OK = 0,
  SVAE_ERR_SHAPE = -1,     /* inconsistent / unsupported shape */
  SVAE_ERR_ALIGN = -2,     /* pointer or leading dimension not 16-byte aligned */
  SVAE_ERR_WORKSPACE = -3, /* workspace too small */
  SVAE_ERR_LAUNCH = -4,    /* hipLaunch failed (see svae_last_error) */
  SVAE_ERR_ARG = -5        /* null pointer / bad enum */
} svae_status;

#define SVAE_MAX_TAPS 128
#define SVAE_MAX_JOINTS 32
#define SVAE_MAX_CHAINS 8
#define SVAE_MAX_CHAIN_LEN 8

int svae_version(void);
/* copies the calling thread's last error text into buf (NUL-terminated) */
void svae_last_error(char* buf, size_t n);

/* ------------------------------------------------------------------ conv / linear --- */
typedef struct {
  int batch;
  int l_in, l_out;   /* sequence lengths; l_out must match the PyTorch formula */
  int c_in, c_out;   /* padded channel counts, multiples of 16 */
  int ld_in, ld_out; /* floats per activation row (>= c_in / c_out, multiple of 4) */
  int kernel, stride, padding, dilation;
  int transposed;    /* 0 = nn.Conv1d, 1 = nn.ConvTranspose1d; Linear: kernel=1,l=1 */
  /* tuning overrides (results are bit-identical for every tile: the per-output summation
   * order over K does not depend on it).  tile[kind] = BM*1000+BN with BM,BN in {64,128}, or
   * 0 for the built-in heuristic; kind 0 fwd, 1 dgrad, 2 wgrad.  Adding 1000000 (fwd/dgrad) selects
   * the LDS-DMA staging variant (global_load_lds, XOR-swizzled LDS image).  tile[2] = 1 selects the tap-fused
   * small-weight gradient kernel (64x64 tiles, all taps of a tile in one workgroup). */
  int tile[3];
  /* 1: the conv's input is nn.Upsample(scale_factor=2, mode="linear", align_corners=False) of a [batch, l_in / 2, c_in] tensor
   * (the decoder's skip path, residual.py:160): l_in stays the UPSAMPLED length, x points to the half-length tensor, and the
   * forward (svae_conv_fwd_split*, halo tile codes 8 / 9) and weight-gradient (svae_conv_wgrad_split, all-taps tile codes) kernels
   * blend its rows while they stage the operand -- the upsampled tensor need not exist.  nn.Conv1d with stride 1 only; the fp32
   * entry points and the other tile codes return SVAE_ERR_SHAPE.  The data-gradient entry points ignore the flag: dx is the
   * gradient with respect to the upsampled input (fold it back with svae_upsample2_bwd).  0 elsewhere. */
  int up2;
} svae_conv_desc;

/* y[b,lo,:] (+)= bias + sum_t x[b,li(lo,t),:] @ w[t]     (residual.py:79-109,137-170,
 * 198,219-222,264,286: every nn.Conv1d / nn.ConvTranspose1d / nn.Linear of the trunk).
 * fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM, 128xBN tiles. */
int svae_conv_fwd(const svae_conv_desc* d, const float* x, const float* w, const float* bias,
                  float* y, int accumulate, void* stream);
/* dx (+)= conv_backward_input(dy, w)      (autograd of the above) */
int svae_conv_dgrad(const svae_conv_desc* d, const float* dy, const float* w, float* dx,
                    int accumulate, void* stream);
/* The same two calls with an optional split-K workspace (the Linear heads: few output tiles, long reduction): partial
 * tiles go to `ws` (svae_conv_splitk_workspace(d, kind) bytes, kind 0 = fwd / 1 = dgrad; 0 = no split for this
 * geometry) and are summed in a fixed order.  ws == NULL or too small: identical to the calls above. */
size_t svae_conv_splitk_workspace(const svae_conv_desc* d, int kind);
int svae_conv_fwd_ws(const svae_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                     int accumulate, void* ws, size_t ws_bytes, void* stream);
int svae_conv_dgrad_ws(const svae_conv_desc* d, const float* dy, const float* w, float* dx, int accumulate,
                       void* ws, size_t ws_bytes, void* stream);
/* dw (+)= conv_backward_weight(x, dy); split-K partial slabs in `ws`, reduced in a fixed
 * order (bit-reproducible).  db (optional, may be NULL) (+)= column sums of dy. */
size_t svae_conv_wgrad_workspace(const svae_conv_desc* d);
int svae_conv_wgrad(const svae_conv_desc* d, const float* x, const float* dy, float* dw,
                    float* db, void* ws, size_t ws_bytes, int accumulate, void* stream);

/* ---- split-bf16 variants of the three contractions (scrubvae_amd/csrc/gemm_bf16s.hip) -----
 * Same geometry, layouts and results contract as svae_conv_fwd / _dgrad / _wgrad, but the
 * products run on the bf16 matrix cores: each fp32 operand is split into `pieces` (1..3) bf16
 * pieces and the cross products with i + j < pieces are accumulated in fp32.  pieces = 3 is
 * fp32-accurate (dropped terms O(2^-24)); 2 -> O(2^-16); 1 = plain bf16 operands.
 * `wsplit` (svae_conv_split_bytes(d) bytes, caller-owned) holds the weight pieces written by
 * svae_conv_split_weights from the fp32 master weights w[tap][c_in][c_out]; refresh it whenever
 * the weights change.  d->tile[0..1]: V*1000000 + BM*1000 + BN, V = kernel variant (waves per
 * workgroup / LDS buffering, see gemm_bf16s.hip; 16 / 17 / 18 (18: on the 16x16x32 MFMA): the 12-wave halo kernel with DMA-only loader waves, 256-row
 * tiles; 19 / 29: the 8-wave halo kernel on 256 x 160 / 256 x 256 tiles).  d->tile[2] for the split weight gradient: V*1000000 + BM*1000 + BN with V a bit set -- 1: single LDS buffer,
 * 2: XCD-aware workgroup order, 4: all taps of a tile in one workgroup (contiguous 5 / 6-tap geometries, 2 pieces;
 * SVAE_ERR_SHAPE otherwise), 8: that kernel on the 16x16x32 MFMA shape, 16: the taps folded into the dY columns (transposed
 * convs) or into the X channel rows (convs) of the tile -- one tile padding for all taps.  Results do not depend on the tile code beyond
 * the summation order (deterministic for a given code). */
size_t svae_conv_split_bytes(const svae_conv_desc* d);
int svae_conv_split_weights(const svae_conv_desc* d, const float* w, void* wsplit, void* stream);
/* the same for up to SVAE_MAX_SPLIT_TASKS convolutions in one launch (all layers of a model, once per step) */
#define SVAE_MAX_SPLIT_TASKS 48
typedef struct {
  const float* w;   /* fp32 master weights [kernel][c_in][c_out] (padded channel counts) */
  void* wsplit;     /* svae_conv_split_bytes() bytes */
  int kernel, c_in, c_out;
} svae_split_task;
int svae_conv_split_weights_batched(const svae_split_task* tasks, int n, void* stream);
int svae_conv_fwd_split(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                        float* y, int accumulate, int pieces, void* stream);
/* pieces = SVAE_PIECES_F16X2 (forward only): two fp16 pieces per operand (22 of the 24 significand bits), three cross products --
 * fp32-class accuracy (~2^-22 per product) at half the matrix-core work of pieces = 3; the weights' fp16 planes (of 2^10 w, undone in
 * the epilogue) are part of the svae_conv_split_weights output.  Operands beyond fp16's range (|x| > 65504) saturate. */
#define SVAE_PIECES_F16X2 22
/* The same launch with the train-mode BatchNorm statistics of the conv output fused into its epilogue (nn.BatchNorm1d right
 * behind the conv, residual.py:88,112,146,173): bn_part[tile][2][c_out] = per-column (sum y, sum y^2) over the valid rows of
 * row tile `tile`, y = the value written (bias and, with accumulate, the previous content included) -- the layout of
 * svae_bn_stats_partial with svae_conv_fwd_stats_tiles(d) in place of svae_bn_chunks(rows); feed it to
 * svae_bn_stats_finalize / svae_bn_reduce_partials.  Removes one full read of the conv output per BatchNorm. */
int svae_conv_fwd_stats_tiles(const svae_conv_desc* d);
int svae_conv_fwd_split_stats(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                              float* y, int accumulate, int pieces, float* bn_part, void* stream);
/* The same launch for a descriptor with up2 = 1 (conv behind nn.Upsample(x2, linear): residual.py:153-170) that additionally
 * writes the upsampled operand rows it blends to up_out[batch * l_in][c_in] (each row once; ld_in == c_in) -- the tensor the
 * conv's weight gradient reads in the backward pass -- so that no separate svae_upsample2_fwd launch (one more read of x) is
 * needed.  up_out = NULL: nothing is written; bn_part as above (NULL: off). */
int svae_conv_fwd_split_up2(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias,
                            float* y, int accumulate, int pieces, float* bn_part, float* up_out, void* stream);
int svae_conv_dgrad_split(const svae_conv_desc* d, const float* dy, const void* wsplit, float* dx,
                          int accumulate, int pieces, void* stream);
/* The same launch when dx is the gradient with respect to the OUTPUT of a BatchNorm1d + PReLU / Tanh stage (the `add` / `residual.1-2`
 * pairs of residual.py:88-89,112-113,146-147,173-174) whose saved input is f->x ([rows][ld_in], the layout of dx): the first pass of that
 * stage's backward -- (sum du, sum du * xhat) per channel and the PReLU slope's partial, what svae_affine_prelu_bwd_partial computes
 * from a re-read of dx and x -- comes out of the epilogue: part[tile][2][c_in], dalpha_part[2 * (tile * col_blocks + col_block)] =
 * the (hi, lo) float pair of the tile's fp64 slope partial (2 * tiles * col_blocks floats), with
 * svae_conv_dgrad_stats_tiles(d, &col_blocks) row tiles.  scale / shift (and mean / rstd) NULL: bare activation; alpha NULL: tanh.
 * With accumulate the sums are those of the accumulated value (the launch that writes dx last carries them). */
typedef struct {
  const float* x;
  const float* scale; const float* shift; const float* mean; const float* rstd;
  const float* alpha;
  float* part;
  float* dalpha_part;   /* may be NULL (tanh) */
} svae_bn_bwd_fuse;
int svae_conv_dgrad_stats_tiles(const svae_conv_desc* d, int* col_blocks);
int svae_conv_dgrad_split_bn(const svae_conv_desc* d, const float* dy, const void* wsplit, float* dx,
                             int accumulate, int pieces, const svae_bn_bwd_fuse* f, void* stream);
int svae_conv_wgrad_split(const svae_conv_desc* d, const float* x, const float* dy, float* dw,
                          float* db, void* ws, size_t ws_bytes, int accumulate, int pieces, void* stream);
/* introspection: tile, kernel variant and (halo variant) image rows of the split fwd (0) / dgrad (1) launch */
int svae_conv_split_tile(const svae_conv_desc* d, int kind, int* bm, int* bn, int* variant, int* rmax);

/* introspection: the (BM, BN) workgroup tile the dispatcher uses; kind 0 fwd, 1 dgrad, 2 wgrad */
int svae_conv_tile(const svae_conv_desc* d, int kind, int* bm, int* bn);

/* ------------------------------------------------------------- elementwise / norm --- */
/* E0: ResVAE.normalize_root + input pack (residual.py:428-431,438-451).
 * x6d [rows, 6J], root [rows,3], arena = HOST pointer to 6 floats [2,3] (may be NULL:
 * no root channels)
 * -> x_in [rows, ld] = [x6d | 2(root-a0)/(a1-a0)-1 | 0 pad]. */
int svae_pack_input(const float* x6d, const float* root, const float* arena, float* x_in,
                    long long rows, int n_joints, int ld, void* stream);

/* Train-mode BatchNorm1d statistics (residual.py:88,112,146,173): per-channel partial
 * sums over row chunks.  part [n_chunks][2][C] (sum, sum of squares); n_chunks returned by
 * svae_bn_chunks(rows). */
int svae_bn_chunks(long long rows);
int svae_bn_stats_partial(const float* x, long long rows, int C, int ld, float* part, void* stream);
/* Finalize: sums over chunks in fixed order (fp64), `count` = rows over ALL ranks (the
 * caller all-reduces `part` reduced to [2][C] for sync-BN, see svae_bn_reduce_partials).
 * Writes scale = gamma*rstd, shift = beta-mean*scale, saves mean/rstd, updates running
 * stats with momentum (unbiased var) when running_mean != NULL. */
int svae_bn_reduce_partials(const float* part, int n_chunks, int C, float* sums /*[2][C]*/, void* stream);
int svae_bn_finalize(const float* sums, double count, int C, const float* gamma, const float* beta,
                     float eps, float momentum, float* running_mean, float* running_var,
                     float* mean, float* rstd, float* scale, float* shift, void* stream);
/* single-rank fast path: svae_bn_reduce_partials + svae_bn_finalize in one launch; also
 * increments *num_batches_tracked (int64, may be NULL) */
int svae_bn_stats_finalize(const float* part, int n_chunks, double count, int C, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                           float* scale, float* shift, void* stream);
/* backward: chunk partials -> sums[2][C], plus (+)= dgamma, dbeta (from the sums) and dalpha
 * (from dalpha_part[n_parts]); any of the three may be NULL */
int svae_bn_bwd_reduce(const float* part, int n_chunks, int C, float* sums, float* dgamma, float* dbeta,
                       float* dalpha, const float* dalpha_part, int n_parts, int accumulate, void* stream);
/* eval mode: scale/shift from running stats */
int svae_bn_eval_coeffs(int C, const float* gamma, const float* beta, float eps,
                        const float* running_mean, const float* running_var,
                        float* scale, float* shift, void* stream);
/* y = PReLU(x*scale+shift) with scalar slope *alpha (nn.PReLU(), residual.py:89,113,147,
 * 174,199).  scale/shift NULL => identity affine (conv_in's bare PReLU).  alpha NULL in this and the
 * two backward calls => tanh instead of PReLU (model.activation == "tanh", nn.Tanh() at the same
 * lines): act'(u) = 1 - tanh(u)^2, no slope gradient (dalpha_part is written as zeros). */
int svae_affine_prelu_fwd(const float* x, const float* scale, const float* shift, const float* alpha,
                          float* y, long long rows, int C, int ld, void* stream);
/* backward, pass 1: partial column sums  part[n_chunks][2][C] = (sum du, sum du*xhat),
 * dalpha_part[2 * (chunk * ceil(C/64) + column block)] = (hi, lo) float pair of sum dy*u*[u<=0] formed in fp64 (the slope's gradient
 * sums ~1e6 cancelling terms; 2 * n_chunks * ceil(C/64) floats; the reduction calls below sum all n_parts floats in fp64);
 * du = dy*(u>0?1:alpha), u = x*scale+shift. */
int svae_affine_prelu_bwd_partial(const float* dy, const float* x, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, const float* alpha,
                                  long long rows, int C, int ld, float* part, float* dalpha_part,
                                  void* stream);
/* pass 2: dx = gamma*rstd*(du - s0/count - xhat*s1/count)  (train-mode BN backward);
 * with sums == NULL: dx = du*scale (eval BN / bare PReLU: scale may be NULL => 1).
 * Also (+)= dgamma, dbeta from sums and dalpha from the n_parts entries of dalpha_part when those
 * pointers are given. */
int svae_affine_prelu_bwd_apply(const float* dy, const float* x, const float* scale, const float* shift,
                                const float* mean, const float* rstd, const float* gamma,
                                const float* alpha, const float* sums, double count,
                                float* dx, long long rows, int C, int ld,
                                float* dgamma, float* dbeta, float* dalpha,
                                const float* dalpha_part, int n_parts, int accumulate_param_grads,
                                void* stream);   /* n_parts = entries of dalpha_part */
/* the same pass, also leaving the column sums of dx -- the bias gradient of the conv(s) in front of the stage (autograd of the
 * reference's `bias=True` convs) -- as per-workgroup partials colsum_part[svae_affine_prelu_colsum_rows(rows, C)][C] for
 * svae_colsum_from_partials (needs C / 4 a power of two <= 256; colsum_part may be NULL) */
int svae_affine_prelu_colsum_rows(long long rows, int C);
int svae_affine_prelu_bwd_apply_colsum(const float* dy, const float* x, const float* scale, const float* shift,
                                       const float* mean, const float* rstd, const float* gamma,
                                       const float* alpha, const float* sums, double count,
                                       float* dx, long long rows, int C, int ld,
                                       float* dgamma, float* dbeta, float* dalpha,
                                       const float* dalpha_part, int n_parts, int accumulate_param_grads,
                                       float* colsum_part, void* stream);

/* nn.Upsample(scale_factor=2, mode="linear", align_corners=False) (residual.py:160) */
int svae_upsample2_fwd(const float* x, float* y, int batch, int l_in, int C, int ld, void* stream);
int svae_upsample2_bwd(const float* dy, float* dx, int batch, int l_in, int C, int ld, int accumulate,
                       void* stream);

/* ------------------------------------------------------------------- latent heads --- */
/* E4+S1+L3 (diag): mu/sigma/dmu/dsigma are [B, ldm] (ldm >= z); eps is [B, z]; h [B, ld]: mu at columns [0,z), raw at [raw_off, raw_off+z); sigma = softplus(raw) (residual.py:60-68),
 * z = mu + sigma*eps (residual.py:305-316; eps NULL => z = mu, eval mode),
 * kl_part[blocks] = per-block partial of -0.5*sum(1+2log(sigma)-mu^2-sigma^2)
 * (losses.py:138-146, before the /B). */
int svae_heads_diag_fwd(const float* h, int ld, const float* eps, float* mu, float* sigma, float* z,
                        int ldz, float* kl_part, int batch, int zdim, int raw_off, int ldm, void* stream);
int svae_heads_blocks(int batch, int zdim);
/* dh = [dmu_total | draw]: dmu_total = dmu + dz + kl_scale*mu,
 * draw = (dz*eps + dsigma + kl_scale*(sigma-1/sigma)) * sigmoid(raw). */
int svae_heads_diag_bwd(const float* h, int ld, const float* eps, const float* sigma,
                        const float* dz, int lddz, const float* dmu, const float* dsigma, float kl_scale,
                        float* dh, int batch, int zdim, int raw_off, int ldm, void* stream);

/* Full-Cholesky variant (model.diag = False; CholeskyL residual.py:39-68): raw holds the
 * z(z+1)/2 lower-triangle entries in torch.tril_indices order at h[:, raw_off:], softplus on
 * the diagonal; L [B,z,z] is written densely (zeros above the diagonal); z = L eps + mu;
 * kl_part as above with diag(L L^T) = row sums of squares. */
int svae_heads_tril_fwd(const float* h, int ld, const float* eps, float* mu, int ldm, float* L, float* z,
                        int ldz, float* kl_part, int batch, int zdim, int raw_off, void* stream);
/* dlv (optional, [B,z]): upstream gradient w.r.t. log diag(L L^T) (total-correlation loss) */
int svae_heads_tril_bwd(const float* h, int ld, const float* eps, const float* L, const float* dz, int lddz,
                        const float* dmu, int ldm, float kl_scale, const float* dlv, float* dh, int batch,
                        int zdim, int raw_off, void* stream);

/* model.prior = "beta" (residual.py:223-239,301-302,328-331,453-456; losses.py:198-206).  h [B, ld]: raw alpha at columns [0, z), raw beta
 * at [raw_off, raw_off + z).  Forward: alpha = softplus(raw) + 1, beta likewise, mu = (alpha - 1 + 1e-8) / (alpha + beta - 2 + 2e-8) * 2 - 1
 * (all [B, ldm]); kl_part[svae_heads_blocks] = partials of sum KL(Beta(alpha, beta) || Beta(1, 1)) (before the / B).  The draw
 * x ~ Beta(alpha, beta) is the CALLER's (z = 2 x - 1; torch's sampler as RNG plumbing, injected in parity tests).  Backward:
 * dh = [d raw_alpha | d raw_beta] from dz [B, lddz] (gradient wrt z, through the implicit reparameterisation of x: the reference's
 * Beta.rsample = Dirichlet rsample with torch._dirichlet_grad, restated in the kernel), dmu [B, ldm] (seed on mu; may be NULL) and
 * kl_scale * d KL. */
int svae_heads_beta_fwd(const float* h, int ld, float* alpha, float* beta, float* mu, int ldm, float* kl_part, int batch, int zdim,
                        int raw_off, void* stream);
int svae_heads_beta_bwd(const float* h, int ld, const float* x, const float* alpha, const float* beta, int ldm, const float* dz,
                        int lddz, const float* dmu, float kl_scale, float* dh, int batch, int zdim, int raw_off, void* stream);

/* L5: total_correlation (losses.py:41-101), beta-TCVAE minibatch estimator, z detached.
 * svae_tc_logvar: lv[b,l] = log diag(L L^T) from sigma (diag; 2 log sigma) or a dense L.
 * svae_tc_fwd: loss[j] (TC = mean_j), plus the two log-sum-exp tables the backward reuses
 * (lse_l [B,z], lse_a [B]).  svae_tc_bwd: d_mu[i,:] += weight * dTCsum/dmu, d_lv = weight *
 * dTCsum/dlv (weight = loss_scale / B).  O(B^2 z), nothing of size [B,B,z] is materialised. */
int svae_tc_logvar(const float* sigma, int lds, const float* L, float* lv, int batch, int zdim, void* stream);
int svae_tc_fwd(const float* z, int ldz, const float* mu, int ldm, const float* lv, int batch, int zdim,
                float* lse_l, float* lse_a, float* loss, void* stream);
int svae_tc_bwd(const float* z, int ldz, const float* mu, int ldm, const float* lv, int batch, int zdim,
                const float* lse_l, const float* lse_a, float weight, float* d_mu, int ldd, float* d_lv,
                int ldv, const float* sigma /* optional: emit d/dsigma = d/dlv*2/sigma */, int lds, void* stream);

/* ----------------------------------------------------------------- pose-loss tail --- */
typedef struct {
  int n_joints;
  int n_chains;
  int chain_len[SVAE_MAX_CHAINS];
  int chain[SVAE_MAX_CHAINS][SVAE_MAX_CHAIN_LEN];
} svae_tree;

/* D3 tail + K1 + K3 + L1 + L2 fused, one pass over the decoder output:
 *   y [rows, ld] conv_out pre-activation -> x_hat = tanh(y)            (residual.py:291)
 *   x6d_hat [rows, 6J], root_hat [rows,3] = inv_normalize_root(...)     (residual.py:479-489)
 *   pose_hat = fwd_kin_cont6d_torch(x6d_hat, tree, offsets, root=0, eps=1e-8)
 *                                                     (dataset.py:83-116, quaternion.py:337-353)
 *   jpe partial  = sum (target_pose - pose_hat)^2     (losses.py:148-171, before /(B*3*J))
 *   root partial = sum (root_hat - root)^2            (losses.py:216-219, before /B)
 *   dy [rows, ld] = jpe_scale * d jpe_sum/dy + root_scale * d root_sum/dy   (analytic
 *   reverse-mode through the kinematic chains, the 6D->matrix map and tanh).
 * `arena` is a HOST pointer to 6 floats (NULL: no root channels).
 * input_is_pre_tanh = 0: `y` already holds x_hat (no tanh, dy = d/dx_hat): the stand-alone
 * mpjpe_loss form.
 * pose_out (optional) [rows, J, 3] receives pose_hat (used to synthesise target_pose and by
 * the evaluation path).
 * loss_part [blocks][2].  dy may be NULL (eval: forward only).  ext_dx6d/ext_droot
 * (optional) are extra upstream grads w.r.t. x6d_hat/root_hat added before the tanh
 * backward (used by the rotation loss and by autograd callers). */
int svae_tail_blocks(long long rows);
int svae_pose_tail(const float* y, int ld, const float* offsets, const float* target_pose,
                   const float* root, const float* arena, const svae_tree* tree,
                   float jpe_scale, float root_scale, const float* ext_dx6d, const float* ext_droot,
                   float* x6d_hat, float* root_hat, float* loss_part, float* dy, float* pose_out,
                   long long rows, int input_is_pre_tanh, void* stream);

/* L4: stable_rotation_loss (losses.py:123-136, rotation_conversion.py:469-488):
 * part[blocks] partial sums of 2*asin(clamp(|R(x_hat)-R(x)|_F/2^1.5)); dx6d_hat (optional)
 * = scale * d/dx_hat. n = rows*J six-vectors. */
int svae_rot_loss(const float* x6d, const float* x6d_hat, float scale, float* part, float* dx6d_hat,
                  long long n, void* stream);
int svae_rot_blocks(long long n);

/* Bias gradients of a whole step in two launches: out_t[c] (+)= sum over rows of x_t[:, c]. */
#define SVAE_MAX_COLSUM_TASKS 48
typedef struct {
  const float* x; /* [rows, ld] */
  float* out;     /* [C] */
  long long rows;
  int C, ld;
} svae_colsum_task;
/* out[c] (+)= sum over rows of part[rows][C] for up to SVAE_MAX_COLSUM_TASKS partial arrays in one launch (fp64, fixed order) */
typedef struct {
  const float* part;
  float* out;
  int rows, C;
} svae_colsum_part_task;
int svae_colsum_from_partials(const svae_colsum_part_task* tasks, int n, int accumulate, void* stream);
size_t svae_colsum_batched_workspace(const svae_colsum_task* tasks, int n);
int svae_colsum_batched(const svae_colsum_task* tasks, int n, void* ws, size_t ws_bytes, int accumulate,
                        void* stream);

/* --------------------------------------------------------- preprocessing --- */
/* inv_kin (dataset.py:11-46, forward_indices=[1,0]) + root centring / "midfwd" re-orientation (dataset.py:385-404)
 * + quaternion_to_cont6d (quaternion.py:291-334) + get_segment_len (dataset.py:279-296) + heading of the window's
 * middle frame (dataset.py:234-241,258-265), one thread per frame.
 *   pose [frames][J][3] (frames = windows*window, windowed order), unit_offset = HOST pointer [J][3]
 *   -> x6d [frames][J][6]; offsets [frames][J][3], root [frames][3], heading [frames/window][2] (each may be NULL).
 * truncate_len != 0 reproduces the reference when OFFSET is an integer array (segment lengths truncated toward
 * zero on assignment, dataset.py:289-294).  Frame 0 gets the identity root quaternion (dataset.py:31). */
int svae_inv_kin(const float* pose, const float* unit_offset_host, const svae_tree* tree, int window, int midfwd,
                 int centre_root, int truncate_len, float* x6d, float* offsets, float* root, float* heading,
                 long long frames, void* stream);
/* get_speed_parts (dataset.py:133-163) + limbs averaged (:373-375): pose [windows][W][J][3] -> out [windows][3].
 * parts_host: the part joint lists concatenated (HOST), part_len_host[n_parts] their lengths (HOST). */
int svae_speed_parts(const float* pose, const int* parts_host, const int* part_len_host, int n_parts, int W, int J,
                     float* out, long long windows, void* stream);

/* ------------------------------------------------------------- MLP-ensemble scrubber heads --- */
/* G2/G3/A1: MLPEnsemble (disentangle.py:583-632) = up to four small MLPs (Linear/ReLU chains of <= 3 Linears) on the same input,
 * as ONE launch forward and one (+ a reduction launch) backward; GRScrubber (:635-660) feeds it mu, AdvNetScrubber (:663-684)
 * cat([mu;mu],[v;v_shuffle]).  The input is assembled in the kernel: columns [0,n0) = src0[b][0..n0), columns [n0,n0+n1) =
 * src1[b][0..n1); halves = 2 appends a second copy of the batch (rows batch..2*batch-1) whose column `shuf_col` of src1 is read
 * from row perm[b] (AdvNetScrubber.shuffle, :678-684; perm = int64 device array) or, when shuf_vals != NULL, is shuf_vals[b].  Weights are TIO Linear weights
 * [K][N] (K, N = features padded to multiples of 16, pads zero) -- the model's own parameter storage, nothing repacked. */
#define SVAE_ENS_MEMBERS 4
#define SVAE_ENS_MAX_LAYERS 3
typedef struct {
  const float* w;   /* [K][N] */
  const float* b;   /* [N] */
  float* dw;        /* gradient destinations; NULL (both) = frozen parameters: no weight gradients for this member */
  float* db;
  int K, N;
} svae_ens_layer;
typedef struct {
  svae_ens_layer layer[SVAE_ENS_MAX_LAYERS];
  int n_layers;
  float* out;          /* fwd: [rows][N_last] pre-activation outputs of the last Linear (rows = batch * halves) */
  const float* d_out;  /* bwd: gradient with respect to `out`, same shape */
} svae_ens_member;
typedef struct {
  svae_ens_member member[SVAE_ENS_MEMBERS];
  int n_members;
  const float* src0; int ld0, n0;
  const float* src1; int ld1, n1;
  const long long* perm; int shuf_col;
  int batch, halves;
  const float* shuf_vals;  /* optional [batch]: the shuffled column's values for the second copy, used instead of src1[perm[b]]
                            * (data parallel: the permutation runs over the GLOBAL batch, the values come from other ranks) */
} svae_ens_desc;
int svae_ens_fwd(const svae_ens_desc* d, void* stream);
/* Backward: recomputes the hidden activations, then writes parameter gradients (dw/db of every non-frozen member; summed over
 * row tiles in a fixed order, no atomics) and the input gradient: d_src0[b][k] += coef * sum over members and halves of
 * d/d input[b][k] for k < n0 (d_src0 may be NULL), and, when gx_raw != NULL, gx_raw[row][k] = sum over members for all rows and
 * all K_0 input columns.  coef = -alpha is the gradient reversal (disentangle.py:541-556). */
size_t svae_ens_bwd_workspace(const svae_ens_desc* d);
int svae_ens_bwd(const svae_ens_desc* d, float* d_src0, int ld_d, float coef, float* gx_raw, void* ws, size_t ws_bytes,
                 int accumulate_param_grads, void* stream);
/* Losses of all members in one launch (losses.py:267-309).  kind 0: sum((out - target)^2) over [rows][C]; 1: CrossEntropy(sum) vs
 * int32 labels; 2: the adversarial net's CrossEntropy applied to softmax(out) with class = (row >= rows/2) (double-softmax quirk,
 * disentangle.py:675 + losses.py:304-307; C = 2).  outs / dpred / loss_w / grad_s are HOST arrays of n_members entries:
 * part[m * nb + j] = loss_w[m] * (sum over the rows of block j), nb = svae_rowloss_blocks(rows); dpred[m] (may be NULL) =
 * grad_s[m] * d loss_m / d out_m. */
int svae_ens_loss(int kind, const float* const* outs, float* const* dpred, const float* loss_w, const float* grad_s, int n_members,
                  const float* target, int ld_t, const int* labels, int rows, int C, int ld, float* part, void* stream);

/* ------------------------------------------------------------------------- optimizer --- */
/* O1: torch.optim.AdamW / Adam step over one flat fp32 buffer (trainer.py:60-65,165).
 * step_t = 1-based step count; decoupled != 0 => AdamW. grad_scale multiplies g first
 * (1/world_size after a sum all-reduce, or the clip coefficient). */
int svae_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1,
                   float beta2, float eps, float weight_decay, int step_t, int decoupled,
                   float grad_scale, void* stream);
/* hipGraph-capturable form: hyper (device, 4 floats) = {lr, lr/(1-beta1^t), 1/sqrt(1-beta2^t), t}.  The caller writes
 * lr (stream-ordered) when the schedule changes it; svae_adam_advance, captured in front of the step, does t += 1 and
 * derives the two bias-correction terms ON THE DEVICE, so queued replays never race a host buffer. */
int svae_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const float* hyper,
                       float beta1, float beta2, float eps, float weight_decay, int decoupled,
                       float grad_scale, void* stream);
int svae_adam_advance(float* hyper, float beta1, float beta2, void* stream);
/* torch.nn.utils.clip_grad_norm_ (trainer.py:164): g *= min(1, max_norm / (sqrt(*sumsq) + 1e-6)); sumsq = device scalar
 * holding the sum of squares of ALL gradients.  Returns without touching g when the clip does not bite (a NaN norm poisons every
 * gradient, as torch's clamp(NaN, max=1) * g does).  norm_out (may be NULL): the total norm sqrt(*sumsq), the function's return value
 * in torch. */
int svae_clip_grads(float* g, long long n, const float* sumsq, float max_norm, float* norm_out, void* stream);

/* Batched small dense solves for the streaming scrubbers (reference: MovingAvgLeastSquares.forward,
 * src/scrubvae/model/disentangle.py:466-486 -- torch.linalg.solve(Sxx + l2 I, Sxy) twice per step; direct_lsq_loss,
 * src/scrubvae/train/losses.py:173-179).  System s: (A[s] + diag(diag)) X[s] = B[s], A row-major [n][n], B / X row-major
 * [n][nrhs], strides in elements between systems; diag may be NULL; n, nrhs <= 64.  LU with partial pivoting, fp32. */
int svae_small_solve(const float* A, long long strideA, const float* diag, const float* B, long long strideB, float* X,
                     long long strideX, int n, int nrhs, int batch, void* stream);
/* Gaussian log-likelihoods of the streaming quadratic discriminants (reference: QuadraticDiscriminantFilter.cgll,
 * src/scrubvae/model/disentangle.py:129-134 -- torch.linalg.solve + torch.logdet per (mean, covariance) pair, 4 pairs per class
 * and step, evaluate_loss :186-232).  x [batch][ldx] (first n columns), mean [pairs][n], S [pairs][n][n] row-major, n <= 64:
 *   ll[p][b] = -0.5 (logdet S_p + r^T S_p^-1 r),  r = x_b - mean_p   (NaN for det < 0, +inf for a singular S, as torch.logdet gives)
 *   grad[p][b][:] = d ll[p][b] / d x_b = -0.5 (S_p^-1 + S_p^-T) r     (grad may be NULL) */
int svae_gauss_ll(const float* x, int ldx, const float* mean, const float* S, float* ll, float* grad, int batch, int n, int pairs,
                  void* stream);
/* Kernel-density mutual information between latent means and conditioning variables (reference: MutInfoEstimator.forward,
 * src/scrubvae/model/disentangle.py:278-317; loss `mcmi`, src/scrubvae/train/losses.py:221-225).  x [batch][ldx] (zx columns),
 * y [batch][ldy] (dy columns), centres xs [centres][zx], ys [centres][dy]; var: one value (var_per_centre = 0, "sphere",
 * logAx[1]) or [centres][zx] ("diagonal", logAx[centres]); zx, dy <= 64.
 *   val[b] = lse_s a_s - lse_s b_s - lse_s c_s  (the reference's three un-normalised log-sum-exps; the loss is mean_b val[b])
 *   grad[b][:] = d val[b] / d x_b               (grad may be NULL) */
int svae_kde_mi(const float* x, int ldx, const float* y, int ldy, const float* xs, const float* ys, const float* var,
                int var_per_centre, const float* logAx, float logAy, float gamma, float* val, float* grad, int batch, int centres,
                int zx, int dy, void* stream);
/* sum of squares partials for clip_grad_norm_ (trainer.py:164): part[svae_sumsq_blocks(n)] */
int svae_sumsq_blocks(long long n);
int svae_sumsq_partial(const float* x, long long n, float* part, void* stream);
/* out[0..k) = sum over rows of part[rows][k] in fixed order (fp64 accumulate) * scale */
int svae_reduce_rows(const float* part, int rows, int k, float scale, float* out, int accumulate, void* stream);
/* the same with one scale per column (host array scales[k], k <= 8): the jpe and root terms -- summed by one tail launch,
 * normalised by 1 / (B 3 J) and 1 / B (losses.py:171,216-219) -- in one launch */
int svae_reduce_rows_scaled(const float* part, int rows, int k, const float* scales, float* out, void* stream);
/* L6 (losses.py:320-322): out[0] = sum_i weights[i] * terms[i] over the loss terms' device scalars, in index order (fp32, like the
 * reference's running total); weights = the loss_scale entries on the host, n <= SVAE_MAX_LOSS_TERMS; terms with weight 0 are
 * skipped as in the reference */
#define SVAE_MAX_LOSS_TERMS 48
int svae_loss_total(const float* terms, const float* weights, int n, float* out, void* stream);

/* ----------------------------------------------------------- small elementwise ops --- */
int svae_relu_fwd(const float* x, float* y, long long n, void* stream);
int svae_relu_bwd(const float* dy, const float* y, float* dx, long long n, void* stream);
int svae_axpy(float a, const float* x, float* y, long long n, void* stream); /* y += a*x */
int svae_fill(float* x, float v, long long n, void* stream);
/* G3: sum of squared error vs target, rows x C (pred ld, target ld_t); part[blocks];
 * dpred (optional) = scale*2*(pred-target) */
int svae_mse_sum(const float* pred, int ld, const float* target, int ld_t, int rows, int C, float scale,
                 float* part, float* dpred, void* stream);
int svae_rowloss_blocks(int rows);
/* CrossEntropyLoss(reduction="sum") on logits with integer labels (losses.py:271-273) */
int svae_ce_sum(const float* logits, int ld, const int* labels, int rows, int C, float scale,
                float* part, float* dlogits, void* stream);
/* A1: softmax then CrossEntropyLoss(sum) on the softmax OUTPUT against one-hot class
 * `cls(row) = row >= rows/2` (double softmax, disentangle.py:675 + losses.py:297-307) */
int svae_double_softmax_ce_sum(const float* logits, int ld, int rows, float scale, float* part,
                               float* dlogits, void* stream);

#ifdef __cplusplus
}
#endif
#endif
