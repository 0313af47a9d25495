// Fused decoder tail + pose losses for gfx950 (HBM-bound, one pass):
//   tanh(conv_out) -> x6d_hat/root_hat unpack (reference residual.py:291,479-489)
//   -> 6D->matrix (quaternion.py:337-353, eps=1e-8, columns [x y z])
//   -> forward kinematics along the chains (dataset.py:83-116: R restarts from R(joint 0)
//      on every chain, pose[c_i] = R*offset[c_i] + pose[c_{i-1}])
//   -> JPE / root squared-error partial sums (losses.py:148-171,216-219)
//   -> analytic reverse pass (subtree-summed position grads, chain product backward,
//      6D->matrix backward, tanh backward) producing d loss / d conv_out.
// One wave (64 lanes) per 64 consecutive frames; the [64][ld] tile of decoder outputs and the
// per-frame offsets/targets are staged through LDS with coalesced 256-byte wave accesses,
// rows padded to an odd stride so that lane-per-frame accesses are bank-conflict free;
// loss terms are reduced with wave shuffles.  The ~300 small torch ops the reference
// launches for this stage become one kernel.
#include "svae_internal.h"
#include <cstdlib>

namespace svae {

struct TailArgs {
  const float* y;
  const float* offsets;
  const float* target;
  const float* root;
  const float* ext_dx6d;
  const float* ext_droot;
  float* x6d_hat;
  float* root_hat;
  float* loss_part;
  float* dy;
  float* pose_out;
  long long rows;
  int ld, J, ldt, ldo;
  int has_arena;
  int pre_tanh;
  float a0[3], a1[3];
  float jpe_scale, root_scale;
  svae_tree tree;
};

struct M3 {
  float m[9];  // row-major
};

__device__ __forceinline__ M3 mul(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) c.m[r * 3 + k] = a.m[r * 3] * b.m[k] + a.m[r * 3 + 1] * b.m[3 + k] + a.m[r * 3 + 2] * b.m[6 + k];
  return c;
}
// a^T * b
__device__ __forceinline__ M3 mul_tn(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k) c.m[r * 3 + k] = a.m[r] * b.m[k] + a.m[3 + r] * b.m[3 + k] + a.m[6 + r] * b.m[6 + k];
  return c;
}
// a * b^T
__device__ __forceinline__ M3 mul_nt(const M3& a, const M3& b) {
  M3 c;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      c.m[r * 3 + k] = a.m[r * 3] * b.m[k * 3] + a.m[r * 3 + 1] * b.m[k * 3 + 1] + a.m[r * 3 + 2] * b.m[k * 3 + 2];
  return c;
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

constexpr float K1_EPS = 1e-8f;

// cont6d_to_matrix: columns [x y z]
__device__ __forceinline__ M3 c6_to_mat(const float* a) {
  const float nx = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const float ix = 1.f / (nx + K1_EPS);
  const float x[3] = {a[0] * ix, a[1] * ix, a[2] * ix};
  float zr[3];
  cross3(x, a + 3, zr);
  const float nz = sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2]);
  const float iz = 1.f / (nz + K1_EPS);
  const float z[3] = {zr[0] * iz, zr[1] * iz, zr[2] * iz};
  float yv[3];
  cross3(z, x, yv);
  M3 M;
#pragma unroll
  for (int r = 0; r < 3; ++r) { M.m[r * 3] = x[r]; M.m[r * 3 + 1] = yv[r]; M.m[r * 3 + 2] = z[r]; }
  return M;
}

// backward of c6_to_mat: given dM returns d a[6]
__device__ __forceinline__ void c6_to_mat_bwd(const float* a, const M3& dM, float* da) {
  const float nx = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const float ix = 1.f / (nx + K1_EPS);
  const float x[3] = {a[0] * ix, a[1] * ix, a[2] * ix};
  float zr[3];
  cross3(x, a + 3, zr);
  const float nz = sqrtf(zr[0] * zr[0] + zr[1] * zr[1] + zr[2] * zr[2]);
  const float iz = 1.f / (nz + K1_EPS);
  const float z[3] = {zr[0] * iz, zr[1] * iz, zr[2] * iz};
  float gx[3] = {dM.m[0], dM.m[3], dM.m[6]}, gy[3] = {dM.m[1], dM.m[4], dM.m[7]}, gz[3] = {dM.m[2], dM.m[5], dM.m[8]};
  float t[3];
  // y = z x x_
  cross3(x, gy, t);  // dz += x x gy
  gz[0] += t[0]; gz[1] += t[1]; gz[2] += t[2];
  cross3(gy, z, t);  // dx += gy x z
  gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
  // z = zr/(nz+eps)
  const float dzr_dot = zr[0] * gz[0] + zr[1] * gz[1] + zr[2] * gz[2];
  const float kz = nz > 0.f ? dzr_dot * iz * iz / nz : 0.f;
  const float gzr[3] = {gz[0] * iz - zr[0] * kz, gz[1] * iz - zr[1] * kz, gz[2] * iz - zr[2] * kz};
  // zr = x x yr
  cross3(a + 3, gzr, t);  // dx += yr x gzr
  gx[0] += t[0]; gx[1] += t[1]; gx[2] += t[2];
  cross3(gzr, x, da + 3);  // dyr = gzr x x
  // x = xr/(nx+eps)
  const float dx_dot = a[0] * gx[0] + a[1] * gx[1] + a[2] * gx[2];
  const float kx = nx > 0.f ? dx_dot * ix * ix / nx : 0.f;
  da[0] = gx[0] * ix - a[0] * kx;
  da[1] = gx[1] * ix - a[1] * kx;
  da[2] = gx[2] * ix - a[2] * kx;
}

// One workgroup per 64 consecutive frames, one WAVE PER KINEMATIC CHAIN (lane = frame): the chains only couple
// through (a) the position of the joint a chain starts from and (b) the gradient of the shared root rotation,
// so the expensive per-chain work (rotation products forward, chain-product backward) runs in parallel waves
// over the same LDS tile; the cheap coupling steps run on wave 0 between barriers.  A single wave per workgroup
// (the previous version) left the CU with one resident wave because the tile takes ~90 KB of LDS.
// TR = frames per workgroup (lane = frame, TR <= 64).  Measured (MI355X, B = 4096, the round-1 layout with 104 KB of LDS per 64-frame
// tile = ONE workgroup per CU): 64 frames 566 us, 48 frames 686 us, 32 frames 863 us -- fewer frames per workgroup leave lanes of the
// per-chain waves idle and that costs more than the extra co-residency buys: the kernel is bound by its per-wave dependent LDS / VALU
// chains (72 % of the wave cycles in s_waitcnt), not by HBM.  The tile below takes 75 KB (targets not staged, one root-rotation
// buffer); a second co-resident workgroup additionally needs <= 168 VGPRs (the kernel uses 194): capping them
// (__launch_bounds__(512, 3), 57 spilled registers) measured no faster (599 -> 636 us), so the cap is not set.  What the
// per-frame cost really is made of -- ~12 k dependent VALU operations per frame behind scalar tree look-ups and ds_read_b32
// operands, issued by 1.5 waves per SIMD -- is what a next round has to restructure (frames x joints parallelism inside a wave).
// Round 2 also built the opposite extreme and measured it SLOWER (730 us vs 604 us, B = 4096): a lane owns a frame end to end, a
// wave 64 frames with no barrier, every operand read per lane straight from global memory (8-byte pieces of the lane's own 576-byte
// row), only the joint positions in LDS (17.7 KB per wave, 8 waves per CU).  Correct (it passed test_pose_tail), but each such load
// instruction touches 64 different cache lines and the 8 waves' rows (295 KB) do not stay in the 32 KB L1: every 8-byte piece
// re-fetches its line from L2, ~16x the bytes.  The LDS tile IS the coalescing buffer; what it costs is occupancy.
template <int TR>
__global__ __launch_bounds__(64 * SVAE_MAX_CHAINS) void pose_tail_kernel(const TailArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nth = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int J = g.J, J3 = 3 * J, C6 = 6 * J;
  const int ldt = g.ldt, ldo = g.ldo;
  float* tile = smem;                // [TR][ldt] tanh outputs, later gradients
  float* offs = tile + TR * ldt;     // [TR][ldo]
  float* pose = offs + TR * ldo;     // [TR][ldo] positions (relative to the chain start, then absolute), then position gradients
  float* dm0 = pose + TR * ldo;      // [TR][9] gradient of the shared root rotation, accumulated chain by chain
  float* red = dm0 + TR * 9;         // [SVAE_MAX_CHAINS] per-wave partial sums
  const long long r0 = (long long)blockIdx.x * TR;
  const int nrows = (int)((g.rows - r0) < TR ? (g.rows - r0) : TR);
  const int n_chains = g.tree.n_chains;

  // ---- phase 0 (all threads): stage the y tile (tanh applied) and the offsets.  The targets are NOT staged: they are read once,
  // coalesced, where the position error is formed (phase 2b) -- 17 KB of LDS less per tile, which together with the single
  // root-rotation buffer lets TWO workgroups share a CU (75 KB each; the kernel is bound by its per-wave dependent LDS / VALU
  // chains, so the second workgroup's waves nearly double the throughput)
  {
    const int f4_per_row = g.ld / 4;
    const int total = nrows * f4_per_row;
    const float* src = g.y + r0 * g.ld;
    for (int e = tid; e < total; e += nth) {
      const int rr = e / f4_per_row, c = (e - rr * f4_per_row) * 4;
      const float4 v = *reinterpret_cast<const float4*>(src + (long long)e * 4);
      float* d = tile + rr * ldt + c;
      if (g.pre_tanh) { d[0] = tanhf(v.x); d[1] = tanhf(v.y); d[2] = tanhf(v.z); d[3] = tanhf(v.w); }
      else { d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w; }
    }
    const int tot3 = nrows * J3;
    const float* so = g.offsets + r0 * J3;
    for (int e = tid; e < tot3; e += nth) {
      const int rr = e / J3, c = e - rr * J3;
      offs[rr * ldo + c] = so[e];
    }
    for (int e = tid; e < TR * 9; e += nth) dm0[e] = 0.f;
  }
  __syncthreads();
  // ---- write x6d_hat / root_hat (coalesced, all threads)
  {
    const int tot6 = nrows * C6;
    float* dst = g.x6d_hat + r0 * C6;
    for (int e = tid; e < tot6; e += nth) {
      const int rr = e / C6, c = e - rr * C6;
      dst[e] = tile[rr * ldt + c];
    }
    if (g.has_arena) {
      float* dr = g.root_hat + r0 * 3;
      for (int e = tid; e < nrows * 3; e += nth) {
        const int rr = e / 3, k = e - rr * 3;
        dr[e] = 0.5f * (tile[rr * ldt + C6 + k] + 1.f) * (g.a1[k] - g.a0[k]) + g.a0[k];
      }
    }
  }

  const bool active = lane < nrows;
  const int lrow = lane < TR ? lane : 0;  // (lanes past the tile are inactive; keep their pointers inside it)
  float* my = tile + lrow * ldt;
  float* myo = offs + lrow * ldo;
  float* myp = pose + lrow * ldo;
  const bool do_bwd = g.dy != nullptr;
  const bool chain_wave = wave < n_chains && g.tree.chain_len[wave < n_chains ? wave : 0] >= 2;
  const int ch = wave < n_chains ? wave : 0;
  const int len = g.tree.chain_len[ch];

  // ---- phase 1 (wave = chain): rotations along the chain, positions relative to the chain's first joint
  float a6[6];
  M3 M0;
  if (active) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a6[k] = my[k];
    M0 = c6_to_mat(a6);
  }
  M3 Rc = M0;  // the chain's running rotation product; after phase 1: through the chain's LAST joint (phase 3 walks it back)
  if (active && chain_wave) {
    M3& R = Rc;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int i = 1; i < len; ++i) {
      const int j = g.tree.chain[ch][i];
#pragma unroll
      for (int k = 0; k < 6; ++k) a6[k] = my[6 * j + k];
      R = mul(R, c6_to_mat(a6));
      const float o0 = myo[3 * j], o1 = myo[3 * j + 1], o2 = myo[3 * j + 2];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        acc[r] += R.m[r * 3] * o0 + R.m[r * 3 + 1] * o1 + R.m[r * 3 + 2] * o2;
        myp[3 * j + r] = acc[r];
      }
    }
  }
  __syncthreads();

  // ---- phase 2a (wave 0): absolute positions in chain order; (wave 1, or 0 if there is only one): root loss
  float rl = 0.f;
  if (wave == 0 && active) {
    myp[0] = 0.f; myp[1] = 0.f; myp[2] = 0.f;
    for (int c = 0; c < n_chains; ++c) {
      const int cl = g.tree.chain_len[c];
      const int head = g.tree.chain[c][0];
      const float h0 = myp[3 * head], h1 = myp[3 * head + 1], h2 = myp[3 * head + 2];
      for (int i = 1; i < cl; ++i) {
        const int j = g.tree.chain[c][i];
        myp[3 * j] += h0; myp[3 * j + 1] += h1; myp[3 * j + 2] += h2;
      }
    }
  }
  const int root_wave = nth > 64 ? 1 : 0;
  if (wave == root_wave && active && g.has_arena) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float xh = my[C6 + k];
      const float half = 0.5f * (g.a1[k] - g.a0[k]);
      const float rh = (xh + 1.f) * half + g.a0[k];
      const float d = rh - g.root[(r0 + lane) * 3 + k];
      rl += d * d;
      if (do_bwd) {
        float gk = 2.f * g.root_scale * d;
        if (g.ext_droot) gk += g.ext_droot[(r0 + lane) * 3 + k];
        my[C6 + k] = g.pre_tanh ? gk * half * (1.f - xh * xh) : gk * half;
      }
    }
  }
  if (wave == root_wave && active && do_bwd) {
    for (int c = C6 + (g.has_arena ? 3 : 0); c < g.ld; ++c) my[c] = 0.f;  // padding columns carry no gradient
  }
  __syncthreads();

  // ---- phase 2b (all threads, coalesced): position error against the targets read straight from global memory, squared-error
  // partial sums, and the direct position gradients written in place of the positions
  {
    float jpe = 0.f;
    const int tot3 = nrows * J3;
    const float* stg = g.target + r0 * J3;
    float* dp = g.pose_out ? g.pose_out + r0 * J3 : nullptr;
    for (int e = tid; e < tot3; e += nth) {
      const int rr = e / J3, c = e - rr * J3;
      const float pv = pose[rr * ldo + c];
      if (dp) dp[e] = pv;
      const float d = pv - stg[e];
      jpe += d * d;
      pose[rr * ldo + c] = 2.f * g.jpe_scale * d;
    }
    jpe = wave_sum(jpe);
    if (lane == 0) red[wave] = jpe;
  }
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int w = 0; w < nth / 64; ++w) t += red[w];
    g.loss_part[blockIdx.x * 2] = t;
  }
  // ---- phase 2c (wave 0): subtree-summed position gradients
  if (do_bwd && wave == 0 && active) {
    for (int c = n_chains - 1; c >= 0; --c) {
      const int cl = g.tree.chain_len[c];
      for (int i = cl - 1; i >= 1; --i) {
        const int j = g.tree.chain[c][i], par = g.tree.chain[c][i - 1];
#pragma unroll
        for (int r = 0; r < 3; ++r) myp[3 * par + r] += myp[3 * j + r];
      }
    }
  }
  if (do_bwd) __syncthreads();

  // ---- phase 3 (wave = chain): chain-product backward; joints i >= 1 of a chain occur once as a rotation
  M3 carry;
#pragma unroll
  for (int k = 0; k < 9; ++k) carry.m[k] = 0.f;
  if (do_bwd && active && chain_wave) {
    // The backward of joint i needs the prefix product P_{i-1} = M_0 M_c1 ... M_c(i-1).  Keeping all of them (an array of up to 7
    // matrices, 63 registers) put the kernel at 194 VGPRs -- one 6-wave workgroup per CU.  They are walked BACK instead from the
    // full product phase 1 left in Rc:  P_{i-1} = P_i M_ci^-1, and M^-1 = diag(1 / |column|^2) M^T because the columns of a
    // cont6d matrix are mutually orthogonal by construction (x, z = x x b / |.|, y = z x x) and differ from unit length only by the
    // eps of the normalisations: one extra 3x3 product per joint, rounding-level differences (<= 1e-6) in the gradients.
    M3 P = Rc;
#pragma unroll
    for (int i = SVAE_MAX_CHAIN_LEN - 1; i >= 1; --i) {
      if (i < len) {
        const int j = g.tree.chain[ch][i];
        M3 D = carry;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int k = 0; k < 3; ++k) D.m[r * 3 + k] += myp[3 * j + r] * myo[3 * j + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) a6[k] = my[6 * j + k];
        const M3 Mj = c6_to_mat(a6);
        {  // P <- P Mj^-1 (prefix product up to the joint in front of j)
          float inv[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) inv[c] = 1.f / (Mj.m[c] * Mj.m[c] + Mj.m[3 + c] * Mj.m[3 + c] + Mj.m[6 + c] * Mj.m[6 + c]);
          M3 Q;
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k)  // (P Mj^-1)[r][k] = sum_c P[r][c] inv[c] Mj[k][c]
              Q.m[r * 3 + k] = P.m[r * 3] * inv[0] * Mj.m[k * 3] + P.m[r * 3 + 1] * inv[1] * Mj.m[k * 3 + 1] + P.m[r * 3 + 2] * inv[2] * Mj.m[k * 3 + 2];
          P = Q;
        }
        const M3 dMj = mul_tn(P, D);
        carry = mul_nt(D, Mj);
        float da[6];
        c6_to_mat_bwd(a6, dMj, da);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          float gk = da[k];
          if (g.ext_dx6d) gk += g.ext_dx6d[(r0 + lane) * C6 + 6 * j + k];
          my[6 * j + k] = g.pre_tanh ? gk * (1.f - a6[k] * a6[k]) : gk;
        }
      }
    }
  }
  if (do_bwd) {
    // the root rotation collects every chain's carry, chain by chain in a fixed order (one 2 KB buffer instead of one per chain)
    for (int c = 0; c < n_chains; ++c) {
      if (wave == c && chain_wave && active) {
        float* d = dm0 + lane * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) d[k] += carry.m[k];
      }
      __syncthreads();
    }
  }
  if (do_bwd && wave == 0 && active) {  // joint 0
    M3 dM0;
    const float* d = dm0 + lane * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) dM0.m[k] = d[k];
    float a0[6], da0[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) a0[k] = my[k];
    c6_to_mat_bwd(a0, dM0, da0);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float gk = da0[k];
      if (g.ext_dx6d) gk += g.ext_dx6d[(r0 + lane) * C6 + k];
      my[k] = g.pre_tanh ? gk * (1.f - a0[k] * a0[k]) : gk;
    }
  }
  if (wave == root_wave) {
    rl = wave_sum(rl);
    if (lane == 0) g.loss_part[blockIdx.x * 2 + 1] = rl;
  }
  __syncthreads();
  if (do_bwd) {
    const int f4_per_row = g.ld / 4;
    const int total = nrows * f4_per_row;
    float* dst = g.dy + r0 * g.ld;
    for (int e = tid; e < total; e += nth) {
      const int rr = e / f4_per_row, c = (e - rr * f4_per_row) * 4;
      const float* sp = tile + rr * ldt + c;
      *reinterpret_cast<float4*>(dst + (long long)e * 4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
    }
  }
}

// ------------------------------------------------------------------------ rotation loss
__device__ __forceinline__ void rot6d_rows(const float* d6, float* m /*9, rows b1,b2,b3*/) {
  const float n1 = fmaxf(sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]), 1e-12f);
  const float b1[3] = {d6[0] / n1, d6[1] / n1, d6[2] / n1};
  const float dt = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
  const float u[3] = {d6[3] - dt * b1[0], d6[4] - dt * b1[1], d6[5] - dt * b1[2]};
  const float n2 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float b3[3];
  cross3(b1, b2, b3);
#pragma unroll
  for (int k = 0; k < 3; ++k) { m[k] = b1[k]; m[3 + k] = b2[k]; m[6 + k] = b3[k]; }
}

// backward of rot6d_rows w.r.t. d6 given g (9)
__device__ __forceinline__ void rot6d_rows_bwd(const float* d6, const float* g, float* dd) {
  const float n1r = sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]);
  const float n1 = fmaxf(n1r, 1e-12f);
  const float b1[3] = {d6[0] / n1, d6[1] / n1, d6[2] / n1};
  const float dt = b1[0] * d6[3] + b1[1] * d6[4] + b1[2] * d6[5];
  const float u[3] = {d6[3] - dt * b1[0], d6[4] - dt * b1[1], d6[5] - dt * b1[2]};
  const float n2r = sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  const float n2 = fmaxf(n2r, 1e-12f);
  const float b2[3] = {u[0] / n2, u[1] / n2, u[2] / n2};
  float gb1[3] = {g[0], g[1], g[2]}, gb2[3] = {g[3], g[4], g[5]};
  const float gb3[3] = {g[6], g[7], g[8]};
  float t[3];
  cross3(b2, gb3, t);  // b3 = b1 x b2: db1 += b2 x g3
  gb1[0] += t[0]; gb1[1] += t[1]; gb1[2] += t[2];
  cross3(gb3, b1, t);  // db2 += g3 x b1
  gb2[0] += t[0]; gb2[1] += t[1]; gb2[2] += t[2];
  // b2 = u / n2
  float gu[3];
  if (n2r > 1e-12f) {
    const float d = b2[0] * gb2[0] + b2[1] * gb2[1] + b2[2] * gb2[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) gu[k] = (gb2[k] - b2[k] * d) / n2;
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) gu[k] = gb2[k] / n2;
  }
  // u = a2 - (b1.a2) b1
  const float gub1 = gu[0] * b1[0] + gu[1] * b1[1] + gu[2] * b1[2];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dd[3 + k] = gu[k] - gub1 * b1[k];
    gb1[k] += -dt * gu[k] - gub1 * d6[3 + k];
  }
  // b1 = a1 / n1
  if (n1r > 1e-12f) {
    const float d = b1[0] * gb1[0] + b1[1] * gb1[1] + b1[2] * gb1[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) dd[k] = (gb1[k] - b1[k] * d) / n1;
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) dd[k] = gb1[k] / n1;
  }
}

__global__ __launch_bounds__(256) void rot_loss_kernel(const float* __restrict__ x, const float* __restrict__ xh, float scale,
                                                        float* __restrict__ part, float* __restrict__ dxh, long long n) {
  __shared__ float red4[4];
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f;
  if (i < n) {
    float a[6], b[6], m1[9], m2[9];
#pragma unroll
    for (int k = 0; k < 6; ++k) { a[k] = x[i * 6 + k]; b[k] = xh[i * 6 + k]; }
    rot6d_rows(a, m1);
    rot6d_rows(b, m2);
    float ss = 0.f, d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) { d[k] = m2[k] - m1[k]; ss += d[k] * d[k]; }
    const float fro = sqrtf(ss);
    const float c = 0.35355339059327373f;  // 2^-1.5
    float s = fro * c;
    const float lim = 1.f - 1e-7f;
    const bool clamped = s > lim;
    if (clamped) s = lim;
    loss = 2.f * asinf(s);
    if (dxh) {
      float gm[9];
      // d loss/d s = 2/sqrt(1-s^2) (0 when clamped); ds/dm2 = c * d/fro
      const float gs = (!clamped && fro > 0.f) ? scale * 2.f / sqrtf(1.f - s * s) * c / fro : 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) gm[k] = gs * d[k];
      float dd[6];
      rot6d_rows_bwd(b, gm, dd);
#pragma unroll
      for (int k = 0; k < 6; ++k) dxh[i * 6 + k] = dd[k];
    }
  }
  const float t = block_sum_256(loss, red4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

}  // namespace svae

using namespace svae;

// frames per workgroup of the tail kernel: 64 (75 KB of LDS: two workgroups per CU); SVAE_TAIL_ROWS=64|48|32 overrides it for experiments
// (fewer frames per workgroup measured slower: idle lanes in the per-chain waves cost more than the extra co-residency buys)
static int tail_rows() {
  static int tr = 0;
  if (tr == 0) {
    const char* e = getenv("SVAE_TAIL_ROWS");
    const int v = e ? atoi(e) : 64;
    tr = (v == 64 || v == 48 || v == 32) ? v : 64;
  }
  return tr;
}

extern "C" int svae_tail_blocks(long long rows) { const int tr = tail_rows(); return (int)((rows + tr - 1) / tr); }

extern "C" int svae_pose_tail(const float* y, int ld, const float* offsets, const float* target_pose, const float* root,
                              const float* arena_host, const svae_tree* tree, float jpe_scale, float root_scale,
                              const float* ext_dx6d, const float* ext_droot, float* x6d_hat, float* root_hat, float* loss_part,
                              float* dy, float* pose_out, long long rows, int input_is_pre_tanh, void* stream) {
  SVAE_REQUIRE(y && offsets && target_pose && tree && x6d_hat && loss_part && rows > 0, SVAE_ERR_ARG, "pose_tail: null pointer");
  const int J = tree->n_joints;
  SVAE_REQUIRE(J >= 1 && J <= SVAE_MAX_JOINTS && tree->n_chains >= 0 && tree->n_chains <= SVAE_MAX_CHAINS, SVAE_ERR_SHAPE,
               "pose_tail: bad tree");
  SVAE_REQUIRE(ld % 4 == 0 && ld >= 6 * J + (arena_host ? 3 : 0) && aligned16(y) && (!dy || aligned16(dy)), SVAE_ERR_ALIGN,
               "pose_tail: ld %d too small or misaligned", ld);
  SVAE_REQUIRE(!arena_host || (root && root_hat), SVAE_ERR_ARG, "pose_tail: arena without root buffers");
  // every joint other than 0 must be produced exactly once, heads must already exist
  int seen[SVAE_MAX_JOINTS] = {0};
  seen[0] = 1;
  for (int c = 0; c < tree->n_chains; ++c) {
    const int len = tree->chain_len[c];
    SVAE_REQUIRE(len >= 1 && len <= SVAE_MAX_CHAIN_LEN, SVAE_ERR_SHAPE, "pose_tail: chain %d length %d", c, len);
    for (int i = 0; i < len; ++i) SVAE_REQUIRE(tree->chain[c][i] >= 0 && tree->chain[c][i] < J, SVAE_ERR_SHAPE, "pose_tail: joint id");
    SVAE_REQUIRE(seen[tree->chain[c][0]], SVAE_ERR_SHAPE, "pose_tail: chain %d starts at a joint not yet placed", c);
    for (int i = 1; i < len; ++i) {
      SVAE_REQUIRE(!seen[tree->chain[c][i]], SVAE_ERR_SHAPE, "pose_tail: joint %d placed twice", tree->chain[c][i]);
      seen[tree->chain[c][i]] = 1;
    }
  }
  for (int j = 0; j < J; ++j) SVAE_REQUIRE(seen[j], SVAE_ERR_SHAPE, "pose_tail: joint %d not covered by the kinematic tree", j);
  TailArgs g;
  memset(&g, 0, sizeof(g));
  g.y = y; g.offsets = offsets; g.target = target_pose; g.root = root; g.ext_dx6d = ext_dx6d; g.ext_droot = ext_droot;
  g.x6d_hat = x6d_hat; g.root_hat = root_hat; g.loss_part = loss_part; g.dy = dy; g.pose_out = pose_out;
  g.rows = rows; g.ld = ld; g.J = J;
  g.ldt = ld | 1;
  g.ldo = (3 * J) | 1;
  g.has_arena = arena_host != nullptr;
  g.pre_tanh = input_is_pre_tanh;
  if (arena_host)
    for (int k = 0; k < 3; ++k) { g.a0[k] = arena_host[k]; g.a1[k] = arena_host[3 + k]; }
  g.jpe_scale = jpe_scale; g.root_scale = root_scale;
  g.tree = *tree;
  const int n_waves = tree->n_chains >= 2 ? tree->n_chains : 2;
  const int tr = tail_rows();
  const size_t smem = (size_t)(tr * g.ldt + 2 * tr * g.ldo + tr * 9 + SVAE_MAX_CHAINS) * sizeof(float);
  SVAE_REQUIRE(smem <= 160 * 1024, SVAE_ERR_SHAPE, "pose_tail: LDS tile %zu B exceeds 160 KiB", smem);
  static DeviceOnce attr_set;
  int attr_dev;
  if (attr_set.need(&attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)pose_tail_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pose_tail_kernel<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)pose_tail_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    SVAE_REQUIRE(e == hipSuccess, SVAE_ERR_LAUNCH, "pose_tail: hipFuncSetAttribute(MaxDynamicSharedMemorySize): %s", hipGetErrorString(e));
    attr_set.done(attr_dev);
  }
  const dim3 grid(svae_tail_blocks(rows)), block(64 * n_waves);
  if (tr == 64) hipLaunchKernelGGL(pose_tail_kernel<64>, grid, block, smem, (hipStream_t)stream, g);
  else if (tr == 48) hipLaunchKernelGGL(pose_tail_kernel<48>, grid, block, smem, (hipStream_t)stream, g);
  else hipLaunchKernelGGL(pose_tail_kernel<32>, grid, block, smem, (hipStream_t)stream, g);
  return check_launch("pose_tail");
}

extern "C" int svae_rot_blocks(long long n) { return (int)((n + 255) / 256); }

extern "C" int svae_rot_loss(const float* x6d, const float* x6d_hat, float scale, float* part, float* dx6d_hat, long long n,
                             void* stream) {
  SVAE_REQUIRE(x6d && x6d_hat && part && n > 0, SVAE_ERR_ARG, "rot_loss: bad args");
  hipLaunchKernelGGL(rot_loss_kernel, dim3(svae_rot_blocks(n)), dim3(256), 0, (hipStream_t)stream, x6d, x6d_hat, scale, part,
                     dx6d_hat, n);
  return check_launch("rot_loss");
}
