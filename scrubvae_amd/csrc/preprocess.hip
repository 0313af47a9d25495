// Pose preprocessing on the device: the step in front of the training path.
//
//   inv_kin_kernel      dataset.py:11-46 (inv_kin: pose -> local quaternions along the kinematic chains)
//                       + dataset.py:385-404 (root centring, "midfwd" re-orientation of the root quaternion and
//                         of the root trajectory by the yaw of the window's middle frame)
//                       + quaternion.py:291-334 (quaternion -> first two columns of the rotation matrix)
//                       + dataset.py:279-296 (get_segment_len: unit offsets scaled by the segment lengths)
//                       + dataset.py:234-241, 258-265 (heading = [sin yaw, cos yaw] of the middle frame)
//   speed_parts_kernel  dataset.py:133-163 (get_speed_parts) + :373-375 (limbs averaged)
//
// HBM-bound, fp32 (the reference computes the pose differences in float64 numpy and every quaternion helper in
// float32 torch).  A frame is 3J floats in and 12J + 3 out.
#include "svae_internal.h"

namespace svae {

struct Q4 { float w, x, y, z; };

__device__ __forceinline__ Q4 qmul(const Q4& q, const Q4& r) {  // Hamilton product q*r (quaternion.py:34-52)
  Q4 o;
  o.w = q.w * r.w - q.x * r.x - q.y * r.y - q.z * r.z;
  o.x = q.x * r.w + q.w * r.x - q.z * r.y + q.y * r.z;
  o.y = q.y * r.w + q.z * r.x + q.w * r.y - q.x * r.z;
  o.z = q.z * r.w - q.y * r.x + q.x * r.y + q.w * r.z;
  return o;
}
__device__ __forceinline__ Q4 qinv(const Q4& q) { return Q4{q.w, -q.x, -q.y, -q.z}; }
// quaternion.py:409-420: rotation taking v0 to v1
__device__ __forceinline__ Q4 qbetween(const float* v0, const float* v1) {
  Q4 q;
  q.x = v0[1] * v1[2] - v0[2] * v1[1];
  q.y = v0[2] * v1[0] - v0[0] * v1[2];
  q.z = v0[0] * v1[1] - v0[1] * v1[0];
  q.w = sqrtf((v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]) * (v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2])) +
        (v0[0] * v1[0] + v0[1] * v1[1] + v0[2] * v1[2]);
  const float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  q.w /= n; q.x /= n; q.y /= n; q.z /= n;
  return q;
}
// quaternion.py:291-334: columns 0 and 1 of the rotation matrix
__device__ __forceinline__ void q_to_cont6d(const Q4& q, float* o) {
  const float two_s = 2.0f / (q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  o[0] = 1.f - two_s * (q.y * q.y + q.z * q.z);
  o[1] = two_s * (q.x * q.y + q.z * q.w);
  o[2] = two_s * (q.x * q.z - q.y * q.w);
  o[3] = two_s * (q.x * q.y - q.z * q.w);
  o[4] = 1.f - two_s * (q.x * q.x + q.z * q.z);
  o[5] = two_s * (q.y * q.z + q.x * q.w);
}
__device__ __forceinline__ void normalize3(float* v) {
  const float n = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  v[0] /= n; v[1] /= n; v[2] /= n;
}

struct InvKinArgs {
  const float* pose;   // [frames][J][3]
  float* x6d;          // [frames][J][6]
  float* offsets;      // [frames][J][3] or null
  float* root;         // [frames][3] or null
  float* heading;      // [frames / window][2] or null
  long long frames;
  int J, window;
  int midfwd;          // 1: rotate root quaternion / root trajectory by the middle-frame yaw
  int centre_root;     // 1: subtract the middle frame's root (x, y)
  int truncate_len;    // 1: segment lengths truncated toward zero (integer OFFSET array in the reference)
  int parent[SVAE_MAX_JOINTS];
  float uoff[SVAE_MAX_JOINTS][3];
  svae_tree tree;
};

// One workgroup per 64 consecutive frames, one wave per kinematic chain (lane = frame): every chain starts from the
// frame's root quaternion, so the chains are independent.  The pose rows come in and the x6d / offset rows go out
// through LDS tiles with fully coalesced accesses (a block of 64 frames is one contiguous span of each array; the
// odd row strides make the lane-per-frame phase bank-conflict free).
__global__ __launch_bounds__(64 * SVAE_MAX_CHAINS) void inv_kin_kernel(const InvKinArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nth = blockDim.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int J = g.J, J3 = 3 * J, J6 = 6 * J;
  const int ldp = J3 | 1, ldx = J6 | 1;
  float* tp = smem;             // [64][ldp] pose
  float* tx = tp + 64 * ldp;    // [64][ldx] x6d
  float* to = tx + 64 * ldx;    // [64][ldp] offsets
  const long long f0 = (long long)blockIdx.x * 64;
  const int nf = (int)((g.frames - f0) < 64 ? (g.frames - f0) : 64);

  for (int e = tid; e < nf * J3; e += nth) {
    const int rr = e / J3, c = e - rr * J3;
    tp[rr * ldp + c] = g.pose[f0 * J3 + e];
  }
  __syncthreads();

  const bool active = lane < nf;
  const long long f = f0 + lane;
  const float* p = tp + lane * ldp;
  if (active) {
    const long long win = f / g.window;
    const float* pm = g.pose + (win * g.window + g.window / 2) * J3;  // middle frame of this window (may be outside the tile)
    float fw[3] = {pm[3] - pm[0], pm[4] - pm[1], pm[5] - pm[2]};
    normalize3(fw);
    const float yaw = -atan2f(fw[1], fw[0]);
    const Q4 fwd_q = {cosf(0.5f * yaw), 0.f, 0.f, sinf(0.5f * yaw)};
    // root quaternion: forward_indices = [1, 0] -> pose[0] - pose[1], rotated onto +x; frame 0 of the array is identity
    float fr[3] = {p[0] - p[3], p[1] - p[4], p[2] - p[5]};
    normalize3(fr);
    const float ex[3] = {1.f, 0.f, 0.f};
    Q4 root_q = qbetween(fr, ex);
    if (f == 0) root_q = Q4{1.f, 0.f, 0.f, 0.f};
    float* out = tx + lane * ldx;
    if (wave == 0) {
      if (g.heading && f == win * g.window) {
        g.heading[win * 2] = sinf(yaw);
        g.heading[win * 2 + 1] = cosf(yaw);
      }
      const Q4 q0 = g.midfwd ? qmul(fwd_q, root_q) : root_q;
      q_to_cont6d(q0, out);
      if (g.root) {
        float r[3] = {p[0], p[1], p[2]};
        if (g.centre_root) { r[0] -= pm[0]; r[1] -= pm[1]; }
        if (g.midfwd) {  // qrot(fwd_q, r), quaternion.py:55-74
          const float qv[3] = {fwd_q.x, fwd_q.y, fwd_q.z};
          const float uv[3] = {qv[1] * r[2] - qv[2] * r[1], qv[2] * r[0] - qv[0] * r[2], qv[0] * r[1] - qv[1] * r[0]};
          const float uuv[3] = {qv[1] * uv[2] - qv[2] * uv[1], qv[2] * uv[0] - qv[0] * uv[2], qv[0] * uv[1] - qv[1] * uv[0]};
#pragma unroll
          for (int k = 0; k < 3; ++k) r[k] = r[k] + 2.f * (fwd_q.w * uv[k] + uuv[k]);
        }
        float* ro = g.root + f * 3;
        ro[0] = r[0]; ro[1] = r[1]; ro[2] = r[2];
      }
    }
    for (int c = wave; c < g.tree.n_chains; c += (nth >> 6)) {
      Q4 R = root_q;
      const int len = g.tree.chain_len[c];
      for (int i = 0; i + 1 < len; ++i) {
        const int a = g.tree.chain[c][i], b = g.tree.chain[c][i + 1];
        float v[3] = {p[3 * b] - p[3 * a], p[3 * b + 1] - p[3 * a + 1], p[3 * b + 2] - p[3 * a + 2]};
        normalize3(v);
        const Q4 rot = qbetween(g.uoff[b], v);
        const Q4 loc = qmul(qinv(R), rot);
        q_to_cont6d(loc, out + 6 * b);
        R = qmul(R, loc);
      }
    }
  }
  if (g.offsets) {  // get_segment_len: one (frame, joint) per thread-iteration
    for (int e = tid; e < nf * J; e += nth) {
      const int rr = e / J, j = e - rr * J;
      const float* q = tp + rr * ldp;
      float* o = to + rr * ldp + 3 * j;
      if (j == 0) {
        o[0] = g.uoff[0][0]; o[1] = g.uoff[0][1]; o[2] = g.uoff[0][2];
      } else {
        const int a = g.parent[j];
        const float d0 = q[3 * j] - q[3 * a], d1 = q[3 * j + 1] - q[3 * a + 1], d2 = q[3 * j + 2] - q[3 * a + 2];
        const float len = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float v = len * g.uoff[j][k];
          o[k] = g.truncate_len ? truncf(v) : v;
        }
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < nf * J6; e += nth) {
    const int rr = e / J6, c = e - rr * J6;
    g.x6d[f0 * J6 + e] = tx[rr * ldx + c];
  }
  if (g.offsets)
    for (int e = tid; e < nf * J3; e += nth) {
      const int rr = e / J3, c = e - rr * J3;
      g.offsets[f0 * J3 + e] = to[rr * ldp + c];
    }
}

// get_speed_parts: one wave per window.  out[win][3] = [root speed, spine+head, mean(arms, legs)].
// The reference's `centered_pose[:, part[0]:part[0]+1]` subtraction (dataset.py:147) indexes the window axis and
// cancels under the frame difference, so each part's speed is the mean over its joints part[1:] and the W-1 frame
// pairs of |d/dt (pose_j - pose_0)|.
struct SpeedArgs {
  const float* pose;  // [windows][W][J][3]
  float* out;         // [windows][3]
  long long windows;
  int W, J;
  int n_parts;
  int part_len[4];
  int part[4][SVAE_MAX_JOINTS];
};

__global__ __launch_bounds__(64) void speed_parts_kernel(const SpeedArgs g) {
  const long long win = blockIdx.x;
  const int lane = threadIdx.x;
  const float* base = g.pose + win * g.W * g.J * 3;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int w = lane; w + 1 < g.W; w += 64) {
    const float* a = base + (long long)w * g.J * 3;
    const float* b = a + g.J * 3;
    const float dr[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]};
    acc[0] += sqrtf(dr[0] * dr[0] + dr[1] * dr[1] + dr[2] * dr[2]);
    for (int pi = 0; pi < g.n_parts; ++pi) {
      float s = 0.f;
      for (int k = 1; k < g.part_len[pi]; ++k) {
        const int j = g.part[pi][k];
        const float d0 = (b[3 * j] - b[0]) - (a[3 * j] - a[0]);
        const float d1 = (b[3 * j + 1] - b[1]) - (a[3 * j + 1] - a[1]);
        const float d2 = (b[3 * j + 2] - b[2]) - (a[3 * j + 2] - a[2]);
        s += sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
      }
      acc[1 + pi] += s;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = wave_sum(acc[i]);
  if (lane == 0) {
    const float nw = (float)(g.W - 1);
    float sp[4];
    sp[0] = acc[0] / nw;
    for (int pi = 0; pi < g.n_parts; ++pi) sp[1 + pi] = acc[1 + pi] / (nw * (float)(g.part_len[pi] - 1));
    float limbs = 0.f;
    for (int pi = 1; pi < g.n_parts; ++pi) limbs += sp[1 + pi];
    g.out[win * 3] = sp[0];
    g.out[win * 3 + 1] = sp[1];
    g.out[win * 3 + 2] = g.n_parts > 1 ? limbs / (float)(g.n_parts - 1) : 0.f;
  }
}

}  // namespace svae

using namespace svae;

extern "C" int svae_inv_kin(const float* pose, const float* unit_offset_host, const svae_tree* tree, int window, int midfwd,
                            int centre_root, int truncate_len, float* x6d, float* offsets, float* root, float* heading,
                            long long frames, void* stream) {
  SVAE_REQUIRE(pose && unit_offset_host && tree && x6d, SVAE_ERR_ARG, "inv_kin: null pointer");
  SVAE_REQUIRE(tree->n_joints >= 2 && tree->n_joints <= SVAE_MAX_JOINTS, SVAE_ERR_SHAPE, "inv_kin: %d joints not in [2,%d]",
               tree->n_joints, SVAE_MAX_JOINTS);
  SVAE_REQUIRE(tree->n_chains >= 1 && tree->n_chains <= SVAE_MAX_CHAINS, SVAE_ERR_SHAPE, "inv_kin: bad chain count");
  SVAE_REQUIRE(window >= 1 && frames >= 0 && frames % window == 0, SVAE_ERR_SHAPE, "inv_kin: frames %lld not a multiple of window %d",
               frames, window);
  if (frames == 0) return SVAE_OK;
  InvKinArgs g;
  memset(&g, 0, sizeof(g));
  g.pose = pose; g.x6d = x6d; g.offsets = offsets; g.root = root; g.heading = heading;
  g.frames = frames; g.J = tree->n_joints; g.window = window;
  g.midfwd = midfwd; g.centre_root = centre_root; g.truncate_len = truncate_len;
  g.tree = *tree;
  g.parent[0] = 0;
  for (int j = 1; j < g.J; ++j) g.parent[j] = 0;
  for (int c = 0; c < tree->n_chains; ++c) {
    SVAE_REQUIRE(tree->chain_len[c] >= 1 && tree->chain_len[c] <= SVAE_MAX_CHAIN_LEN, SVAE_ERR_SHAPE, "inv_kin: chain %d length", c);
    for (int i = 0; i < tree->chain_len[c]; ++i)
      SVAE_REQUIRE(tree->chain[c][i] >= 0 && tree->chain[c][i] < g.J, SVAE_ERR_SHAPE, "inv_kin: joint index out of range");
    for (int i = 1; i < tree->chain_len[c]; ++i) g.parent[tree->chain[c][i]] = tree->chain[c][i - 1];
  }
  for (int j = 0; j < g.J; ++j)
    for (int k = 0; k < 3; ++k) g.uoff[j][k] = unit_offset_host[3 * j + k];
  const int n_waves = tree->n_chains >= 2 ? tree->n_chains : 2;
  const int J3 = 3 * g.J, J6 = 6 * g.J;
  const size_t smem = (size_t)64 * (2 * (J3 | 1) + (J6 | 1)) * sizeof(float);
  hipLaunchKernelGGL(inv_kin_kernel, dim3((unsigned)((frames + 63) / 64)), dim3(64 * n_waves), smem, (hipStream_t)stream, g);
  return check_launch("inv_kin");
}

extern "C" int svae_speed_parts(const float* pose, const int* parts_host, const int* part_len_host, int n_parts, int W, int J,
                                float* out, long long windows, void* stream) {
  SVAE_REQUIRE(pose && parts_host && part_len_host && out, SVAE_ERR_ARG, "speed_parts: null pointer");
  SVAE_REQUIRE(n_parts >= 1 && n_parts <= 3 && W >= 2 && J >= 1 && J <= SVAE_MAX_JOINTS, SVAE_ERR_SHAPE, "speed_parts: bad shape");
  if (windows == 0) return SVAE_OK;
  SpeedArgs g;
  memset(&g, 0, sizeof(g));
  g.pose = pose; g.out = out; g.windows = windows; g.W = W; g.J = J; g.n_parts = n_parts;
  int o = 0;
  for (int p = 0; p < n_parts; ++p) {
    SVAE_REQUIRE(part_len_host[p] >= 2 && part_len_host[p] <= SVAE_MAX_JOINTS, SVAE_ERR_SHAPE, "speed_parts: part %d length", p);
    g.part_len[p] = part_len_host[p];
    for (int k = 0; k < part_len_host[p]; ++k) {
      SVAE_REQUIRE(parts_host[o + k] >= 0 && parts_host[o + k] < J, SVAE_ERR_SHAPE, "speed_parts: joint index out of range");
      g.part[p][k] = parts_host[o + k];
    }
    o += part_len_host[p];
  }
  hipLaunchKernelGGL(speed_parts_kernel, dim3((unsigned)windows), dim3(64), 0, (hipStream_t)stream, g);
  return check_launch("speed_parts");
}
