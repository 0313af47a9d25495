// Split-bf16 implicit-GEMM convolution kernels for gfx950 (MI355X).
//
// Same contractions, gather plans and layouts as gemm_f32.hip (reference:
// src/scrubvae/model/residual.py:79-109,137-170,198,219-222,264,286 and their autograd), but
// the products run on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, 16x the fp32 MFMA rate):
// every fp32 operand is split into P bf16 pieces  x = x0 + x1 + x2  (each piece the leading
// 8 significant bits of what is left, so three pieces hold all 24 bits of an fp32 exactly), and
// the cross products with i + j < P are accumulated in fp32:
//
//   P = 3 : 6 MFMAs per 16-deep k-step; dropped terms are O(2^-24) relative -- the result is as
//           accurate as the fp32 MFMA path (tests/studies/precision_bf16_split.py) at 6/16 of its
//           matrix-core time;
//   P = 2 : 3 MFMAs (error O(2^-16));   P = 1 : plain bf16 operands (error O(2^-8)).
//
// Activations stay fp32 in HBM and are split on the way into LDS (11 VALU per pair of elements);
// weights are split once per pass by split_weights(_batched)_kernel into piece planes pre-tiled in the
// order the GEMMs stage them, for both uses:
//   Wf[piece][tap][c_in/32][c_out][32]  (forward: k = c_in)     Wd[piece][tap][c_out/32][c_in][32]  (dgrad: k = c_out)
// so the weight operand needs no conversion, no transposed LDS read, and loads as contiguous 128-byte lines.
// LDS image of a tile: [piece][row][32 k] bf16, 16-byte chunks XOR-swizzled by the row so that the
// ds_read_b128 operand fetches are conflict-free without padding.
//
// Kernels: gather_gemm_bf16s_kernel (A tile re-staged per tap; 4 / 8 waves, single / double LDS buffer),
// gather_gemm_bf16s_ws_kernel (producer / consumer waves), gather_halo_bf16s_kernel (one A image per channel
// block shared by all taps), wgrad_gemm_bf16s_kernel (both operands transposed into LDS), split_weights*.
#include "gemm_common.h"
#include <type_traits>

namespace svae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned f2u(float f) { return __builtin_bit_cast(unsigned, f); }
__device__ __forceinline__ float u2f(unsigned u) { return __builtin_bit_cast(float, u); }
// {hi16(b), hi16(a)}: the truncated-bf16 pair (a in the low half = lower k)
__device__ __forceinline__ unsigned pack_hi(float a, float b) { return __builtin_amdgcn_perm(f2u(b), f2u(a), 0x07060302u); }
__device__ __forceinline__ unsigned pack_rne(float a, float b) {
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float residual(float x) { return x - u2f(f2u(x) & 0xffff0000u); }  // exact

// split 4 consecutive-k floats into P pieces of 4 bf16 (8 bytes each)
template <int P>
__device__ __forceinline__ void split4(float4 v, uint2 (&out)[P]) {
  float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (p == P - 1) {
      out[p].x = pack_rne(x[0], x[1]);
      out[p].y = pack_rne(x[2], x[3]);
    } else {
      out[p].x = pack_hi(x[0], x[1]);
      out[p].y = pack_hi(x[2], x[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) x[i] = residual(x[i]);
    }
  }
}

// ---- fp16 pieces (H = true, P = 2): x = h0 + h1 + O(2^-22 x), h0 = fp16(x) (11 significant bits), h1 = fp16(x - h0): two pieces hold
// 22 of the 24 bits of an fp32 against 16 for two bf16 pieces, so the THREE cross products h0*g0 + h0*g1 + h1*g0 are accurate to
// ~2^-22 per product -- fp32-class accuracy at half the matrix-core work of the 6-product bf16 split (the fp16 and bf16 MFMAs run at
// the same rate).  fp16 has a narrow exponent range: operands above 65504 saturate (round-toward-zero conversion: no infinities)
// and second pieces below 6e-5 go subnormal (absolute error <= 3e-8, negligible next to O(1) activations); the WEIGHTS, whose
// second pieces would sit there, are scaled by 2^10 when they are split and the accumulators by 2^-10 in the epilogue (both exact).
// Used for the forward pass only (precision "f16x3b3"): gradients span too many decades for unscaled fp16.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_WSCALE = 1024.f, F16_OSCALE = 1.f / 1024.f;
__device__ __forceinline__ unsigned pack_h2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(a, b)); }
__device__ __forceinline__ float h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ void split4h(float4 v, uint2 (&out)[2]) {
  const unsigned a = pack_h2(v.x, v.y), b = pack_h2(v.z, v.w);
  out[0].x = a;
  out[0].y = b;
  const float r0 = v.x - h2f((unsigned short)(a & 0xffffu)), r1 = v.y - h2f((unsigned short)(a >> 16));
  const float r2 = v.z - h2f((unsigned short)(b & 0xffffu)), r3 = v.w - h2f((unsigned short)(b >> 16));
  out[1].x = pack_h2(r0, r1);
  out[1].y = pack_h2(r2, r3);
}
template <int P, bool H>
__device__ __forceinline__ void split4x(float4 v, uint2 (&out)[P]) {
  if constexpr (H) { static_assert(P == 2, "fp16 pieces: two"); split4h(v, out); }
  else split4<P>(v, out);
}

// 16-byte chunk c of LDS row `row` (64-byte rows = 32 bf16) sits at chunk position c ^ swz(row):
// the ds_read_b128 operand fetch of 32 consecutive rows is then conflict-free without padding
__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ f32x16 mfma_bf16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// acc += sum over kept cross products of the pieces (small terms first)
template <int P>
__device__ __forceinline__ f32x16 mfma_split(const uint4 (&a)[P], const uint4 (&b)[P], f32x16 acc) {
#pragma unroll
  for (int s = P - 1; s >= 0; --s)
#pragma unroll
    for (int i = s; i >= 0; --i) acc = mfma_bf16(a[i], b[s - i], acc);
  return acc;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gptr_t;

__device__ __forceinline__ f32x16 mfma_f16(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <int P, bool H>
__device__ __forceinline__ f32x16 mfma_splitx(const uint4 (&a)[P], const uint4 (&b)[P], f32x16 acc) {
  if constexpr (H) {
#pragma unroll
    for (int s = P - 1; s >= 0; --s)
#pragma unroll
      for (int i = s; i >= 0; --i) acc = mfma_f16(a[i], b[s - i], acc);
    return acc;
  } else {
    return mfma_split<P>(a, b, acc);
  }
}

constexpr int SBK = 32;  // K depth of one LDS stage: a 128-byte line of every gathered fp32 row

struct SplitGatherArgs {
  GatherArgs g;               // g.W / g.ldW / g.w_tap_stride unused
  const unsigned short* Wp;   // weight pieces, pre-tiled [piece][tap][k/32][n][32] (zero padded in k)
  long long w_piece_stride;   // elements between piece planes
  long long rowsA;            // rows of the gathered operand (batch * Lin): bound of the halo image
  int KB;                     // 32-deep k blocks per tap
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

// ------------------------------------------------------------------ gather GEMM (fwd / dgrad)
// BM x BN tile per workgroup of WR x WC waves.  NSTAGE = 2: double-buffered LDS, one barrier per
// K stage (register prefetch two stages ahead);  NSTAGE = 1: one LDS buffer, two barriers per
// stage, half the LDS -> more workgroups per CU hide each other's conversion / barrier phases.
template <int BM, int BN, int P, int WR, int WC, int NSTAGE, int MINW, bool H = false>
__global__ __launch_bounds__(64 * WR * WC, MINW) void gather_gemm_bf16s_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int NTH = 64 * WR * WC;
  constexpr int WM = BM / WR, MT = WM / 32, WN = BN / WC, NT = WN / 32;
  static_assert(MT >= 1 && NT >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  constexpr int RPP = NTH / 8;          // A rows per pass (8 lanes x float4 per row)
  static_assert(BM % RPP == 0, "A rows must divide over the passes");
  constexpr int APASS = BM / RPP;
  constexpr int CH = 4;                 // 16-byte chunks per bf16 row
  constexpr int B_CHUNKS = P * BN * CH;
  constexpr int BPASS = (B_CHUNKS + NTH - 1) / NTH;
  constexpr bool B_EXACT = B_CHUNKS % NTH == 0;
  constexpr int ROWB = SBK * 2;         // bytes per LDS row
  constexpr int A_PIECE = BM * ROWB, B_PIECE = BN * ROWB;
  constexpr int STAGE = P * (A_PIECE + B_PIECE);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];
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

  const int akq = tid & 7;
  long long a_off[APASS];
  int a_j[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const long long m = m0 + (tid >> 3) + RPP * i;
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
  // B chunk owned by this thread in pass i: (piece, row, chunk); rows past N read row 0 and are zeroed
  int b_lds[BPASS];
  long long b_goff[BPASS];
  bool b_in[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int idx = tid + NTH * i;
    const int piece = (idx / (BN * CH)) % P, rem = idx % (BN * CH);
    const int row = rem / CH, ch = rem % CH;
    b_in[i] = (B_EXACT || idx < B_CHUNKS) && n0 + row < g.N;
    b_goff[i] = (long long)piece * sa.w_piece_stride + (long long)(b_in[i] ? n0 + row : 0) * SBK + ch * 8;
    b_lds[i] = P * A_PIECE + piece * B_PIECE + row * ROWB + ((ch ^ swz(row)) << 4);
  }
  const long long kb_stride = (long long)g.N * SBK;          // one (tap, k-block) slab of a piece plane
  const long long tap_stride = kb_stride * sa.KB;

  const int nk = ntaps * sa.KB;

  float4 ra[APASS];
  uint4 rb[BPASS];
  bool ra_ok[APASS];
  int nx_c0 = 0, nx_ti = 0;
  int nx_tb = ntaps > 0 ? tap_base[0] : 0;
  long long nx_woff = ntaps > 0 ? (long long)tap_w[0] * tap_stride : 0;
  auto load_tile = [&]() {
    const int c0 = nx_c0;
    const int tb = nx_tb;
    const unsigned short* wt = sa.Wp + nx_woff + (long long)(c0 >> 5) * kb_stride;
    {
      nx_c0 += SBK;
      const bool wrap = nx_c0 >= g.Kc;
      nx_c0 = wrap ? 0 : nx_c0;
      nx_ti += wrap ? 1 : 0;
      const int tic = nx_ti < ntaps ? nx_ti : ntaps - 1;
      nx_tb = tap_base[tic];
      nx_woff = (long long)tap_w[tic] * tap_stride;
    }
    const bool kq_ok = c0 + akq * 4 < g.Kc;
    const int cq = kq_ok ? c0 + akq * 4 : 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int li = a_j[i] + tb;
      const bool ok = li >= 0 && li < g.Lin;
      const int lic = ok ? li : 0;
      ra[i] = *reinterpret_cast<const float4*>(g.A + a_off[i] + (long long)lic * g.ldA + cq);
      ra_ok[i] = ok && kq_ok;
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[i] = *reinterpret_cast<const uint4*>(wt + b_goff[i]);
  };
  auto store_tile = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int row = (tid >> 3) + RPP * i;
      uint2 pc[P];
      split4x<P, H>(ra_ok[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f), pc);
      const int off = row * ROWB + (((akq >> 1) ^ swz(row)) << 4) + ((akq & 1) << 3);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + p * A_PIECE + off) = pc[p];
    }
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const uint4 v = b_in[i] ? rb[i] : make_uint4(0u, 0u, 0u, 0u);
      if constexpr (B_EXACT) {
        *reinterpret_cast<uint4*>(st + b_lds[i]) = v;
      } else {
        if (tid + NTH * i < B_CHUNKS) *reinterpret_cast<uint4*>(st + b_lds[i]) = v;
      }
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](const unsigned char* st, int ks) {
    uint4 av[MT][P], bv[NT][P];
    const int ch = ks * 2 + h;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = wr * WM + mt * 32 + lr;
#pragma unroll
      for (int p = 0; p < P; ++p)
        av[mt][p] = *reinterpret_cast<const uint4*>(st + p * A_PIECE + row * ROWB + ((ch ^ swz(row)) << 4));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int row = wc * WN + nt * 32 + lr;
#pragma unroll
      for (int p = 0; p < P; ++p)
        bv[nt][p] = *reinterpret_cast<const uint4*>(st + P * A_PIECE + p * B_PIECE + row * ROWB + ((ch ^ swz(row)) << 4));
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_splitx<P, H>(av[mt], bv[nt], acc[mt][nt]);
  };

  if constexpr (NSTAGE == 2) {
    if (nk > 0) {
      load_tile();
      store_tile(0);
      if (nk > 1) load_tile();
    }
    __syncthreads();
    auto k_tile = [&](int kt, auto do_store, auto do_load) {
      const unsigned char* st = smem + (kt & 1) * STAGE;
      if constexpr (decltype(do_store)::value) store_tile((kt & 1) ^ 1);
      if constexpr (decltype(do_load)::value) load_tile();
      compute(st, 0);
      compute(st, 1);
      __syncthreads();
    };
    int kt = 0;
    for (; kt + 2 < nk; ++kt) k_tile(kt, std::true_type{}, std::true_type{});
    if (kt + 1 < nk) { k_tile(kt, std::true_type{}, std::false_type{}); ++kt; }
    if (kt < nk) k_tile(kt, std::false_type{}, std::false_type{});
  } else {
    if (nk > 0) load_tile();
    __syncthreads();  // rowoff
    for (int kt = 0; kt < nk; ++kt) {
      store_tile(0);
      __syncthreads();
      if (kt + 1 < nk) load_tile();
      compute(smem, 0);
      compute(smem, 1);
      __syncthreads();
    }
  }

  tile_epilogue<MT, NT, WM, WN, WR, BN>(g, acc, rowoff, n0, wr, wc, lr, h, reinterpret_cast<float*>(smem), tid, NTH, H ? F16_OSCALE : 1.f);
}

// ------------------------------------------------- gather GEMM, wave-specialised (fwd / dgrad)
// In the kernel above all waves of a workgroup convert, then all run MFMAs: barrier-synchronised
// waves sharing a SIMD are in the same phase, so the VALU/LDS-write time of the split adds to the
// matrix-core time instead of hiding under it.  Here the roles are separated: waves 0-3 (one per
// SIMD) are PRODUCERS -- they gather + split the next A tile and stage the next weight tile -- and
// CWR x CWC CONSUMER waves only read operands from LDS and issue MFMAs.  Double-buffered LDS, one
// barrier per K stage, executed once per stage by both roles.
// DBG (timing experiments only, wrong results): 1 = producers issue no global loads, 2 = producers skip the
// split (stage the raw bits), 4 = consumers issue no MFMAs, 8 = consumers issue no LDS reads
template <int BM, int BN, int P, int CWR, int CWC, int D, int DBG = 0, bool H = false>
__global__ __launch_bounds__(64 * (4 + CWR * CWC)) void gather_gemm_bf16s_ws_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int WM = BM / CWR, MT = WM / 32, WN = BN / CWC, NT = WN / 32;
  static_assert(MT >= 1 && NT >= 1 && WM % 32 == 0 && WN % 32 == 0, "consumer tile must be a multiple of 32x32");
  constexpr int NPT = 256;              // producer threads
  constexpr int RPP = NPT / 8;
  constexpr int APASS = BM / RPP;
  constexpr int CH = 4;
  constexpr int B_CHUNKS = P * BN * CH;
  constexpr int BPASS = (B_CHUNKS + NPT - 1) / NPT;
  constexpr bool B_EXACT = B_CHUNKS % NPT == 0;
  constexpr int ROWB = SBK * 2;
  constexpr int A_PIECE = BM * ROWB, B_PIECE = BN * ROWB;
  constexpr int STAGE = P * (A_PIECE + B_PIECE);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
  __shared__ long long rowoff[BM];

  const int tid = threadIdx.x;
  int bx = blockIdx.x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = blockIdx.y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  const int nk = ntaps * sa.KB;

  if (tid < NPT) {
    // ================================================================== producer waves
    const int* __restrict__ tap_base = g.base[phase];
    const int* __restrict__ tap_w = g.widx[phase];
    const int akq = tid & 7;
    long long a_off[APASS];
    int a_j[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const long long m = m0 + (tid >> 3) + RPP * i;
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
    int b_lds[BPASS];
    long long b_goff[BPASS];
    bool b_in[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int idx = tid + NPT * i;
      const int piece = (idx / (BN * CH)) % P, rem = idx % (BN * CH);
      const int row = rem / CH, ch = rem % CH;
      b_in[i] = (B_EXACT || idx < B_CHUNKS) && n0 + row < g.N;
      b_goff[i] = (long long)piece * sa.w_piece_stride + (long long)(b_in[i] ? n0 + row : 0) * SBK + ch * 8;
      b_lds[i] = P * A_PIECE + piece * B_PIECE + row * ROWB + ((ch ^ swz(row)) << 4);
    }
    const long long kb_stride = (long long)g.N * SBK;
    const long long tap_stride = kb_stride * sa.KB;

    // D register sets: the tile staged in stage kt was requested D stages earlier, so a producer
    // wave (which has no MFMA phase to wait behind) never sits on a cold load
    f32x4 ra[D][APASS];
    u32x4 rb[D][BPASS];
    bool ra_ok[D][APASS];
    int nx_c0 = 0, nx_ti = 0;
    int nx_tb = ntaps > 0 ? tap_base[0] : 0;
    long long nx_woff = ntaps > 0 ? (long long)tap_w[0] * tap_stride : 0;
    auto load_tile = [&](auto slot_c) {
      constexpr int S = decltype(slot_c)::value;
      const int c0 = nx_c0;
      const int tb = nx_tb;
      const unsigned short* wt = sa.Wp + nx_woff + (long long)(c0 >> 5) * kb_stride;
      {
        nx_c0 += SBK;
        const bool wrap = nx_c0 >= g.Kc;
        nx_c0 = wrap ? 0 : nx_c0;
        nx_ti += wrap ? 1 : 0;
        const int tic = nx_ti < ntaps ? nx_ti : ntaps - 1;
        nx_tb = tap_base[tic];
        nx_woff = (long long)tap_w[tic] * tap_stride;
      }
      const bool kq_ok = c0 + akq * 4 < g.Kc;
      const int cq = kq_ok ? c0 + akq * 4 : 0;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int li = a_j[i] + tb;
        const bool ok = li >= 0 && li < g.Lin;
        const int lic = ok ? li : 0;
        if constexpr (!(DBG & 1)) ra[S][i] = *reinterpret_cast<const f32x4*>(g.A + a_off[i] + (long long)lic * g.ldA + cq);
        ra_ok[S][i] = ok && kq_ok;
      }
#pragma unroll
      for (int i = 0; i < BPASS; ++i)
        if constexpr (!(DBG & 1)) rb[S][i] = *reinterpret_cast<const u32x4*>(wt + b_goff[i]);
    };
    auto store_tile = [&](int buf, auto slot_c) {
      constexpr int S = decltype(slot_c)::value;
      unsigned char* st = smem + buf * STAGE;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const u32x4 v = b_in[i] ? rb[S][i] : u32x4{0u, 0u, 0u, 0u};
        if constexpr (B_EXACT) {
          *reinterpret_cast<u32x4*>(st + b_lds[i]) = v;
        } else {
          if (tid + NPT * i < B_CHUNKS) *reinterpret_cast<u32x4*>(st + b_lds[i]) = v;
        }
      }
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int row = (tid >> 3) + RPP * i;
        uint2 pc[P];
        const f32x4 a = ra[S][i];
        if constexpr (DBG & 2) {
#pragma unroll
          for (int p = 0; p < P; ++p) pc[p] = make_uint2(f2u(a.x) + p, f2u(a.y));
        } else {
          split4x<P, H>(ra_ok[S][i] ? make_float4(a.x, a.y, a.z, a.w) : make_float4(0.f, 0.f, 0.f, 0.f), pc);
        }
        const int off = row * ROWB + (((akq >> 1) ^ swz(row)) << 4) + ((akq & 1) << 3);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + p * A_PIECE + off) = pc[p];
      }
    };
    // tile t lives in register set t % D.  Loads past the last tile are harmless re-reads of the
    // last tap (clamped addresses) that are never staged.
    if (nk > 0) {
      load_tile(std::integral_constant<int, 0>{});
      if constexpr (D > 1) load_tile(std::integral_constant<int, 1 % D>{});
      if constexpr (D > 2) load_tile(std::integral_constant<int, 2 % D>{});
      store_tile(0, std::integral_constant<int, 0>{});
      load_tile(std::integral_constant<int, 0>{});  // tile D
    }
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    // stage kt stages tile kt+1 (set (kt+1) % D) and requests tile kt+1+D into the same set
    // unconditional on purpose: a conditional load makes hipcc fall back to vmcnt(0) waits, which
    // collapses the prefetch distance.  Past the end the loads re-read the last tap (clamped, in
    // bounds) and the store fills the LDS buffer nobody reads any more.
    auto stage = [&](int kt, auto slot_c) {
      store_tile((kt & 1) ^ 1, slot_c);
      load_tile(slot_c);
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);  // hipcc otherwise hoists the NEXT stage's conversion above this barrier -> vmcnt(0)
    };
    int kt = 0;
    for (; kt + D <= nk; kt += D) {
      stage(kt, std::integral_constant<int, 1 % D>{});
      if constexpr (D > 1) stage(kt + 1, std::integral_constant<int, 2 % D>{});
      if constexpr (D > 2) stage(kt + 2, std::integral_constant<int, 3 % D>{});
    }
    if (kt < nk) { stage(kt, std::integral_constant<int, 1 % D>{}); ++kt; }
    if constexpr (D > 2) { if (kt < nk) { stage(kt, std::integral_constant<int, 2 % D>{}); ++kt; } }
    return;
  }

  // ==================================================================== consumer waves
  const int ctid = tid - NPT;
  const int wave = ctid >> 6, lane = ctid & 63;
  const int wr = wave / CWC, wc = wave % CWC;
  const int lr = lane & 31, h = lane >> 5;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int a_addr[MT], b_addr[NT], a_sw[MT], b_sw[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = wr * WM + mt * 32 + lr;
    a_addr[mt] = row * ROWB;
    a_sw[mt] = swz(row);
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wc * WN + nt * 32 + lr;
    b_addr[nt] = P * A_PIECE + row * ROWB;
    b_sw[nt] = swz(row);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const unsigned char* st = smem + (kt & 1) * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 av[MT][P], bv[NT][P];
      const int ch = ks * 2 + h;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (DBG & 8) av[mt][p] = make_uint4(kt, ks, mt, p);
          else av[mt][p] = *reinterpret_cast<const uint4*>(st + p * A_PIECE + a_addr[mt] + ((ch ^ a_sw[mt]) << 4));
        }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (DBG & 8) bv[nt][p] = make_uint4(kt, ks, nt, p);
          else bv[nt][p] = *reinterpret_cast<const uint4*>(st + p * B_PIECE + b_addr[nt] + ((ch ^ b_sw[nt]) << 4));
        }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (DBG & 4) {
#pragma unroll
            for (int p = 0; p < P; ++p) acc[mt][nt][p] += __builtin_bit_cast(float, av[mt][p].x ^ bv[nt][p].y);
          } else {
            acc[mt][nt] = mfma_splitx<P, H>(av[mt], bv[nt], acc[mt][nt]);
          }
        }
    }
    __syncthreads();
  }

  tile_epilogue<MT, NT, WM, WN, CWR, BN>(g, acc, rowoff, n0, wr, wc, lr, h, reinterpret_cast<float*>(smem), ctid, 64 * CWR * CWC, H ? F16_OSCALE : 1.f);
}

// ----------------------------------------------------- gather GEMM with a halo image (fwd / dgrad)
// The implicit-GEMM kernels above re-stage (load, split, write) the A tile for every tap although
// consecutive taps read the SAME input rows shifted by one: a k-tap convolution converts every
// activation k times.  Here one LDS image per 32-channel block holds all input rows a workgroup's
// BM output rows touch over all taps (the tile's rows plus a halo: BM*stride + k - 1 rows); each
// tap then reads that image with a row offset, and only the weight tile is staged per tap.  Rows
// that a tap reaches across a sample boundary (conv padding) hold the neighbouring sample's data
// and are zeroed in registers by the reader.  A-side global loads, split VALU and LDS writes drop
// by ~k.  K loop order: channel block outer, tap inner.
template <int BM, int BN, int P, int WR, int WC, int RMAX, bool H = false>
__global__ __launch_bounds__(64 * WR * WC) void gather_halo_bf16s_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int NTH = 64 * WR * WC;
  constexpr int WM = BM / WR, MT = WM / 32, WN = BN / WC, NT = WN / 32;
  static_assert(MT >= 1 && NT >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  constexpr int RPP = NTH / 8;
  constexpr int APASS = (RMAX + RPP - 1) / RPP;
  constexpr int CH = 4;
  constexpr int B_CHUNKS = P * BN * CH;
  constexpr int BPASS = (B_CHUNKS + NTH - 1) / NTH;
  constexpr bool B_EXACT = B_CHUNKS % NTH == 0;
  constexpr int ROWB = SBK * 2;
  constexpr int A_PIECE = RMAX * ROWB, B_PIECE = BN * ROWB;
  constexpr int A_IMG = P * A_PIECE, B_STAGE = P * B_PIECE;
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_IMG + 2 * B_STAGE];
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

  // ---- extent of the image (uniform): anchor a(m) = b*Lin + j*sj, image row 0 = a(m0) + min base
  int bmin = 1 << 30, bmax = -(1 << 30);
  for (int t = 0; t < ntaps; ++t) {
    const int b = tap_base[t];
    bmin = b < bmin ? b : bmin;
    bmax = b > bmax ? b : bmax;
  }
  const long long b0 = m0 / nj;
  const long long amin = b0 * g.Lin + (long long)(m0 - b0 * nj) * g.sj;
  const long long ml = (m0 + BM < Mp ? m0 + BM : Mp) - 1;
  const long long bl = ml / nj;
  const long long amax = bl * g.Lin + (long long)(ml - bl * nj) * g.sj;
  const int R = (int)(amax - amin) + bmax - bmin + 1;  // <= RMAX (checked on the host)
  const long long gbase = amin + bmin;

  // ---- A image staging: 8 lanes per row, RPP rows per pass
  const int akq = tid & 7;
  long long a_goff[APASS];
  bool a_row_ok[APASS];
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = (tid >> 3) + RPP * i;
    const long long grow = gbase + r;
    a_row_ok[i] = r < R && grow >= 0 && grow < sa.rowsA;
    a_goff[i] = (a_row_ok[i] ? grow : 0) * (long long)g.ldA;
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
  int b_lds[BPASS];
  long long b_goff[BPASS];
  bool b_in[BPASS];
#pragma unroll
  for (int i = 0; i < BPASS; ++i) {
    const int idx = tid + NTH * i;
    const int piece = (idx / (BN * CH)) % P, rem = idx % (BN * CH);
    const int row = rem / CH, ch = rem % CH;
    b_in[i] = (B_EXACT || idx < B_CHUNKS) && n0 + row < g.N;
    b_goff[i] = (long long)piece * sa.w_piece_stride + (long long)(b_in[i] ? n0 + row : 0) * SBK + ch * 8;
    b_lds[i] = A_IMG + piece * B_PIECE + row * ROWB + ((ch ^ swz(row)) << 4);
  }
  const long long kb_stride = (long long)g.N * SBK;
  const long long tap_stride = kb_stride * sa.KB;
  const int ns = ntaps * sa.KB;

  float4 ra[APASS];
  bool ra_ok[APASS];
  uint4 rb[BPASS];
  auto load_a = [&](int kb) {
    const int c0 = kb * SBK;
    const bool kq_ok = c0 + akq * 4 < g.Kc;
    const int cq = kq_ok ? c0 + akq * 4 : 0;
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(g.A + a_goff[i] + cq);
      ra_ok[i] = a_row_ok[i] && kq_ok;
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int r = (tid >> 3) + RPP * i;
      uint2 pc[P];
      split4x<P, H>(ra_ok[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f), pc);
      const int off = r * ROWB + (((akq >> 1) ^ swz(r)) << 4) + ((akq & 1) << 3);
      if ((i + 1) * RPP <= RMAX || r < RMAX) {
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(smem + p * A_PIECE + off) = pc[p];
      }
    }
  };
  int nx_tap = 0, nx_kb = 0;
  auto load_b = [&]() {
    const int kbc = nx_kb < sa.KB ? nx_kb : sa.KB - 1;  // loads past the last stage re-read the last block
    const unsigned short* wt = sa.Wp + (long long)tap_w[nx_tap] * tap_stride + (long long)kbc * kb_stride;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) rb[i] = *reinterpret_cast<const uint4*>(wt + b_goff[i]);
    ++nx_tap;
    const bool wrap = nx_tap >= ntaps;
    nx_tap = wrap ? 0 : nx_tap;
    nx_kb += wrap ? 1 : 0;
  };
  auto store_b = [&](int buf) {
    unsigned char* st = smem + buf * B_STAGE;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const uint4 v = b_in[i] ? rb[i] : make_uint4(0u, 0u, 0u, 0u);
      if constexpr (B_EXACT) {
        *reinterpret_cast<uint4*>(st + b_lds[i]) = v;
      } else {
        if (tid + NTH * i < B_CHUNKS) *reinterpret_cast<uint4*>(st + b_lds[i]) = v;
      }
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;
  int ro[MT], jj[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = m0 + wr * WM + mt * 32 + lr;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      ro[mt] = (int)(b * g.Lin + (long long)j * g.sj - amin) - bmin;
      jj[mt] = j * g.sj;
    } else {
      ro[mt] = -bmin;
      jj[mt] = -(1 << 28);
    }
  }
  int b_addr[NT], b_sw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wc * WN + nt * 32 + lr;
    b_addr[nt] = A_IMG + row * ROWB;
    b_sw[nt] = swz(row);
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf, int tap) {
    const int tb = tap_base[tap];
    const unsigned char* bst = smem + buf * B_STAGE;
    int arow[MT];
    bool aval[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      arow[mt] = ro[mt] + tb;  // image row = a(m) - amin + (base - bmin)
      aval[mt] = (unsigned)(jj[mt] + tb) < (unsigned)g.Lin;
    }
    // operand fragments of BOTH 16-deep k-steps are requested up front (one exposed LDS latency per stage, then 2 x
    // MT x NT x 6 MFMAs back to back); padding rows are zeroed with an AND mask (a select on the loaded value would be
    // turned into a divergent branch around the ds_read)
    uint4 av[2][MT][P], bv[2][NT][P];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ch = ks * 2 + h;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned mask = aval[mt] ? 0xffffffffu : 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          uint4 v = *reinterpret_cast<const uint4*>(smem + p * A_PIECE + arow[mt] * ROWB + ((ch ^ swz(arow[mt])) << 4));
          v.x &= mask; v.y &= mask; v.z &= mask; v.w &= mask;
          av[ks][mt][p] = v;
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p)
          bv[ks][nt][p] = *reinterpret_cast<const uint4*>(bst + p * B_PIECE + b_addr[nt] + ((ch ^ b_sw[nt]) << 4));
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_splitx<P, H>(av[ks][mt], bv[ks][nt], acc[mt][nt]);
  };

  if (ns > 0) {
    load_a(0);
    load_b();
    store_b(0);
    load_b();
  }
  int tap = 0, kb = 0;
  for (int s = 0; s < ns; ++s) {
    if (tap == 0) {  // new channel block: publish its image (the barrier also publishes weight stage s)
      store_a();
      __syncthreads();
      load_a(kb + 1 < sa.KB ? kb + 1 : kb);
    }
    store_b((s & 1) ^ 1);
    load_b();
    compute(s & 1, tap);
    __syncthreads();
    if (++tap == ntaps) { tap = 0; ++kb; }
  }
  if (ns == 0) __syncthreads();  // rowoff

  tile_epilogue<MT, NT, WM, WN, WR, BN>(g, acc, rowoff, n0, wr, wc, lr, h, reinterpret_cast<float*>(smem), tid, NTH, H ? F16_OSCALE : 1.f);
}

// ------------------------------------------- halo-image gather GEMM, wave-specialised (fwd / dgrad)
// In gather_halo_bf16s_kernel every wave stages AND computes: hipcc sinks the weight-tile loads of the next stage to the end of
// the current one and waits for them (vmcnt(0)) at the top of the next, so every stage exposes one L2 round trip plus the LDS
// writes on all eight waves at once -- the matrix cores idle for ~2,000 of a stage's ~5,500 cycles (rocprof: 0.43-0.55 busy).
// Here the roles are split: WR x WC CONSUMER waves issue nothing but operand fetches (ds_read_b128) and MFMAs; two PRODUCER waves
// (dispatched last: they land on SIMDs 0 and 2 beside two consumers each) feed them:
//   * the pre-split weight tile of the next stage goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip,
//     no ds_write; the XOR swizzle of the image is applied to the per-lane SOURCE address), double-buffered;
//   * the activation image of the NEXT 32-channel block is loaded a block ahead into producer registers, split into bf16
//     pieces and written to the second image buffer while the consumers work on the current one (two image buffers).
// One barrier per stage, shared by both roles; the consumers never wait for global memory.
#ifdef SVAE_ABLATION_KERNELS
// DBG & 16 (diagnostic build only): s_memtime stamps of the stage loop of workgroup (STAMP_BLOCK, 0), written to a buffer of their
// own ([wave][stage < 64][8] ticks) that nothing else reads; the waits for the stamps' scalar loads are deferred to the end of the
// stage so that no stamp drains the operand fetches in flight.
__device__ unsigned long long* g_stamp_buf = nullptr;
#define SVAE_STAMP(i) do { if constexpr ((DBG & 16) != 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(stamp_t[i])); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define SVAE_STAMP_FLUSH(wave_, s_, n_) do { if constexpr ((DBG & 16) != 0) {                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stamp_t[0]), "+s"(stamp_t[1]), "+s"(stamp_t[2]), "+s"(stamp_t[3]), "+s"(stamp_t[4]), "+s"(stamp_t[5])); \
    if (g_stamp_buf && blockIdx.x == STAMP_BLOCK && blockIdx.y == 0 && (s_) < 64 && (threadIdx.x & 63) == 0)                                          \
      for (int i_ = 0; i_ < (n_); ++i_) g_stamp_buf[((wave_) * 64 + (s_)) * 8 + i_] = stamp_t[i_];                                                     \
  } } while (0)
constexpr int STAMP_BLOCK = 37;
// the 12-wave kernel logs to LDS (low 32 bits, stages 8..23) and dumps the log once after the stage loop: a per-stage flush to
// global memory costs ~750 cycles and, in the loader waves, queues behind the DMA requests
#define SVAE_LFLUSH(wave_, s_) do { if constexpr ((DBG & 16) != 0) {                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(stamp_t[0]), "+s"(stamp_t[1]), "+s"(stamp_t[2]), "+s"(stamp_t[3]), "+s"(stamp_t[4]), "+s"(stamp_t[5])); \
    if ((s_) >= 8 && (s_) < 24 && (threadIdx.x & 63) == 0)                                                                                            \
      for (int i_ = 0; i_ < 6; ++i_) slog[((wave_) * 16 + (s_) - 8) * 6 + i_] = (unsigned)stamp_t[i_];                                                \
  } } while (0)
#else
#define SVAE_STAMP(i) do { } while (0)
#define SVAE_STAMP_FLUSH(wave_, s_, n_) do { } while (0)
#define SVAE_LFLUSH(wave_, s_) do { } while (0)
#endif
__device__ __attribute__((aligned(16))) unsigned short halo_zero_chunk[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // source of weight rows past N

// DBG (timing experiments only, wrong results; python scrubvae_amd/build.py --ablation): 1 = producers issue no global loads / DMA,
// 2 = consumers issue no LDS operand fetches, 4 = consumers issue no MFMAs, 8 = no per-stage barrier in the consumers' loop
// PIPE: pin the fetch / multiply blocks of the consumers' software pipeline with sched_barrier (hipcc otherwise sinks every operand
// fetch down to its first use and the ds_read latency is exposed again).  Costs registers: both k-steps' fragments stay live, which
// fits the 168 VGPRs of a 10-wave workgroup with 2 pieces (64 x 64 wave tile) or with 3 pieces on a 64 x 32 wave tile (BN = 64).
// NB: weight-tile buffers.  2: the tile of stage s + 1 is requested in interval s and must have landed by its end -- with 2 pieces
// an interval's matrix work (~0.7 us) is shorter than the DMA's L2 round trip (~1 us), the producers become the critical path.
// 3: the tile of stage s + 2 is requested in interval s; the producers wait with a COUNTED vmcnt that leaves this interval's
// requests in flight (bare s_barrier + explicit waits: __syncthreads() would drain them all).
template <int BM, int BN, int P, int WR, int WC, int RMAX, int DBG = 0, bool PIPE = false, int NB = 2, bool H = false>
__global__ __launch_bounds__(64 * (WR * WC + 2)) void gather_halo_ws_bf16s_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int NCT = 64 * WR * WC;   // consumer threads
  constexpr int NPT = 128;            // producer threads (2 waves)
  constexpr int WM = BM / WR, MT = WM / 32, WN = BN / WC, NT = WN / 32;
  static_assert(MT >= 1 && NT >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(BM <= NCT, "rowoff is filled by the consumer threads");
  constexpr int ROWB = SBK * 2;
  constexpr int A_PIECE = RMAX * ROWB, B_PIECE = BN * ROWB;
  constexpr int A_IMG = P * A_PIECE, B_STAGE = P * B_PIECE;
  static_assert(2 * A_IMG + NB * B_STAGE + BM * 8 <= 160 * 1024, "LDS budget");
  static_assert(NB == 2 || NB == 3, "2 or 3 weight buffers");
  constexpr int RPP = NPT / 8;                     // image rows per producer pass (8 lanes x float4 per row)
  constexpr int APASS = (RMAX + RPP - 1) / RPP;
  constexpr int B_INSTR = B_STAGE / 1024;          // 1-KiB DMA wave-instructions per weight stage
  static_assert(B_STAGE % 2048 == 0, "weight stage must split evenly over the two producer waves");
  constexpr int BPW = B_INSTR / 2;                 // per producer wave
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_IMG + NB * B_STAGE];
  __shared__ long long rowoff[BM];
  unsigned char* const bbase = smem + 2 * A_IMG;

  const int tid = threadIdx.x;
  int bx = blockIdx.x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = blockIdx.y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  // tap tables in closed form (gemm_common.h: plan_is_affine, checked on the host): no scalar loads inside the stage loop
  const int tb0 = g.base0[phase], tbs = g.bstep[phase], tw0 = g.w0[phase], tws = g.wstep[phase];
  const int tbl = tb0 + (ntaps - 1) * tbs;
  const int bmin = tb0 < tbl ? tb0 : tbl, bmax = tb0 < tbl ? tbl : tb0;
  const long long b0 = m0 / nj;
  const long long amin = b0 * g.Lin + (long long)(m0 - b0 * nj) * g.sj;
  const long long ml = (m0 + BM < Mp ? m0 + BM : Mp) - 1;
  const long long bl = ml / nj;
  const long long amax = bl * g.Lin + (long long)(ml - bl * nj) * g.sj;
  const int R = (int)(amax - amin) + bmax - bmin + 1;  // <= RMAX (checked on the host)
  const long long gbase = amin + bmin;
  const long long kb_stride = (long long)g.N * SBK;
  const long long tap_stride = kb_stride * sa.KB;
  const int ns = ntaps * sa.KB;

  if (tid >= NCT) {
    // ================================================================== producer waves
    const int ptid = tid - NCT;
    const int pw = __builtin_amdgcn_readfirstlane(ptid >> 6), lane = ptid & 63;
    [[maybe_unused]] unsigned long long stamp_t[6] = {0, 0, 0, 0, 0, 0};
    // weight DMA: wave-instruction i of this wave fills LDS bytes [(2 i + pw) KiB, +1 KiB) of the stage: chunk q = 64 (2 i + pw) + lane
    long long b_src[BPW];
    bool b_ok[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const int q = 64 * (2 * i + pw) + lane;
      const int piece = q / (BN * 4), rem = q - piece * (BN * 4);
      const int row = rem >> 2, slot = rem & 3;
      b_ok[i] = n0 + row < g.N;
      b_src[i] = (long long)piece * sa.w_piece_stride + (long long)(n0 + row) * SBK + ((slot ^ swz(row)) << 3);
    }
    auto dma_b = [&](int tap, int kb, int buf) {
      const unsigned short* wt = sa.Wp + (long long)(tw0 + tap * tws) * tap_stride + (long long)kb * kb_stride;
#pragma unroll
      for (int i = 0; i < BPW; ++i) {
        const unsigned short* src = b_ok[i] ? wt + b_src[i] : halo_zero_chunk;
        if constexpr (!(DBG & 1))
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(bbase + buf * B_STAGE + (2 * i + pw) * 1024), 16, 0, 0);
      }
    };
    // activation image: 8 lanes per row, RPP rows per pass
    const int akq = ptid & 7;
    long long a_goff[APASS];
    bool a_row_ok[APASS];
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int r = (ptid >> 3) + RPP * i;
      const long long grow = gbase + r;
      a_row_ok[i] = r < R && grow >= 0 && grow < sa.rowsA;
      a_goff[i] = (a_row_ok[i] ? grow : 0) * (long long)g.ldA;
    }
    float4 ra[APASS];
    bool ra_kq;
    auto load_a = [&](int kb) {
      const int c0 = kb * SBK;
      ra_kq = c0 + akq * 4 < g.Kc;
      const int cq = ra_kq ? c0 + akq * 4 : 0;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        if constexpr (DBG & 1) ra[i] = make_float4(1.f + i, 2.f, 3.f + cq, 4.f);
        else ra[i] = *reinterpret_cast<const float4*>(g.A + a_goff[i] + cq);
      }
    };
    auto store_a = [&](int buf) {
      unsigned char* img = smem + buf * A_IMG;
#pragma unroll
      for (int i = 0; i < APASS; ++i) {
        const int r = (ptid >> 3) + RPP * i;
        uint2 pc[P];
        split4x<P, H>((a_row_ok[i] && ra_kq) ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f), pc);
        const int off = r * ROWB + (((akq >> 1) ^ swz(r)) << 4) + ((akq & 1) << 3);
        if ((i + 1) * RPP <= RMAX || r < RMAX) {
#pragma unroll
          for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(img + p * A_PIECE + off) = pc[p];
        }
      }
    };
    if constexpr (NB == 2) {
      if (ns > 0) {
        load_a(0);
        dma_b(0, 0, 0);
        store_a(0);
        load_a(sa.KB > 1 ? 1 : 0);
      }
      __syncthreads();  // (hipcc waits for the DMA in flight here)
      int tap = 0, kb = 0;
      for (int s = 0; s < ns; ++s) {
        SVAE_STAMP(0);
        // stage s + 1's weight tile
        int ntap = tap + 1, nkb = kb;
        if (ntap == ntaps) { ntap = 0; ++nkb; }
        if (s + 1 < ns) dma_b(ntap, nkb, (s + 1) & 1);
        SVAE_STAMP(1);
        if (tap == 0 && kb + 1 < sa.KB) {  // the next channel block's image, into the buffer the consumers left one block ago
          store_a((kb + 1) & 1);
          SVAE_STAMP(2);
          load_a(kb + 2 < sa.KB ? kb + 2 : kb + 1);
        } else SVAE_STAMP(2);
        SVAE_STAMP(3);
        if constexpr ((DBG & 16) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SVAE_STAMP(4);
        __syncthreads();
        SVAE_STAMP(5);
        SVAE_STAMP_FLUSH(WR * WC + pw, s, 6);
        tap = ntap;
        kb = nkb;
      }
    } else {
      // (tap, block) of stages s + 1 and s + 2
      int t1 = 0, k1 = 0;
      auto advance = [&](int& t, int& k) { if (++t == ntaps) { t = 0; ++k; } };
      if (ns > 0) {
        load_a(0);
        dma_b(0, 0, 0);
        advance(t1, k1);
        if (ns > 1) dma_b(t1, k1, 1);
        store_a(0);
        load_a(sa.KB > 1 ? 1 : 0);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      int tap = 0, kb = 0;
      int t2 = t1, k2 = k1;
      advance(t2, k2);
      for (int s = 0; s < ns; ++s) {
        const bool do_dma = s + 2 < ns;
        const bool do_img = tap == 0 && kb + 1 < sa.KB;
        SVAE_STAMP(0);
        if (do_img) store_a((kb + 1) & 1);  // (its loads were requested a whole channel block ago)
        SVAE_STAMP(1);
        if (do_dma) dma_b(t2, k2, (s + 2) % 3);
        SVAE_STAMP(2);
        if (do_img) load_a(kb + 2 < sa.KB ? kb + 2 : kb + 1);
        SVAE_STAMP(3);
        // everything older than THIS interval's requests has landed: stage s + 1's weight tile (requested one interval ago) is complete
        if (do_dma && do_img) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BPW + APASS) : "memory");
        else if (do_dma) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BPW) : "memory");
        else if (do_img) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(APASS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        SVAE_STAMP(4);
        __builtin_amdgcn_s_barrier();
        SVAE_STAMP(5);
        SVAE_STAMP_FLUSH(WR * WC + pw, s, 6);
        advance(tap, kb);
        advance(t2, k2);
      }
    }
    return;
  }

  // ==================================================================== consumer waves
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
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;
  int ro[MT], jj[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = m0 + wr * WM + mt * 32 + lr;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      ro[mt] = (int)(b * g.Lin + (long long)j * g.sj - amin) - bmin;
      jj[mt] = j * g.sj;
    } else {
      ro[mt] = -bmin;
      jj[mt] = -(1 << 28);
    }
  }
  int b_addr[NT], b_sw[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wc * WN + nt * 32 + lr;
    b_addr[nt] = row * ROWB;
    b_sw[nt] = swz(row);
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  __syncthreads();
  {
    uint4 av[2][MT][P], bv[2][NT][P];
    [[maybe_unused]] unsigned long long stamp_t[6] = {0, 0, 0, 0, 0, 0};
    int tap = 0, kb = 0;
    // stage whose operands are being fetched: tap offset, validity masks, image / weight-stage base
    int tb = 0;
    const unsigned char* img = smem;
    const unsigned char* bst = bbase;
    auto begin_stage = [&](int s) {
      tb = tb0 + tap * tbs;
      img = smem + (kb & 1) * A_IMG;
      bst = bbase + (NB == 2 ? (s & 1) : (s % 3)) * B_STAGE;
      if (++tap == ntaps) { tap = 0; ++kb; }
    };
    // operand fragments of 16-deep k-step ks of the current stage (raw: the padding-row mask is applied by the consumer of the
    // registers, so nothing in here waits for a load)
    unsigned amask[2][MT];
    auto fetch = [&](auto ks_c, int s) {
      constexpr int ks = decltype(ks_c)::value;
      const int ch = ks * 2 + h;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int arow = ro[mt] + tb;
        amask[ks][mt] = (unsigned)(jj[mt] + tb) < (unsigned)g.Lin ? 0xffffffffu : 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (DBG & 2) av[ks][mt][p] = make_uint4(s + arow, ks, mt, p);
          else av[ks][mt][p] = *reinterpret_cast<const uint4*>(img + p * A_PIECE + arow * ROWB + ((ch ^ swz(arow)) << 4));
        }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if constexpr (DBG & 2) bv[ks][nt][p] = make_uint4(s, ks + b_addr[nt], nt, p);
          else bv[ks][nt][p] = *reinterpret_cast<const uint4*>(bst + p * B_PIECE + b_addr[nt] + ((ch ^ b_sw[nt]) << 4));
        }
    };
    auto mma = [&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned mask = amask[ks][mt];
#pragma unroll
        for (int p = 0; p < P; ++p) { av[ks][mt][p].x &= mask; av[ks][mt][p].y &= mask; av[ks][mt][p].z &= mask; av[ks][mt][p].w &= mask; }
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          if constexpr (DBG & 4) {
#pragma unroll
            for (int p = 0; p < P; ++p) acc[mt][nt][p] += __builtin_bit_cast(float, av[ks][mt][p].x ^ bv[ks][nt][p].y);
          } else {
            acc[mt][nt] = mfma_splitx<P, H>(av[ks][mt], bv[ks][nt], acc[mt][nt]);
          }
        }
    };
    // Software pipeline over half stages: the operand fetches of one 16-deep k-step are issued in front of the MFMAs of the
    // previous one, so every ds_read has a k-step's worth of matrix work (MT x NT x products MFMAs) to land behind:
    //   interval s:  fetch k0(s) | mma k1(s-1) | fetch k1(s) | mma k0(s) | barrier
    // Both fetches of stage s fall between barrier s-1 (which published its weight tile) and barrier s (after which the
    // producers may overwrite it), the same hand-over protocol as a fetch-everything-then-multiply loop.
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    for (int s = 0; s < ns; ++s) {
      SVAE_STAMP(0);
      begin_stage(s);
      fetch(K0{}, s);
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      SVAE_STAMP(1);
      if (s > 0) mma(K1{});
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      SVAE_STAMP(2);
      fetch(K1{}, s);
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      SVAE_STAMP(3);
      mma(K0{});
      if constexpr (PIPE) __builtin_amdgcn_sched_barrier(0);
      SVAE_STAMP(4);
      __syncthreads();
      SVAE_STAMP(5);
      SVAE_STAMP_FLUSH(wave, s, 6);
    }
    if (ns > 0) mma(K1{});
  }
  tile_epilogue<MT, NT, WM, WN, WR, BN>(g, acc, rowoff, n0, wr, wc, lr, h, reinterpret_cast<float*>(smem), tid, NCT, H ? F16_OSCALE : 1.f);
}

// ------------------------------------------- halo-image gather GEMM, 8 consumer + 4 loader waves (fwd / dgrad)
// In-kernel stamps of gather_halo_ws_bf16s_kernel (tools/stamp_halo.py) showed its two producer waves to be the critical
// path of every stage: 8 LDS-DMA wave-instructions cost a producer ~1,700 cycles beside the consumers' operand traffic (the matrix
// work of a stage is 1,536 cycles per SIMD), and every ntaps-th stage additionally carries the split of a whole activation image
// (~3,000 cycles); the consumers idle at the barrier for a third of the time.  Here
//   * FOUR loader waves (one per SIMD) issue nothing but LDS-DMA: a quarter of every weight tile each, and the fp32 rows of the
//     next channel block's activation image, a slice of ceil(rows / ntaps) rows per stage, into a small raw staging ring.  No wave
//     of this kernel has a VGPR-destination global load in its stage loop, so hipcc inserts no vmcnt(0) drains: every
//     vector-memory wait is a counted s_waitcnt vmcnt(N) -- "everything older than this interval's requests has landed";
//   * the CONSUMERS convert the raw slice that landed an interval ago (ds_read_b128 -> bf16 / fp16 pieces -> 2 ds_write_b64 per
//     thread and stage: the split of an image is spread over its ntaps stages and over 512 threads instead of stalling two waves);
//   * STAG: the consumers' second half (waves 4-7, the SIMD partners of waves 0-3) runs the stage loop rotated by a quarter stage --
//       waves 0-3:  fetch k0(s) | mma k1(s-1) | fetch k1(s) | mma k0(s)   | barrier
//       waves 4-7:  mma k0(s-1) | fetch k0(s) | mma k1(s-1) | fetch k1(s) | barrier
//     so on every SIMD one wave fetches operands while its partner multiplies (MI355X_MICROARCH: two waves per SIMD, item 9).
// Hand-over protocol (one barrier per stage, shared by all 12 waves):
//   weight tile of stage s+2: requested in interval s, landed by barrier s+1 (counted wait of interval s+1), read in interval s+2,
//     its buffer (3 of them) re-requested in interval s+1 at the earliest -- after barrier s, which every consumer reaches with its
//     operand reads of stage s complete (lgkmcnt(0));
//   raw slice `tap` of block kb+1: requested in interval s-2 into staging slot s%3, landed by barrier s-1, converted by the consumers
//     in interval s = (kb, tap) into image buffer (kb+1)&1, which the consumers left at the end of block kb-1; the conversion's LDS
//     writes are complete at barrier s (lgkmcnt(0)); slot s%3 is re-requested in interval s+1.
// Two bf16 or fp16 pieces only (three weight buffers + the staging ring do not fit beside three-piece images).
// s_barrier without the fence of __syncthreads() (which would drain the LDS-DMA in flight); the empty asm statements keep hipcc
// from moving LDS accesses across it
__device__ __forceinline__ void bare_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N, int MAXN>
__device__ __forceinline__ void wait_vmcnt_rt(int n) {  // s_waitcnt vmcnt(n) for a wave-uniform runtime n <= MAXN
  if (n == N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  else if constexpr (N < MAXN) wait_vmcnt_rt<N + 1, MAXN>(n);
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MAXN) : "memory");
}

__device__ __attribute__((aligned(16))) float halo_zero_f32[4] = {0.f, 0.f, 0.f, 0.f};  // source of image rows / channels out of range

template <int BM, int BN, int RMAX, int STG, bool H, bool STAG = true, int DBG = 0>
__global__ __launch_bounds__(768) void gather_halo_ws4_bf16s_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int P = 2, NB = 3, WR = 4, WC = 2, NPW = 4;
  constexpr int NCT = 64 * WR * WC;   // consumer threads
  constexpr int WM = BM / WR, MT = WM / 32, WN = BN / WC, NT = WN / 32;
  static_assert(MT >= 1 && NT >= 1 && WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(BM <= NCT, "rowoff is filled by the consumer threads");
  static_assert(RMAX % 8 == 0 && STG % 8 == 0, "a DMA wave-instruction carries 8 image rows");
  constexpr int ROWB = SBK * 2;                    // bytes of an image / weight row per piece
  constexpr int RAWB = SBK * 4;                    // bytes of a raw fp32 row
  constexpr int A_PIECE = (RMAX + 1) * ROWB, B_PIECE = BN * ROWB;  // image row RMAX: zeros, the target of operand reads that hit conv padding
  constexpr int ZROW = RMAX * ROWB;
  constexpr int A_IMG = P * A_PIECE, B_STAGE = P * B_PIECE;
  constexpr int STG_BYTES = STG * RAWB;
  static_assert(RMAX * RAWB <= A_IMG, "the prologue stages the raw image of block 0 in image buffer 1");
  static_assert(2 * A_IMG + NB * B_STAGE + 3 * STG_BYTES + BM * 8 <= 160 * 1024, "LDS budget");
  constexpr int B_INSTR = B_STAGE / 1024;          // 1-KiB DMA wave-instructions per weight stage
  static_assert(B_STAGE % (1024 * NPW) == 0, "weight stage must split evenly over the loader waves");
  constexpr int BPW = B_INSTR / NPW;               // per loader wave
  constexpr int SPW = (STG / 8 + NPW - 1) / NPW;   // raw-slice wave-instructions per loader wave, at most
  constexpr int CPT = (STG * 8 + NCT - 1) / NCT;   // float4 conversions per consumer thread and stage, at most
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_IMG + NB * B_STAGE + 3 * STG_BYTES];
  __shared__ long long rowoff[BM];
  [[maybe_unused]] __shared__ unsigned slog[(DBG & 16) ? 12 * 16 * 6 : 1];
  unsigned char* const bbase = smem + 2 * A_IMG;
  unsigned char* const sbase = bbase + NB * B_STAGE;

  const int tid = threadIdx.x;
  // (An XCD-aware tile order -- every XCD a contiguous run of the column-major tile list, so that its workgroups stream the same
  //  weight tiles -- was measured 2-8 % SLOWER: each XCD then reads every activation image once per column tile from beyond its
  //  L2, where the launch order already gives an XCD 8 row tiles x 4 column tiles at a time, each line shared by 4-8 workgroups.)
  const int tile_x = blockIdx.x, tile_y = blockIdx.y;
  int bx = tile_x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = tile_y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  const int tb0 = g.base0[phase], tbs = g.bstep[phase], tw0 = g.w0[phase], tws = g.wstep[phase];
  const int tbl = tb0 + (ntaps - 1) * tbs;
  const int bmin = tb0 < tbl ? tb0 : tbl, bmax = tb0 < tbl ? tbl : tb0;
  const long long b0 = m0 / nj;
  const long long amin = b0 * g.Lin + (long long)(m0 - b0 * nj) * g.sj;
  const long long ml = (m0 + BM < Mp ? m0 + BM : Mp) - 1;
  const long long bl = ml / nj;
  const long long amax = bl * g.Lin + (long long)(ml - bl * nj) * g.sj;
  const int R = (int)(amax - amin) + bmax - bmin + 1;  // <= RMAX (checked on the host)
  const long long gbase = amin + bmin;
  const long long kb_stride = (long long)g.N * SBK;
  const long long tap_stride = kb_stride * sa.KB;
  const int ns = ntaps * sa.KB;
  const int SR = (((RMAX + ntaps - 1) / ntaps) + 7) & ~7;  // image rows per slice (<= STG, checked on the host); ntaps slices cover RMAX

  if (tid >= NCT) {
    // ================================================================== loader waves (LDS-DMA only)
    const int ptid = tid - NCT;
    const int pw = __builtin_amdgcn_readfirstlane(ptid >> 6), lane = ptid & 63;
    [[maybe_unused]] unsigned long long stamp_t[6] = {0, 0, 0, 0, 0, 0};
    // weight DMA: wave-instruction i of this wave fills LDS bytes [(NPW i + pw) KiB, +1 KiB) of the stage: chunk q = 64 (NPW i + pw) + lane
    long long b_src[BPW];
    bool b_ok[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const int q = 64 * (NPW * i + pw) + lane;
      const int piece = q / (BN * 4), rem = q - piece * (BN * 4);
      const int row = rem >> 2, slot = rem & 3;
      b_ok[i] = n0 + row < g.N;
      b_src[i] = (long long)piece * sa.w_piece_stride + (long long)(n0 + row) * SBK + ((slot ^ swz(row)) << 3);
    }
    auto dma_b = [&](int tap, int kb, int buf) {
      const unsigned short* wt = sa.Wp + (long long)(tw0 + tap * tws) * tap_stride + (long long)kb * kb_stride;
#pragma unroll
      for (int i = 0; i < BPW; ++i) {
        const unsigned short* src = b_ok[i] ? wt + b_src[i] : halo_zero_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(bbase + buf * B_STAGE + (NPW * i + pw) * 1024), 16, 0, 0);
      }
    };
    // raw activation rows: a wave-instruction carries 8 rows x 128 bytes (lane -> row lane / 8, channels 4 (lane % 8) ..+3), linear in LDS
    const int lrow = lane >> 3, lch = (lane & 7) * 4;
    const long long rlo_ll = -gbase, rhi_ll = sa.rowsA - gbase;
    const int row_lo = rlo_ll > 0 ? (int)(rlo_ll < RMAX ? rlo_ll : RMAX) : 0;          // image rows [row_lo, row_hi) exist in A
    const int row_hi = rhi_ll < R ? (int)(rhi_ll > 0 ? rhi_ll : 0) : R;
    const float* const a_lane = g.A + (gbase + lrow) * (long long)g.ldA + lch;
    // rows [r0, r0 + 8 n) of channel block kb -> dst (n wave-instructions split over the loader waves); returns this wave's count
    auto dma_rows = [&](int r0, int n, int kb, unsigned char* dst) -> int {
      const bool ch_ok = kb * SBK + lch < g.Kc;
      const float* base = a_lane + kb * SBK;
      int cnt = 0;
      for (int j = pw; j < n; j += NPW) {
        const int r = r0 + 8 * j + lrow;
        const bool ok = ch_ok && r >= row_lo && r < row_hi;
        const float* src = ok ? base + (long long)(r0 + 8 * j) * g.ldA : halo_zero_f32;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
        ++cnt;
      }
      return cnt;
    };
    // what stage c = (kq, tq) needs: its weight tile (buffer c % 3) and the raw slice the consumers convert during it -- slice tq of
    // block kq+1 (staging slot c % 3); returns this wave's number of wave-instructions
    int tq = 0, kq = 0;
    auto request = [&](int c) -> int {
      if (c >= ns) return 0;
      dma_b(tq, kq, c % NB);
      int n = BPW;
      if (kq + 1 < sa.KB) n += dma_rows(tq * SR, SR / 8, kq + 1, sbase + (c % 3) * STG_BYTES);
      if (++tq == ntaps) { tq = 0; ++kq; }
      return n;
    };
    // prologue: the raw image of block 0 whole (into image buffer 1), then the requests of stages 0 and 1
    if (ns > 0) dma_rows(0, RMAX / 8, 0, smem + A_IMG);
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bare_barrier();  // A: raw image of block 0 landed
    bare_barrier();  // B: the consumers converted it
    for (int s = 0; s < ns; ++s) {
      SVAE_STAMP(0);
      const int n = request(s + 2);
      SVAE_STAMP(1);
      // everything older than THIS interval's requests has landed: what stage s+1 needs
      wait_vmcnt_rt<0, BPW + SPW>(n);
      SVAE_STAMP(2);
      bare_barrier();
      SVAE_STAMP(3);
      SVAE_LFLUSH(WR * WC + pw, s);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ==================================================================== consumer waves
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
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;
  int ro[MT], jj[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const long long m = m0 + wr * WM + mt * 32 + lr;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      ro[mt] = (int)(b * g.Lin + (long long)j * g.sj - amin) - bmin;
      jj[mt] = j * g.sj;
    } else {
      ro[mt] = -bmin;
      jj[mt] = -(1 << 28);
    }
  }
  // LDS byte offsets (from smem) of this lane's weight fragments, k-step 0, weight buffer 0; k-step 1 = the same ^ 32 (the 16-byte
  // chunk index is 2 ks + h, XOR-swizzled by the row: flipping its bit 1 flips address bit 5)
  int b_off0[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int row = wc * WN + nt * 32 + lr;
    b_off0[nt] = 2 * A_IMG + row * ROWB + ((h ^ swz(row)) << 4);
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // raw fp32 rows [r0, r0 + nrows) at `raw` (128 bytes per row) -> piece planes of image `img`: item idx = tid + NCT i covers
  // channels 4 (idx % 8) ..+3 of row idx / 8
  const int cchunk = tid & 7;
  auto convert = [&](const unsigned char* raw, int r0, int nrows, unsigned char* img, int iters) {
    for (int i = 0; i < iters; ++i) {
      const int rl = (tid >> 3) + (NCT / 8) * i;
      const int r = r0 + rl;
      if (rl < nrows && r < RMAX) {
        const float4 v = *reinterpret_cast<const float4*>(raw + rl * RAWB + cchunk * 16);
        uint2 pc[P];
        split4x<P, H>(v, pc);
        const int off = r * ROWB + (((cchunk >> 1) ^ swz(r)) << 4) + ((cchunk & 1) << 3);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(img + p * A_PIECE + off) = pc[p];
      }
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // rowoff
  bare_barrier();  // A
  if (ns > 0) convert(smem + A_IMG, 0, RMAX, smem, (RMAX * 8 + NCT - 1) / NCT);
  if (tid < 4 * P) *reinterpret_cast<uint4*>(smem + (tid >> 2) * A_PIECE + ZROW + (tid & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bare_barrier();  // B
  // (the zero row of image buffer 1, where the raw image of block 0 sat until now: complete at the first stage barrier, read from block 1 on)
  if (tid < 4 * P) *reinterpret_cast<uint4*>(smem + A_IMG + (tid >> 2) * A_PIECE + ZROW + (tid & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
  {
    uint4 av[2][MT][P], bv[2][NT][P];
    [[maybe_unused]] unsigned long long stamp_t[6] = {0, 0, 0, 0, 0, 0};
    int tap = 0, kb = 0;
    int tb = 0;
    int a_off[MT], b_off[NT];
    // the slice converted in interval s = (kb, tap): slice `tap` of block kb+1, from staging slot s % 3.  Two steps so that the raw
    // read has a k-step of matrix work to land behind: raw_read(s) next to the operand fetches, raw_write() after the multiply
    float4 rawv[CPT];
    auto raw_read = [&](int s) {
      if (kb + 1 >= sa.KB) return;
      const unsigned char* raw = sbase + (s % 3) * STG_BYTES;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int rl = (tid >> 3) + (NCT / 8) * i;
        if (rl < SR) rawv[i] = *reinterpret_cast<const float4*>(raw + rl * RAWB + cchunk * 16);
      }
    };
    auto raw_write = [&]() {
      if (kb + 1 >= sa.KB) return;
      unsigned char* dst = smem + ((kb + 1) & 1) * A_IMG;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int rl = (tid >> 3) + (NCT / 8) * i;
        const int r = tap * SR + rl;
        if (rl < SR && r < RMAX) {
          uint2 pc[P];
          split4x<P, H>(rawv[i], pc);
          const int off = r * ROWB + (((cchunk >> 1) ^ swz(r)) << 4) + ((cchunk & 1) << 3);
#pragma unroll
          for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(dst + p * A_PIECE + off) = pc[p];
        }
      }
    };
    // operand addresses of stage s, k-step 0.  An output row whose tap falls into the conv padding reads the image's zero row
    // instead of being masked after the fetch (no VALU work on the fragments at all)
    auto begin_stage = [&](int s) {
      tb = tb0 + tap * tbs;
      const int img_off = (kb & 1) * A_IMG, bst_off = (s % NB) * B_STAGE;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int arow = ro[mt] + tb;
        const bool valid = (unsigned)(jj[mt] + tb) < (unsigned)g.Lin;
        a_off[mt] = img_off + (valid ? arow * ROWB + ((h ^ swz(arow)) << 4) : ZROW);
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b_off[nt] = b_off0[nt] + bst_off;
    };
    auto end_stage = [&]() { if (++tap == ntaps) { tap = 0; ++kb; } };
    auto fetch = [&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int p = 0; p < P; ++p) av[ks][mt][p] = *reinterpret_cast<const uint4*>(smem + (a_off[mt] ^ (ks * 32)) + p * A_PIECE);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) bv[ks][nt][p] = *reinterpret_cast<const uint4*>(smem + (b_off[nt] ^ (ks * 32)) + p * B_PIECE);
    };
    auto mma = [&](auto ks_c) {
      constexpr int ks = decltype(ks_c)::value;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_splitx<P, H>(av[ks][mt], bv[ks][nt], acc[mt][nt]);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
#ifdef SVAE_ABLATION_KERNELS
    [[maybe_unused]] unsigned long long clk_t0 = 0, clk_r0 = 0;
    if constexpr ((DBG & 16) != 0) {  // in-kernel clock: shader cycles per 100 MHz reference tick over the whole stage loop
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(clk_t0), "=s"(clk_r0));
    }
#endif
    if (!STAG || wave < 4) {
      for (int s = 0; s < ns; ++s) {
        SVAE_STAMP(0);
        begin_stage(s);
        fetch(K0{});
        raw_read(s);
        SVAE_STAMP(1);
        if (s > 0) mma(K1{});
        SVAE_STAMP(2);
        raw_write();
        fetch(K1{});
        SVAE_STAMP(3);
        mma(K0{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this stage's operand reads and conversion writes have left the LDS queue
        SVAE_STAMP(4);
        bare_barrier();
        SVAE_STAMP(5);
        SVAE_LFLUSH(wave, s);
        end_stage();
      }
      if (ns > 0) mma(K1{});
    } else {
      for (int s = 0; s < ns; ++s) {
        SVAE_STAMP(0);
        if (s > 0) mma(K0{});
        SVAE_STAMP(1);
        begin_stage(s);
        fetch(K0{});
        raw_read(s);
        SVAE_STAMP(2);
        if (s > 0) mma(K1{});
        SVAE_STAMP(3);
        raw_write();
        fetch(K1{});
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SVAE_STAMP(4);
        bare_barrier();
        SVAE_STAMP(5);
        SVAE_LFLUSH(wave, s);
        end_stage();
      }
      if (ns > 0) { mma(K0{}); mma(K1{}); }
    }
#ifdef SVAE_ABLATION_KERNELS
    if constexpr ((DBG & 16) != 0) {
      unsigned long long t1, r1;
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
      __syncthreads();  // (the loader waves have exited; their log entries were complete at their last barrier)
      if (g_stamp_buf && blockIdx.x == STAMP_BLOCK && blockIdx.y == 0) {
        for (int i = tid; i < 12 * 16 * 6; i += NCT) g_stamp_buf[8 + i] = slog[i];
        if (tid == 0) {
          g_stamp_buf[0] = t1 - clk_t0;
          g_stamp_buf[1] = r1 - clk_r0;
          g_stamp_buf[2] = (unsigned long long)ns;
        }
      }
    }
#endif
  }
  tile_epilogue<MT, NT, WM, WN, WR, BN>(g, acc, rowoff, n0, wr, wc, lr, h, reinterpret_cast<float*>(smem), tid, NCT, H ? F16_OSCALE : 1.f,
                                        tile_x, tile_y);
}

// ---------------------------------------------------------------------------- weight split
// w[tap][c_in][c_out] fp32 -> bf16 piece planes, pre-tiled in the order the GEMM stages them:
//   plane(n, k) at [piece][tap][k/32][n][k%32], k zero-padded to a multiple of 32.
//   Wf: n = c_out, k = c_in (forward)      Wd: n = c_in, k = c_out (data gradient)
// One workgroup converts a 32 (n) x 32 (k) block; Wf goes through an LDS transpose.
// outh != NULL (forward orientation): additionally the two fp16 piece planes of 2^10 * w, same tiling
__device__ __forceinline__ void split_block(const float* __restrict__ w, unsigned short* __restrict__ out, int T, int Cin, int Cout,
                                            int to_wd, long long piece_stride, int nb, int kb, int t, float (*tile)[33],
                                            unsigned short* __restrict__ outh = nullptr) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int N = to_wd ? Cin : Cout, K = to_wd ? Cout : Cin;
  const int KB = (K + 31) / 32;
  const int n0 = nb * 32, k0 = kb * 32;
  float v[4];
  if (to_wd) {  // source rows are n (= c_in), contiguous in k (= c_out)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + ty + 8 * i, k = k0 + tx;
      v[i] = (n < N && k < K) ? w[((long long)t * Cin + n) * Cout + k] : 0.f;
    }
  } else {      // source rows are k (= c_in), contiguous in n (= c_out): transpose
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + ty + 8 * i, n = n0 + tx;
      tile[ty + 8 * i][tx] = (n < N && k < K) ? w[((long long)t * Cin + k) * Cout + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = tile[tx][ty + 8 * i];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i;
    if (n >= N) continue;
    float x = v[i];
    const long long o = (((long long)t * KB + kb) * N + n) * 32 + tx;
    if (outh != nullptr) {
      const float xs = v[i] * F16_WSCALE;
      const _Float16 h0 = (_Float16)xs;            // round to nearest: the residual is exact in fp32
      const _Float16 h1 = (_Float16)(xs - (float)h0);
      outh[o] = __builtin_bit_cast(unsigned short, h0);
      outh[piece_stride + o] = __builtin_bit_cast(unsigned short, h1);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const unsigned u = p == 2 ? (pack_rne(x, 0.f) & 0xffffu) : (f2u(x) >> 16);
      out[p * piece_stride + o] = (unsigned short)u;
      x = residual(x);
    }
  }
}

__global__ __launch_bounds__(256) void split_weights_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                                            int T, int Cin, int Cout, int to_wd, long long piece_stride,
                                                            unsigned short* __restrict__ outh) {
  __shared__ float tile[32][33];
  split_block(w, out, T, Cin, Cout, to_wd, piece_stride, blockIdx.x, blockIdx.y, blockIdx.z, tile, outh);
}

// all convolutions of a model in ONE launch: block -> (task, direction, tap, k block, n block)
struct SplitTasks {
  const float* w[SVAE_MAX_SPLIT_TASKS];
  unsigned short* out[SVAE_MAX_SPLIT_TASKS];
  int T[SVAE_MAX_SPLIT_TASKS], Cin[SVAE_MAX_SPLIT_TASKS], Cout[SVAE_MAX_SPLIT_TASKS];
  int first[SVAE_MAX_SPLIT_TASKS + 1];  // first block of each task
  int n;
};

__global__ __launch_bounds__(256) void split_weights_batched_kernel(const SplitTasks ts) {
  __shared__ float tile[32][33];
  int k = 0;
  while (k + 1 < ts.n && (int)blockIdx.x >= ts.first[k + 1]) ++k;
  const int T = ts.T[k], Cin = ts.Cin[k], Cout = ts.Cout[k];
  const int bi = (Cin + 31) / 32, bo = (Cout + 31) / 32;
  int b = blockIdx.x - ts.first[k];
  const int per_dir = T * bi * bo;
  const int to_wd = b >= per_dir;
  b -= to_wd ? per_dir : 0;
  const int t = b / (bi * bo);
  b -= t * bi * bo;
  // Wf: n = c_out (bo blocks), k = c_in (bi blocks);  Wd: n = c_in, k = c_out
  const int nblocks = to_wd ? bi : bo;
  const int kb = b / nblocks, nb = b - kb * nblocks;
  const long long pf = (long long)T * bi * 32 * Cout, pd = (long long)T * bo * 32 * Cin;
  split_block(ts.w[k], ts.out[k] + (to_wd ? 3 * pf : 0), T, Cin, Cout, to_wd, to_wd ? pd : pf, nb, kb, t, tile,
              to_wd ? nullptr : ts.out[k] + 3 * pf + 3 * pd);
}

// ------------------------------------------------------------------------------ weight grad
// dW_t[c][n] = sum_r X[xrow(r,t)][c] * dY[yrow(r,t)][n]: the contraction runs over ROWS, so both
// operands are transposed on the way into LDS: a thread owns 4 reduction rows x 4 channels, splits
// them and writes, per channel, the 4 row-values as 8 bytes of that channel's K-contiguous LDS row
// (image [piece][channel][16 k + pad], 48-byte rows: conflict-free ds_read_b128 operand fetches).
constexpr int WSK = 16;
constexpr int WROWB = 48;
// rows outside the operand (conv padding, the tail past a split's last row) are read from here instead of being masked
// after the load: no per-value selects, no validity flags carried in registers
__device__ __attribute__((aligned(16))) float wgrad_zero_row[4] = {0.f, 0.f, 0.f, 0.f};  // not const: keeps the select in the global address space (global_load, not flat_load)

// NSTAGE = 2: double-buffered LDS, one barrier per stage;  NSTAGE = 1: one buffer, two barriers, half the LDS
// (more workgroups per CU overlap each other's conversion / LDS / MFMA phases)
// WR x WC waves (default 2 x 2 = 256 threads; the 256-row / 256-column tiles run 8 waves).  Both operands are fp32 in HBM
// and every 16-row stage loads (BM + BN) * 64 B through the CU's vector-memory path, the resource this kernel is bound by
// (DESIGN.md 4): a 256 x 256 tile moves half the bytes per FLOP of a 128 x 128 one, and the split over reduction rows
// (slabs) supplies the parallelism that the larger tile takes away.
template <int BM, int BN, int P, int NSTAGE, int WR = 2, int WC = 2>
__global__ __launch_bounds__(64 * WR * WC, (WR * WC > 4 ? 1 : 2)) void wgrad_gemm_bf16s_kernel(const WgradArgs g) {
  constexpr int WM = BM / WR, MT = WM / 32, WN = BN / WC, NT = WN / 32;
  static_assert(BM + BN <= 64 * WR * WC, "one staging unit (4 rows x 4 channels) per thread");
  // channel row c of an operand image starts at c * 48 + (c / 16) * 16 bytes: the extra 16 B per 16 channels keep the
  // staging writes of a 16-lane group (8 channel quads x 2 row groups, see below) on 32 different banks
  constexpr int A_PIECE = BM * WROWB + BM, B_PIECE = BN * WROWB + BN;
  constexpr int STAGE = P * (A_PIECE + B_PIECE);
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSTAGE * STAGE];
  auto lds_row = [](int c) { return c * WROWB + (c >> 4) * 16; };

  const int tid = threadIdx.x;
  // xmap: workgroups are dealt to the 8 XCDs round-robin in launch order and every XCD has an L2 of its own.  All (tap, tile) workgroups
  // of one split read the same reduction rows of X and dY; in launch order they are spread over all 8 XCDs, so every L2 fetches every
  // row from beyond it (measured 6-7x the operand bytes per launch).  Hand each XCD a contiguous run of the split-major workgroup
  // list instead: the workgroups resident on an XCD then share one or two splits' rows.  Bijective for any grid (q, r split).
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (g.xmap) {
    const int gxy = gridDim.x * gridDim.y, total = gxy * gridDim.z;
    const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bzi = w / gxy;
    const int rem = w - bzi * gxy;
    byi = rem / (int)gridDim.x;
    bxi = rem - byi * (int)gridDim.x;
  }
  const int ti = bxi / g.ctiles;
  const int c0 = (bxi - ti * g.ctiles) * BM;
  const int n0 = byi * BN;
  const long long r_begin = (long long)bzi * g.rows_per_split;
  long long r_end = r_begin + g.rows_per_split;
  if (r_end > g.R) r_end = g.R;
  // flat_np > 0 (transposed convs with many taps: the 22-tap output conv): the taps are FOLDED INTO THE COLUMNS of the dY operand --
  // column c' = t * flat_np + c_out of a virtual [rows][T * flat_np] matrix whose row j starts at dY row j + by[0]: the tile columns
  // then pad T * c_out to a multiple of BN once instead of c_out per tap (141 -> 256 per tap with BN = 128 is 44 % idle columns,
  // 22 * 144 -> 3200 is 1 %), X is staged once for 128 flat columns instead of once per tap, and a staging thread masks its rows with
  // the tap of ITS channel quad (flat_np is a multiple of 4: a quad never straddles two taps).
  // flat_mp > 0 is the mirror image for (non-transposed) convs, whose dY rows do not depend on the tap: the taps fold into the ROWS
  // of the weight tile -- channel row m' = t * flat_mp + c_in of a virtual [rows][T * flat_mp] view of X (conv_in: 7 taps x 144
  // channels = 1008 rows in 8 tiles of 128 instead of 7 x 2 half-empty ones), dY is staged once per 128 flat rows.
  const int np = g.flat_np, mp = g.flat_mp;
  const int tbx = g.bx[ti], tby = g.by[ti];

  // staging unit of this thread: operand (A = X channels, B = dY channels), row group rg, channel quad cq
  const bool is_a = tid < BM;
  const bool has_unit = tid < BM + BN;
  const int u = is_a ? tid : tid - BM;
  // 16 consecutive lanes = 8 channel quads x 2 row groups: each quarter-wave of a global_load_dwordx4 reads two whole
  // 128-B lines (8 quads x 16 B of two rows) instead of four half lines
  const int q16 = u >> 4, l16 = u & 15;
  const int nq8 = (is_a ? BM : BN) / 32;  // groups of 8 channel quads in the tile
  const int cq = (l16 & 7) + 8 * (q16 % nq8);
  const int rg = ((l16 >> 3) & 1) + 2 * (q16 / nq8);
  int ch0 = (is_a ? c0 : n0) + cq * 4;
  bool col_ok = has_unit && ch0 < (is_a ? g.Kc : g.N);
  int tb_unit = is_a ? tbx : tby;          // tap offset of this unit's operand rows
  if (mp > 0 && is_a) {
    const int trow = ch0 / mp;             // the tap this X channel quad belongs to (affine tap offsets)
    tb_unit = g.bx[0] + trow * (g.T > 1 ? g.bx[1] - g.bx[0] : 0);
    ch0 -= trow * mp;
    col_ok = has_unit && trow < g.T && ch0 < g.Kc;
  }
  if (np > 0 && !is_a) {
    const int tcol = ch0 / np;             // the tap this dY channel quad belongs to (affine tap offsets: by[t] = by[0] + t (by[1] - by[0]))
    tb_unit = g.by[0] + tcol * (g.T > 1 ? g.by[1] - g.by[0] : 0);
    ch0 -= tcol * np;
  }
  const float* const base = is_a ? g.X : g.dY;
  const int ld = is_a ? g.ldX : g.ldY;
  const int L = is_a ? g.Lx : g.Ly;
  const int s = is_a ? g.sx : g.sy;
  const int tb = tb_unit;
  const long long row_step = (long long)s * ld;                       // next reduction row, same sample
  const long long row_wrap = ((long long)L - (long long)g.nj * s) * ld;  // extra when j wraps into the next sample
  const int qs16 = WSK / g.nj, r16 = WSK % g.nj;
  const long long st_step = ((long long)qs16 * L + (long long)r16 * s) * ld;
  // running state of the first row of this thread's next tile: position in the sample jj, its X / dY rows, pointer, rows left
  int jj, xr0, yr0, left;
  const float* ptr;
  {
    const long long r = r_begin + rg * 4;
    const long long b = r / g.nj;
    jj = (int)(r - b * g.nj);
    ptr = base + (b * L + (long long)jj * s + tb) * (long long)ld + (col_ok ? ch0 : 0);
    const long long l = r_end - r;
    left = l > 0x7fffffffLL ? 0x7fffffff : (int)l;
    xr0 = jj * g.sx + (mp > 0 ? (is_a ? tb_unit : 0) : tbx);
    yr0 = jj * g.sy + (np > 0 ? (is_a ? 0 : tb_unit) : tby);  // flat mode: the X rows are shared by columns of different taps: never masked by one
  }
  const int dx16 = r16 * g.sx, dy16 = r16 * g.sy, wrapx = g.nj * g.sx, wrapy = g.nj * g.sy;
  const int yr_base = np > 0 ? (is_a ? 0 : tb_unit) : tby;
  const int xr_base = mp > 0 ? (is_a ? tb_unit : 0) : tbx;
  const bool y_free = np > 0 && is_a;   // (flat columns, X unit: its rows are valid for every tap)
  const bool x_free = mp > 0 && !is_a;  // (flat rows, dY unit: likewise)
  const int lds_unit = (is_a ? 0 : P * A_PIECE) + lds_row(cq * 4) + rg * 8;
  const float* const zero_row = wgrad_zero_row;

  float4 rv[4];
  // branch-free on purpose (bitwise &, selects): exec-mask branches here fence the MFMAs of the stage behind the whole
  // address computation
  auto load_tile = [&]() {
    int j = jj, xr = xr0, yr = yr0;
    const float* p = ptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = col_ok & (left > i) & (((unsigned)xr < (unsigned)g.Lx) | x_free) & (((unsigned)yr < (unsigned)g.Ly) | y_free);
      rv[i] = *reinterpret_cast<const float4*>(ok ? p : zero_row);
      ++j;
      p += row_step;
      xr += g.sx;
      yr += g.sy;
      const bool wrap = j >= g.nj;
      j = wrap ? 0 : j;
      p += wrap ? row_wrap : 0;
      xr = wrap ? xr_base : xr;
      yr = wrap ? yr_base : yr;
    }
    jj += r16;
    ptr += st_step;
    left -= WSK;
    xr0 += dx16;
    yr0 += dy16;
    const bool w2 = jj >= g.nj;
    jj -= w2 ? g.nj : 0;
    ptr += w2 ? row_wrap : 0;
    xr0 -= w2 ? wrapx : 0;
    yr0 -= w2 ? wrapy : 0;
  };
  auto store_tile = [&](int buf) {
    if (!has_unit) return;
    unsigned char* st = smem + buf * STAGE + lds_unit;
    const float v[4][4] = {{rv[0].x, rv[0].y, rv[0].z, rv[0].w}, {rv[1].x, rv[1].y, rv[1].z, rv[1].w},
                           {rv[2].x, rv[2].y, rv[2].z, rv[2].w}, {rv[3].x, rv[3].y, rv[3].z, rv[3].w}};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      uint2 pc[P];
      split4<P>(make_float4(v[0][c], v[1][c], v[2][c], v[3][c]), pc);
#pragma unroll
      for (int p = 0; p < P; ++p)
        *reinterpret_cast<uint2*>(st + p * (is_a ? A_PIECE : B_PIECE) + c * WROWB) = pc[p];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (int)((r_end - r_begin + WSK - 1) / WSK);
  auto compute = [&](const unsigned char* st, auto mid) {
    uint4 av[MT][P], bv[NT][P];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int p = 0; p < P; ++p)
        av[mt][p] = *reinterpret_cast<const uint4*>(st + p * A_PIECE + lds_row(wr * WM + mt * 32 + lr) + h * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int p = 0; p < P; ++p)
        bv[nt][p] = *reinterpret_cast<const uint4*>(st + P * A_PIECE + p * B_PIECE + lds_row(wc * WN + nt * 32 + lr) + h * 16);
    mid();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_split<P>(av[mt], bv[nt], acc[mt][nt]);
  };
  // loads / stores past the last tile are unconditional on purpose (rows past r_end read the zero row and stage zeros
  // into a buffer nobody reads): conditional loads make hipcc wait vmcnt(0) and lose the prefetch distance
  // Measured and NOT kept (MI355X, B=1024): issuing the next tile's loads before the stage's MFMAs
  // (sched_barrier), and two register slots with loads a whole stage ahead of their use -- both +-0 to -10 %.
  if constexpr (NSTAGE == 2) {
    load_tile();
    store_tile(0);
    load_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      compute(smem + (kt & 1) * STAGE, [&]() {
        store_tile((kt & 1) ^ 1);
        load_tile();
      });
      __syncthreads();
    }
  } else {
    load_tile();
    for (int kt = 0; kt < nk; ++kt) {
      store_tile(0);
      __syncthreads();
      load_tile();
      compute(smem, []() {});
      __syncthreads();
    }
  }

  float* out = g.out + (long long)bzi * g.slab_stride + (long long)ti * g.Kc * g.ldW;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int col = n0 + wc * WN + nt * 32 + lr;
    if (col >= g.N) continue;
    if (np > 0) {  // flat column -> (tap, output channel)
      const int t = col / np;
      col -= t * np;
      out = g.out + (long long)bzi * g.slab_stride + (long long)t * g.Kc * g.ldW;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int c = c0 + wr * WM + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        bool c_ok = c < g.Kc;
        if (mp > 0) {  // flat row -> (tap, input channel)
          const int t = c / mp;
          c -= t * mp;
          c_ok = t < g.T && c < g.Kc;
          out = g.out + (long long)bzi * g.slab_stride + (long long)t * g.Kc * g.ldW;
        }
        if (c_ok) {
          float* dst = out + (long long)c * g.ldW + col;
          float v = acc[mt][nt][r];
          if (g.accumulate) v += *dst;
          *dst = v;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------ weight grad, all taps in one workgroup
// wgrad_gemm_bf16s_kernel gives every tap its own workgroups: the two operand tiles of a 16-row step are loaded, converted and
// transposed into LDS once PER TAP although all T taps multiply the same dY rows with row-shifted views of the same X rows.  Here a
// workgroup owns a BM x BN tile of ALL taps: per 32-row stage the fixed operand F (dY of a conv) and the shifted operand S (X, with
// its (T-1) dil halo rows) are staged once, in their NATURAL [row][channel] order -- no transposing VALU work: the k-contiguous
// MFMA fragments are fetched with gfx950's transposing read ds_read_b64_tr_b16 -- and tap t reads S rows shifted by t dil: a plain
// address offset in this layout.  Per loaded and converted byte the matrix cores do T times the work; the per-tap re-read of both
// operands through the vector-memory path (the resource wgrad_gemm_bf16s_kernel is bound by) disappears.
//   D_t[m][n] = sum_r S[(r ss + t dil - pad)][m] * F[r][n]      r = reduction row (sample b, position j < nj)
// conv:             S = X (m = c_in),  F = dY (n = c_out), D_t = dW_t;
// transposed conv:  S = dY (m = c_out), F = X (n = c_in),  D_t = dW_t^T (TRANS_OUT: the epilogue stores the transpose).
// A tap that leaves its sample (0 <= j ss + t dil - pad < Ls violated) reads a zero region instead of the image: the mask is an
// address select, no VALU work on fragments.  Geometry: Lf == nj and Ls == nj ss + e -- e = 0 for the contiguous convs, e = 1 for the
// (k+1)-tap skip convs behind the upsampler (input 2L, output 2L - 1), e = -1 for the odd-length stride-2 (transposed) convs: the
// image of a stage is still one contiguous run of S rows, and a fragment row's image row moves by e per sample boundary between
// the stage's first row and it (g.e, g.srows).  Two bf16 pieces, T <= TMAX.
// Wave tile 32 (m) x WN (n) x T taps: T x NT accumulators of 32 x 32 (T = 5, NT = 2: 160 VGPRs).
// LDS images: 64-byte chunks (32 channels) of a row XOR-swizzled by the row so that the 4 rows x 64 bytes a half-wave's transposing
// read touches lie on different banks.
template <int BM, int BN, int TMAX, int SS, bool TRANS_OUT>
__global__ __launch_bounds__(512) void wgrad_taps_bf16s_kernel(const WgradTapsArgs g) {
  constexpr int P = 2, KS = 32, NTH = 512;
  constexpr int WR = BM / 32, WC = 8 / WR, WN = BN / WC, NT = WN / 32;
  static_assert(WR * WC == 8 && NT >= 1 && WN % 32 == 0, "8 waves of 32 x (32 NT)");
  constexpr int SR = (KS - 1) * SS + (TMAX - 1) + 1;  // image rows of the shifted operand (dil = 1)
  // image row: [piece 0: channels][piece 1: channels][pad]; the pad makes the 4 rows x 64 bytes a half-wave's transposing read
  // touches (rows SS apart) start 64 bytes apart modulo the 256-byte bank row -- conflict-free without an XOR swizzle, so that a tap
  // shift and the piece are IMMEDIATE offsets of the read instruction
  constexpr int RSS = P * BM * 2 + (SS == 1 ? 64 : 32), RSF = P * BN * 2 + 64;
  constexpr int S_ITEMS = (SR * (BM / 4) + NTH - 1) / NTH, F_ITEMS = (KS * (BN / 4) + NTH - 1) / NTH;
  constexpr int SR_ALLOC = S_ITEMS * (NTH / (BM / 4));  // every staging item owns an image row: no conditional LDS writes
  static_assert(F_ITEMS * (NTH / (BN / 4)) == KS, "the fixed operand's rows divide evenly over the staging items");
  constexpr int S_IMG = SR_ALLOC * RSS, F_IMG = KS * RSF;
  constexpr int STAGE = S_IMG + F_IMG;
  constexpr int ZBYTES = P * BM * 2 + 64;             // zero region: what a masked read may touch at its immediate offsets
  static_assert(2 * STAGE + ZBYTES <= 160 * 1024, "LDS budget");
  static_assert((TMAX - 1) * RSS + BM * 2 < 65536, "tap and piece offsets are 16-bit immediates");
  __shared__ __attribute__((aligned(64))) unsigned char smem[2 * STAGE + ZBYTES];
  constexpr int ZOFF = 2 * STAGE;

  const int tid = threadIdx.x;
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (g.xmap) {  // XCD-aware workgroup order: see wgrad_gemm_bf16s_kernel
    const int gxy = gridDim.x * gridDim.y, total = gxy * gridDim.z;
    const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bzi = w / gxy;
    const int rem = w - bzi * gxy;
    byi = rem / (int)gridDim.x;
    bxi = rem - byi * (int)gridDim.x;
  }
  const int c0 = bxi * BM, n0 = byi * BN;
  const long long r_begin = (long long)bzi * g.rows_per_split;
  long long r_end = r_begin + g.rows_per_split;
  if (r_end > g.R) r_end = g.R;
  constexpr int T = TMAX;  // (a run-time tap count puts the multiplies behind branches and hipcc then keeps every accumulator twice)
  for (int i = tid; i < ZBYTES / 4; i += NTH) reinterpret_cast<unsigned*>(smem + ZOFF)[i] = 0u;

  // ---- staging: item idx = tid + 512 i covers 4 channels (one float4) of image row idx / (channels / 4)
  // The two operands are staged in turn (S: requested in front of a stage's first k-step, written behind it; F: requested there,
  // written behind the second k-step), so that only one of them occupies registers at a time: with 5 taps x 2 column blocks the
  // accumulators alone take 160 of the 256 registers.
  float4 sv[S_ITEMS], fv[F_ITEMS];
  auto load_s = [&](int kt) {
    const long long rk = r_begin + (long long)kt * KS;
    long long srow0 = rk * g.ss - g.pad;
    if (g.e != 0) srow0 += (rk / g.nj) * g.e;  // samples are Ls = nj ss + e rows apart: the image starts at the stage's first sample offset
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      const long long sr = srow0 + row;
      // (select on the POINTER: a conditional load costs an exec-mask branch and a vmcnt(0) in front of the matrix work)
      const bool ok = (row < g.srows) & (sr >= 0) & (sr < g.rowsS) & (c0 + cq * 4 < g.Cs);  // srows = SR + the rows e > 0 adds (<= SR_ALLOC, host-checked)
      sv[i] = *reinterpret_cast<const float4*>(ok ? g.S + sr * g.ldS + c0 + cq * 4 : wgrad_zero_row);
    }
  };
  auto load_f = [&](int kt) {
    const long long r0 = r_begin + (long long)kt * KS;
#pragma unroll
    for (int i = 0; i < F_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BN / 4), cq = idx % (BN / 4);
      const long long fr = r0 + row;
      const bool ok = (fr < r_end) & (n0 + cq * 4 < g.Cf);
      fv[i] = *reinterpret_cast<const float4*>(ok ? g.F + fr * g.ldF + n0 + cq * 4 : wgrad_zero_row);
    }
  };
  auto store_s = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      uint2 pc[P];
      split4<P>(sv[i], pc);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + row * RSS + p * (BM * 2) + cq * 8) = pc[p];
    }
  };
  auto store_f = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < F_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BN / 4), cq = idx % (BN / 4);
      uint2 pc[P];
      split4<P>(fv[i], pc);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + S_IMG + row * RSF + p * (BN * 2) + cq * 8) = pc[p];
    }
  };

  // ---- fragment addressing (transposing reads: lane 4q+p of a 16-lane group supplies row q, channels 4p..4p+3 of a 4 x 16 block;
  //      lane i of the group receives channel i of the 4 rows -- 4 consecutive k of its MFMA operand row)
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int lr = lane & 31, h = lane >> 5;
  const int mhalf = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
  // k-step ks, half j of the fragment: reduction row rr = 16 ks + 8 h + 4 j + q.  Byte offsets (in a stage) at tap 0 / piece 0 /
  // column block 0 (the tap (t RSS), the piece and the column block are immediates) are rebuilt from rr per k-step
  const int rr0 = 8 * h + q4;
  const int a_lane = (wr * 32 + 16 * mhalf + 4 * p4) * 2, f_lane = S_IMG + (wc * WN + 16 * mhalf + 4 * p4) * 2;
  const unsigned njm = (g.nj & (g.nj - 1)) == 0 ? (unsigned)(g.nj - 1) : 0u;  // power-of-two sample length: mask instead of modulo
  int lpos0 = (int)((r_begin + rr0) % g.nj);  // position in its sample of reduction row rr0 of the current stage
  int sp0 = (int)(r_begin % g.nj);            // ... of the stage's first reduction row (wave-uniform)

  f32x16 acc[TMAX][NT];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][nt][r] = 0.f;

  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
  auto tr_read = [&](const unsigned char* p) -> uint2 {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p));
  };
  const int nk = (int)((r_end - r_begin + KS - 1) / KS);
  auto wrap = [&](int v) { return (int)(njm ? ((unsigned)v & njm) : ((unsigned)v % (unsigned)g.nj)); };
  auto compute = [&](int buf, auto mid, auto tail) {
    const int soff = buf * STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      int a_base[2], f_base[2], lbs[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rr = ks * 16 + 4 * j + rr0;
        a_base[j] = soff + rr * (SS * RSS) + a_lane;
        // Ls = nj ss + e (the k+1-tap skip convs: e = 1; odd-length stride-2 (transposed) convs: e = -1): every sample boundary
        // between the stage's first row and row rr shifts the image row by e
        if (g.e != 0) a_base[j] += (int)((unsigned)(sp0 + rr) / (unsigned)g.nj) * g.e * RSS;
        f_base[j] = soff + rr * RSF + f_lane;
        // position in the sample of the tap-0 source row, minus the padding: tap t is inside its sample iff 0 <= lbs + t < Ls
        lbs[j] = wrap(lpos0 + ks * 16 + 4 * j) * SS - g.pad;
      }
      uint4 bv[NT][P];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const uint2 lo = tr_read(smem + f_base[0] + p * (BN * 2) + nt * 64), hi = tr_read(smem + f_base[1] + p * (BN * 2) + nt * 64);
          bv[nt][p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      // taps: operand fetch of tap t+1 issued in front of the multiplies of tap t
      auto fetch_a = [&](auto t_c, uint4 (&av)[P]) {
        constexpr int t = decltype(t_c)::value;
        int off[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) off[j] = (unsigned)(lbs[j] + t) < (unsigned)g.Ls ? a_base[j] + t * RSS : ZOFF;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const uint2 lo = tr_read(smem + off[0] + p * (BM * 2)), hi = tr_read(smem + off[1] + p * (BM * 2));
          av[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      };
      uint4 av0[P], av1[P];
      fetch_a(std::integral_constant<int, 0>{}, av0);
      auto tap = [&](auto t_c, uint4 (&cur)[P], uint4 (&nxt)[P]) {
        constexpr int t = decltype(t_c)::value;
        if constexpr (t + 1 < TMAX) fetch_a(std::integral_constant<int, t + 1>{}, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = mfma_split<P>(cur, bv[nt], acc[t][nt]);
        __builtin_amdgcn_sched_barrier(0);
      };
      tap(std::integral_constant<int, 0>{}, av0, av1);
      tap(std::integral_constant<int, 1>{}, av1, av0);
      tap(std::integral_constant<int, 2>{}, av0, av1);
      tap(std::integral_constant<int, 3>{}, av1, av0);
      tap(std::integral_constant<int, 4>{}, av0, av1);
      if constexpr (TMAX > 5) tap(std::integral_constant<int, 5>{}, av1, av0);
      if (ks == 0) mid(); else tail();
    }
    lpos0 = wrap(lpos0 + KS);  // next stage: the reduction rows advance by KS
    sp0 = wrap(sp0 + KS);
  };

  // stage kt+1 is staged while stage kt is multiplied: S requested before, written between the two k-steps; F requested there,
  // written after the second
  load_s(0);
  load_f(0);
  store_s(0);
  store_f(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    load_s(kt + 1);
    compute(kt & 1,
            [&]() { store_s((kt & 1) ^ 1); load_f(kt + 1); },
            [&]() { store_f((kt & 1) ^ 1); });
    __syncthreads();
  }

  // ---- epilogue: D_t[m][n] -> out[t][.][.] of this split's slab
  float* out = g.out + (long long)bzi * g.slab_stride;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    float* ot = out + (long long)t * (TRANS_OUT ? g.Cf : g.Cs) * g.ldW;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = n0 + wc * WN + nt * 32 + lr;
      if (col >= g.Cf) continue;
      if constexpr (TRANS_OUT) {
        // D_t is dW_t^T: a lane's 4 consecutive accumulator rows are 4 consecutive output channels of input channel `col` -- one
        // 16-byte store each (c_out is a multiple of 16: rows of dW are 64-byte aligned)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
          const int m = c0 + wr * 32 + 8 * rq + 4 * h;
          if (m < g.Cs) {
            float4* dst = reinterpret_cast<float4*>(ot + (long long)col * g.ldW + m);
            float4 v = make_float4(acc[t][nt][4 * rq], acc[t][nt][4 * rq + 1], acc[t][nt][4 * rq + 2], acc[t][nt][4 * rq + 3]);
            if (g.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *dst = v;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = c0 + wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (m < g.Cs) {
            float* dst = ot + (long long)m * g.ldW + col;
            float v = acc[t][nt][r];
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
}

// ---- the same kernel on v_mfma_f32_16x16x32_bf16.  MI355X_MICROARCH (DVFS give-back, item 7): where the chip holds its clock down
// under matrix load -- it does in every GEMM of this step: in-kernel clocks of 1.7-2.0 GHz, tools/stamp_halo.py -- the 16x16x32
// shape sustains a higher clock than 32x32x16 at equal cycles per FLOP.  One MFMA spans the whole 32-row stage (K = 32).  The
// reduction rows are PERMUTED over the operand's k index (the same permutation for both operands: a sum does not care):
// lane group g = lane / 16 (k chunk 8 g ..), read j, row q of the transposing read's block  ->  stage row 16 j + 4 g + q, so that a
// half-wave's read covers 8 CONSECUTIVE rows x 32 bytes -- conflict-free with image rows 32 bytes (mod 256) apart.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma16_bf16(uint4 a, uint4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16_split2(const uint4 (&a)[2], const uint4 (&b)[2], f32x4v acc) {  // small terms first
  acc = mfma16_bf16(a[1], b[0], acc);
  acc = mfma16_bf16(a[0], b[1], acc);
  acc = mfma16_bf16(a[0], b[0], acc);
  return acc;
}

template <int BM, int BN, int TMAX, int SS, bool TRANS_OUT>
__global__ __launch_bounds__(512) void wgrad_taps16_bf16s_kernel(const WgradTapsArgs g) {
  constexpr int P = 2, KS = 32, NTH = 512, T = TMAX;
  constexpr int WR = BM / 32, WC = 8 / WR, WN = BN / WC, MB = 2, NB = WN / 16;
  static_assert(WR * WC == 8 && NB >= 2 && WN % 16 == 0, "8 waves of 32 x (16 NB)");
  constexpr int SR = (KS - 1) * SS + (TMAX - 1) + 1;
  constexpr int RSS = P * BM * 2 + (SS == 1 ? 32 : 16), RSF = P * BN * 2 + 32;  // 8 rows of a half-wave's read: 32 bytes apart mod 256
  constexpr int S_ITEMS = (SR * (BM / 4) + NTH - 1) / NTH, F_ITEMS = (KS * (BN / 4) + NTH - 1) / NTH;
  constexpr int SR_ALLOC = S_ITEMS * (NTH / (BM / 4));
  static_assert(F_ITEMS * (NTH / (BN / 4)) == KS, "the fixed operand's rows divide evenly over the staging items");
  constexpr int S_IMG = SR_ALLOC * RSS, F_IMG = KS * RSF;
  constexpr int STAGE = S_IMG + F_IMG;
  constexpr int ZBYTES = P * BM * 2 + 64;
  static_assert(2 * STAGE + ZBYTES <= 160 * 1024, "LDS budget");
  static_assert((TMAX - 1) * RSS + P * BM * 2 < 65536, "tap and piece offsets are 16-bit immediates");
  __shared__ __attribute__((aligned(64))) unsigned char smem[2 * STAGE + ZBYTES];
  constexpr int ZOFF = 2 * STAGE;

  const int tid = threadIdx.x;
  int bxi = blockIdx.x, byi = blockIdx.y, bzi = blockIdx.z;
  if (g.xmap) {
    const int gxy = gridDim.x * gridDim.y, total = gxy * gridDim.z;
    const int id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, r = total & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    bzi = w / gxy;
    const int rem = w - bzi * gxy;
    byi = rem / (int)gridDim.x;
    bxi = rem - byi * (int)gridDim.x;
  }
  const int c0 = bxi * BM, n0 = byi * BN;
  const long long r_begin = (long long)bzi * g.rows_per_split;
  long long r_end = r_begin + g.rows_per_split;
  if (r_end > g.R) r_end = g.R;
  for (int i = tid; i < ZBYTES / 4; i += NTH) reinterpret_cast<unsigned*>(smem + ZOFF)[i] = 0u;

  float4 sv[S_ITEMS], fv[F_ITEMS];
  auto load_s = [&](int kt) {
    const long long rk = r_begin + (long long)kt * KS;
    long long srow0 = rk * g.ss - g.pad;
    if (g.e != 0) srow0 += (rk / g.nj) * g.e;  // samples are Ls = nj ss + e rows apart: the image starts at the stage's first sample offset
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      const long long sr = srow0 + row;
      const bool ok = (row < g.srows) & (sr >= 0) & (sr < g.rowsS) & (c0 + cq * 4 < g.Cs);  // srows = SR + the rows e > 0 adds (<= SR_ALLOC, host-checked)
      sv[i] = *reinterpret_cast<const float4*>(ok ? g.S + sr * g.ldS + c0 + cq * 4 : wgrad_zero_row);
    }
  };
  auto load_f = [&](int kt) {
    const long long r0 = r_begin + (long long)kt * KS;
#pragma unroll
    for (int i = 0; i < F_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BN / 4), cq = idx % (BN / 4);
      const long long fr = r0 + row;
      const bool ok = (fr < r_end) & (n0 + cq * 4 < g.Cf);
      fv[i] = *reinterpret_cast<const float4*>(ok ? g.F + fr * g.ldF + n0 + cq * 4 : wgrad_zero_row);
    }
  };
  auto store_s = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      uint2 pc[P];
      split4<P>(sv[i], pc);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + row * RSS + p * (BM * 2) + cq * 8) = pc[p];
    }
  };
  auto store_f = [&](int buf) {
    unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < F_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BN / 4), cq = idx % (BN / 4);
      uint2 pc[P];
      split4<P>(fv[i], pc);
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(st + S_IMG + row * RSF + p * (BN * 2) + cq * 8) = pc[p];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int g4 = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
  const int rr0 = 4 * g4 + q4;  // stage row of read j: 16 j + rr0
  const int a_lane = (wr * 32 + 4 * p4) * 2, f_lane = S_IMG + (wc * WN + 4 * p4) * 2;
  const unsigned njm = (g.nj & (g.nj - 1)) == 0 ? (unsigned)(g.nj - 1) : 0u;
  int lpos0 = (int)((r_begin + rr0) % g.nj);
  int sp0 = (int)(r_begin % g.nj);
  auto wrap = [&](int v) { return (int)(njm ? ((unsigned)v & njm) : ((unsigned)v % (unsigned)g.nj)); };

  f32x4v acc[TMAX][MB][NB];
#pragma unroll
  for (int t = 0; t < TMAX; ++t)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][mb][nb][r] = 0.f;

  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
  auto tr_read = [&](const unsigned char* p) -> uint2 {
    return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_tr_ptr)p));
  };
  const int nk = (int)((r_end - r_begin + KS - 1) / KS);
  auto compute = [&](int buf, auto mid, auto tail) {
    const int soff = buf * STAGE;
    int a_base[2], f_base[2], lbs[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = 16 * j + rr0;
      a_base[j] = soff + rr * (SS * RSS) + a_lane;
      if (g.e != 0) a_base[j] += (int)((unsigned)(sp0 + rr) / (unsigned)g.nj) * g.e * RSS;  // see wgrad_taps_bf16s_kernel
      f_base[j] = soff + rr * RSF + f_lane;
      lbs[j] = wrap(lpos0 + 16 * j) * SS - g.pad;
    }
    uint4 bv[NB][P];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const uint2 lo = tr_read(smem + f_base[0] + p * (BN * 2) + nb * 32), hi = tr_read(smem + f_base[1] + p * (BN * 2) + nb * 32);
        bv[nb][p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
      }
    auto tap = [&](auto t_c) {
      constexpr int t = decltype(t_c)::value;
      int off[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) off[j] = (unsigned)(lbs[j] + t) < (unsigned)g.Ls ? a_base[j] + t * RSS : ZOFF;
      uint4 av[MB][P];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const uint2 lo = tr_read(smem + off[0] + p * (BM * 2) + mb * 32), hi = tr_read(smem + off[1] + p * (BM * 2) + mb * 32);
          av[mb][p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[t][mb][nb] = mfma16_split2(av[mb], bv[nb], acc[t][mb][nb]);
    };
    tap(std::integral_constant<int, 0>{});
    tap(std::integral_constant<int, 1>{});
    mid();
    tap(std::integral_constant<int, 2>{});
    tap(std::integral_constant<int, 3>{});
    tap(std::integral_constant<int, 4>{});
    if constexpr (TMAX > 5) tap(std::integral_constant<int, 5>{});
    tail();
    lpos0 = wrap(lpos0 + KS);
    sp0 = wrap(sp0 + KS);
  };

  load_s(0);
  load_f(0);
  store_s(0);
  store_f(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    load_s(kt + 1);
    compute(kt & 1,
            [&]() { store_s((kt & 1) ^ 1); load_f(kt + 1); },
            [&]() { store_f((kt & 1) ^ 1); });
    __syncthreads();
  }

  // ---- epilogue (16x16 tiles: column = lane & 15, row = 4 (lane / 16) + register)
  float* out = g.out + (long long)bzi * g.slab_stride;
#pragma unroll
  for (int t = 0; t < TMAX; ++t) {
    float* ot = out + (long long)t * (TRANS_OUT ? g.Cf : g.Cs) * g.ldW;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = n0 + wc * WN + nb * 16 + (lane & 15);
      if (col >= g.Cf) continue;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int m = c0 + wr * 32 + mb * 16 + 4 * g4;
        if (m >= g.Cs) continue;
        if constexpr (TRANS_OUT) {
          float4* dst = reinterpret_cast<float4*>(ot + (long long)col * g.ldW + m);
          float4 v = make_float4(acc[t][mb][nb][0], acc[t][mb][nb][1], acc[t][mb][nb][2], acc[t][mb][nb][3]);
          if (g.accumulate) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *dst = v;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* dst = ot + (long long)(m + r) * g.ldW + col;
            float v = acc[t][mb][nb][r];
            if (g.accumulate) v += *dst;
            *dst = v;
          }
        }
      }
    }
  }
}

// ------------------------------------------- the 12-wave halo kernel on v_mfma_f32_16x16x32 (tile code 18)
// Same structure as gather_halo_ws4_bf16s_kernel (loaders, staging ring, consumer-side conversion, zero-row reads); the matrix work
// runs on the 16x16x32 shape: 3-11 % faster alone on the deep layers, no change of the whole step.
// Fragment = 16 rows x 32 k: lane l reads the 16-byte chunk l / 16 of row l % 16 -- one ds_read_b128 per 16-row block, piece and stage.
// The chunk swizzle differs from swz(): with chunks 0 / 1 of rows 0-3 / 4-11 / 12-15 in one ds_read_b128 lane group, the map
// row / 4 -> (0, 3, 2, 1) keeps the 16 lanes of a group on 16 different 16-byte slots.
__device__ __forceinline__ int swz16(int row) { return (0 - (row >> 2)) & 3; }
__device__ __forceinline__ f32x4v mfma16_f16(uint4 a, uint4 b, f32x4v c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool H>
__device__ __forceinline__ f32x4v mfma16_split2x(const uint4 (&a)[2], const uint4 (&b)[2], f32x4v acc) {  // small terms first
  if constexpr (H) {
    acc = mfma16_f16(a[1], b[0], acc);
    acc = mfma16_f16(a[0], b[1], acc);
    acc = mfma16_f16(a[0], b[0], acc);
    return acc;
  } else {
    return mfma16_split2(a, b, acc);
  }
}

// tile_epilogue for 16 x 16 accumulator tiles: column = lane % 16, row = 4 (lane / 16) + register
template <int MB, int NBK, int WM, int WN, int WR, int BN>
__device__ __forceinline__ void tile_epilogue16(const GatherArgs& g, f32x4v (&acc)[MB][NBK], const long long* rowoff, int n0, int wr, int wc,
                                                int lane, float* red, int tid, int nth, float oscale, int tile_x, int tile_y) {
  const int l16 = lane & 15, rg = lane >> 4;
  float cs[NBK], cq[NBK];
  double da = 0.0;  // the PReLU slope's partial: a sum of ~1e6 cancelling terms over the launch -- fp64 products and sums
  const bool bwd = g.bn_x != nullptr;        // uniform
  const bool th = g.bn_alpha == nullptr;     // tanh instead of PReLU
  const float slope = (bwd && !th) ? g.bn_alpha[0] : 0.f;
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const int col = n0 + wc * WN + nb * 16 + l16;
    cs[nb] = 0.f;
    cq[nb] = 0.f;
    if (col >= g.N) continue;
    const float bv = g.bias ? g.bias[col] : 0.f;
    float sc = 1.f, sh = 0.f, mu = 0.f, rs = 0.f;
    if (bwd) {
      if (g.bn_scale) { sc = g.bn_scale[col]; sh = g.bn_shift[col]; }
      if (g.bn_mean) { mu = g.bn_mean[col]; rs = g.bn_rstd[col]; }
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr * WM + mb * 16 + 4 * rg + r;
        const long long off = rowoff[row];
        if (off >= 0) {
          float* dst = g.C + off + col;
          float v = acc[mb][nb][r] * oscale + bv;
          if (g.accumulate) v += *dst;
          *dst = v;
          if (bwd) {
            const float x = g.bn_x[off + col];
            const float u = x * sc + sh;
            float du;
            if (th) { const float t = tanhf(u); du = v * (1.f - t * t); }
            else { du = u > 0.f ? v : slope * v; if (!(u > 0.f)) da += (double)v * (double)u; }
            cs[nb] += du;
            cq[nb] += du * (x - mu) * rs;
          } else {
            cs[nb] += v;
            cq[nb] += v * v;
          }
        }
      }
    }
  }
  if (g.stats == nullptr) return;  // uniform
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    cs[nb] += __shfl_xor(cs[nb], 16, 64);
    cq[nb] += __shfl_xor(cq[nb], 16, 64);
    cs[nb] += __shfl_xor(cs[nb], 32, 64);
    cq[nb] += __shfl_xor(cq[nb], 32, 64);
  }
  __syncthreads();  // every wave is past its last LDS operand read: the staging buffers are free
  if (rg == 0) {
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb) {
      const int c = wc * WN + nb * 16 + l16;
      red[(0 * WR + wr) * BN + c] = cs[nb];
      red[(1 * WR + wr) * BN + c] = cq[nb];
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

template <int BM, int BN, int RMAX, int STG, bool H>
__global__ __launch_bounds__(768) void gather_halo_ws4m_bf16s_kernel(const SplitGatherArgs sa) {
  const GatherArgs& g = sa.g;
  constexpr int P = 2, NB = 3, WR = 4, WC = 2, NPW = 4;
  constexpr int NCT = 64 * WR * WC;   // consumer threads
  constexpr int WM = BM / WR, MB = WM / 16, WN = BN / WC, NBK = WN / 16;
  static_assert(MB >= 1 && NBK >= 1 && WM % 16 == 0 && WN % 16 == 0, "wave tile must be a multiple of 16x16");
  static_assert(BM <= NCT, "rowoff is filled by the consumer threads");
  static_assert(RMAX % 8 == 0 && STG % 8 == 0, "a DMA wave-instruction carries 8 image rows");
  constexpr int ROWB = SBK * 2;                    // bytes of an image / weight row per piece
  constexpr int RAWB = SBK * 4;                    // bytes of a raw fp32 row
  constexpr int A_PIECE = (RMAX + 1) * ROWB, B_PIECE = BN * ROWB;  // image row RMAX: zeros, the target of operand reads that hit conv padding
  constexpr int ZROW = RMAX * ROWB;
  constexpr int A_IMG = P * A_PIECE, B_STAGE = P * B_PIECE;
  constexpr int STG_BYTES = STG * RAWB;
  static_assert(RMAX * RAWB <= A_IMG, "the prologue stages the raw image of block 0 in image buffer 1");
  static_assert(2 * A_IMG + NB * B_STAGE + 3 * STG_BYTES + BM * 8 <= 160 * 1024, "LDS budget");
  constexpr int B_INSTR = B_STAGE / 1024;          // 1-KiB DMA wave-instructions per weight stage
  static_assert(B_STAGE % (1024 * NPW) == 0, "weight stage must split evenly over the loader waves");
  constexpr int BPW = B_INSTR / NPW;               // per loader wave
  constexpr int SPW = (STG / 8 + NPW - 1) / NPW;   // raw-slice wave-instructions per loader wave, at most
  constexpr int CPT = (STG * 8 + NCT - 1) / NCT;   // float4 conversions per consumer thread and stage, at most
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * A_IMG + NB * B_STAGE + 3 * STG_BYTES];
  __shared__ long long rowoff[BM];
  unsigned char* const bbase = smem + 2 * A_IMG;
  unsigned char* const sbase = bbase + NB * B_STAGE;

  const int tid = threadIdx.x;
  // (An XCD-aware tile order -- every XCD a contiguous run of the column-major tile list, so that its workgroups stream the same
  //  weight tiles -- was measured 2-8 % SLOWER: each XCD then reads every activation image once per column tile from beyond its
  //  L2, where the launch order already gives an XCD 8 row tiles x 4 column tiles at a time, each line shared by 4-8 workgroups.)
  const int tile_x = blockIdx.x, tile_y = blockIdx.y;
  int bx = tile_x, phase = 0;
  if (bx >= g.blocks_m[0]) { phase = 1; bx -= g.blocks_m[0]; }
  const long long m0 = (long long)bx * BM;
  const int n0 = tile_y * BN;
  const long long Mp = g.M[phase];
  const int nj = g.nj[phase];
  const int ntaps = g.ntaps[phase];
  const int tb0 = g.base0[phase], tbs = g.bstep[phase], tw0 = g.w0[phase], tws = g.wstep[phase];
  const int tbl = tb0 + (ntaps - 1) * tbs;
  const int bmin = tb0 < tbl ? tb0 : tbl, bmax = tb0 < tbl ? tbl : tb0;
  const long long b0 = m0 / nj;
  const long long amin = b0 * g.Lin + (long long)(m0 - b0 * nj) * g.sj;
  const long long ml = (m0 + BM < Mp ? m0 + BM : Mp) - 1;
  const long long bl = ml / nj;
  const long long amax = bl * g.Lin + (long long)(ml - bl * nj) * g.sj;
  const int R = (int)(amax - amin) + bmax - bmin + 1;  // <= RMAX (checked on the host)
  const long long gbase = amin + bmin;
  const long long kb_stride = (long long)g.N * SBK;
  const long long tap_stride = kb_stride * sa.KB;
  const int ns = ntaps * sa.KB;
  const int SR = (((RMAX + ntaps - 1) / ntaps) + 7) & ~7;  // image rows per slice (<= STG, checked on the host); ntaps slices cover RMAX

  if (tid >= NCT) {
    // ================================================================== loader waves (LDS-DMA only)
    const int ptid = tid - NCT;
    const int pw = __builtin_amdgcn_readfirstlane(ptid >> 6), lane = ptid & 63;
    // weight DMA: wave-instruction i of this wave fills LDS bytes [(NPW i + pw) KiB, +1 KiB) of the stage: chunk q = 64 (NPW i + pw) + lane
    long long b_src[BPW];
    bool b_ok[BPW];
#pragma unroll
    for (int i = 0; i < BPW; ++i) {
      const int q = 64 * (NPW * i + pw) + lane;
      const int piece = q / (BN * 4), rem = q - piece * (BN * 4);
      const int row = rem >> 2, slot = rem & 3;
      b_ok[i] = n0 + row < g.N;
      b_src[i] = (long long)piece * sa.w_piece_stride + (long long)(n0 + row) * SBK + ((slot ^ swz16(row)) << 3);
    }
    auto dma_b = [&](int tap, int kb, int buf) {
      const unsigned short* wt = sa.Wp + (long long)(tw0 + tap * tws) * tap_stride + (long long)kb * kb_stride;
#pragma unroll
      for (int i = 0; i < BPW; ++i) {
        const unsigned short* src = b_ok[i] ? wt + b_src[i] : halo_zero_chunk;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(bbase + buf * B_STAGE + (NPW * i + pw) * 1024), 16, 0, 0);
      }
    };
    // raw activation rows: a wave-instruction carries 8 rows x 128 bytes (lane -> row lane / 8, channels 4 (lane % 8) ..+3), linear in LDS
    const int lrow = lane >> 3, lch = (lane & 7) * 4;
    const long long rlo_ll = -gbase, rhi_ll = sa.rowsA - gbase;
    const int row_lo = rlo_ll > 0 ? (int)(rlo_ll < RMAX ? rlo_ll : RMAX) : 0;          // image rows [row_lo, row_hi) exist in A
    const int row_hi = rhi_ll < R ? (int)(rhi_ll > 0 ? rhi_ll : 0) : R;
    const float* const a_lane = g.A + (gbase + lrow) * (long long)g.ldA + lch;
    // rows [r0, r0 + 8 n) of channel block kb -> dst (n wave-instructions split over the loader waves); returns this wave's count
    auto dma_rows = [&](int r0, int n, int kb, unsigned char* dst) -> int {
      const bool ch_ok = kb * SBK + lch < g.Kc;
      const float* base = a_lane + kb * SBK;
      int cnt = 0;
      for (int j = pw; j < n; j += NPW) {
        const int r = r0 + 8 * j + lrow;
        const bool ok = ch_ok && r >= row_lo && r < row_hi;
        const float* src = ok ? base + (long long)(r0 + 8 * j) * g.ldA : halo_zero_f32;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lds_ptr_t)(dst + j * 1024), 16, 0, 0);
        ++cnt;
      }
      return cnt;
    };
    // what stage c = (kq, tq) needs: its weight tile (buffer c % 3) and the raw slice the consumers convert during it -- slice tq of
    // block kq+1 (staging slot c % 3); returns this wave's number of wave-instructions
    int tq = 0, kq = 0;
    auto request = [&](int c) -> int {
      if (c >= ns) return 0;
      dma_b(tq, kq, c % NB);
      int n = BPW;
      if (kq + 1 < sa.KB) n += dma_rows(tq * SR, SR / 8, kq + 1, sbase + (c % 3) * STG_BYTES);
      if (++tq == ntaps) { tq = 0; ++kq; }
      return n;
    };
    // prologue: the raw image of block 0 whole (into image buffer 1), then the requests of stages 0 and 1
    if (ns > 0) dma_rows(0, RMAX / 8, 0, smem + A_IMG);
    request(0);
    request(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bare_barrier();  // A: raw image of block 0 landed
    bare_barrier();  // B: the consumers converted it
    for (int s = 0; s < ns; ++s) {
      const int n = request(s + 2);
      // everything older than THIS interval's requests has landed: what stage s+1 needs
      wait_vmcnt_rt<0, BPW + SPW>(n);
      bare_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // ==================================================================== consumer waves
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
  const int wave = tid >> 6, lane = tid & 63;
  const int wr = wave / WC, wc = wave % WC;
  const int l16 = lane & 15, ch = lane >> 4;  // fragment row within a 16-row block, 16-byte chunk (8 k) of the 32-deep stage
  int ro[MB], jj[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const long long m = m0 + wr * WM + mb * 16 + l16;
    if (m < Mp) {
      const long long b = m / nj;
      const int j = (int)(m - b * nj);
      ro[mb] = (int)(b * g.Lin + (long long)j * g.sj - amin) - bmin;
      jj[mb] = j * g.sj;
    } else {
      ro[mb] = -bmin;
      jj[mb] = -(1 << 28);
    }
  }
  int b_off0[NBK];
#pragma unroll
  for (int nb = 0; nb < NBK; ++nb) {
    const int row = wc * WN + nb * 16 + l16;
    b_off0[nb] = 2 * A_IMG + row * ROWB + ((ch ^ swz16(row)) << 4);
  }
  f32x4v acc[MB][NBK];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NBK; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;

  // raw fp32 rows [r0, r0 + nrows) at `raw` (128 bytes per row) -> piece planes of image `img`: item idx = tid + NCT i covers
  // channels 4 (idx % 8) ..+3 of row idx / 8
  const int cchunk = tid & 7;
  auto convert = [&](const unsigned char* raw, int r0, int nrows, unsigned char* img, int iters) {
    for (int i = 0; i < iters; ++i) {
      const int rl = (tid >> 3) + (NCT / 8) * i;
      const int r = r0 + rl;
      if (rl < nrows && r < RMAX) {
        const float4 v = *reinterpret_cast<const float4*>(raw + rl * RAWB + cchunk * 16);
        uint2 pc[P];
        split4x<P, H>(v, pc);
        const int off = r * ROWB + (((cchunk >> 1) ^ swz16(r)) << 4) + ((cchunk & 1) << 3);
#pragma unroll
        for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(img + p * A_PIECE + off) = pc[p];
      }
    }
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // rowoff
  bare_barrier();  // A
  if (ns > 0) convert(smem + A_IMG, 0, RMAX, smem, (RMAX * 8 + NCT - 1) / NCT);
  if (tid < 4 * P) *reinterpret_cast<uint4*>(smem + (tid >> 2) * A_PIECE + ZROW + (tid & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  bare_barrier();  // B
  // (the zero row of image buffer 1, where the raw image of block 0 sat until now: complete at the first stage barrier, read from block 1 on)
  if (tid < 4 * P) *reinterpret_cast<uint4*>(smem + A_IMG + (tid >> 2) * A_PIECE + ZROW + (tid & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
  {
    uint4 av[MB][P], bv[NBK][P];
    int tap = 0, kb = 0;
    int tb = 0;
    int a_off[MB], b_off[NBK];
    // the slice converted in interval s = (kb, tap): slice `tap` of block kb+1, from staging slot s % 3.  Two steps so that the raw
    // read has a k-step of matrix work to land behind: raw_read(s) next to the operand fetches, raw_write() after the multiply
    float4 rawv[CPT];
    auto raw_read = [&](int s) {
      if (kb + 1 >= sa.KB) return;
      const unsigned char* raw = sbase + (s % 3) * STG_BYTES;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int rl = (tid >> 3) + (NCT / 8) * i;
        if (rl < SR) rawv[i] = *reinterpret_cast<const float4*>(raw + rl * RAWB + cchunk * 16);
      }
    };
    auto raw_write = [&]() {
      if (kb + 1 >= sa.KB) return;
      unsigned char* dst = smem + ((kb + 1) & 1) * A_IMG;
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const int rl = (tid >> 3) + (NCT / 8) * i;
        const int r = tap * SR + rl;
        if (rl < SR && r < RMAX) {
          uint2 pc[P];
          split4x<P, H>(rawv[i], pc);
          const int off = r * ROWB + (((cchunk >> 1) ^ swz16(r)) << 4) + ((cchunk & 1) << 3);
#pragma unroll
          for (int p = 0; p < P; ++p) *reinterpret_cast<uint2*>(dst + p * A_PIECE + off) = pc[p];
        }
      }
    };
    // operand addresses of stage s.  An output row whose tap falls into the conv padding reads the image's zero row
    auto begin_stage = [&](int s) {
      tb = tb0 + tap * tbs;
      const int img_off = (kb & 1) * A_IMG, bst_off = (s % NB) * B_STAGE;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int arow = ro[mb] + tb;
        const bool valid = (unsigned)(jj[mb] + tb) < (unsigned)g.Lin;
        a_off[mb] = img_off + (valid ? arow * ROWB + ((ch ^ swz16(arow)) << 4) : ZROW);
      }
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) b_off[nb] = b_off0[nb] + bst_off;
    };
    auto end_stage = [&]() { if (++tap == ntaps) { tap = 0; ++kb; } };
    auto fetch = [&]() {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int p = 0; p < P; ++p) av[mb][p] = *reinterpret_cast<const uint4*>(smem + a_off[mb] + p * A_PIECE);
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
        for (int p = 0; p < P; ++p) bv[nb][p] = *reinterpret_cast<const uint4*>(smem + b_off[nb] + p * B_PIECE);
    };
    auto mma = [&]() {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb) acc[mb][nb] = mfma16_split2x<H>(av[mb], bv[nb], acc[mb][nb]);
    };
    // One MFMA spans the whole 32-deep stage, so a wave fetches a stage's operands and then multiplies them; the two waves of a SIMD
    // run half a stage apart (waves 0-3: fetch(s) | multiply(s) | barrier; waves 4-7: multiply(s-1) | fetch(s) | barrier), so that
    // one fetches while its partner multiplies.
    if (wave < 4) {
      for (int s = 0; s < ns; ++s) {
        begin_stage(s);
        fetch();
        raw_read(s);
        mma();
        raw_write();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bare_barrier();
        end_stage();
      }
    } else {
      for (int s = 0; s < ns; ++s) {
        raw_read(s);
        if (s > 0) mma();
        begin_stage(s);
        fetch();
        raw_write();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bare_barrier();
        end_stage();
      }
      if (ns > 0) mma();
    }
  }
  tile_epilogue16<MB, NBK, WM, WN, WR, BN>(g, acc, rowoff, n0, wr, wc, lane, reinterpret_cast<float*>(smem), tid, NCT, H ? F16_OSCALE : 1.f,
                                           tile_x, tile_y);
}

int launch_wgrad_taps(const WgradTapsArgs& g, dim3 grid, hipStream_t st, int bm, int bn, int trans_out, int m16) {
  const dim3 block(512);
#define SVAE_WT(BM_, BN_, T_, S_) do {                                                                                     \
    if (m16) {                                                                                                             \
      if (trans_out) hipLaunchKernelGGL((wgrad_taps16_bf16s_kernel<BM_, BN_, T_, S_, true>), grid, block, 0, st, g);      \
      else hipLaunchKernelGGL((wgrad_taps16_bf16s_kernel<BM_, BN_, T_, S_, false>), grid, block, 0, st, g);               \
    } else {                                                                                                               \
      if (trans_out) hipLaunchKernelGGL((wgrad_taps_bf16s_kernel<BM_, BN_, T_, S_, true>), grid, block, 0, st, g);        \
      else hipLaunchKernelGGL((wgrad_taps_bf16s_kernel<BM_, BN_, T_, S_, false>), grid, block, 0, st, g);                 \
    } } while (0)
  if (g.dil != 1 || (g.ss != 1 && g.ss != 2) || (g.T != 5 && g.T != 6)) { set_error("wgrad taps: built for 5 or 6 taps, stride 1 or 2, dilation 1"); return SVAE_ERR_SHAPE; }
#define SVAE_WTT(BM_, BN_, S_) do { if (g.T == 5) SVAE_WT(BM_, BN_, 5, S_); else SVAE_WT(BM_, BN_, 6, S_); } while (0)
  if (bm == 128 && bn == 128 && g.T == 5) { if (g.ss == 1) SVAE_WT(128, 128, 5, 1); else SVAE_WT(128, 128, 5, 2); }
  else if (bm == 64 && bn == 128) { if (g.ss == 1) SVAE_WTT(64, 128, 1); else SVAE_WTT(64, 128, 2); }
  else if (bm == 128 && bn == 64) { if (g.ss == 1) SVAE_WTT(128, 64, 1); else SVAE_WTT(128, 64, 2); }
  else { set_error("wgrad taps: tile %dx%d with %d taps unsupported", bm, bn, g.T); return SVAE_ERR_SHAPE; }
#undef SVAE_WTT
#undef SVAE_WT
  return check_launch("wgrad_taps_bf16s");
}

template <int BM, int BN, int NSTAGE>
static void launch_wgrad_p(const WgradArgs& g, dim3 grid, hipStream_t st, int pieces) {
  if (pieces == 3) hipLaunchKernelGGL((wgrad_gemm_bf16s_kernel<BM, BN, 3, NSTAGE>), grid, dim3(256), 0, st, g);
  else if (pieces == 2) hipLaunchKernelGGL((wgrad_gemm_bf16s_kernel<BM, BN, 2, NSTAGE>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((wgrad_gemm_bf16s_kernel<BM, BN, 1, NSTAGE>), grid, dim3(256), 0, st, g);
}

// 8-wave tiles with a 256 edge (2 or 3 pieces; LDS: 2 pieces 50 / 38 KB per stage for 256x256 / 256x128, 3 pieces 75 / 56 KB)
// (16 waves of 64 x 64 on the 256 x 256 tile measured no better than these 8: 174 vs 164 us on the largest layer)
template <int BM, int BN, int NSTAGE, int WR, int WC>
static int launch_wgrad_big(const WgradArgs& g, dim3 grid, hipStream_t st, int pieces) {
  if (pieces == 2) hipLaunchKernelGGL((wgrad_gemm_bf16s_kernel<BM, BN, 2, NSTAGE, WR, WC>), grid, dim3(64 * WR * WC), 0, st, g);
  else if (pieces == 3 && NSTAGE == 1) hipLaunchKernelGGL((wgrad_gemm_bf16s_kernel<BM, BN, 3, 1, WR, WC>), grid, dim3(64 * WR * WC), 0, st, g);
  else { set_error("wgrad split: the 256-edge tiles are built for 2 pieces (3 pieces: single LDS buffer only)"); return SVAE_ERR_SHAPE; }
  return SVAE_OK;
}

// variant 0: double-buffered LDS; 1: single LDS buffer
int launch_wgrad_split(const WgradArgs& g_in, dim3 grid, hipStream_t st, int bm, int bn, int pieces, int variant) {
  WgradArgs g = g_in;
  g.xmap = (variant >> 1) & 1;  // variants 2 / 3: as 0 / 1 with the XCD-aware workgroup order (bit 16: taps folded into the columns, set up by the caller)
  variant &= 1;
#define SVAE_WG_CASE(BM_, BN_)                                            \
  if (bm == BM_ && bn == BN_) {                                           \
    if (variant == 1) launch_wgrad_p<BM_, BN_, 1>(g, grid, st, pieces);   \
    else launch_wgrad_p<BM_, BN_, 2>(g, grid, st, pieces);                \
  }
  SVAE_WG_CASE(128, 128) else SVAE_WG_CASE(128, 64) else SVAE_WG_CASE(64, 128) else SVAE_WG_CASE(64, 64)
#undef SVAE_WG_CASE
#define SVAE_WG_BIG(BM_, BN_, WR_, WC_)                                                                   \
  else if (bm == BM_ && bn == BN_) {                                                                      \
    if (int e = (variant == 1 ? launch_wgrad_big<BM_, BN_, 1, WR_, WC_>(g, grid, st, pieces)              \
                              : launch_wgrad_big<BM_, BN_, 2, WR_, WC_>(g, grid, st, pieces))) return e;  \
  }
  SVAE_WG_BIG(256, 256, 2, 4) SVAE_WG_BIG(256, 128, 4, 2) SVAE_WG_BIG(128, 256, 2, 4)
#undef SVAE_WG_BIG
  else { set_error("wgrad split: tile %dx%d unsupported", bm, bn); return SVAE_ERR_SHAPE; }
  return check_launch("wgrad_gemm_bf16s");
}

// ------------------------------------------------------------------------------ host side
// layout of the split-weight buffer: [Wf: 3 bf16 planes][Wd: 3 bf16 planes][Wfh: 2 fp16 planes of 2^10 w], plane sizes in elements
static inline long long plane_f(const svae_conv_desc* d) { return (long long)d->kernel * ((d->c_in + 31) / 32) * 32 * d->c_out; }
static inline long long plane_d(const svae_conv_desc* d) { return (long long)d->kernel * ((d->c_out + 31) / 32) * 32 * d->c_in; }

template <int BM, int BN, int WR, int WC, int NSTAGE, int MINW>
static void launch_split_p(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces) {
  const dim3 block(64 * WR * WC);
  if (pieces == 3) hipLaunchKernelGGL((gather_gemm_bf16s_kernel<BM, BN, 3, WR, WC, NSTAGE, MINW>), grid, block, 0, st, sa);
  else if (pieces == SVAE_PIECES_F16X2) hipLaunchKernelGGL((gather_gemm_bf16s_kernel<BM, BN, 2, WR, WC, NSTAGE, MINW, true>), grid, block, 0, st, sa);
  else if (pieces == 2) hipLaunchKernelGGL((gather_gemm_bf16s_kernel<BM, BN, 2, WR, WC, NSTAGE, MINW>), grid, block, 0, st, sa);
  else hipLaunchKernelGGL((gather_gemm_bf16s_kernel<BM, BN, 1, WR, WC, NSTAGE, MINW>), grid, block, 0, st, sa);
}

template <int BM, int BN, int CWR, int CWC, int D>
static void launch_split_ws(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces) {
  const dim3 block(64 * (4 + CWR * CWC));
  if (pieces == 3) hipLaunchKernelGGL((gather_gemm_bf16s_ws_kernel<BM, BN, 3, CWR, CWC, D>), grid, block, 0, st, sa);
  else if (pieces == SVAE_PIECES_F16X2) hipLaunchKernelGGL((gather_gemm_bf16s_ws_kernel<BM, BN, 2, CWR, CWC, D, 0, true>), grid, block, 0, st, sa);
  else if (pieces == 2) hipLaunchKernelGGL((gather_gemm_bf16s_ws_kernel<BM, BN, 2, CWR, CWC, D>), grid, block, 0, st, sa);
  else hipLaunchKernelGGL((gather_gemm_bf16s_ws_kernel<BM, BN, 1, CWR, CWC, D>), grid, block, 0, st, sa);
}

#ifdef SVAE_ABLATION_KERNELS  // timing experiments with parts of the work removed (wrong results): python scrubvae_amd/build.py --ablation
static void launch_split_dbg(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int dbg) {
  const dim3 block(64 * 12);
#define SVAE_DBG_CASE(N) case N: hipLaunchKernelGGL((gather_gemm_bf16s_ws_kernel<128, 128, 3, 4, 2, 2, N>), grid, block, 0, st, sa); break;
  switch (dbg) {
    SVAE_DBG_CASE(0) SVAE_DBG_CASE(1) SVAE_DBG_CASE(2) SVAE_DBG_CASE(3) SVAE_DBG_CASE(4) SVAE_DBG_CASE(8) SVAE_DBG_CASE(12)
    SVAE_DBG_CASE(5) SVAE_DBG_CASE(13) SVAE_DBG_CASE(15) SVAE_DBG_CASE(7)
    default: break;
  }
#undef SVAE_DBG_CASE
}
#endif

// upper bound of the halo image rows over all BM-row tiles of both phases
static int halo_rows(const GatherArgs& g, int bm) {
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

template <int BN>
static int launch_halo(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  if (pieces != 3 && pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the halo kernel is built for 2 or 3 pieces"); return SVAE_ERR_SHAPE; }
  if (rows > 264) { set_error("split gather: halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  if (pieces == SVAE_PIECES_F16X2) {
    if (rows <= 160) hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 2, 4, 2, 160, true>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 2, 4, 2, 264, true>), grid, dim3(512), 0, st, sa);
  } else if (pieces == 3) {
    if (rows <= 160) hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 3, 4, 2, 160>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 3, 4, 2, 264>), grid, dim3(512), 0, st, sa);
  } else {
    if (rows <= 160) hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 2, 4, 2, 160>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, 2, 4, 2, 264>), grid, dim3(512), 0, st, sa);
  }
  return SVAE_OK;
}

// V = 9: the halo kernel on 256-row tiles (code 9128NNN; 8 waves of 64 x 64): half the weight-piece bytes per FLOP through
// the vector-memory path; pays once the problem has >= 2 x 256 such row tiles (batch >= 2048 for the deep layers)
template <int BN>
static int launch_halo256(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  if (pieces != 3 && pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the halo kernel is built for 2 or 3 pieces"); return SVAE_ERR_SHAPE; }
  if (rows > 528 || (pieces == 3 && rows > 320 && BN > 64)) { set_error("split gather: 256-row halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  if (pieces == SVAE_PIECES_F16X2) {
    if (rows <= 320) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 320, true>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 528, true>), grid, dim3(512), 0, st, sa);
  } else if (pieces == 3) {
    if (rows <= 320) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 3, 4, 2, 320>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 64, 3, 4, 2, 528>), grid, dim3(512), 0, st, sa);
  } else {
    if (rows <= 320) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 320>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 528>), grid, dim3(512), 0, st, sa);
  }
  return SVAE_OK;
}

// V = 19: the halo kernel on 256 x 160 tiles, 8 x 1 waves of 32 x 160 (code 19128128; the fields of the code are placeholders).  For
// the output conv (141 -> 144 channels): ONE column tile instead of three 64-wide ones -- 10 % instead of 25 % of the matrix work on
// padding columns, and the activation image of a channel block staged once instead of three times.  Two pieces (bf16 or fp16).
static int launch_halo256_wide(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  if (pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the 256 x 160 halo tile is built for 2 pieces"); return SVAE_ERR_SHAPE; }
  if (rows > 528) { set_error("split gather: 256-row halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  if (pieces == SVAE_PIECES_F16X2) {
    if (rows <= 320) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 160, 2, 8, 1, 320, true>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 160, 2, 8, 1, 528, true>), grid, dim3(512), 0, st, sa);
  } else {
    if (rows <= 320) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 160, 2, 8, 1, 320>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 160, 2, 8, 1, 528>), grid, dim3(512), 0, st, sa);
  }
  return SVAE_OK;
}

// V = 10 / 11: the wave-specialised halo kernel (8 consumer + 2 producer waves) on 128- / 256-row tiles (row field of the code: 128).
// Instantiated where two image buffers + two weight stages fit the 160 KiB of LDS.
template <int BM, int BN>
static int launch_halo_ws(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  const dim3 block(64 * 10);
  if (pieces != 3 && pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the halo kernels are built for 2 or 3 pieces"); return SVAE_ERR_SHAPE; }
#define SVAE_HWS(P_, R_) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<BM, BN, P_, 4, 2, R_>), grid, block, 0, st, sa)
#define SVAE_HWSH(R_) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<BM, BN, 2, 4, 2, R_, 0, false, 2, true>), grid, block, 0, st, sa)
  if constexpr (BM == 128) {
    if (rows <= 160) { if (pieces == 3) SVAE_HWS(3, 160); else if (pieces == 2) SVAE_HWS(2, 160); else SVAE_HWSH(160); }
    else if (rows <= 264) { if (pieces == 3) SVAE_HWS(3, 264); else if (pieces == 2) SVAE_HWS(2, 264); else SVAE_HWSH(264); }
    else { set_error("split gather: halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  } else {
    if (rows <= 264) { if (pieces == 3) SVAE_HWS(3, 264); else if (pieces == 2) SVAE_HWS(2, 264); else SVAE_HWSH(264); }
    else if (rows <= 320 && pieces == 2) SVAE_HWS(2, 320);
    else if (rows <= 320 && pieces == SVAE_PIECES_F16X2) SVAE_HWSH(320);
    else { set_error("split gather: 256-row halo image of %d rows does not fit twice", rows); return SVAE_ERR_SHAPE; }
  }
#undef SVAE_HWS
#undef SVAE_HWSH
  return SVAE_OK;
}

// V = 12 / 13: the same kernel with three weight-tile buffers (the producers request two stages ahead): 2 pieces on BN = 128 or 64,
// 3 pieces on BN = 64 (LDS)
template <int BM, int BN>
static int launch_halo_ws_pipe(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  const dim3 block(64 * 10);
  if (pieces == SVAE_PIECES_F16X2) {  // three weight buffers, fp16 pieces
#define SVAE_HWPH(R_) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<BM, BN, 2, 4, 2, R_, 0, false, 3, true>), grid, block, 0, st, sa)
    constexpr int R0h = BM == 128 ? 160 : 264;
    if (rows <= R0h) SVAE_HWPH(R0h);
    else if (BM == 128 && rows <= 264) SVAE_HWPH(264);
    else { set_error("split gather: halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
#undef SVAE_HWPH
    return SVAE_OK;
  }
  if (pieces != 3 && pieces != 2) { set_error("split gather: the halo kernels are built for 2 or 3 pieces"); return SVAE_ERR_SHAPE; }
  if (pieces == 3 && BN != 64) { set_error("split gather: three weight buffers with 3 pieces exist for 64-column tiles only"); return SVAE_ERR_SHAPE; }
#define SVAE_HWP(P_, R_) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<BM, BN, P_, 4, 2, R_, 0, false, 3>), grid, block, 0, st, sa)
  constexpr int R0 = BM == 128 ? 160 : 264;
  if (rows <= R0) {
    if (pieces == 2) SVAE_HWP(2, R0);
    else if constexpr (BN == 64) SVAE_HWP(3, R0);
  } else if (BM == 128 && rows <= 264) {
    if (pieces == 2) SVAE_HWP(2, 264);
    else if constexpr (BN == 64) SVAE_HWP(3, 264);
  } else { set_error("split gather: halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
#undef SVAE_HWP
  return SVAE_OK;
}

// V = 14 / 15: FOUR consumer waves with 128 x 64 wave tiles (one per SIMD, 256 registers each) + 2 producers on the 256 x 128 tile:
// every operand fragment feeds 4 or 2 MFMA groups instead of 2, i.e. 25 % fewer LDS fragment bytes per MFMA -- the resource the
// ablations show is NOT overlapped with the matrix pipe.  14: compiler-scheduled pipeline, 15: pinned (sched_barrier).  2 pieces only.
template <bool PIPE_>
static int launch_halo_ws_fat(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  const dim3 block(64 * 6);
  if (pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the 128 x 64 wave tiles are built for 2 pieces"); return SVAE_ERR_SHAPE; }
#define SVAE_HWF(R_, H_) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<256, 128, 2, 2, 2, R_, 0, PIPE_, 3, H_>), grid, block, 0, st, sa)
  if (rows <= 264) { if (pieces == 2) SVAE_HWF(264, false); else SVAE_HWF(264, true); }
  else if (rows <= 320) { if (pieces == 2) SVAE_HWF(320, false); else SVAE_HWF(320, true); }
  else { set_error("split gather: 256-row halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
#undef SVAE_HWF
  return SVAE_OK;
}

// V = 16 / 17: gather_halo_ws4_bf16s_kernel (8 consumer + 4 loader waves, 256-row tiles; row field of the code: 128) with / without
// the quarter-stage stagger of the consumers' second half;  V = 18: the same kernel on v_mfma_f32_16x16x32 (gather_halo_ws4m_bf16s_kernel).  Two pieces (bf16 or fp16) only.  The raw staging ring holds
// three slices of ceil(image rows / ntaps) rows: 88 rows beside 264-row images (ntaps >= 3), 64 beside 320-row images (ntaps >= 5).
static int ws4_slice_rows(const GatherArgs& g, int rmax) {
  int worst = 0;
  for (int p = 0; p < 2; ++p) {
    if (g.M[p] <= 0 || g.ntaps[p] <= 0) continue;
    const int sr = (((rmax + g.ntaps[p] - 1) / g.ntaps[p]) + 7) & ~7;
    worst = sr > worst ? sr : worst;
  }
  return worst;
}
template <int BN, bool STAG, bool M16 = false>
static int launch_halo_ws4(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  const dim3 block(768);
  if (pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the 12-wave halo kernel is built for 2 pieces"); return SVAE_ERR_SHAPE; }
  const int rmax = rows <= 264 ? 264 : 320;
  if (rows > 320 || ws4_slice_rows(sa.g, rmax) > (rmax == 264 ? 88 : 64)) {
    set_error("split gather: 256-row halo image of %d rows / its raw slices do not fit", rows);
    return SVAE_ERR_SHAPE;
  }
#define SVAE_HW4(R_, S_, H_)                                                                                         \
  do {                                                                                                               \
    if constexpr (M16) hipLaunchKernelGGL((gather_halo_ws4m_bf16s_kernel<256, BN, R_, S_, H_>), grid, block, 0, st, sa); \
    else hipLaunchKernelGGL((gather_halo_ws4_bf16s_kernel<256, BN, R_, S_, H_, STAG>), grid, block, 0, st, sa);       \
  } while (0)
  if (pieces == 2) { if (rmax == 264) SVAE_HW4(264, 88, false); else SVAE_HW4(320, 64, false); }
  else { if (rmax == 264) SVAE_HW4(264, 88, true); else SVAE_HW4(320, 64, true); }
#undef SVAE_HW4
  return SVAE_OK;
}

// tile code V*1000000 + BM*1000 + BN.  V = 0: 4 waves, double-buffered LDS;  1: 4 waves, single LDS buffer;
// 2: 8 waves (4x2), single buffer;  3: 8 waves, double-buffered (BM = 128 only);
// 4: wave-specialised, 4 producer + 8 consumer waves, 2 tiles in flight;  5: same with 4 consumers;
// 6 / 7: as 4 / 5 with 3 tiles in flight;  8: halo image kernel (BM = 128, 8 waves)
static int launch_split_gather(SplitGatherArgs& sa, hipStream_t st, int code, int pieces) {
  GatherArgs& g = sa.g;
  Tile t;
  if (!decode_tile(code, t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; }
  const int v = t.dma;
#ifdef SVAE_ABLATION_KERNELS
  if ((v == 37 || v == 38) && (pieces == 2 || pieces == SVAE_PIECES_F16X2)) {  // stamped gather_halo_ws4_bf16s_kernel (tools/stamp_halo.py)
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    dim3 gridw(nb, (g.N + 127) / 128), block(768);
    if (halo_rows(g, 256) > 264) { set_error("ablation: image does not fit"); return SVAE_ERR_SHAPE; }
    if (v == 37) {
      if (pieces == 2) hipLaunchKernelGGL((gather_halo_ws4_bf16s_kernel<256, 128, 264, 88, false, true, 16>), gridw, block, 0, st, sa);
      else hipLaunchKernelGGL((gather_halo_ws4_bf16s_kernel<256, 128, 264, 88, true, true, 16>), gridw, block, 0, st, sa);
    } else {
      if (pieces == 2) hipLaunchKernelGGL((gather_halo_ws4_bf16s_kernel<256, 128, 264, 88, false, false, 16>), gridw, block, 0, st, sa);
      else hipLaunchKernelGGL((gather_halo_ws4_bf16s_kernel<256, 128, 264, 88, true, false, 16>), gridw, block, 0, st, sa);
    }
    return check_launch("gather_halo_ws4_bf16s<stamps>");
  }
  if (((v >= 20 && v < 28) || v == 36) && (pieces == 3 || pieces == 2)) {  // timing experiments on the 256 x 128 wave-specialised halo kernel: V = 20 + DBG
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    dim3 gridw(nb, (g.N + 127) / 128), block(640);
    if (halo_rows(g, 256) > 264) { set_error("ablation: image does not fit"); return SVAE_ERR_SHAPE; }
#define SVAE_HD(D_) case D_: if (pieces == 3) hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<256, 128, 3, 4, 2, 264, D_>), gridw, block, 0, st, sa); \
                             else hipLaunchKernelGGL((gather_halo_ws_bf16s_kernel<256, 128, 2, 4, 2, 264, D_, false, 3>), gridw, block, 0, st, sa); break;
    switch (v - 20) { SVAE_HD(0) SVAE_HD(1) SVAE_HD(2) SVAE_HD(3) SVAE_HD(4) SVAE_HD(5) SVAE_HD(6) SVAE_HD(7) SVAE_HD(16) default: break; }
#undef SVAE_HD
    return check_launch("gather_halo_ws_bf16s<dbg>");
  }
#endif
  if (v == 19) {  // 256 x 160 tiles
    if (t.bm != 128 || t.bn != 128) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    if (int e = launch_halo256_wide(sa, dim3(nb, (g.N + 159) / 160), st, pieces, halo_rows(g, 256))) return e;
    return check_launch("gather_halo_bf16s<256,160>");
  }
  if (v == 16 || v == 17 || v == 18) {
    if (t.bm != 128 || (t.bn != 128 && t.bn != 64)) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    dim3 gridw(nb, (g.N + t.bn - 1) / t.bn);
    const int rows = halo_rows(g, 256);
    if (!plan_is_affine(g)) { set_error("split gather: tap tables are not arithmetic progressions"); return SVAE_ERR_SHAPE; }
    int e;
    if (v == 16) e = t.bn == 128 ? launch_halo_ws4<128, true>(sa, gridw, st, pieces, rows) : launch_halo_ws4<64, true>(sa, gridw, st, pieces, rows);
    else if (v == 18) e = t.bn == 128 ? launch_halo_ws4<128, true, true>(sa, gridw, st, pieces, rows) : launch_halo_ws4<64, true, true>(sa, gridw, st, pieces, rows);
    else e = t.bn == 128 ? launch_halo_ws4<128, false>(sa, gridw, st, pieces, rows) : launch_halo_ws4<64, false>(sa, gridw, st, pieces, rows);
    if (e) return e;
    return check_launch("gather_halo_ws4_bf16s");
  }
  if (v >= 10 && v <= 15) {
    if (t.bm != 128 || (v >= 14 && t.bn != 128)) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    const int bmr = (v == 11 || v >= 13) ? 256 : 128;
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + bmr - 1) / bmr);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    dim3 gridw(nb, (g.N + t.bn - 1) / t.bn);
    const int rows = halo_rows(g, bmr);
    if (!plan_is_affine(g)) { set_error("split gather: tap tables are not arithmetic progressions"); return SVAE_ERR_SHAPE; }
    int e;
    if (v == 10) e = t.bn == 128 ? launch_halo_ws<128, 128>(sa, gridw, st, pieces, rows) : launch_halo_ws<128, 64>(sa, gridw, st, pieces, rows);
    else if (v == 11) e = t.bn == 128 ? launch_halo_ws<256, 128>(sa, gridw, st, pieces, rows) : launch_halo_ws<256, 64>(sa, gridw, st, pieces, rows);
    else if (v == 12) e = t.bn == 128 ? launch_halo_ws_pipe<128, 128>(sa, gridw, st, pieces, rows) : launch_halo_ws_pipe<128, 64>(sa, gridw, st, pieces, rows);
    else if (v == 14) e = launch_halo_ws_fat<false>(sa, gridw, st, pieces, rows);
    else if (v == 15) e = launch_halo_ws_fat<true>(sa, gridw, st, pieces, rows);
    else e = t.bn == 128 ? launch_halo_ws_pipe<256, 128>(sa, gridw, st, pieces, rows) : launch_halo_ws_pipe<256, 64>(sa, gridw, st, pieces, rows);
    if (e) return e;
    return check_launch("gather_halo_ws_bf16s");
  }
  if (v == 9) {  // 256-row halo tiles, encoded with a 128 row field
    if (t.bm != 128) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    dim3 grid9(nb, (g.N + t.bn - 1) / t.bn);
    if (int e = (t.bn == 128 ? launch_halo256<128>(sa, grid9, st, pieces, halo_rows(g, 256))
                             : launch_halo256<64>(sa, grid9, st, pieces, halo_rows(g, 256)))) return e;
    return check_launch("gather_halo_bf16s<256>");
  }
  for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + t.bm - 1) / t.bm);
  const int bm = g.blocks_m[0] + g.blocks_m[1];
  if (bm == 0) return SVAE_OK;
  dim3 grid(bm, (g.N + t.bn - 1) / t.bn);
#ifdef SVAE_ABLATION_KERNELS
#define SVAE_ABLATION_CASE(BM_, BN_) \
  else if (v >= 10 && v < 26 && BM_ == 128 && BN_ == 128 && pieces == 3) launch_split_dbg(sa, grid, st, v - 10);
#else
#define SVAE_ABLATION_CASE(BM_, BN_)
#endif
#define SVAE_SPLIT_CASE(BM_, BN_)                                                          \
  if (t.bm == BM_ && t.bn == BN_) {                                                        \
    if (v == 0) launch_split_p<BM_, BN_, 2, 2, 2, 1>(sa, grid, st, pieces);                \
    else if (v == 1) launch_split_p<BM_, BN_, 2, 2, 1, 2>(sa, grid, st, pieces);           \
    else if (v == 2 && BM_ == 128) launch_split_p<128, BN_, 4, 2, 1, 2>(sa, grid, st, pieces); \
    else if (v == 3 && BM_ == 128) launch_split_p<128, BN_, 4, 2, 2, 1>(sa, grid, st, pieces); \
    else if (v == 4 && BM_ == 128) launch_split_ws<128, BN_, 4, 2, 2>(sa, grid, st, pieces);  \
    else if (v == 4 && BN_ == 128) launch_split_ws<64, 128, 2, 4, 2>(sa, grid, st, pieces);   \
    else if (v == 5) launch_split_ws<BM_, BN_, 2, 2, 2>(sa, grid, st, pieces);                \
    else if (v == 6 && BM_ == 128) launch_split_ws<128, BN_, 4, 2, 3>(sa, grid, st, pieces);  \
    else if (v == 6 && BN_ == 128) launch_split_ws<64, 128, 2, 4, 3>(sa, grid, st, pieces);   \
    else if (v == 7) launch_split_ws<BM_, BN_, 2, 2, 3>(sa, grid, st, pieces);                \
    else if (v == 8 && BM_ == 128) { if (int e = launch_halo<BN_>(sa, grid, st, pieces, halo_rows(g, 128))) return e; } \
    SVAE_ABLATION_CASE(BM_, BN_)                                                           \
    else { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; } \
  }
  SVAE_SPLIT_CASE(128, 128) else SVAE_SPLIT_CASE(128, 64) else SVAE_SPLIT_CASE(64, 128) else SVAE_SPLIT_CASE(64, 64)
#undef SVAE_SPLIT_CASE
  return check_launch("gather_gemm_bf16s");
}

}  // namespace svae

using namespace svae;

#ifdef SVAE_ABLATION_KERNELS
extern "C" int svae_debug_stamp_buffer(void* buf) {  // diagnostic build only: [10 waves][64 stages][8] ticks
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(svae::g_stamp_buf), &p, sizeof(p)) == hipSuccess ? SVAE_OK : SVAE_ERR_LAUNCH;
}
#endif

extern "C" size_t svae_conv_split_bytes(const svae_conv_desc* d) {
  if (validate(d)) return 0;
  return (size_t)(3 * (plane_f(d) + plane_d(d)) + 2 * plane_f(d)) * sizeof(unsigned short);
}

extern "C" int svae_conv_split_weights(const svae_conv_desc* d, const float* w, void* wsplit, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(w && wsplit && aligned16(w) && aligned16(wsplit), SVAE_ERR_ARG, "conv_split_weights: null / unaligned pointer");
  unsigned short* out = (unsigned short*)wsplit;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(split_weights_kernel, dim3((d->c_out + 31) / 32, (d->c_in + 31) / 32, d->kernel), dim3(256), 0, st, w, out,
                     d->kernel, d->c_in, d->c_out, 0, plane_f(d), out + 3 * plane_f(d) + 3 * plane_d(d));
  hipLaunchKernelGGL(split_weights_kernel, dim3((d->c_in + 31) / 32, (d->c_out + 31) / 32, d->kernel), dim3(256), 0, st, w,
                     out + 3 * plane_f(d), d->kernel, d->c_in, d->c_out, 1, plane_d(d), (unsigned short*)nullptr);
  return check_launch("split_weights");
}

// row tiles of the split forward launch for the descriptor's current tile choice
extern "C" int svae_conv_fwd_stats_tiles(const svae_conv_desc* d) {
  if (validate(d)) return 0;
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.N = d->c_out;
  build_plan(g, d, !d->transposed, d->l_out, d->l_in);
  Tile t;
  if (!decode_tile(d->tile[0], t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; }
  const int bm = (t.dma == 9 || t.dma == 11 || t.dma >= 13) ? 256 : t.bm;
  return (int)((g.M[0] + bm - 1) / bm + (g.M[1] + bm - 1) / bm);
}

extern "C" int svae_conv_fwd_split(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                   int accumulate, int pieces, void* stream) {
  return svae_conv_fwd_split_stats(d, x, wsplit, bias, y, accumulate, pieces, nullptr, stream);
}

extern "C" int svae_conv_fwd_split_stats(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                         int accumulate, int pieces, float* bn_part, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(x && wsplit && y, SVAE_ERR_ARG, "conv_fwd_split: null pointer");
  SVAE_REQUIRE(aligned16(x) && aligned16(wsplit) && aligned16(y), SVAE_ERR_ALIGN, "conv_fwd_split: pointers must be 16-byte aligned");
  SVAE_REQUIRE((pieces >= 1 && pieces <= 3) || pieces == SVAE_PIECES_F16X2, SVAE_ERR_ARG, "conv_fwd_split: pieces %d not in 1..3 / 22", pieces);
  SplitGatherArgs sa;
  memset(&sa, 0, sizeof(sa));
  GatherArgs& g = sa.g;
  sa.Wp = (const unsigned short*)wsplit;  // Wf planes
  if (pieces == SVAE_PIECES_F16X2) sa.Wp += 3 * plane_f(d) + 3 * plane_d(d);  // the fp16 planes
  sa.w_piece_stride = plane_f(d);
  sa.KB = (d->c_in + 31) / 32;
  sa.rowsA = (long long)d->batch * d->l_in;
  g.A = x; g.bias = bias; g.C = y;
  g.Kc = d->c_in; g.ldA = d->ld_in; g.ldC = d->ld_out;
  g.N = d->c_out;
  g.accumulate = accumulate;
  g.stats = bn_part;
  build_plan(g, d, /*strided=*/!d->transposed, d->l_out, d->l_in);
  return launch_split_gather(sa, (hipStream_t)stream, d->tile[0], pieces);
}

// row tiles / column blocks of the split data-gradient launch for the descriptor's current tile choice
extern "C" int svae_conv_dgrad_stats_tiles(const svae_conv_desc* d, int* col_blocks) {
  if (validate(d)) return 0;
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.N = d->c_in;
  build_plan(g, d, d->transposed != 0, d->l_in, d->l_out);
  Tile t;
  if (!decode_tile(d->tile[1], t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; }
  const int bm = (t.dma == 9 || t.dma == 11 || t.dma >= 13) ? 256 : t.bm;
  if (col_blocks) *col_blocks = t.dma == 19 ? (g.N + 159) / 160 : (g.N + t.bn - 1) / t.bn;
  return (int)((g.M[0] + bm - 1) / bm + (g.M[1] + bm - 1) / bm);
}

extern "C" int svae_conv_dgrad_split(const svae_conv_desc* d, const float* dy, const void* wsplit, float* dx, int accumulate,
                                     int pieces, void* stream) {
  return svae_conv_dgrad_split_bn(d, dy, wsplit, dx, accumulate, pieces, nullptr, stream);
}

extern "C" int svae_conv_dgrad_split_bn(const svae_conv_desc* d, const float* dy, const void* wsplit, float* dx, int accumulate,
                                        int pieces, const svae_bn_bwd_fuse* f, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(dy && wsplit && dx, SVAE_ERR_ARG, "conv_dgrad_split: null pointer");
  SVAE_REQUIRE(aligned16(dy) && aligned16(wsplit) && aligned16(dx), SVAE_ERR_ALIGN, "conv_dgrad_split: pointers must be 16-byte aligned");
  SVAE_REQUIRE(pieces >= 1 && pieces <= 3, SVAE_ERR_ARG, "conv_dgrad_split: pieces %d not in 1..3", pieces);
  SplitGatherArgs sa;
  memset(&sa, 0, sizeof(sa));
  GatherArgs& g = sa.g;
  sa.Wp = (const unsigned short*)wsplit + 3 * plane_f(d);  // Wd planes
  sa.w_piece_stride = plane_d(d);
  sa.KB = (d->c_out + 31) / 32;
  sa.rowsA = (long long)d->batch * d->l_out;
  g.A = dy; g.bias = nullptr; g.C = dx;
  g.Kc = d->c_out; g.ldA = d->ld_out; g.ldC = d->ld_in;
  g.N = d->c_in;
  g.accumulate = accumulate;
  if (f) {
    SVAE_REQUIRE(f->x && f->part && (!f->scale || f->shift) && (!f->mean || f->rstd), SVAE_ERR_ARG, "conv_dgrad_split_bn: bad fuse descriptor");
    g.stats = f->part; g.bn_x = f->x; g.bn_scale = f->scale; g.bn_shift = f->shift; g.bn_mean = f->mean; g.bn_rstd = f->rstd;
    g.bn_alpha = f->alpha; g.bn_dalpha = f->dalpha_part;
  }
  build_plan(g, d, /*strided=*/d->transposed != 0, d->l_in, d->l_out);
  return launch_split_gather(sa, (hipStream_t)stream, d->tile[1], pieces);
}

/* kernel the split dispatcher uses for kind 0 (fwd) / 1 (dgrad): workgroup tile, variant and -- for the halo
 * variant -- the image rows it is instantiated with (0 otherwise) */
extern "C" int svae_conv_split_tile(const svae_conv_desc* d, int kind, int* bm, int* bn, int* variant, int* rmax) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(bm && bn && variant && rmax && (kind == 0 || kind == 1), SVAE_ERR_ARG, "conv_split_tile: bad args");
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  if (kind == 0) { g.N = d->c_out; build_plan(g, d, !d->transposed, d->l_out, d->l_in); }
  else { g.N = d->c_in; build_plan(g, d, d->transposed != 0, d->l_in, d->l_out); }
  Tile t;
  if (!decode_tile(d->tile[kind], t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; }
  *bm = t.bm; *bn = t.bn; *variant = t.dma; *rmax = 0;
  if (t.dma == 8) { const int r = halo_rows(g, 128); *rmax = r <= 160 ? 160 : 264; }
  if (t.dma == 9) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 320 ? 320 : 528; }
  if (t.dma == 10 || t.dma == 12) { const int r = halo_rows(g, 128); *rmax = r <= 160 ? 160 : 264; }
  if (t.dma == 11) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma == 13) { *bm = 256; *rmax = 264; }
  if (t.dma == 14 || t.dma == 15) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma >= 16 && t.dma <= 18) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma == 19) { const int r = halo_rows(g, 256); *bm = 256; *bn = 160; *rmax = r <= 320 ? 320 : 528; }
  return SVAE_OK;
}

extern "C" int svae_conv_split_weights_batched(const svae_split_task* tasks, int n, void* stream) {
  SVAE_REQUIRE(tasks && n > 0 && n <= SVAE_MAX_SPLIT_TASKS, SVAE_ERR_ARG, "split_weights_batched: 1..%d tasks", SVAE_MAX_SPLIT_TASKS);
  SplitTasks ts;
  memset(&ts, 0, sizeof(ts));
  int total = 0;
  for (int i = 0; i < n; ++i) {
    const svae_split_task& k = tasks[i];
    SVAE_REQUIRE(k.w && k.wsplit && aligned16(k.w) && aligned16(k.wsplit) && k.kernel >= 1 && k.c_in > 0 && k.c_out > 0 &&
                     k.c_in % 16 == 0 && k.c_out % 16 == 0,
                 SVAE_ERR_ARG, "split_weights_batched: bad task %d", i);
    ts.w[i] = k.w; ts.out[i] = (unsigned short*)k.wsplit;
    ts.T[i] = k.kernel; ts.Cin[i] = k.c_in; ts.Cout[i] = k.c_out;
    ts.first[i] = total;
    total += 2 * k.kernel * ((k.c_in + 31) / 32) * ((k.c_out + 31) / 32);
  }
  ts.first[n] = total;
  ts.n = n;
  hipLaunchKernelGGL(split_weights_batched_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, ts);
  return check_launch("split_weights_batched");
}
