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
// block shared by all taps; 8 / 10 / 12 waves), split_weights*.  The weight-gradient kernels live in wgrad_bf16s.hip.
#include "split_gather.h"

namespace svae {


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
// UP: the gathered operand is the x2 linear upsample (align_corners = false; reference residual.py:160, the decoder's skip path) of
// the half-length tensor g.A points to: image row (b, ro) = 0.75 x[b, ro / 2] + 0.25 x[b, ro / 2 -/+ 1] (clamped at the sample's
// ends) is blended from two source rows on its way into LDS -- the upsampled tensor never exists in HBM.
template <int BM, int BN, int P, int WR, int WC, int RMAX, bool H = false, bool UP = false>
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
  // UP: the SOURCE rows of the image (flat rows (gbase >> 1) - 1 .. of the half-length tensor: RMAX / 2 + 3 of them at most) are what
  // is fetched -- half as many requests as image rows -- and parked as raw fp32 rows in LDS; the image rows are blended from there
  constexpr int SRC_ROWS = UP ? RMAX / 2 + 8 : 0, SPASS = UP ? (SRC_ROWS + RPP - 1) / RPP : 0, RAW_BYTES = SRC_ROWS * SBK * 4;
  constexpr int RA_N = UP ? SPASS : APASS;
  __shared__ __attribute__((aligned(16))) unsigned char smem[A_IMG + 2 * B_STAGE + RAW_BYTES];
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
  long long a_goff[RA_N];                      // UP: offsets of the SOURCE rows this thread fetches
  bool s_ok[UP ? SPASS : 1];
  int a_pq[UP ? APASS : 1];                    // UP: raw rows (p | q << 16) image row i blends: 0.75 raw[p] + 0.25 raw[q]
  bool a_row_ok[APASS];
  bool up_store[UP ? APASS : 1];
  long long own_end = 0;  // UP: this tile owns the upsampled rows [amin, own_end): up to the next tile's first anchor / the end of the tensor
  if constexpr (UP) {
    const long long mn = m0 + BM;
    if (mn < Mp) { const long long bn_ = mn / nj; own_end = bn_ * g.Lin + (mn - bn_ * nj) * g.sj; }
    else own_end = sa.rowsA;
  }
#pragma unroll
  for (int i = 0; i < APASS; ++i) {
    const int r = (tid >> 3) + RPP * i;
    const long long grow = gbase + r;
    a_row_ok[i] = r < R && grow >= 0 && grow < sa.rowsA;
    if constexpr (UP) {
      up_store[i] = sa.up_out != nullptr && blockIdx.y == 0 && a_row_ok[i] && grow >= amin && grow < own_end;
      // image row (b, ro) = 0.75 x[b, ro / 2] + 0.25 x[b, ro / 2 -/+ 1] (clamped at the sample's ends); flat source row of the
      // first term: grow >> 1 (samples are Lin = 2 L image rows and L source rows long)
      const long long gr = a_row_ok[i] ? grow : gbase;
      const int ro = (int)(gr % g.Lin);
      const int pp = (int)((gr >> 1) - ((gbase >> 1) - 1));
      const int qq = (gr & 1) ? (ro == g.Lin - 1 ? pp : pp + 1) : (ro == 0 ? pp : pp - 1);
      a_pq[i] = a_row_ok[i] ? (pp | (qq << 16)) : 0;
    } else {
      a_goff[i] = (a_row_ok[i] ? grow : 0) * (long long)g.ldA;
    }
  }
  if constexpr (UP) {
    const long long s0 = (gbase >> 1) - 1, s_end = sa.rowsA >> 1;
#pragma unroll
    for (int i = 0; i < SPASS; ++i) {
      const int j = (tid >> 3) + RPP * i;
      const long long src = s0 + j;
      s_ok[i] = j < SRC_ROWS && src >= 0 && src < s_end;
      a_goff[i] = (s_ok[i] ? src : 0) * (long long)g.ldA;
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

  float4 ra[RA_N];
  bool ra_ok[RA_N];
  uint4 rb[BPASS];
  int st_c = 0;        // UP: channel offset / validity of the values load_a fetched last (store_a's blend and by-product store)
  bool st_kq = false;
  auto load_a = [&](int kb) {
    const int c0 = kb * SBK;
    const bool kq_ok = c0 + akq * 4 < g.Kc;
    const int cq = kq_ok ? c0 + akq * 4 : 0;
    if constexpr (UP) { st_c = cq; st_kq = kq_ok; }
#pragma unroll
    for (int i = 0; i < RA_N; ++i) {
      ra[i] = *reinterpret_cast<const float4*>(g.A + a_goff[i] + cq);
      if constexpr (UP) ra_ok[i] = s_ok[i] && kq_ok; else ra_ok[i] = a_row_ok[i] && kq_ok;
    }
  };
  auto store_a = [&]() {
    [[maybe_unused]] unsigned char* const raw = smem + A_IMG + 2 * B_STAGE;
    if constexpr (UP) {  // park the source rows, then blend the image rows from them
#pragma unroll
      for (int i = 0; i < SPASS; ++i) {
        const int j = (tid >> 3) + RPP * i;
        if ((i + 1) * RPP <= SRC_ROWS || j < SRC_ROWS)
          *reinterpret_cast<float4*>(raw + j * (SBK * 4) + akq * 16) = ra_ok[i] ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
      const int r = (tid >> 3) + RPP * i;
      uint2 pc[P];
      float4 v;
      bool v_ok;
      if constexpr (UP) {
        const float4 pv = *reinterpret_cast<const float4*>(raw + (a_pq[i] & 0xffff) * (SBK * 4) + akq * 16);
        const float4 qv = *reinterpret_cast<const float4*>(raw + (a_pq[i] >> 16) * (SBK * 4) + akq * 16);
        v = up2_blend4(pv, qv);
        v_ok = a_row_ok[i] && st_kq;
        // by-product: the upsampled rows this tile OWNS (first row's anchor up to the next tile's) go to sa.up_out -- the operand of the
        // conv's weight gradient in the backward pass -- from the first column tile only; halo rows belong to the neighbours
        if (up_store[i] && v_ok) *reinterpret_cast<float4*>(sa.up_out + (gbase + r) * (long long)g.ldA + st_c) = v;
      } else {
        v = ra[i];
        v_ok = ra_ok[i];
      }
      split4x<P, H>(v_ok ? v : make_float4(0.f, 0.f, 0.f, 0.f), pc);
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
    if constexpr (MT * NT >= 8) {
      // 64 x 128 wave tiles (the 256 x 256 workgroup tile): 128 accumulator registers -- the fragments of ONE k-step at a time
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int ch = ks * 2 + h;
        uint4 av[MT][P], bv[NT][P];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned mask = aval[mt] ? 0xffffffffu : 0u;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            uint4 v = *reinterpret_cast<const uint4*>(smem + p * A_PIECE + arow[mt] * ROWB + ((ch ^ swz(arow[mt])) << 4));
            v.x &= mask; v.y &= mask; v.z &= mask; v.w &= mask;
            av[mt][p] = v;
          }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int p = 0; p < P; ++p)
            bv[nt][p] = *reinterpret_cast<const uint4*>(bst + p * B_PIECE + b_addr[nt] + ((ch ^ b_sw[nt]) << 4));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = mfma_splitx<P, H>(av[mt], bv[nt], acc[mt][nt]);
      }
      return;
    }
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

template <int BN>
static int launch_halo(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  if (pieces != 3 && pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the halo kernel is built for 2 or 3 pieces"); return SVAE_ERR_SHAPE; }
  if (rows > 264) { set_error("split gather: halo image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  if (sa.g.up) {  // fused x2 upsample of the gathered operand
#define SVAE_HUP(P_, H_) do { if (rows <= 160) hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, P_, 4, 2, 160, H_, true>), grid, dim3(512), 0, st, sa); \
                              else hipLaunchKernelGGL((gather_halo_bf16s_kernel<128, BN, P_, 4, 2, 264, H_, true>), grid, dim3(512), 0, st, sa); } while (0)
    if (pieces == SVAE_PIECES_F16X2) SVAE_HUP(2, true); else if (pieces == 3) SVAE_HUP(3, false); else SVAE_HUP(2, false);
#undef SVAE_HUP
    return SVAE_OK;
  }
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
  if (sa.g.up) {  // fused x2 upsample of the gathered operand (images of up to 320 rows: 6-tap convs on 256-row tiles need 298)
    if (rows > 320) { set_error("split gather: the fused upsample exists for 256-row halo images of <= 320 rows (%d)", rows); return SVAE_ERR_SHAPE; }
    if (pieces == SVAE_PIECES_F16X2) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 320, true, true>), grid, dim3(512), 0, st, sa);
    else if (pieces == 3) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 3, 4, 2, 320, false, true>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, BN, 2, 4, 2, 320, false, true>), grid, dim3(512), 0, st, sa);
    return SVAE_OK;
  }
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

// V = 29: the halo kernel on 256 x 256 tiles, 4 x 2 waves of 64 x 128 (code 29128128; the fields of the code are placeholders): a
// quarter fewer operand bytes per multiply through the CU's fetch path than 256 x 128.  Two pieces, images of <= 320 rows.
static int launch_halo256x256(const SplitGatherArgs& sa, dim3 grid, hipStream_t st, int pieces, int rows) {
  if (pieces != 2 && pieces != SVAE_PIECES_F16X2) { set_error("split gather: the 256 x 256 halo tile is built for 2 pieces"); return SVAE_ERR_SHAPE; }
  if (rows > 320) { set_error("split gather: 256 x 256 halo tile: image of %d rows does not fit", rows); return SVAE_ERR_SHAPE; }
  if (sa.g.up) {  // fused x2 upsample of the gathered operand
    if (pieces == SVAE_PIECES_F16X2) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 256, 2, 4, 2, 320, true, true>), grid, dim3(512), 0, st, sa);
    else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 256, 2, 4, 2, 320, false, true>), grid, dim3(512), 0, st, sa);
    return SVAE_OK;
  }
  if (pieces == SVAE_PIECES_F16X2) hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 256, 2, 4, 2, 320, true>), grid, dim3(512), 0, st, sa);
  else hipLaunchKernelGGL((gather_halo_bf16s_kernel<256, 256, 2, 4, 2, 320>), grid, dim3(512), 0, st, sa);
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

// tile code V*1000000 + BM*1000 + BN.  V = 0: 4 waves, double-buffered LDS;  1: 4 waves, single LDS buffer;
// 2: 8 waves (4x2), single buffer;  3: 8 waves, double-buffered (BM = 128 only);
// 4: wave-specialised, 4 producer + 8 consumer waves, 2 tiles in flight;  5: same with 4 consumers;
// 6 / 7: as 4 / 5 with 3 tiles in flight;  8: halo image kernel (BM = 128, 8 waves)
static int launch_split_gather(SplitGatherArgs& sa, hipStream_t st, int code, int pieces) {
  GatherArgs& g = sa.g;
  Tile t;
  if (!decode_tile(code, t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; if (g.up) { t.bm = 128; t.dma = 8; } }
  const int v = t.dma;
  if (g.up && v != 8 && v != 9 && v != 29) {
    set_error("split gather: the fused x2 upsample of the input exists in the halo kernels (tile codes 8 / 9 / 29), not in code %d", code);
    return SVAE_ERR_SHAPE;
  }
  if (v >= 10 && v != 19 && v != 29) {
    bool handled = false;
    const int e = launch_split_halo_ws(sa, st, t, code, pieces, &handled);
    if (handled) return e;
  }
  if (v == 29) {  // 256 x 256 tiles
    if (t.bm != 128 || t.bn != 128) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    if (int e = launch_halo256x256(sa, dim3(nb, (g.N + 255) / 256), st, pieces, halo_rows(g, 256))) return e;
    return check_launch("gather_halo_bf16s<256,256>");
  }
  if (v == 19) {  // 256 x 160 tiles
    if (t.bm != 128 || t.bn != 128) { set_error("split gather: tile code %d unsupported", code); return SVAE_ERR_SHAPE; }
    for (int p = 0; p < 2; ++p) g.blocks_m[p] = (int)((g.M[p] + 255) / 256);
    const int nb = g.blocks_m[0] + g.blocks_m[1];
    if (nb == 0) return SVAE_OK;
    if (int e = launch_halo256_wide(sa, dim3(nb, (g.N + 159) / 160), st, pieces, halo_rows(g, 256))) return e;
    return check_launch("gather_halo_bf16s<256,160>");
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
  if (!decode_tile(d->tile[0], t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; if (d->up2) { t.bm = 128; t.dma = 8; } }  // as launch_split_gather
  const int bm = (t.dma == 9 || t.dma == 11 || t.dma >= 13) ? 256 : t.bm;  // (13 .. 19 and 29: 256-row tiles)
  return (int)((g.M[0] + bm - 1) / bm + (g.M[1] + bm - 1) / bm);
}

extern "C" int svae_conv_fwd_split(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                   int accumulate, int pieces, void* stream) {
  return svae_conv_fwd_split_stats(d, x, wsplit, bias, y, accumulate, pieces, nullptr, stream);
}

extern "C" int svae_conv_fwd_split_stats(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                         int accumulate, int pieces, float* bn_part, void* stream) {
  return svae_conv_fwd_split_up2(d, x, wsplit, bias, y, accumulate, pieces, bn_part, nullptr, stream);
}

extern "C" int svae_conv_fwd_split_up2(const svae_conv_desc* d, const float* x, const void* wsplit, const float* bias, float* y,
                                       int accumulate, int pieces, float* bn_part, float* up_out, void* stream) {
  if (int e = validate(d)) return e;
  SVAE_REQUIRE(up_out == nullptr || (d->up2 && aligned16(up_out) && d->ld_in == d->c_in), SVAE_ERR_ARG,
               "conv_fwd_split_up2: up_out needs an up2 descriptor with ld_in == c_in and a 16-byte aligned buffer");
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
  g.up = d->up2;
  sa.up_out = up_out;
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
  if (col_blocks) *col_blocks = t.dma == 19 ? (g.N + 159) / 160 : (t.dma == 29 ? (g.N + 255) / 256 : (g.N + t.bn - 1) / t.bn);
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
  if (!decode_tile(d->tile[kind], t)) { t = pick_tile(g.M[0], g.M[1], g.N); t.dma = 1; if (d->up2 && kind == 0) { t.bm = 128; t.dma = 8; } }
  *bm = t.bm; *bn = t.bn; *variant = t.dma; *rmax = 0;
  if (t.dma == 8) { const int r = halo_rows(g, 128); *rmax = r <= 160 ? 160 : 264; }
  if (t.dma == 9) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 320 ? 320 : 528; }
  if (t.dma == 10 || t.dma == 12) { const int r = halo_rows(g, 128); *rmax = r <= 160 ? 160 : 264; }
  if (t.dma == 11) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma == 13) { *bm = 256; *rmax = 264; }
  if (t.dma == 14 || t.dma == 15) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma >= 16 && t.dma <= 18) { const int r = halo_rows(g, 256); *bm = 256; *rmax = r <= 264 ? 264 : 320; }
  if (t.dma == 19) { const int r = halo_rows(g, 256); *bm = 256; *bn = 160; *rmax = r <= 320 ? 320 : 528; }
  if (t.dma == 29) { *bm = 256; *bn = 256; *rmax = 320; }
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
