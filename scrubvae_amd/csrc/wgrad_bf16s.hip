// Split-bf16 weight-gradient kernels for gfx950 (MI355X): the per-tap kernel (both operands transposed into LDS), the all-taps
// kernels on the 32x32x16 and 16x16x32 matrix-core shapes, and their launchers.  Same contractions as gemm_f32.hip
// (autograd of src/scrubvae/model/residual.py:79-109,137-170,198,264,286); operand splitting as in gemm_bf16s.hip.
#include "split_common.h"

namespace svae {

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
template <int BM, int BN, int TMAX, int SS, bool TRANS_OUT, bool UP = false>
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
  float4 sv2[UP ? S_ITEMS : 1];  // UP: the second source row of every image row (see load_s)
  auto load_s = [&](int kt) {
    const long long rk = r_begin + (long long)kt * KS;
    long long srow0 = rk * g.ss - g.pad;
    if (g.e != 0) srow0 += (rk / g.nj) * g.e;  // samples are Ls = nj ss + e rows apart: the image starts at the stage's first sample offset
    unsigned up_q0 = 0, up_r0 = 0;  // UP: srow0 = (up_q0 - 1) Ls + up_r0, 0 <= up_r0 < Ls (srow0 >= -pad > -Ls)
    if constexpr (UP) {
      const unsigned t0 = (unsigned)(srow0 + g.Ls);
      up_q0 = t0 / (unsigned)g.Ls;
      up_r0 = t0 - up_q0 * (unsigned)g.Ls;
    }
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      const long long sr = srow0 + row;
      // (select on the POINTER: a conditional load costs an exec-mask branch and a vmcnt(0) in front of the matrix work)
      const bool ok = (row < g.srows) & (sr >= 0) & (sr < g.rowsS) & (c0 + cq * 4 < g.Cs);  // srows = SR + the rows e > 0 adds (<= SR_ALLOC, host-checked)
      if constexpr (UP) {
        // S is the x2 linear upsample (align_corners = false; reference residual.py:160) of the half-length tensor g.S points to:
        // row (b, ro) of it = 0.75 x[b, ro / 2] + 0.25 x[b, ro / 2 -/+ 1] (clamped at the sample's ends), blended in store_s.
        // (sample, position) of image row `row`: one wave-uniform division per stage (up_q0 / up_r0 above), then a multiply-shift
        // per item -- per-item divisions made the staging VALU work the kernel's bottleneck (+30 %)
        const unsigned rl = up_r0 + (unsigned)row, wq = (rl * g.up_inv) >> 20;  // exact: rl * Ls < 2^20 (host-checked)
        const int ro = (int)(rl - wq * (unsigned)g.Ls), L = g.Ls >> 1, ii = ro >> 1;
        const int i2 = (ro & 1) ? (ii + 1 < L ? ii + 1 : L - 1) : (ii > 0 ? ii - 1 : 0);
        const float* base = g.S + ((long long)up_q0 - 1 + wq) * L * g.ldS + c0 + cq * 4;
        sv[i] = *reinterpret_cast<const float4*>(ok ? base + (long long)ii * g.ldS : wgrad_zero_row);
        sv2[i] = *reinterpret_cast<const float4*>(ok ? base + (long long)i2 * g.ldS : wgrad_zero_row);
      } else {
        sv[i] = *reinterpret_cast<const float4*>(ok ? g.S + sr * g.ldS + c0 + cq * 4 : wgrad_zero_row);
      }
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
      if constexpr (UP) {
        const float4 p = sv[i], q = sv2[i];
        split4<P>(up2_blend4(p, q), pc);
      } else {
        split4<P>(sv[i], pc);
      }
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

template <int BM, int BN, int TMAX, int SS, bool TRANS_OUT, bool UP = false>
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
  float4 sv2[UP ? S_ITEMS : 1];  // UP: the second source row of every image row (see load_s)
  auto load_s = [&](int kt) {
    const long long rk = r_begin + (long long)kt * KS;
    long long srow0 = rk * g.ss - g.pad;
    if (g.e != 0) srow0 += (rk / g.nj) * g.e;  // samples are Ls = nj ss + e rows apart: the image starts at the stage's first sample offset
    unsigned up_q0 = 0, up_r0 = 0;  // UP: srow0 = (up_q0 - 1) Ls + up_r0, 0 <= up_r0 < Ls (srow0 >= -pad > -Ls)
    if constexpr (UP) {
      const unsigned t0 = (unsigned)(srow0 + g.Ls);
      up_q0 = t0 / (unsigned)g.Ls;
      up_r0 = t0 - up_q0 * (unsigned)g.Ls;
    }
#pragma unroll
    for (int i = 0; i < S_ITEMS; ++i) {
      const int idx = tid + NTH * i, row = idx / (BM / 4), cq = idx % (BM / 4);
      const long long sr = srow0 + row;
      const bool ok = (row < g.srows) & (sr >= 0) & (sr < g.rowsS) & (c0 + cq * 4 < g.Cs);  // srows = SR + the rows e > 0 adds (<= SR_ALLOC, host-checked)
      if constexpr (UP) {
        // S is the x2 linear upsample (align_corners = false; reference residual.py:160) of the half-length tensor g.S points to:
        // row (b, ro) of it = 0.75 x[b, ro / 2] + 0.25 x[b, ro / 2 -/+ 1] (clamped at the sample's ends), blended in store_s.
        // (sample, position) of image row `row`: one wave-uniform division per stage (up_q0 / up_r0 above), then a multiply-shift
        // per item -- per-item divisions made the staging VALU work the kernel's bottleneck (+30 %)
        const unsigned rl = up_r0 + (unsigned)row, wq = (rl * g.up_inv) >> 20;  // exact: rl * Ls < 2^20 (host-checked)
        const int ro = (int)(rl - wq * (unsigned)g.Ls), L = g.Ls >> 1, ii = ro >> 1;
        const int i2 = (ro & 1) ? (ii + 1 < L ? ii + 1 : L - 1) : (ii > 0 ? ii - 1 : 0);
        const float* base = g.S + ((long long)up_q0 - 1 + wq) * L * g.ldS + c0 + cq * 4;
        sv[i] = *reinterpret_cast<const float4*>(ok ? base + (long long)ii * g.ldS : wgrad_zero_row);
        sv2[i] = *reinterpret_cast<const float4*>(ok ? base + (long long)i2 * g.ldS : wgrad_zero_row);
      } else {
        sv[i] = *reinterpret_cast<const float4*>(ok ? g.S + sr * g.ldS + c0 + cq * 4 : wgrad_zero_row);
      }
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
      if constexpr (UP) {
        const float4 p = sv[i], q = sv2[i];
        split4<P>(up2_blend4(p, q), pc);
      } else {
        split4<P>(sv[i], pc);
      }
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
  if (g.up) {  // the shifted operand is the x2 linear upsample of g.S, blended while it is staged: the (k+1)-tap skip convs of the decoder
    if (g.T != 6 || g.ss != 1 || trans_out || (g.Ls & 1) || g.Ls > 512 || g.pad >= g.Ls || g.rowsS >= 0x7fffffffLL || g.up_inv != ((1u << 20) + g.Ls - 1) / g.Ls ||
        !((bm == 64 && bn == 128) || (bm == 128 && bn == 64))) {
      set_error("wgrad taps: the fused upsample exists for 6-tap stride-1 convs on 64x128 / 128x64 tiles");
      return SVAE_ERR_SHAPE;
    }
#define SVAE_WTU(BM_, BN_) do {                                                                                            \
    if (m16) hipLaunchKernelGGL((wgrad_taps16_bf16s_kernel<BM_, BN_, 6, 1, false, true>), grid, block, 0, st, g);          \
    else hipLaunchKernelGGL((wgrad_taps_bf16s_kernel<BM_, BN_, 6, 1, false, true>), grid, block, 0, st, g); } while (0)
    if (bm == 64) SVAE_WTU(64, 128); else SVAE_WTU(128, 64);
#undef SVAE_WTU
    return check_launch("wgrad_taps_bf16s<up>");
  }
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

}  // namespace svae
