// Wave-specialised halo-image kernels of the split-precision implicit GEMM for gfx950 (forward / data-gradient): 8 consumer + 2
// producer waves (tile codes 10 - 15), 8 consumer + 4 DMA-only loader waves on the 32x32x16 (16 / 17) and 16x16x32 (18) matrix-core
// shapes.  Contractions, plans and layouts: gemm_bf16s.hip.
#include "split_gather.h"
#include <type_traits>

namespace svae {

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

int launch_split_halo_ws(SplitGatherArgs& sa, hipStream_t st, const Tile& t, int code, int pieces, bool* handled) {
  GatherArgs& g = sa.g;
  const int v = t.dma;
  *handled = true;
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
  *handled = false;
  return SVAE_OK;
}

}  // namespace svae

using namespace svae;

#ifdef SVAE_ABLATION_KERNELS
extern "C" int svae_debug_stamp_buffer(void* buf) {  // diagnostic build only: [10 waves][64 stages][8] ticks
  unsigned long long* p = (unsigned long long*)buf;
  return hipMemcpyToSymbol(HIP_SYMBOL(svae::g_stamp_buf), &p, sizeof(p)) == hipSuccess ? SVAE_OK : SVAE_ERR_LAUNCH;
}
#endif
