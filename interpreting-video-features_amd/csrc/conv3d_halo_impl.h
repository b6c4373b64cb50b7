// (implementation header: included by conv3d_halo.hip / conv3d_halo_x6.hip / conv3d_halo_bf16.hip, one operand mode each)
// Stride-1 k x k x k (k <= 4) convolution, split-bf16 MFMA, with the input
// neighbourhood held in LDS: the kernel that carries the 3x3x3 Inception branches, their
// backward-data, and the stem's 4x4x4 depth-to-space backward-data.
//
// The plain implicit GEMM (conv3d.hip) re-loads every input element once per tap and per
// output-channel tile; at split-bf16 MFMA speed that load stream (~10 TB/s through L1/L2),
// not the matrix cores, bounds it.  Here a workgroup owns a TT x 8 x 8 box of output
// positions; per 32-channel chunk it stages the (TT+k-1) x (8+k-1) x (8+k-1) input halo
// ONCE (global -> registers -> split hi/lo bf16 -> LDS), then sweeps all k^3 taps over it:
// a tap only changes the LDS row offset of the A fragments.  Only the small weight tile
// (BN x 32) streams per tap, double-buffered so its loads fly under the previous tap's MFMAs
// (one barrier per tap).  Input bytes per MAC drop ~8x for 3x3x3.
#pragma once
#include <algorithm>
#include <type_traits>

#include "conv_common.h"

namespace ivf {

// Diagnostic build only (make stamps -> libivf_hip_stamps.so, tools/halo_stamps.py): thread 0 of every
// workgroup adds its cycles per phase to g_halo_stamps; the product build compiles none of it.
#ifdef IVF_HALO_STAMPS
static __device__ unsigned long long g_halo_stamps[8];
#define IVF_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define IVF_STAMP_RT(var) const unsigned long long var = __builtin_amdgcn_s_memrealtime()   // 100 MHz
#define IVF_STAMP_ADD(slot, t1, t0) \
  do { if (threadIdx.x == 0) atomicAdd(&g_halo_stamps[slot], (t1) - (t0)); } while (0)
#else
#define IVF_STAMP(var) do {} while (0)
#define IVF_STAMP_RT(var) do {} while (0)
#define IVF_STAMP_ADD(slot, t1, t0) do {} while (0)
#endif

// Position (h, w) inside the 8 x 8 plane of a tile for plane-row p in [0,64): two 32-row MFMA
// tiles (4 h-rows x 8 w each).  Which lane bit feeds which coordinate bit is free as long as
// the A operand and the epilogue agree; the choice per halo width minimises LDS bank conflicts
// of the 16-byte fragment reads (rows are 80 B apart; 3-way with the natural order, 2-way so).
__device__ __forceinline__ void tile_hw(int p, int hw_pitch, int* h, int* w) {
  const int b0 = p & 1, b1 = (p >> 1) & 1, b2 = (p >> 2) & 1, b3 = (p >> 3) & 1, b4 = (p >> 4) & 1;
  const int half = p >> 5;
  int hl, ww;
  if (hw_pitch == 10) {          // k = 3
    ww = b0 | (b2 << 1) | (b1 << 2);
    hl = b3 | (b4 << 1);
  } else if (hw_pitch == 11) {   // k = 4
    ww = b0 | (b2 << 1) | (b3 << 2);
    hl = b1 | (b4 << 1);
  } else {
    ww = p & 7;
    hl = (p >> 3) & 3;
  }
  *h = half * 4 + hl;
  *w = ww;
}

// Box row -> (t, h, w) inside a TT x TH x TW box.
//  * 4 x 8 x 8 boxes: a 32-row MFMA tile is one h-row of the box across its 4 planes (4 t x 8 w).
//    The halo plane stride of a 3x3x3 conv is 100 rows = 4 (mod 16), so the four planes' 8-row runs
//    start at 16-byte bank slots 0, 4, 8, 12 of the 256-byte LDS bank row: putting planes {0,2} on
//    one ds_read_b128 lane group ({0-3,12-15,20-27}, MI355X_MICROARCH.md) and {1,3} on the other
//    makes the A-fragment reads conflict-free (the earlier one-plane tiles could not be: one slot
//    is hit three times by any 4 h x 8 w cell set at row pitch 10).  For k = 2 and 4 this order is
//    as good as the old one (2-way).
//  * other 8 x 8 boxes (TT = 2) keep the permuted one-plane order above;
//  * other plane shapes (4 x 14: the 28 x 28 and 14 x 14 maps of Mixed_3*/4*, which 8 x 8 boxes
//    cover with 23 % of their rows outside the map) use raster order.
template <int TT, int TH, int TW>
__device__ __forceinline__ void box_pos(int row, int hw_pitch, int* t, int* h, int* w) {
  if constexpr (TT == 4 && TW == 8) {   // (any TH: a tile is one h-row of the box)
    const int li = row & 31, q = li >> 2, ql = q & 3;
    const int sel = ((ql + 1) >> 1) & 1;
    *t = (q >> 2) ? 3 - sel : sel;
    *w = (ql >> 1) * 4 + (li & 3);
    *h = row >> 5;
  } else if constexpr (TH == 8 && TW == 8) {
    tile_hw(row & 63, hw_pitch, h, w);
    *t = row >> 6;
  } else {
    const int p = row % (TH * TW);
    *t = row / (TH * TW);
    *h = p / TW;
    *w = p % TW;
  }
}

// LDS rows per halo plane.  The 4 x 8 x 8 boxes want the plane stride = 4 (mod 8) rows (see
// box_pos): 100 for k = 3 as it is, 121 -> 124 for k = 4, 81 -> 84 for k = 2 (pad rows are
// never staged nor read).
template <int TT, int TH, int TW>
__host__ __device__ constexpr int halo_plane_rows(int HH, int HW) {
  int ps = HH * HW;
  if (TT == 4 && TW == 8) ps += (4 - ps % 8 + 8) % 8;
  return ps;
}

// KS = 2 splits the taps of a chunk between two groups of waves (each wave then owns a
// bigger output sub-tile, i.e. fewer LDS fragment reads per MFMA: what narrow outputs such as
// the stem's 32-column backward-data need); the two partial sums meet in LDS at the end.
// BKH = channels per chunk: 32, or 16 to halve the LDS footprint so that TWO workgroups are
// resident per CU and one's halo staging / weight pipeline overlaps the other's MFMAs.
// TPS = taps per barrier interval ("step"): the weight tiles of TPS taps sit in each LDS buffer and the waves sweep
// them without meeting.  Every barrier costs the MFMA pipe about 1000 idle cycles (the waves re-issue their fragment
// reads together; profiles/r02_halo_phase_stamps.txt), which a narrow tile (32 columns: 6 MFMAs per wave per
// tap) cannot amortise over one tap.
// DMA = the weight tiles go from global memory straight into LDS (global_load_lds_dwordx4: no register ring, no
// ds_write_b128, no wait on the data inside the step): unpadded 64-byte rows, the 16-byte slot of a row XOR-ed
// with (row >> 2) & 3 -- the permutation is applied to the SOURCE address of every lane (the DMA writes lane i's
// 16 bytes at M0 + 16 i), and a lane's fragment address is a per-lane constant, so the tap loop pays nothing for it.
// AM = operand mode (conv_common.h): NPA activation planes and NPB weight planes sit in LDS, the tap loop issues the
// mode's MFMA sequence per 16-deep k-step (3 / 2 / 6 passes); AM_BF16 stages bf16 activations as they are and its
// epilogue stores bf16 (the depth-to-space forms always write fp32: the stem's input gradient).
template <int AM, int TT, int BN, int WROWS, int WCOLS, int KS, int BKH, int TH, int TW, int TPS, bool DMA>
__global__ __launch_bounds__((TT * TH * TW / WROWS) * (BN / WCOLS) * KS * 64) void conv3d_halo_kernel(ConvKArgs a,
                                                                                                int tilesT,
                                                                                                int tilesH,
                                                                                                int tilesW) {
  constexpr int BM = TT * TH * TW;
  constexpr int NPA = OpPlanes<AM>::A, NPB = OpPlanes<AM>::B;
  static_assert(BM % WROWS == 0 && WROWS % 32 == 0 && BN % WCOLS == 0 && WCOLS % 32 == 0, "tile");
  static_assert((KS * BN * (BKH / 8)) % 64 == 0, "weight-stream slots must fill whole waves");
  constexpr int WM = BM / WROWS, WN = BN / WCOLS;
  constexpr int NT = WM * WN * KS * 64;
  constexpr int TM = WROWS / 32, TN = WCOLS / 32;
  constexpr int ROWB = (BKH + 8) * 2;   // bytes per LDS row per plane (80 / 48: conflict-free 16-byte reads)
  constexpr int ROWB_B = DMA ? BKH * 2 : ROWB;   // weight rows: unpadded + swizzled in DMA mode
  static_assert(!DMA || BKH == 32, "the LDS-DMA weight stream is written for 32-channel chunks");
  constexpr int G4 = BKH / 4;           // float4 groups per halo row
  constexpr int G8 = BKH / 8;           // 16-byte weight groups per row per plane
  constexpr int BLOADS = (TPS * KS * BN * G8 + NT - 1) / NT;   // 16-byte weight loads per thread per plane per step

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  IVF_STAMP(st_begin);
  IVF_STAMP_RT(rt_begin);
  const int HT = TT + a.kT - 1, HH = TH + a.kH - 1, HW = TW + a.kW - 1;
  const int PS = halo_plane_rows<TT, TH, TW>(HH, HW);   // LDS rows per halo plane (>= HH * HW)
  const int HR = HT * PS;
  unsigned char* a_hi = smem;                              // NPA planes of HR rows, APL bytes apart
  const size_t APL = (size_t)HR * ROWB;
  unsigned char* b_base = smem + (size_t)NPA * HR * ROWB;   // [2 buffers][TPS taps][KS groups][NPB planes][BN rows]
  int* rowoff = reinterpret_cast<int*>(b_base + (size_t)2 * NPB * TPS * KS * BN * ROWB_B);   // [HR] input offset of a halo row / in_ld, or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: tap decode and tile bases stay on the scalar unit
  const int wk = wave / (WM * WN);               // tap group of this wave
  const int wm = (wave % (WM * WN)) / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // tile decode: n-tile fastest, then w, h, t, b
  int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int nt = tile % a.ntiles;
  tile /= a.ntiles;
  const int tw = tile % tilesW;
  tile /= tilesW;
  const int th = tile % tilesH;
  tile /= tilesH;
  const int tt = tile % tilesT;
  const int b = tile / tilesT;
  const int t0 = tt * TT, h0 = th * TH, w0 = tw * TW;
  const int n0 = nt * BN;

  // A-fragment base rows of this lane (one per 32-row MFMA tile of the wave)
  int arow[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int r = wm * WROWS + i * 32 + li;
    int pt, ph, pw;
    box_pos<TT, TH, TW>(r, HW, &pt, &ph, &pw);
    arow[i] = pt * PS + ph * HW + pw;
  }
  const int ntaps = a.kT * a.kH * a.kW;
  const int ntg = (ntaps + KS - 1) / KS;         // tap group g handles taps [g*ntg, (g+1)*ntg)
  const int nsteps = (ntg + TPS - 1) / TPS;      // barrier intervals per chunk; step s = group-local taps [s*TPS, (s+1)*TPS)
  const int khw = a.kH * a.kW;
  // halo row -> input position (decoded once; the chunk loop only adds the channel offset)
  for (int row = tid; row < HR; row += NT) {
    int ht = row / PS;
    int rem = row - ht * PS;
    int hh = rem / HW;
    int hw = rem - hh * HW;
    int ti = t0 - a.pT + ht, hi = h0 - a.pH + hh, wi = w0 - a.pW + hw;
    bool ok = hh < HH && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
    rowoff[row] = ok ? ((b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi : -1;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Weight tiles stream through a register ring PF taps deep (their L2 latency is longer than
  // one tap of MFMAs and there is a single workgroup per CU, so nothing else would hide it) and
  // a double-buffered LDS tile.
  constexpr int PF = 3;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rb[PF][NPB][BLOADS] = {};
  // Per-thread constants of the weight stream, decoded once: the tap loop only adds the tap and the chunk.  (The
  // scalar unit is what a narrow tile runs out of: 75 scalar + 28 vector instructions per tap beside 6 MFMAs
  // measured 53 % scalar-busy at 36 % MFMA-busy on the stem's backward-data.)
  // Every thread loads every step, from an address clamped into the weight array, with no validity branch and no
  // zero fill: a slot outside the step, tap range, layer rows or channels holds finite weights of some other
  // position, which meet staged zeros (channels), are skipped (taps) or land in columns nobody stores (rows).
  // Straight-line loads are what lets the compiler count them (`s_waitcnt vmcnt(N)`): behind a branch it waited
  // for vmcnt(0) right after issuing the prefetch, i.e. exposed a full L2 round trip per tap.
  int b_src[BLOADS];    // element offset min(n, Cout - 1) * ldw into wbf (0 for threads without a slot)
  int b_dst[BLOADS];    // byte offset in one LDS weight buffer [TPS][KS][hi, lo][BN rows]; -1 = no slot
  int b_tap[BLOADS];    // sub-tap + group * ntg: tap of step s = s * TPS + b_tap
  int b_c[BLOADS];      // channel offset inside the chunk
#pragma unroll
  for (int q = 0; q < BLOADS; ++q) {
    const int idx = tid + q * NT;
    const int sub = idx / (KS * BN * G8);
    const int r1 = idx - sub * (KS * BN * G8);
    const int grp = r1 / (BN * G8);
    const int rem = r1 - grp * (BN * G8);
    const int row = BKH == 32 ? perm8(rem / G8) : rem / G8, g2 = rem % G8;
    const int n = min(n0 + row, a.Cout - 1);
    b_src[q] = sub < TPS ? n * a.ldw : 0;
    b_dst[q] = sub < TPS ? ((sub * KS + grp) * NPB * BN + row) * ROWB_B + 16 * g2 : -1;
    b_tap[q] = sub < TPS ? sub + grp * ntg : 0;
    b_c[q] = 8 * g2;
  }
  // Kept straight-line on purpose: no scalar branch around the loads or the stores (lanes without a slot are
  // masked off, which is not a branch for a block this short), and the loop below issues them every step.  Then
  // the compiler can count (`s_waitcnt vmcnt(4)` before a slot is stored); behind any control-flow join it falls
  // back to vmcnt(0) right after issuing the prefetch.  (Hand-issued asm loads were tried: the compiler copies
  // asm outputs between registers while the load is still in flight.)
  auto load_b = [&](int slot, int step, int c0) __attribute__((always_inline)) {
    const int s0 = min(step, nsteps - 1) * TPS;
#pragma unroll
    for (int q = 0; q < BLOADS; ++q) {
      const int tap = min(s0 + b_tap[q], ntaps - 1);
      int c = c0 + b_c[q];
      c = c < a.Cin ? c : 0;
      const unsigned short* p = a.wbf + (size_t)(b_src[q] + tap * a.Cin + c);
      if (b_dst[q] >= 0) {
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) rb[slot][pl][q] = *reinterpret_cast<const u32x4*>(p + pl * a.w_lo_off);
      }
    }
  };
  auto store_b = [&](int slot, int buf) __attribute__((always_inline)) {
    unsigned char* bb = b_base + (size_t)buf * (TPS * KS * NPB * BN * ROWB_B);
#pragma unroll
    for (int q = 0; q < BLOADS; ++q) {
      if (b_dst[q] >= 0) {
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) *reinterpret_cast<u32x4*>(bb + b_dst[q] + pl * BN * ROWB_B) = rb[slot][pl][q];
      }
    }
  };
  // tap -> byte offset of its halo row shift, advanced tap by tap on the scalar unit (no divisions in the loop);
  // this wave's tap group starts at tap wk * ntg
  int g_kw, g_kh, g_toff;
  {
    const int tap0 = wk * ntg;
    const int kt = tap0 / khw;
    const int rem = tap0 - kt * khw;
    g_kh = rem / a.kW;
    g_kw = rem - g_kh * a.kW;
    g_toff = (kt * PS + g_kh * HW + g_kw) * ROWB;
  }
  int t_kw = 0, t_kh = 0, t_off = 0;   // running state of the current chunk
  auto tap_advance = [&]() __attribute__((always_inline)) {
    t_off += ROWB;
    if (++t_kw == a.kW) {
      t_kw = 0;
      t_off += (HW - a.kW) * ROWB;
      if (++t_kh == a.kH) {
        t_kh = 0;
        t_off += (PS - a.kH * HW) * ROWB;
      }
    }
  };
  int abase[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) abase[i] = arow[i] * ROWB + 16 * lh;
  // weight-fragment offsets of this lane inside a (tap, group) tile: column tile j, 16-channel slice ks
  auto b_off = [&](int j, int ks) __attribute__((always_inline)) {
    const int row = wn * WCOLS + j * 32 + li;
    if constexpr (DMA) return row * ROWB_B + 16 * ((2 * ks + lh) ^ ((row >> 2) & 3));
    else return row * ROWB_B + ks * 32 + 16 * lh;
  };
  // ---- LDS-DMA weight stream: piece p (16 bytes) of a buffer = [sub][group][plane][row][physical slot], i.e. byte 16 p
  constexpr int NP = TPS * KS * NPB * BN * (BKH / 8), NI = NP / 64, NWV = NT / 64, MAXI = (NI + NWV - 1) / NWV;
  int d_base[MAXI], d_tap[MAXI], d_c[MAXI];
  if constexpr (DMA) {
#pragma unroll
    for (int k = 0; k < MAXI; ++k) {
      const int p = (wave + k * NWV) * 64 + lane;
      const int slot = p % G8, row = (p / G8) % BN, plane = (p / (G8 * BN)) % NPB;
      const int grp = (p / (G8 * BN * NPB)) % KS, sb = p / (G8 * BN * NPB * KS);
      d_base[k] = plane * (int)a.w_lo_off + min(n0 + row, a.Cout - 1) * a.ldw;
      d_tap[k] = sb + grp * ntg;
      d_c[k] = 8 * (slot ^ ((row >> 2) & 3));
    }
  }
  auto dma_b = [&](int step, int buf, int c0) __attribute__((always_inline)) {
    if constexpr (DMA) {
#pragma unroll
      for (int k = 0; k < MAXI; ++k) {
        if (wave + k * NWV >= NI) continue;   // (wave-uniform)
        const int tap = min(step * TPS + d_tap[k], ntaps - 1);
        int c = c0 + d_c[k];
        c = c < a.Cin ? c : 0;
        const unsigned short* g = a.wbf + (size_t)(d_base[k] + tap * a.Cin + c);
        // the compiler's own LDS-DMA builtin: it owns M0 (destination = wave-uniform LDS base, lane i writes 16 bytes
        // at base + 16 i) and knows that LDS is written
        unsigned char* dst = b_base + (size_t)buf * (TPS * KS * NPB * BN * ROWB_B) + (size_t)(wave + k * NWV) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };
  // One step = the TPS taps of an LDS weight buffer.  Narrow wave tiles (one MFMA tile, 6 MFMAs per tap) run a
  // software pipeline over the (tap, 16-channel slice) units of the step: the fragments of unit u+1 are in flight
  // while unit u's MFMAs issue, so a wave exposes one LDS round trip per step instead of two per slice.
  constexpr int NKS = BKH / 16;
  constexpr bool PIPELINED = (TM == 1 && TN == 1);
  auto mma_tap = [&](int step, int buf, int nks) __attribute__((always_inline)) {
    const int lt0 = step * TPS;
    const int nsub = min(TPS, min(ntg - lt0, ntaps - wk * ntg - lt0));   // valid taps of this wave's group in the step
    if (nsub <= 0) return;
    const unsigned char* bstep = b_base + (size_t)(buf * TPS * KS + wk) * NPB * BN * ROWB_B;   // + sub * KS * NPB * BN * ROWB_B
    if constexpr (PIPELINED) {   // (a partial last chunk runs all NKS slices: channels past Cin are staged as zeros)
      bf16x8 fa[2][NPA], fb[2][NPB];
      auto issue = [&](int slot, int sub, int ks) __attribute__((always_inline)) {
        const int aoff = abase[0] + t_off + ks * 32;
        const unsigned char* bh = bstep + (size_t)sub * KS * NPB * BN * ROWB_B + b_off(0, ks);
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) fa[slot][pl] = *reinterpret_cast<const bf16x8*>(a_hi + pl * APL + aoff);
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) fb[slot][pl] = *reinterpret_cast<const bf16x8*>(bh + (size_t)pl * BN * ROWB_B);
      };
      issue(0, 0, 0);
#pragma unroll
      for (int u = 0; u < TPS * NKS; ++u) {
        if (u / NKS < nsub) {
          if (u + 1 < TPS * NKS && (u + 1) / NKS < nsub) {
            if ((u + 1) % NKS == 0) tap_advance();
            issue((u + 1) & 1, (u + 1) / NKS, (u + 1) % NKS);
          }
          mma_planes<AM>(fa[u & 1], fb[u & 1], acc[0][0]);
          // keep the source order: next unit's fragment reads, then this unit's MFMAs
          __builtin_amdgcn_sched_group_barrier(0x100, NPA + NPB, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, AM == AM_X6 ? 6 : (AM == AM_BF16 ? 2 : 3), 0);
        }
      }
      tap_advance();
    } else {
#pragma unroll
      for (int sub = 0; sub < TPS; ++sub) {
        if (sub >= nsub) break;
        const unsigned char* bh = bstep + (size_t)sub * KS * NPB * BN * ROWB_B;
        for (int ks = 0; ks < nks; ++ks) {
          bf16x8 fa[TM][NPA];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const int off = abase[i] + t_off + ks * 32;
#pragma unroll
            for (int pl = 0; pl < NPA; ++pl) fa[i][pl] = *reinterpret_cast<const bf16x8*>(a_hi + pl * APL + off);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int off = b_off(j, ks);
            bf16x8 fb[NPB];
#pragma unroll
            for (int pl = 0; pl < NPB; ++pl) fb[pl] = *reinterpret_cast<const bf16x8*>(bh + (size_t)pl * BN * ROWB_B + off);
#pragma unroll
            for (int i = 0; i < TM; ++i) mma_planes<AM>(fa[i], fb, acc[i][j]);
          }
        }
        tap_advance();
      }
    }
  };

  // halo staging, synchronous in batches of 4 loads per thread: 8 float4 groups per halo row (fetching the NEXT
  // chunk's halo into registers under the current chunk's taps was measured twice -- 256-thread tiles in round 2,
  // the 512-/1024-thread three-plane tiles in round 3 with 16-28 registers to spare: nothing gained either time;
  // what an ablated staging phase saves (4-12 %) is its memory traffic, not its latency)
  constexpr int NSTG = 4;
  float4 stg[NSTG];
  // (rows are visited in perm8 order, so the item range is padded to whole blocks of 8 rows)
  const int ngroups = BKH == 32 ? ((HR + 7) & ~7) * G4 : HR * G4;
  int stage_base = 0;
  auto stage_load = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      int idx = stage_base + u * NT + tid;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < ngroups) {
        int row = BKH == 32 ? perm8(idx / G4) : idx / G4, g = idx % G4;
        if (row >= HR) row = -1;
        int c = c0 + 4 * g;
        int pos = row >= 0 ? rowoff[row] : -1;
        if (pos >= 0 && c < a.Cin) v = load_act4<AM>(a.in, (size_t)pos * a.in_ld + a.in_coff + c);
      }
      stg[u] = v;
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NSTG; ++u) {
      int idx = stage_base + u * NT + tid;
      const int srow = BKH == 32 ? perm8(idx / G4) : idx / G4;
      if (idx < ngroups && srow < HR) {
        stage_planes<AM>(a_hi + srow * ROWB + 8 * (idx % G4), APL, stg[u]);
      }
    }
  };
  __syncthreads();   // rowoff table is complete

  for (int c0 = 0; c0 < a.Cin; c0 += BKH) {
    // weight tiles of the first PF taps start flying before the halo is staged
    if constexpr (!DMA) {
      load_b(0, 0, c0);
      load_b(1, 1, c0);
      load_b(2, 2, c0);
    }
    __syncthreads();   // everyone is done with the previous chunk's halo and weight buffers
    dma_b(0, 0, c0);   // (DMA mode: the first step's tiles land in buffer 0 while the halo is staged)
    IVF_STAMP(st_s0);
    for (stage_base = 0; stage_base < ngroups; stage_base += NSTG * NT) {
      stage_load(c0);
      stage_store();
    }
    stage_base = 0;
    // everything the compiler knows to be in flight has landed by now; saying so keeps its conservative
    // `s_waitcnt vmcnt(0)` (staging registers reused by the fragment reads) out of the tap loop, where it would
    // drain the weight prefetch every step
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    if constexpr (!DMA) store_b(0, 0);
    __syncthreads();
    IVF_STAMP(st_s1);
    IVF_STAMP_ADD(1, st_s1, st_s0);

    const int cw = min(BKH, a.Cin - c0);
    const int nks = (cw + 15) >> 4;
    t_kw = g_kw, t_kh = g_kh, t_off = g_toff;
    // tap loop unrolled by PF so the register ring is statically indexed: at tap (slot u)
    // the LDS buffer tap&1 holds its weights; slot u is refilled with tap+PF, and slot u+1's
    // weights (tap+1, loaded PF-1 taps ago) move to the other LDS buffer after the MFMAs.
    auto tap_body = [&](auto U, int tap0) __attribute__((always_inline)) {
      constexpr int u = decltype(U)::value;
      const int tap = tap0 + u;   // step index within the chunk (the last round may run past nsteps: no MFMAs then)
      load_b(u, tap + PF, c0);
      if (tap < nsteps) mma_tap(tap, tap & 1, nks);
      store_b((u + 1) % PF, (tap + 1) & 1);
      __syncthreads();
    };
    static_assert(PF == 3, "tap loop is unrolled by hand for a 3-deep ring");
    if constexpr (DMA) {
      // step t: the tiles of step t+1 start towards the other buffer (free since the barrier that ended step t-1),
      // the MFMAs of step t run, then the wave waits for its own DMA and everybody meets
      for (int tap = 0; tap < nsteps; ++tap) {
        if (tap + 1 < nsteps) dma_b(tap + 1, (tap + 1) & 1, c0);
        mma_tap(tap, tap & 1, nks);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    } else {
      for (int tap0 = 0; tap0 < nsteps; tap0 += PF) {
        tap_body(std::integral_constant<int, 0>{}, tap0);
        tap_body(std::integral_constant<int, 1>{}, tap0);
        tap_body(std::integral_constant<int, 2>{}, tap0);
      }
    }
    IVF_STAMP(st_s2);
    IVF_STAMP_ADD(2, st_s2, st_s1);
  }
  IVF_STAMP(st_loop_end);

  if constexpr (KS == 2) {
    // meet the two tap groups: group 1 parks its partial sums in LDS (the halo is dead now),
    // group 0 adds them and owns the epilogue
    float* red = reinterpret_cast<float*>(smem);
    const int slot = (wm * WN + wn) * TM * TN;
    if (wk == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) red[((slot + i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((slot + i * TN + j) * 16 + r) * 64 + lane];
    }
    __syncthreads();   // the reduction buffer may be reused below
  }

  // Depth-to-space output of a full 2x2x2 block conv with 4-channel pixels (the stem's
  // backward-data): every lane quad holds one 16-byte pixel and a wave's direct stores land as
  // 32-byte pieces.  Park the tile in LDS as [2TT][16][16][4] and write whole 256-byte pixel
  // rows instead.
  if (TH == 8 && TW == 8 && BN == 32 && a.d2s && a.Cout == 32 && a.bsT == 2 && a.bsH == 2 && a.bsW == 2 && !a.accumulate && !a.relu &&
      a.dC == 4 && (a.out_ld & 3) == 0 && (a.out_coff & 3) == 0) {
    float* ot = reinterpret_cast<float*>(smem);
    if (wk == 0) {
      const int n = wn * WCOLS + li;          // TN == 1 for BN == 32
      const int par = n >> 2, c = n & 3;
      const int pt = par >> 2, ph = (par >> 1) & 1, pw = par & 1;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = wm * WROWS + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          int bt, bh, bw;
          box_pos<TT, TH, TW>(row, HW, &bt, &bh, &bw);
          const int tt2 = 2 * bt + pt, hh2 = 2 * bh + ph, ww2 = 2 * bw + pw;
          ot[((tt2 * 16 + hh2) * 16 + ww2) * 4 + c] = acc[i][0][r];
        }
    }
    __syncthreads();
    for (int p = tid; p < 2 * TT * 256; p += NT) {
      const int t = 2 * t0 + (p >> 8), h = 2 * h0 + ((p >> 4) & 15), w = 2 * w0 + (p & 15);
      if (t < a.dT && h < a.dH && w < a.dW)
        *reinterpret_cast<float4*>(a.out + ((size_t)((b * a.dT + t) * a.dH + h) * a.dW + w) * a.out_ld + a.out_coff) =
            *reinterpret_cast<const float4*>(ot + (size_t)p * 4);
    }
    IVF_STAMP(st_end_d2s);
    IVF_STAMP_RT(rt_end_d2s);
    IVF_STAMP_ADD(5, rt_end_d2s, rt_begin);
    IVF_STAMP_ADD(3, st_end_d2s, st_loop_end);
    IVF_STAMP_ADD(0, st_end_d2s, st_begin);
    IVF_STAMP_ADD(4, st_begin + 1, st_begin);      // workgroup count
    return;
  }
  if (wk != 0) return;

  conv_epilogue<AM, TM, TN>(
      a, acc,
      [&](int row) {
        int pt, ph, pw;
        box_pos<TT, TH, TW>(row, HW, &pt, &ph, &pw);
        int t = t0 + pt, h = h0 + ph, w = w0 + pw;
        if (t >= a.To || h >= a.Ho || w >= a.Wo) return -1;
        return ((b * a.To + t) * a.Ho + h) * a.Wo + w;
      },
      wm * WROWS, n0 + wn * WCOLS, li, lh);
  IVF_STAMP(st_end);
  IVF_STAMP_RT(rt_end);
  IVF_STAMP_ADD(5, rt_end, rt_begin);
  IVF_STAMP_ADD(3, st_end, st_loop_end);
  IVF_STAMP_ADD(0, st_end, st_begin);
  IVF_STAMP_ADD(4, st_begin + 1, st_begin);        // workgroup count
}

#if defined(IVF_HALO_STAMPS) && defined(IVF_HALO_STAMPS_OWNER)
}  // namespace ivf
// slots: 0 total, 1 halo staging (+ first weight tile), 2 tap loops, 3 reduction + epilogue, 4 workgroups,
// 5 total in 100 MHz s_memrealtime ticks (slot 0 / slot 5 x 100 MHz = the clock the chip held)
extern "C" int ivf_debug_halo_stamps(unsigned long long* out8, int reset) {
  IVF_CHECK_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(ivf::g_halo_stamps), 8 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[8] = {0};
    IVF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(ivf::g_halo_stamps), z, sizeof(z)));
  }
  return IVF_OK;
}
namespace ivf {
#endif

template <int AM, int TT, int BN, int WROWS, int WCOLS, int KS = 1, int BKH = 32, int TH = 8, int TW = 8, int TPS = 1, bool DMA = false>
static int launch_halo(ConvKArgs& a, int variant_id, hipStream_t s) {
  constexpr int NPA = OpPlanes<AM>::A, NPB = OpPlanes<AM>::B;
  constexpr int NT = (TT * TH * TW / WROWS) * (BN / WCOLS) * KS * 64;
  constexpr int ROWB = (BKH + 8) * 2;
  constexpr int ROWB_B = DMA ? BKH * 2 : ROWB;
  const int HR = (TT + a.kT - 1) * halo_plane_rows<TT, TH, TW>(TH + a.kH - 1, TW + a.kW - 1);
  size_t shm = (size_t)NPA * HR * ROWB + (size_t)2 * TPS * KS * NPB * BN * ROWB_B + (size_t)HR * sizeof(int);
  // buffers that reuse the LDS from offset 0 once the tap loops are done (everything staged is dead by then): the
  // tap-split reduction [BM][BN] floats, the depth-to-space tile [2 TT][16][16][4] floats
  if (KS == 2) shm = std::max(shm, (size_t)TT * TH * TW * BN * 4);
  if (a.d2s) shm = std::max(shm, (size_t)2 * TT * 256 * 4 * sizeof(float));
  if (shm > 160 * 1024) {
    set_error("conv3d_halo: %zu bytes of LDS needed", shm);
    return IVF_ERR_UNSUPPORTED;
  }
  static LdsAttrOnce once;
  IVF_PROPAGATE(raise_lds_limit(reinterpret_cast<const void*>(&conv3d_halo_kernel<AM, TT, BN, WROWS, WCOLS, KS, BKH, TH, TW, TPS, DMA>), 160 * 1024, once));
  const int tilesT = cdiv(a.To, TT), tilesH = cdiv(a.Ho, TH), tilesW = cdiv(a.Wo, TW);
  a.ntiles = cdiv(a.Cout, BN);
  a.mtiles = a.B * tilesT * tilesH * tilesW;
  dim3 grid(a.mtiles * a.ntiles);
  prof_name(IVF_CONV_HALO_BASE + variant_id, "conv3d_halo_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,%d,%s>", AM, TT, BN, WROWS, WCOLS, KS,
            BKH, TH, TW, TPS, DMA ? "true" : "false");
  const bool timed = prof_begin(s, IVF_CONV_HALO_BASE + variant_id);
  hipLaunchKernelGGL((conv3d_halo_kernel<AM, TT, BN, WROWS, WCOLS, KS, BKH, TH, TW, TPS, DMA>), grid, dim3(NT), shm, s, a, tilesT, tilesH, tilesW);
  if (timed) prof_end(s);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}


// Variant table (ivf_conv3d_desc.variant = IVF_CONV_HALO_BASE + index).
//            TT  BN  wave rows x cols  tap groups
//  0: 4 192  32 x 96  1      1: 4 128  64 x 64  1     2: 4 128  32 x 64  1     3: 4  96  32 x 96  1
//  4: 4  64  32 x 64  1      5: 4  64  64 x 64  2     6: 4  32  32 x 32  1     7: 4  32  64 x 32  2
//  8: 2 192  32 x 96  1      9: 2 128  32 x 64  1    10: 2  96  32 x 96  1    11: 2  64  32 x 64  1
// 12: 2  32  32 x 32  1     13: 2  64  64 x 64  2    14: 2 128  64 x 64  1
// 15: 4  64  32 x 64  2     16: 4  96  32 x 96  2    17: 4  32  32 x 32  2
// 16-channel chunks (two workgroups per CU):
// 18: 2 192  32 x 96  1     19: 2 128  32 x 64  1    20: 4  96  32 x 96  1    21: 4  64  32 x 64  1
// 22: 2  96  32 x 96  1
// 4 x 4 x 14 boxes (224 rows = 7 MFMA row tiles; exact on 28- and 14-wide maps):
// 23: 192  32 x 96  1     24: 128  32 x 64  1     25:  96  32 x 96  1     26:  64  32 x 64  1
// 27:  32  32 x 32  1     28:  64  32 x 32  1     29:  32  32 x 32  2     30: 128  32 x 128 1
// narrow outputs, 2-frame boxes with 16-channel chunks (two or three workgroups per CU: one's per-tap barrier
// and LDS round trip hide under another's MFMAs -- what the 32-column stem backward-data is short of):
// 31: 2 32 64 x 32 2    32: 2 32 32 x 32 2    33: 2 32 32 x 32 1    34: 2 64 32 x 64 2    35: 2 64 64 x 64 2
// 36: 4 32 64 x 32 2 (16)   37: 4 32 32 x 32 2 (16)
// several taps per barrier interval (narrow outputs; 16-channel chunks leave the LDS room):
// 38: 4 32 32 x 32 2 (16) x4 taps   39: 4 32 64 x 32 2 (16) x4   40: 4 32 32 x 32 2 (16) x2   41: 2 32 32 x 32 2 (32) x2
// 42: 4 64 32 x 64 1 (16) x2        43: 4 64 32 x 64 1 (32) x2   44: 4 32 32 x 32 1 (16) x4   45: 4 96 32 x 96 1 (16) x2
// 46: 4 32 32 x 32 2 (32) x2 (k <= 3)   47: 4x4x14 box, 64 32 x 64 1 (32) x2
// 4 x 7 x 8 boxes (7 one-h-row tiles, conflict-free like the 4 x 8 x 8 ones; exact in H on 28- and 14-row maps):
// 48: 192 32 x 96 1 (32)   49: 128 32 x 64 1 (32)   50: 96 32 x 96 1 (16)   51: 64 32 x 64 1 (16)
// weight tiles by LDS-DMA (global_load_lds_dwordx4, swizzled 64-byte rows):
// 52: 4 192 32 x 96 1 (32) 8x8   53: same, 4x14   54: 4 128 32 x 64 1 (32) 4x14   55: 4 128 32 x 64 1 (32) 8x8
// 56: 4 32 32 x 32 2 (32) 8x8    57: 4 128 32 x 64 1 (32) 7x8   58: 4 96 32 x 96 1 (32) 8x8   59: 4 64 32 x 64 1 (32) 8x8
// wide tiles on 16-channel chunks (what fits beside the THREE activation planes of the 6-pass mode):
// 60: 4 192 32 x 96 1 (16) 8x8   61: 4 128 32 x 64 1 (16) 8x8   62: 4 192 32 x 96 1 (16) 4x14   63: 4 128 32 x 64 1 (16) 4x14
// 64: 4 96 32 x 96 1 (16) 4x14   65: 4 64 32 x 64 1 (16) 4x14    66: 4 192 32 x 96 1 (16) 7x8    67: 4 128 32 x 64 1 (16) 7x8
// the same wide tiles on 8 waves of 64-row wave tiles (two waves per SIMD with 256 registers each, 9 fragment reads per
// 12 MFMA groups instead of 12 per 9):
// 68: 4 192 64 x 96 1 (16) 8x8   69: 4 128 64 x 64 1 (16) 8x8
constexpr int HALO_NUM_VARIANTS = 70;

// A variant is built for an operand mode only if its LDS footprint fits for the 3x3x3 case (the 6-pass mode holds
// three activation planes: mostly the 16-channel-chunk variants remain); anything else reports IVF_ERR_UNSUPPORTED.
template <int AM, int TT, int BN, int KS, int BKH, int TH, int TW, int TPS, bool DMA>
constexpr bool halo_variant_built() {
  constexpr int ROWB = (BKH + 8) * 2, ROWB_B = DMA ? BKH * 2 : ROWB;
  constexpr int HH = TH + 2, HW = TW + 2;
  constexpr int PS = HH * HW + ((TT == 4 && TW == 8) ? (4 - (HH * HW) % 8 + 8) % 8 : 0);
  constexpr long HR = (long)(TT + 2) * PS;
  constexpr long shm = OpPlanes<AM>::A * HR * ROWB + 2L * TPS * KS * OpPlanes<AM>::B * BN * ROWB_B + HR * 4;
  return shm <= 160 * 1024;
}
template <int AM, int TT, int BN, int WROWS, int WCOLS, int KS = 1, int BKH = 32, int TH = 8, int TW = 8, int TPS = 1, bool DMA = false>
static int launch_halo_if(ConvKArgs& a, int variant_id, hipStream_t s) {
  if constexpr (halo_variant_built<AM, TT, BN, KS, BKH, TH, TW, TPS, DMA>()) {
    return launch_halo<AM, TT, BN, WROWS, WCOLS, KS, BKH, TH, TW, TPS, DMA>(a, variant_id, s);
  } else {
    set_error("conv3d_halo: variant %d needs more than 160 KB of LDS in this arithmetic mode", variant_id);
    return IVF_ERR_UNSUPPORTED;
  }
}

template <int AM>
int conv_halo_launch_variant_am(ConvKArgs& a, int v, hipStream_t s) {
  switch (v) {
    case 0: return launch_halo_if<AM, 4, 192, 32, 96>(a, 0, s);
    case 1: return launch_halo_if<AM, 4, 128, 64, 64>(a, 1, s);
    case 2: return launch_halo_if<AM, 4, 128, 32, 64>(a, 2, s);
    case 3: return launch_halo_if<AM, 4, 96, 32, 96>(a, 3, s);
    case 4: return launch_halo_if<AM, 4, 64, 32, 64>(a, 4, s);
    case 5: return launch_halo_if<AM, 4, 64, 64, 64, 2>(a, 5, s);
    case 6: return launch_halo_if<AM, 4, 32, 32, 32>(a, 6, s);
    case 7: return launch_halo_if<AM, 4, 32, 64, 32, 2>(a, 7, s);
    case 8: return launch_halo_if<AM, 2, 192, 32, 96>(a, 8, s);
    case 9: return launch_halo_if<AM, 2, 128, 32, 64>(a, 9, s);
    case 10: return launch_halo_if<AM, 2, 96, 32, 96>(a, 10, s);
    case 11: return launch_halo_if<AM, 2, 64, 32, 64>(a, 11, s);
    case 12: return launch_halo_if<AM, 2, 32, 32, 32>(a, 12, s);
    case 13: return launch_halo_if<AM, 2, 64, 64, 64, 2>(a, 13, s);
    case 14: return launch_halo_if<AM, 2, 128, 64, 64>(a, 14, s);
    case 15: return launch_halo_if<AM, 4, 64, 32, 64, 2>(a, 15, s);
    case 16: return launch_halo_if<AM, 4, 96, 32, 96, 2>(a, 16, s);
    case 17: return launch_halo_if<AM, 4, 32, 32, 32, 2>(a, 17, s);
    case 18: return launch_halo_if<AM, 2, 192, 32, 96, 1, 16>(a, 18, s);
    case 19: return launch_halo_if<AM, 2, 128, 32, 64, 1, 16>(a, 19, s);
    case 20: return launch_halo_if<AM, 4, 96, 32, 96, 1, 16>(a, 20, s);
    case 21: return launch_halo_if<AM, 4, 64, 32, 64, 1, 16>(a, 21, s);
    case 22: return launch_halo_if<AM, 2, 96, 32, 96, 1, 16>(a, 22, s);
    case 23: return launch_halo_if<AM, 4, 192, 32, 96, 1, 32, 4, 14>(a, 23, s);
    case 24: return launch_halo_if<AM, 4, 128, 32, 64, 1, 32, 4, 14>(a, 24, s);
    case 25: return launch_halo_if<AM, 4, 96, 32, 96, 1, 32, 4, 14>(a, 25, s);
    case 26: return launch_halo_if<AM, 4, 64, 32, 64, 1, 32, 4, 14>(a, 26, s);
    case 27: return launch_halo_if<AM, 4, 32, 32, 32, 1, 32, 4, 14>(a, 27, s);
    case 28: return launch_halo_if<AM, 4, 64, 32, 32, 1, 32, 4, 14>(a, 28, s);
    case 29: return launch_halo_if<AM, 4, 32, 32, 32, 2, 32, 4, 14>(a, 29, s);
    case 30: return launch_halo_if<AM, 4, 128, 32, 128, 1, 32, 4, 14>(a, 30, s);
    case 31: return launch_halo_if<AM, 2, 32, 64, 32, 2, 16>(a, 31, s);
    case 32: return launch_halo_if<AM, 2, 32, 32, 32, 2, 16>(a, 32, s);
    case 33: return launch_halo_if<AM, 2, 32, 32, 32, 1, 16>(a, 33, s);
    case 34: return launch_halo_if<AM, 2, 64, 32, 64, 2, 16>(a, 34, s);
    case 35: return launch_halo_if<AM, 2, 64, 64, 64, 2, 16>(a, 35, s);
    case 36: return launch_halo_if<AM, 4, 32, 64, 32, 2, 16>(a, 36, s);
    case 37: return launch_halo_if<AM, 4, 32, 32, 32, 2, 16>(a, 37, s);
    case 38: return launch_halo_if<AM, 4, 32, 32, 32, 2, 16, 8, 8, 4>(a, 38, s);
    case 39: return launch_halo_if<AM, 4, 32, 64, 32, 2, 16, 8, 8, 4>(a, 39, s);
    case 40: return launch_halo_if<AM, 4, 32, 32, 32, 2, 16, 8, 8, 2>(a, 40, s);
    case 41: return launch_halo_if<AM, 2, 32, 32, 32, 2, 32, 8, 8, 2>(a, 41, s);
    case 42: return launch_halo_if<AM, 4, 64, 32, 64, 1, 16, 8, 8, 2>(a, 42, s);
    case 43: return launch_halo_if<AM, 4, 64, 32, 64, 1, 32, 8, 8, 2>(a, 43, s);
    case 44: return launch_halo_if<AM, 4, 32, 32, 32, 1, 16, 8, 8, 4>(a, 44, s);
    case 45: return launch_halo_if<AM, 4, 96, 32, 96, 1, 16, 8, 8, 2>(a, 45, s);
    case 46: return launch_halo_if<AM, 4, 32, 32, 32, 2, 32, 8, 8, 2>(a, 46, s);
    case 47: return launch_halo_if<AM, 4, 64, 32, 64, 1, 32, 4, 14, 2>(a, 47, s);
    case 48: return launch_halo_if<AM, 4, 192, 32, 96, 1, 32, 7, 8>(a, 48, s);
    case 49: return launch_halo_if<AM, 4, 128, 32, 64, 1, 32, 7, 8>(a, 49, s);
    case 50: return launch_halo_if<AM, 4, 96, 32, 96, 1, 16, 7, 8>(a, 50, s);
    case 51: return launch_halo_if<AM, 4, 64, 32, 64, 1, 16, 7, 8>(a, 51, s);
    case 52: return launch_halo_if<AM, 4, 192, 32, 96, 1, 32, 8, 8, 1, true>(a, 52, s);
    case 53: return launch_halo_if<AM, 4, 192, 32, 96, 1, 32, 4, 14, 1, true>(a, 53, s);
    case 54: return launch_halo_if<AM, 4, 128, 32, 64, 1, 32, 4, 14, 1, true>(a, 54, s);
    case 55: return launch_halo_if<AM, 4, 128, 32, 64, 1, 32, 8, 8, 1, true>(a, 55, s);
    case 56: return launch_halo_if<AM, 4, 32, 32, 32, 2, 32, 8, 8, 1, true>(a, 56, s);
    case 57: return launch_halo_if<AM, 4, 128, 32, 64, 1, 32, 7, 8, 1, true>(a, 57, s);
    case 58: return launch_halo_if<AM, 4, 96, 32, 96, 1, 32, 8, 8, 1, true>(a, 58, s);
    case 59: return launch_halo_if<AM, 4, 64, 32, 64, 1, 32, 8, 8, 1, true>(a, 59, s);
    case 60: return launch_halo_if<AM, 4, 192, 32, 96, 1, 16>(a, 60, s);
    case 61: return launch_halo_if<AM, 4, 128, 32, 64, 1, 16>(a, 61, s);
    case 62: return launch_halo_if<AM, 4, 192, 32, 96, 1, 16, 4, 14>(a, 62, s);
    case 63: return launch_halo_if<AM, 4, 128, 32, 64, 1, 16, 4, 14>(a, 63, s);
    case 64: return launch_halo_if<AM, 4, 96, 32, 96, 1, 16, 4, 14>(a, 64, s);
    case 65: return launch_halo_if<AM, 4, 64, 32, 64, 1, 16, 4, 14>(a, 65, s);
    case 66: return launch_halo_if<AM, 4, 192, 32, 96, 1, 16, 7, 8>(a, 66, s);
    case 67: return launch_halo_if<AM, 4, 128, 32, 64, 1, 16, 7, 8>(a, 67, s);
    case 68: return launch_halo_if<AM, 4, 192, 64, 96, 1, 16>(a, 68, s);
    case 69: return launch_halo_if<AM, 4, 128, 64, 64, 1, 16>(a, 69, s);
  }
  set_error("conv3d_halo: unknown variant %d", v);
  return IVF_ERR_BAD_ARG;
}

}  // namespace ivf
