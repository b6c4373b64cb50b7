// Stride-1 k x k x k (k <= 4) convolution, split-bf16 (3-pass) MFMA, with the input
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
#include "conv_common.h"

namespace ivf {

constexpr int TH = 8, TW = 8;
constexpr int ROWB = LDS_ROW_BF * 2;  // 80 bytes per LDS row per plane

template <int TT, int BN, int WROWS, int WCOLS>
__global__ __launch_bounds__((TT * 64 / WROWS) * (BN / WCOLS) * 64) void conv3d_halo_kernel(ConvKArgs a, int tilesT,
                                                                                           int tilesH, int tilesW) {
  constexpr int BM = TT * TH * TW;
  constexpr int WM = BM / WROWS, WN = BN / WCOLS;
  constexpr int NT = WM * WN * 64;
  constexpr int TM = WROWS / 32, TN = WCOLS / 32;
  constexpr int BLOADS = (BN * 4 + NT - 1) / NT;   // 16-byte weight loads per thread per plane per tap

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HT = TT + a.kT - 1, HH = TH + a.kH - 1, HW = TW + a.kW - 1;
  const int HR = HT * HH * HW;
  unsigned char* a_hi = smem;
  unsigned char* a_lo = smem + (size_t)HR * ROWB;
  unsigned char* b_base = smem + (size_t)2 * HR * ROWB;   // [2 buffers][hi, lo][BN rows]
  int* rowoff = reinterpret_cast<int*>(b_base + (size_t)4 * BN * ROWB);   // [HR] input offset of a halo row / in_ld, or -1

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
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
    arow[i] = ((r >> 6) * HH + ((r >> 3) & 7)) * HW + (r & 7);
  }
  const int ntaps = a.kT * a.kH * a.kW;
  const int khw = a.kH * a.kW;
  // halo row -> input position (decoded once; the chunk loop only adds the channel offset)
  for (int row = tid; row < HR; row += NT) {
    int hw = row % HW;
    int r2 = row / HW;
    int hh = r2 % HH;
    int ht = r2 / HH;
    int ti = t0 - a.pT + ht, hi = h0 - a.pH + hh, wi = w0 - a.pW + hw;
    bool ok = (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi;
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
  uint4 rbh[PF][BLOADS], rbl[PF][BLOADS];
  auto load_b = [&](int slot, int tap, int c0) {
#pragma unroll
    for (int q = 0; q < BLOADS; ++q) {
      int idx = tid + q * NT;
      int row = idx >> 2, g2 = idx & 3;
      int n = n0 + row;
      int c = c0 + 8 * g2;
      uint4 h = make_uint4(0u, 0u, 0u, 0u), l = h;
      if (row < BN && n < a.Cout && c < a.Cin) {
        const unsigned short* p = a.wbf + (size_t)n * a.ldw + (size_t)tap * a.Cin + c;
        h = *reinterpret_cast<const uint4*>(p);
        l = *reinterpret_cast<const uint4*>(p + a.w_lo_off);
      }
      rbh[slot][q] = h;
      rbl[slot][q] = l;
    }
  };
  auto store_b = [&](int slot, int buf) {
    unsigned char* bh = b_base + (size_t)buf * 2 * BN * ROWB;
    unsigned char* bl = bh + (size_t)BN * ROWB;
#pragma unroll
    for (int q = 0; q < BLOADS; ++q) {
      int idx = tid + q * NT;
      int row = idx >> 2, g2 = idx & 3;
      if (row < BN) {
        *reinterpret_cast<uint4*>(bh + row * ROWB + 16 * g2) = rbh[slot][q];
        *reinterpret_cast<uint4*>(bl + row * ROWB + 16 * g2) = rbl[slot][q];
      }
    }
  };
  auto mma_tap = [&](int tap, int buf, int nks) {
    const int kt = tap / khw;
    const int rem = tap - kt * khw;
    const int kh = rem / a.kW;
    const int kw = rem - kh * a.kW;
    const int toff = ((kt * HH + kh) * HW + kw) * ROWB;
    const unsigned char* bh = b_base + (size_t)buf * 2 * BN * ROWB;
    const unsigned char* bl = bh + (size_t)BN * ROWB;
    for (int ks = 0; ks < nks; ++ks) {
      bf16x8 fah[TM], fal[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int off = (a.dbg & 2) ? 16 * lh : arow[i] * ROWB + toff + ks * 32 + 16 * lh;
        fah[i] = *reinterpret_cast<const bf16x8*>(a_hi + off);
        fal[i] = *reinterpret_cast<const bf16x8*>(a_lo + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int off = (a.dbg & 4) ? 16 * lh : (wn * WCOLS + j * 32 + li) * ROWB + ks * 32 + 16 * lh;
        bf16x8 fbh = *reinterpret_cast<const bf16x8*>(bh + off);
        bf16x8 fbl = *reinterpret_cast<const bf16x8*>(bl + off);
        if (a.dbg & 1) {   // ablation: no MFMAs, keep the fragment reads alive
#pragma unroll
          for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fah[i]), "v"(fal[i]));
          asm volatile("" ::"v"(fbh), "v"(fbl));
          continue;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fbh, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbl, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[i], fbh, acc[i][j], 0, 0, 0);
        }
      }
    }
  };

  for (int c0 = 0; c0 < a.Cin; c0 += BK) {
    // weight tiles of the first PF taps start flying before the halo is staged
#pragma unroll
    for (int u = 0; u < PF; ++u)
      if (u < ntaps) load_b(u, u, c0);
    __syncthreads();   // everyone is done with the previous chunk's halo and weight buffers
    // ---- stage the halo for channels [c0, c0+32): 8 float4 groups per row
    const int ngroups = HR * 8;
    for (int base = 0; base < ngroups; base += 4 * NT) {
      float4 v[4];
      int dst[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        int idx = base + u * NT + tid;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[u] = -1;
        if (idx < ngroups) {
          int row = idx >> 3, g = idx & 7;
          int c = c0 + 4 * g;
          int pos = rowoff[row];
          dst[u] = row * ROWB + 8 * g;
          if (pos >= 0 && c < a.Cin)
            v[u] = *reinterpret_cast<const float4*>(a.in + (size_t)pos * a.in_ld + a.in_coff + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (dst[u] >= 0) {
          uint2 h, l;
          split4(v[u], &h, &l);
          *reinterpret_cast<uint2*>(a_hi + dst[u]) = h;
          *reinterpret_cast<uint2*>(a_lo + dst[u]) = l;
        }
      }
    }
    store_b(0, 0);
    __syncthreads();

    const int cw = min(BK, a.Cin - c0);
    const int nks = (cw + 15) >> 4;
    // tap loop unrolled by PF so the register ring is statically indexed: at tap (slot u)
    // the LDS buffer tap&1 holds its weights; slot u is refilled with tap+PF, and slot u+1's
    // weights (tap+1, loaded PF-1 taps ago) move to the other LDS buffer after the MFMAs.
    for (int tap0 = 0; tap0 < ntaps; tap0 += PF) {
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const int tap = tap0 + u;
        if (tap < ntaps) {
          if (tap + PF < ntaps && !(a.dbg & 8)) load_b(u, tap + PF, c0);
          if (!(a.dbg & 32)) mma_tap(tap, tap & 1, nks);
          if (tap + 1 < ntaps && !(a.dbg & 8)) store_b((u + 1) % PF, (tap + 1) & 1);
          if (!(a.dbg & 16)) __syncthreads();
        }
      }
    }
  }

  conv_epilogue<TM, TN>(
      a, acc,
      [&](int row) {
        int t = t0 + (row >> 6), h = h0 + ((row >> 3) & 7), w = w0 + (row & 7);
        if (t >= a.To || h >= a.Ho || w >= a.Wo) return -1;
        return ((b * a.To + t) * a.Ho + h) * a.Wo + w;
      },
      wm * WROWS, n0 + wn * WCOLS, li, lh);
}

template <int TT, int BN, int WROWS, int WCOLS>
static int launch_halo(ConvKArgs& a, hipStream_t s) {
  constexpr int NT = (TT * 64 / WROWS) * (BN / WCOLS) * 64;
  const int HR = (TT + a.kT - 1) * (TH + a.kH - 1) * (TW + a.kW - 1);
  const size_t shm = (size_t)2 * HR * ROWB + (size_t)2 * 2 * BN * ROWB + (size_t)HR * sizeof(int);
  if (shm > 160 * 1024) {
    set_error("conv3d_halo: %zu bytes of LDS needed", shm);
    return IVF_ERR_UNSUPPORTED;
  }
  static bool attr_set = false;
  if (!attr_set) {
    IVF_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_halo_kernel<TT, BN, WROWS, WCOLS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int tilesT = cdiv(a.To, TT), tilesH = cdiv(a.Ho, TH), tilesW = cdiv(a.Wo, TW);
  a.ntiles = cdiv(a.Cout, BN);
  a.mtiles = a.B * tilesT * tilesH * tilesW;
  dim3 grid(a.mtiles * a.ntiles);
  const bool timed = prof_begin(s, BN >= 128 ? 0 : (BN >= 64 ? 1 : 2));
  hipLaunchKernelGGL((conv3d_halo_kernel<TT, BN, WROWS, WCOLS>), grid, dim3(NT), shm, s, a, tilesT, tilesH, tilesW);
  if (timed) prof_end(s);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

int conv_halo_supported(const ConvKArgs& a) {
  if (a.sT != 1 || a.sH != 1 || a.sW != 1) return 0;
  if (a.kT * a.kH * a.kW <= 1 || a.kT > 4 || a.kH > 4 || a.kW > 4) return 0;
  if (a.Cin % 8) return 0;
  return 1;
}

int conv_halo_launch(ConvKArgs& a, hipStream_t s) {
  // Output-channel tile width: every tile re-stages the halo, so weigh padded columns against
  // the number of tiles (a staging pass costs about as much as ~40 columns of MFMA work).
  static const int widths[5] = {192, 128, 96, 64, 32};
  int best = 32, best_cost = 1 << 30;
  for (int w : widths) {
    int cost = cdiv(a.Cout, w) * (w + 40);
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  const bool deep = a.To >= 4;   // 4-frame boxes when the map has them, else 2-frame boxes
  switch (best) {
    case 192: return deep ? launch_halo<4, 192, 32, 96>(a, s) : launch_halo<2, 192, 32, 96>(a, s);
    case 128: return deep ? launch_halo<4, 128, 64, 64>(a, s) : launch_halo<2, 128, 32, 64>(a, s);
    case 96: return deep ? launch_halo<4, 96, 32, 96>(a, s) : launch_halo<2, 96, 32, 96>(a, s);
    case 64: return deep ? launch_halo<4, 64, 32, 64>(a, s) : launch_halo<2, 64, 32, 64>(a, s);
    default: return deep ? launch_halo<4, 32, 32, 32>(a, s) : launch_halo<2, 32, 32, 32>(a, s);
  }
}

}  // namespace ivf
