// 3-D convolution forward / backward-data as implicit GEMM on the gfx950 fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 FMA chains, so results track
// the reference's fp32 CPU path to rounding).
//
// Replaces, on the saliency path, torch.nn.Conv3d + F.pad + BatchNorm3d(eval) +
// ReLU of the reference's Unit3D (video_features_pytorch/models/I3D_doubled.py:83-118)
// and the autograd backward-data of the same (loss.backward(),
// FindMasksComparison_I3D_smth.py:213).
//
// Data layout: activations are channels-last [B, T, H, W, ld] fp32 so the GEMM K
// dimension (taps x input channels) is contiguous per tap and every global load
// is a 16-byte vector of 4 channels.  Weights are packed [Cout][K] with
// K = ((kt*kH + kh)*kW + kw)*Cin + ci (ci fastest), BN(eval) folded into a
// per-channel scale/shift applied in the epilogue.
//
// GEMM view: D[m][n] = sum_k A[m][k] * Wp[n][k], m = output position
// (b,to,ho,wo), n = output channel.  Tile BM x BN x 32, 256 threads = 4 waves,
// each wave owns (BM/WM) x (BN/WN) as 32x32 MFMA tiles.  A and W chunks are
// staged global -> registers -> LDS ([rows][32+4] floats: the +4 pad makes the
// ds_read_b128 fragment reads conflict-free), the next chunk's global loads are
// in flight while the current chunk's MFMAs run.
#include <cstdlib>

#include "conv_common.h"

namespace ivf {

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv3d_igemm_kernel(ConvKArgs a) {
  constexpr int TM = BM / WM / 32;  // 32x32 tiles per wave along M
  constexpr int TN = BN / WN / 32;
  constexpr int AROWS = BM / 32;    // rows of A each thread stages per chunk
  constexpr int BROWS = BN / 32;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(TM >= 1 && TN >= 1, "tile");

  __shared__ __attribute__((aligned(16))) float lds_a[BM * LDS_LD];
  __shared__ __attribute__((aligned(16))) float lds_b[BN * LDS_LD];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
  const int mt = tile / a.ntiles;
  const int nt = tile - mt * a.ntiles;
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  // staging role: float4 group g of the 32-wide chunk, rows r0 + 32 j
  const int g = tid & 7;
  const int r0 = tid >> 3;

  // decode this thread's A rows once
  int a_base[AROWS];   // element offset of (b, ti0, hi0, wi0) before tap offset; may be "virtual"
  int a_t0[AROWS], a_h0[AROWS], a_w0[AROWS];
#pragma unroll
  for (int j = 0; j < AROWS; ++j) {
    int m = m0 + r0 + 32 * j;
    if (m < a.M) {
      int wo = m % a.Wo;
      int t1 = m / a.Wo;
      int ho = t1 % a.Ho;
      int t2 = t1 / a.Ho;
      int to = t2 % a.To;
      int b = t2 / a.To;
      a_t0[j] = to * a.sT - a.pT;
      a_h0[j] = ho * a.sH - a.pH;
      a_w0[j] = wo * a.sW - a.pW;
      a_base[j] = b * a.Ti;
    } else {
      a_t0[j] = -100000;  // every tap out of range -> zero rows
      a_h0[j] = 0;
      a_w0[j] = 0;
      a_base[j] = 0;
    }
  }
  const float* wrow[BROWS];
#pragma unroll
  for (int j = 0; j < BROWS; ++j) {
    int n = n0 + r0 + 32 * j;
    wrow[j] = (n < a.Cout) ? a.w + (size_t)n * a.K : nullptr;
  }

  float4 ra[AROWS], rb[BROWS];
  const int khw = a.kH * a.kW;

  auto load_chunk = [&](int k0) {
    int kk = k0 + 4 * g;
    bool kvalid = kk < a.K;
    int tap = kvalid ? kk / a.Cin : 0;
    int ci = kk - tap * a.Cin;
    int kt = tap / khw;
    int rem = tap - kt * khw;
    int kh = rem / a.kW;
    int kw = rem - kh * a.kW;
#pragma unroll
    for (int j = 0; j < AROWS; ++j) {
      int ti = a_t0[j] + kt, hi = a_h0[j] + kh, wi = a_w0[j] + kw;
      bool ok = kvalid && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi &&
                (unsigned)wi < (unsigned)a.Wi;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) {
        size_t pos = (size_t)((a_base[j] + ti) * a.Hi + hi) * a.Wi + wi;
        v = (a.in2 && kk >= a.K0)
                ? *reinterpret_cast<const float4*>(a.in2 + pos * a.in2_ld + a.in2_coff + (kk - a.K0))
                : *reinterpret_cast<const float4*>(a.in + pos * a.in_ld + a.in_coff + ci);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (kvalid && wrow[j]) v = *reinterpret_cast<const float4*>(wrow[j] + kk);
      rb[j] = v;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (a.K + BK - 1) / BK;
  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
#pragma unroll
    for (int j = 0; j < AROWS; ++j)
      *reinterpret_cast<float4*>(&lds_a[(r0 + 32 * j) * LDS_LD + 4 * g]) = ra[j];
#pragma unroll
    for (int j = 0; j < BROWS; ++j)
      *reinterpret_cast<float4*>(&lds_b[(r0 + 32 * j) * LDS_LD + 4 * g]) = rb[j];
    __syncthreads();
    if (c + 1 < nchunks) load_chunk((c + 1) * BK);

#pragma unroll
    for (int k8 = 0; k8 < BK / 8; ++k8) {
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const float4*>(
            &lds_a[(wm * (BM / WM) + i * 32 + li) * LDS_LD + k8 * 8 + 4 * lh]);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const float4*>(
            &lds_b[(wn * (BN / WN) + j * 32 + li) * LDS_LD + k8 * 8 + 4 * lh]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  conv_epilogue<AM_X3, TM, TN>(a, acc, [&](int row) { int m = m0 + row; return m < a.M ? m : -1; }, wm * (BM / WM),
                               n0 + wn * (BN / WN), li, lh);   // (AM_X3: fp32 storage)
}

// ------------------------------------------------------------------ split-bf16 variants
// fp32-level accuracy on the bf16 matrix cores: x = hi + lo with hi = bf16(x), lo = bf16(x - hi);
// a*b ~= a_lo*b_hi + a_hi*b_lo + a_hi*b_hi (the dropped lo*lo term is 2^-18 relative), each
// product exact in fp32, fp32 accumulation in the MFMA.  3 x v_mfma_f32_32x32x16_bf16 per 16-deep
// k-step = 96 cycles against 512 for the fp32 MFMA form.  Activations stay fp32 in HBM and are
// split while they are staged into LDS; weights are pre-split by the pack kernels.
// AM (conv_common.h) selects the operand planes: AM_X3 as above, AM_X6 the 3-way split with six passes (fp32-class
// products), AM_BF16 bf16 activations staged as they are with two passes and bf16 stores.
// PW = pointwise (1x1x1, stride 1, no padding: every 1x1x1 unit of the Inception modules and their fused
// backward GEMM): A is a plain [M][K] matrix, so the per-chunk tap decode (three integer divisions and the bounds
// tests per thread: 12-41 vector instructions per MFMA, measured) drops out; rows are addressed through two
// offsets computed once, loads are unconditional (rows past M and the k tail are clamped: they meet zero weights
// or land in rows nobody stores).
template <int AM, int BM, int BN, int WM, int WN, bool PW>
__global__ __launch_bounds__(256) void conv3d_igemm_bf16_kernel(ConvKArgs a) {
  constexpr int NPA = OpPlanes<AM>::A, NPB = OpPlanes<AM>::B;
  constexpr int TM = BM / WM / 32;
  constexpr int TN = BN / WN / 32;
  constexpr int AROWS = BM / 32;
  constexpr int BROWS = (BN + 63) / 64;   // weight rows each thread stages per plane (8 bf16 per load)
  static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile");
  static_assert((NPA * BM + NPB * BN) * LDS_ROW_BF * 2 <= 64 * 1024, "static LDS");

  __shared__ __attribute__((aligned(16))) unsigned short a_pl[NPA][BM * LDS_ROW_BF];
  __shared__ __attribute__((aligned(16))) unsigned short b_pl[NPB][BN * LDS_ROW_BF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, a.mtiles * a.ntiles);
  const int mt = tile / a.ntiles;
  const int nt = tile - mt * a.ntiles;
  const int m0 = mt * BM;
  const int n0 = nt * BN;

  const int g = tid & 7;            // A: 4-channel group of the 32-wide chunk
  const int r0 = perm8(tid >> 3);   //    rows r0 + 32 j (perm8: conflict-free LDS staging writes)
  const int g2 = tid & 3;           // B: group of 8 bf16
  const int q0 = perm8(tid >> 2);   //    rows q0 + 64 j

  int a_base[AROWS], a_t0[AROWS], a_h0[AROWS], a_w0[AROWS];
  size_t a_o1[AROWS], a_o2[AROWS];   // PW: element offsets of the row in `in` / `in2` (the latter minus K0)
#pragma unroll
  for (int j = 0; j < AROWS; ++j) {
    int m = m0 + r0 + 32 * j;
    if constexpr (PW) {
      const size_t mc = (size_t)min(m, a.M - 1);
      a_o1[j] = mc * a.in_ld + a.in_coff;
      a_o2[j] = a.in2 ? mc * a.in2_ld + a.in2_coff - a.K0 : a_o1[j];
      a_base[j] = a_t0[j] = a_h0[j] = a_w0[j] = 0;
    } else if (m < a.M) {
      int wo = m % a.Wo;
      int t1 = m / a.Wo;
      int ho = t1 % a.Ho;
      int t2 = t1 / a.Ho;
      int to = t2 % a.To;
      int b = t2 / a.To;
      a_t0[j] = to * a.sT - a.pT;
      a_h0[j] = ho * a.sH - a.pH;
      a_w0[j] = wo * a.sW - a.pW;
      a_base[j] = b * a.Ti;
      // element offset of the window origin (may lie before the tensor: wraps, and is only used added to an in-range tap)
      a_o1[j] = (size_t)(((long long)((a_base[j] + a_t0[j]) * a.Hi + a_h0[j]) * a.Wi + a_w0[j]) * a.in_ld + a.in_coff);
      a_o2[j] = 0;
    } else {
      a_t0[j] = -100000; a_h0[j] = 0; a_w0[j] = 0; a_base[j] = 0;
      a_o1[j] = a_o2[j] = 0;
    }
  }
  const unsigned short* wrow[BROWS];
#pragma unroll
  for (int j = 0; j < BROWS; ++j) {
    int n = n0 + q0 + 64 * j;
    wrow[j] = (q0 + 64 * j < BN && n < a.Cout) ? a.wbf + (size_t)n * a.ldw : nullptr;
  }

  float4 ra[AROWS];
  uint4 rb[NPB][BROWS];
  // tap decode of k = 4 g (chunk 0) and the per-chunk step BK in the same mixed radix (Cin, kW, kH, kT)
  int d_ci, d_kw, d_kh, d_kt, s_ci, s_kw, s_kh, s_kt;
  {
    const int khw = a.kH * a.kW;
    int tap = (4 * g) / a.Cin;
    d_ci = 4 * g - tap * a.Cin;
    d_kt = tap / khw;
    int rem = tap - d_kt * khw;
    d_kh = rem / a.kW;
    d_kw = rem - d_kh * a.kW;
    tap = BK / a.Cin;
    s_ci = BK - tap * a.Cin;
    s_kt = tap / khw;
    rem = tap - s_kt * khw;
    s_kh = rem / a.kW;
    s_kw = rem - s_kh * a.kW;
  }

  auto load_chunk = [&](int k0) {
    if constexpr (PW) {
      const int kk = min(k0 + 4 * g, a.K - 4);
      const bool second = a.in2 && kk >= a.K0;
#pragma unroll
      for (int j = 0; j < AROWS; ++j)
        ra[j] = second ? load_act4<AM>(a.in2, a_o2[j] + kk) : load_act4<AM>(a.in, a_o1[j] + kk);
    } else {
      // this thread's k = k0 + 4 g as (tap (kt, kh, kw), channel): kept as running state, advanced by one chunk per
      // call -- chunks are visited in order -- with carries instead of three integer divisions per chunk (on the
      // 4-channel stem, 8 taps per chunk, the decode was a third of the vector instructions of a VALU-bound kernel)
      const bool kvalid = d_kt < a.kT;
      const size_t tapoff = (size_t)((long long)((d_kt * a.Hi + d_kh) * a.Wi + d_kw) * a.in_ld + d_ci);   // same for every row
#pragma unroll
      for (int j = 0; j < AROWS; ++j) {
        int ti = a_t0[j] + d_kt, hi = a_h0[j] + d_kh, wi = a_w0[j] + d_kw;
        bool ok = kvalid && (unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi &&
                  (unsigned)wi < (unsigned)a.Wi;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = load_act4<AM>(a.in, a_o1[j] + tapoff);
        ra[j] = v;
      }
      d_ci += s_ci;
      int cw = s_kw, ch = s_kh;
      if (d_ci >= a.Cin) { d_ci -= a.Cin; ++cw; }
      d_kw += cw;
      if (d_kw >= a.kW) { d_kw -= a.kW; ++ch; }
      d_kh += ch;
      if (d_kh >= a.kH) { d_kh -= a.kH; ++d_kt; }
      d_kt += s_kt;
    }
    int kb = k0 + 8 * g2;
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
#pragma unroll
      for (int pl = 0; pl < NPB; ++pl) {
        uint4 h = make_uint4(0u, 0u, 0u, 0u);
        if (kb < a.ldw && wrow[j]) h = *reinterpret_cast<const uint4*>(wrow[j] + pl * a.w_lo_off + kb);
        rb[pl][j] = h;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nchunks = (a.K + BK - 1) / BK;
  load_chunk(0);
  for (int c = 0; c < nchunks; ++c) {
#pragma unroll
    for (int j = 0; j < AROWS; ++j)
      stage_planes<AM>(reinterpret_cast<unsigned char*>(&a_pl[0][(r0 + 32 * j) * LDS_ROW_BF + 4 * g]),
                       sizeof(a_pl[0]), ra[j]);
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      if (q0 + 64 * j < BN) {
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) *reinterpret_cast<uint4*>(&b_pl[pl][(q0 + 64 * j) * LDS_ROW_BF + 8 * g2]) = rb[pl][j];
      }
    }
    __syncthreads();
    if (c + 1 < nchunks) load_chunk((c + 1) * BK);

#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 fa[TM][NPA], fb[TN][NPB];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int off = (wm * (BM / WM) + i * 32 + li) * LDS_ROW_BF + ks * 16 + 8 * lh;
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) fa[i][pl] = *reinterpret_cast<const bf16x8*>(&a_pl[pl][off]);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        int off = (wn * (BN / WN) + j * 32 + li) * LDS_ROW_BF + ks * 16 + 8 * lh;
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) fb[j][pl] = *reinterpret_cast<const bf16x8*>(&b_pl[pl][off]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_planes<AM>(fa[i], fb[j], acc[i][j]);
    }
    __syncthreads();
  }
  conv_epilogue<AM, TM, TN>(a, acc, [&](int row) { int m = m0 + row; return m < a.M ? m : -1; }, wm * (BM / WM),
                            n0 + wn * (BN / WN), li, lh);
}

// ------------------------------------------------------------------ weight packing
// Reference layout [Cout][Cin][kT][kH][kW] (I3D_doubled.py:64-71) ->
// forward pack  Wf[co][(tap)*CinP + ci], zero for ci >= Cin (CinP = padded Cin)
// store one packed weight: fp32, or split into the hi / lo bf16 planes ([rows][ldw] each)
__device__ __forceinline__ void store_packed(float* out, int math, size_t row, int k, int ldw, size_t rows,
                                             float v) {
  if (math == 0) {
    out[row * ldw + k] = v;
  } else {
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    unsigned h = pk_bf16(v, 0.f) & 0xffffu;
    const float r1 = v - __uint_as_float(h << 16);
    unsigned l = pk_bf16(r1, 0.f) & 0xffffu;
    o[row * ldw + k] = (unsigned short)h;
    o[rows * (size_t)ldw + row * ldw + k] = (unsigned short)l;   // lo (2 planes) or mid (3 planes)
    if (math == IVF_MATH_BF16X6)
      o[2 * rows * (size_t)ldw + row * ldw + k] = (unsigned short)(pk_bf16(r1 - __uint_as_float(l << 16), 0.f) & 0xffffu);
  }
}

// rows [row0, row0 + Cout) of a packed matrix with rows_total rows (units packed side by side)
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout,
                                int Cin, int CinP, int taps, int ldw, int math, int row0, int rows_total) {
  size_t total = (size_t)Cout * ldw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int k = i % ldw;
    int co = i / ldw;
    int ci = k % CinP;
    int tap = k / CinP;
    float v = (tap < taps && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f;
    store_packed(out, math, row0 + co, k, ldw, rows_total, v);
  }
}

// backward-data pack for a stride-1 conv: a conv over dY with flipped taps,
// Wb[ci][(tapf)*Cout + co] = scale[co] * W[co][ci][kT-1-kt][kH-1-kh][kW-1-kw]
// 1x1x1 units of one Inception module packed side by side for a single backward GEMM:
// out[ci][koff + co] = scale[co] * w[co][ci], rows of length ldw (columns outside [koff,koff+Cout)
// belong to the other units or are padding zeroed by the first unit, koff == 0).
__global__ void pack_bwd_fused1x1_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                         float* __restrict__ out, int Cout, int Cin, int CinRows, int koff,
                                         int ktotal, int ldw, int math) {
  // every unit writes its own columns; the unit that ends at ktotal also zeroes the row padding
  const int span = Cout + ((koff + Cout == ktotal) ? ldw - ktotal : 0);
  size_t total = (size_t)CinRows * span;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int kk = i % span;
    int ci = i / span;
    float v = (kk < Cout && ci < Cin) ? (scale ? scale[kk] : 1.f) * w[(size_t)kk * Cin + ci] : 0.f;
    store_packed(out, math, ci, koff + kk, ldw, CinRows, v);
  }
}

__global__ void pack_bwd_s1_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                   float* __restrict__ out, int Cout, int Cin, int CinRows, int kT,
                                   int kH, int kW, int ldw, int math) {
  int taps = kT * kH * kW;
  size_t total = (size_t)CinRows * ldw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int k = i % ldw;
    int ci = i / ldw;
    int co = k % Cout;
    int tapf = k / Cout;
    int tap = taps - 1 - tapf;  // flipping all three dims == reversing the flat tap index
    float v = 0.f;
    if (tapf < taps && ci < Cin) v = (scale ? scale[co] : 1.f) * w[((size_t)co * Cin + ci) * taps + tap];
    store_packed(out, math, ci, k, ldw, CinRows, v);
  }
}

// backward-data pack for stride 2 in every strided dim, as a stride-1 conv over dY
// producing 2x2x2 output blocks (depth-to-space): for block offset d in [0,kb)
// (kb = ceil(k/2)+... see host), parity p, the forward tap is k = p + pad - 2*(d - dlo)
// rows n = ((pt*2+ph)*2+pw)*CinP + ci ; cols (dt*KH+dh)*KW+dw)*Cout + co
struct D2SPack {
  int kT, kH, kW;     // forward kernel
  int sT, sH, sW;     // forward strides (1 or 2 each)
  int pT, pH, pW;     // forward front pads
  int KT, KH, KW;     // block-conv kernel extents
  int oT, oH, oW;     // block-conv front pads (dY index = blk + d - o)
};
__global__ void pack_bwd_d2s_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                    float* __restrict__ out, int Cout, int Cin, int CinP, D2SPack p, int ldw,
                                    int math) {
  int taps = p.kT * p.kH * p.kW;
  int btaps = p.KT * p.KH * p.KW;
  size_t total = (size_t)8 * CinP * ldw;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int k = i % ldw;
    int n = i / ldw;
    int co = k % Cout;
    int bt = k / Cout;
    if (bt >= btaps) {
      store_packed(out, math, n, k, ldw, (size_t)8 * CinP, 0.f);
      continue;
    }
    int ci = n % CinP, par = n / CinP;
    int pt = par >> 2, ph = (par >> 1) & 1, pw = par & 1;
    int dw = bt % p.KW, dh = (bt / p.KW) % p.KH, dt = bt / (p.KW * p.KH);
    // input index x = s*blk + par (par < s), output index o = blk + d - off;
    // forward: x = o*s - pad + k  =>  k = par + pad - s*(d - off)
    int kt = pt + p.pT - p.sT * (dt - p.oT);
    int kh = ph + p.pH - p.sH * (dh - p.oH);
    int kw = pw + p.pW - p.sW * (dw - p.oW);
    bool ok = ci < Cin && pt < p.sT && ph < p.sH && pw < p.sW && kt >= 0 && kt < p.kT && kh >= 0 &&
              kh < p.kH && kw >= 0 && kw < p.kW;
    float v = 0.f;
    if (ok) {
      int tap = (kt * p.kH + kh) * p.kW + kw;
      v = (scale ? scale[co] : 1.f) * w[((size_t)co * Cin + ci) * taps + tap];
    }
    store_packed(out, math, n, k, ldw, (size_t)8 * CinP, v);
  }
}

// BN(eval) fold: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
// (I3D_doubled.py:75 eps=1e-3; torch batch_norm eval formula).
__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean,
                               const float* var, float eps, float* scale, float* shift, int C) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    float s = gamma[c] / sqrtf(var[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - mean[c] * s;
  }
}

template <int AM, int BM, int BN, int WM, int WN>
static void launch_bf16(ConvKArgs& a, bool pw, dim3 grid, hipStream_t s) {
  if (pw)
    hipLaunchKernelGGL((conv3d_igemm_bf16_kernel<AM, BM, BN, WM, WN, true>), grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((conv3d_igemm_bf16_kernel<AM, BM, BN, WM, WN, false>), grid, dim3(256), 0, s, a);
}

template <int BM, int BN, int WM, int WN>
static int launch_variant(ConvKArgs& a, int math, hipStream_t s) {
  a.mtiles = cdiv(a.M, BM);
  a.ntiles = cdiv(a.Cout, BN);
  dim3 grid(a.mtiles * a.ntiles);
  // profiler class: 1..3 fp32 tiles, 4..6 split-bf16 tiles, 7..9 the wide split-bf16 tiles, 10..11 the small ones
  const int cls = BM == 64 ? (BN == 64 ? 10 : 11)
                  : BN > 128 ? (BN == 256 ? 7 : (BN == 192 ? 8 : 9))
                             : IVF_CONV_IGEMM_BASE + (BN == 128 ? 0 : (BN == 64 ? 1 : 2)) + (math ? 3 : 0);
  constexpr bool wide = BN > 128 || BM == 64;          // split-bf16 only
  constexpr bool fits_x6 = (3 * BM + 3 * BN) * LDS_ROW_BF * 2 <= 64 * 1024;
  if (wide && math == IVF_MATH_FP32) {
    set_error("conv3d: wide implicit-GEMM tiles exist for the split-bf16 modes only");
    return IVF_ERR_UNSUPPORTED;
  }
  if (!fits_x6 && math == IVF_MATH_BF16X6) {
    set_error("conv3d: this implicit-GEMM tile does not fit the LDS with three operand planes");
    return IVF_ERR_UNSUPPORTED;
  }
  if (math == IVF_MATH_FP32)
    prof_name(cls, "conv3d_igemm_kernel<%d,%d,%d,%d>", BM, BN, WM, WN);
  // a plain GEMM over the pixels: no taps, no strides, no padding
  const bool pw = a.kT * a.kH * a.kW == 1 && a.sT == 1 && a.sH == 1 && a.sW == 1 && a.pT == 0 && a.pH == 0 && a.pW == 0 &&
                  a.To == a.Ti && a.Ho == a.Hi && a.Wo == a.Wi && a.K >= 4 && a.K % 4 == 0 && !a.d2s;
  if (a.in2 && !pw && math != IVF_MATH_FP32) {
    set_error("conv3d: a second input needs the plain-GEMM form (1x1x1, stride 1, no padding, K a multiple of 4)");
    return IVF_ERR_UNSUPPORTED;
  }
  if (math != IVF_MATH_FP32)
    prof_name(cls, "conv3d_igemm_bf16_kernel<%d,%d,%d,%d,%d,%s>", math == IVF_MATH_BF16X6 ? AM_X6 : (math == IVF_MATH_BF16ACT ? AM_BF16 : AM_X3),
              BM, BN, WM, WN, pw ? "true" : "false");
  const bool timed = prof_begin(s, cls);
  if (math == IVF_MATH_FP32) {
    if constexpr (!wide) hipLaunchKernelGGL((conv3d_igemm_kernel<BM, BN, WM, WN>), grid, dim3(256), 0, s, a);
  } else if (math == IVF_MATH_BF16X3) {
    launch_bf16<AM_X3, BM, BN, WM, WN>(a, pw, grid, s);
  } else if (math == IVF_MATH_BF16ACT) {
    launch_bf16<AM_BF16, BM, BN, WM, WN>(a, pw, grid, s);
  } else {
    if constexpr (fits_x6) launch_bf16<AM_X6, BM, BN, WM, WN>(a, pw, grid, s);
  }
  if (timed) prof_end(s);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

int conv_launch(ConvKArgs& a, int math, hipStream_t s) {
  // pick BN to minimise padded columns; ties go to the wider tile
  int n128 = cdiv(a.Cout, 128) * 128, n64 = cdiv(a.Cout, 64) * 64, n32 = cdiv(a.Cout, 32) * 32;
  if (n128 <= n64 && n128 <= n32) return launch_variant<128, 128, 2, 2>(a, math, s);
  if (n64 <= n32) return launch_variant<128, 64, 4, 1>(a, math, s);
  return launch_variant<128, 32, 4, 1>(a, math, s);
}

int conv_igemm_launch_variant(ConvKArgs& a, int math, int v, hipStream_t s) {
  switch (v) {
    case 0: return launch_variant<128, 128, 2, 2>(a, math, s);
    case 1: return launch_variant<128, 64, 4, 1>(a, math, s);
    case 2: return launch_variant<128, 32, 4, 1>(a, math, s);
    // wide tiles for the 1x1x1 convs (HBM-bound: the input rows are read once instead of once
    // per 64-column tile)
    case 3: return launch_variant<128, 256, 4, 1>(a, math, s);
    case 4: return launch_variant<128, 192, 4, 1>(a, math, s);
    case 5: return launch_variant<128, 160, 4, 1>(a, math, s);
    // small tiles: twice the workgroups per CU for the latency-bound short-K GEMMs
    case 6: return launch_variant<64, 64, 2, 2>(a, math, s);
    case 7: return launch_variant<64, 128, 2, 2>(a, math, s);
  }
  set_error("conv3d: unknown implicit-GEMM variant %d", v);
  return IVF_ERR_BAD_ARG;
}

static inline int pack_ldw(int K, int math) { return math ? (K + 7) / 8 * 8 : K; }
// floats a packed [rows][ldw] matrix occupies: fp32, or 2 / 3 bf16 planes
static inline size_t pack_floats(size_t rows, int ldw, int math) {
  return math ? (rows * ldw * math_planes(math) + 1) / 2 : rows * ldw;
}

// Extent / front offset of the backward conv along one dim.  Input x = s*blk + par
// receives from outputs o with forward tap k = x + pad - s*o in [0,k): o = blk + d,
// d in [dlo, dhi], dlo = ceil((pad-(k-1))/s), dhi = floor((s-1+pad)/s).
static void bwd_span(int k, int st, int pad, int* K, int* off) {
  if (st == 1) { *K = k; *off = k - 1 - pad; return; }
  int lo_num = pad - (k - 1);
  int dlo = lo_num >= 0 ? (lo_num + st - 1) / st : -((-lo_num) / st);
  int dhi = (st - 1 + pad) / st;
  *K = dhi - dlo + 1;
  *off = -dlo;
}

static int check_desc(const ivf_conv3d_desc* d) {
  IVF_CHECK_ARG(d != nullptr, "conv3d: null descriptor");
  IVF_CHECK_ARG(d->B > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0, "conv3d: bad input dims");
  IVF_CHECK_ARG(d->Cin > 0 && d->Cin % 4 == 0, "conv3d: Cin (%d) must be a positive multiple of 4", d->Cin);
  const int cin_first = d->in2 ? d->K0 : d->Cin;
  IVF_CHECK_ARG(d->in_ld % 4 == 0 && d->in_coff % 4 == 0 && d->in_coff + cin_first <= d->in_ld,
                "conv3d: input channel window [%d,+%d) must be 4-aligned inside ld %d", d->in_coff,
                cin_first, d->in_ld);
  IVF_CHECK_ARG(d->Cout > 0 && d->out_coff >= 0, "conv3d: bad Cout");
  IVF_CHECK_ARG(d->kT > 0 && d->kH > 0 && d->kW > 0 && d->sT > 0 && d->sH > 0 && d->sW > 0,
                "conv3d: bad kernel/stride");
  IVF_CHECK_ARG(d->To > 0 && d->Ho > 0 && d->Wo > 0, "conv3d: bad output dims");
  long long in_elems = (long long)d->B * d->Ti * d->Hi * d->Wi * d->in_ld;
  IVF_CHECK_ARG(in_elems < (1ll << 40), "conv3d: input too large");
  IVF_CHECK_ARG((long long)d->B * d->To * d->Ho * d->Wo < (1ll << 31) &&
                    (long long)d->B * d->Ti * d->Hi * d->Wi < (1ll << 31),
                "conv3d: position count exceeds int32");
  return IVF_OK;
}

}  // namespace ivf

using namespace ivf;

extern "C" int ivf_conv3d(const ivf_conv3d_desc* d, const float* in, const float* w_packed,
                          const float* scale, const float* shift, const float* relu_mask,
                          float* out, ivf_stream_t stream) {
  IVF_PROPAGATE(check_desc(d));
  IVF_CHECK_ARG(in && w_packed && out, "conv3d: null pointer");
  ConvKArgs a;
  a.in = in; a.w = w_packed; a.out = out; a.scale = scale; a.shift = shift; a.mask = relu_mask;
  a.B = d->B; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.Cin = d->Cin; a.in_ld = d->in_ld;
  a.in_coff = d->in_coff;
  a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.Cout = d->Cout; a.out_ld = d->out_ld;
  a.out_coff = d->out_coff; a.mask_ld = d->mask_ld; a.mask_coff = d->mask_coff;
  a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.sT = d->sT; a.sH = d->sH; a.sW = d->sW;
  a.pT = d->pT; a.pH = d->pH; a.pW = d->pW;
  a.K = d->kT * d->kH * d->kW * d->Cin;
  a.M = d->B * d->To * d->Ho * d->Wo;
  a.in2 = d->in2; a.in2_ld = d->in2_ld; a.in2_coff = d->in2_coff; a.K0 = d->K0;
  if (d->in2) {
    IVF_CHECK_ARG(d->kT * d->kH * d->kW == 1, "conv3d: a second input is only defined for 1x1x1 convs");
    IVF_CHECK_ARG(d->K0 > 0 && d->K0 < d->Cin && d->K0 % 4 == 0 && d->in2_ld % 4 == 0 && d->in2_coff % 4 == 0 &&
                      d->in2_coff + (d->Cin - d->K0) <= d->in2_ld,
                  "conv3d: second-input channel window must be 4-aligned inside its row");
  }
  a.out2 = d->out2; a.out2_ld = d->out2_ld; a.out2_coff = d->out2_coff; a.N0 = d->N0;
  if (d->out2) {
    IVF_CHECK_ARG(!d->accumulate && relu_mask == nullptr && !d->d2s,
                  "conv3d: a second output window is a forward-epilogue feature (no accumulate / relu_mask / d2s)");
    IVF_CHECK_ARG(d->N0 > 0 && d->N0 < d->Cout && d->out2_coff >= 0 && d->out2_coff + (d->Cout - d->N0) <= d->out2_ld &&
                      d->out_coff + d->N0 <= d->out_ld,
                  "conv3d: output windows [0,N0) / [N0,Cout) must fit their rows");
    IVF_CHECK_ARG(d->variant == IVF_CONV_AUTO || (d->variant >= IVF_CONV_IGEMM_BASE && d->variant < IVF_CONV_PIX4),
                  "conv3d: a second output window is served by the implicit-GEMM tiles only");
  }
  a.gbo = d->gate_out; a.gbo2 = d->gate_out2; a.gbi = d->gate_in;
  a.gbo_ld = d->gate_out_ld; a.gbo_coff = d->gate_out_coff; a.gbo2_ld = d->gate_out2_ld;
  a.gbi_ld = d->gate_in_ld; a.gbi_coff = d->gate_in_coff;
  if (d->gate_out || d->gate_out2 || d->gate_in) {
    IVF_CHECK_ARG(!d->d2s && (d->Cout & 7) == 0 && (d->out_ld & 3) == 0 && (d->out_coff & 3) == 0,
                  "conv3d: 1-bit gates need Cout %% 8 == 0 and 4-aligned output rows (the 16-byte epilogue)");
    IVF_CHECK_ARG(d->variant != IVF_CONV_PIX4, "conv3d: the pix4 kernel does not record 1-bit gates");
    IVF_CHECK_ARG(!d->out2 || ((d->N0 | d->out2_ld | d->out2_coff) & 3) == 0, "conv3d: 1-bit gates with out2 need 4-aligned windows");
  }
  if (d->gate_out || d->gate_out2) {
    IVF_CHECK_ARG(!d->accumulate && relu_mask == nullptr && !d->gate_in,
                  "conv3d: gate_out / gate_out2 are forward-epilogue records (no accumulate / relu_mask / gate_in)");
    IVF_CHECK_ARG(!d->gate_out || ((d->gate_out_coff & 7) == 0 && d->gate_out_ld > 0), "conv3d: gate_out window must be 8-aligned");
    IVF_CHECK_ARG(!d->gate_out2 || (d->out2 && (d->N0 & 7) == 0 && (d->out2_coff & 7) == 0 && d->gate_out2_ld > 0),
                  "conv3d: gate_out2 needs out2 with 8-aligned N0 / out2_coff");
  }
  if (d->gate_in)
    IVF_CHECK_ARG(relu_mask == nullptr && (d->gate_in_coff & 3) == 0 && d->gate_in_ld > 0,
                  "conv3d: gate_in replaces relu_mask (give one of them) and needs a 4-aligned channel offset");
  IVF_CHECK_ARG(d->math >= IVF_MATH_FP32 && d->math <= IVF_MATH_BF16ACT, "conv3d: math must be one of IVF_MATH_* (0..3)");
  if (d->math == IVF_MATH_BF16ACT) {
    // bf16 storage on both sides, except: the 4-channel-pixel strided conv reads fp32 pixels (pix4 kernel only), a
    // depth-to-space backward writes fp32 (LDS-halo kernel only)
    IVF_CHECK_ARG(!(d->d2s && (d->accumulate || d->relu)), "conv3d: bf16act depth-to-space output is a plain fp32 store");
  }
  a.ldw = pack_ldw(a.K, d->math);
  a.wbf = reinterpret_cast<const unsigned short*>(w_packed);
  a.w_lo_off = (long)d->Cout * a.ldw;
  a.relu = d->relu; a.accumulate = d->accumulate; a.d2s = d->d2s;
  a.dT = d->dT; a.dH = d->dH; a.dW = d->dW; a.dC = d->dC;
  a.bsT = d->bsT; a.bsH = d->bsH; a.bsW = d->bsW;
  if (d->d2s) {
    IVF_CHECK_ARG(relu_mask == nullptr, "conv3d: relu_mask unsupported with depth-to-space");
    IVF_CHECK_ARG(d->bsT >= 1 && d->bsT <= 2 && d->bsH >= 1 && d->bsH <= 2 && d->bsW >= 1 && d->bsW <= 2,
                  "conv3d: block strides must be 1 or 2");
    IVF_CHECK_ARG(d->Cout % 8 == 0 && d->Cout <= 32, "conv3d: depth-to-space needs Cout = 8 * Cpad <= 32");
    IVF_CHECK_ARG(d->dT > 0 && d->dH > 0 && d->dW > 0 && d->dC > 0 && d->dC <= d->Cout / 8,
                  "conv3d: bad depth-to-space dims");
    IVF_CHECK_ARG(d->out_coff + d->dC <= d->out_ld, "conv3d: d2s output window outside ld");
  } else if (!d->out2) {
    IVF_CHECK_ARG(d->out_coff + d->Cout <= d->out_ld, "conv3d: output window outside ld");
  }
  if (d->math == 0) IVF_CHECK_ARG(a.ldw == a.K, "conv3d: internal ldw");
  static const bool no_halo = getenv("IVF_NO_HALO") != nullptr;   // A/B switch for measurements
  const bool bf = d->math != IVF_MATH_FP32;
  // bf16act: which kernels can serve the two mixed-storage ends of the network
  const bool act_stem = d->math == IVF_MATH_BF16ACT && conv_pix4_supported(a);   // fp32 pixels in: pix4 only
  const bool act_d2s = d->math == IVF_MATH_BF16ACT && d->d2s;                     // fp32 out: LDS-halo only
  if (d->variant != IVF_CONV_AUTO) {
    // explicit kernel variant (set by the plan's tuner)
    if (d->variant == IVF_CONV_PIX4) {
      IVF_CHECK_ARG(bf && conv_pix4_supported(a), "conv3d: pix4 variant needs a split-bf16 mode, 4-channel pixels, stride (1|2,2,2), k <= 7");
      return conv_pix4_launch(a, d->math, IVF_CONV_PIX4, (hipStream_t)stream);
    }
    IVF_CHECK_ARG(!act_stem, "conv3d: bf16act reads fp32 pixels through the pix4 kernel only");
    if (d->variant >= IVF_CONV_HALO_BASE) {
      IVF_CHECK_ARG(bf && conv_halo_supported(a), "conv3d: halo variant needs a split-bf16 mode, stride 1, k in 2..4");
      return conv_halo_launch_variant(a, d->math, d->variant - IVF_CONV_HALO_BASE, (hipStream_t)stream);
    }
    IVF_CHECK_ARG(!act_d2s, "conv3d: bf16act depth-to-space (fp32 output) is served by the LDS-halo kernel only");
    return conv_igemm_launch_variant(a, d->math, d->variant - IVF_CONV_IGEMM_BASE, (hipStream_t)stream);
  }
  if (act_stem) {
    IVF_CHECK_ARG(!d->out2 && !a.gbo, "conv3d: the pix4 kernel has no second output window / gate record");
    return conv_pix4_launch(a, d->math, IVF_CONV_PIX4, (hipStream_t)stream);
  }
  if (bf && (!no_halo || act_d2s) && !d->out2 && conv_halo_supported(a)) return conv_halo_launch(a, d->math, (hipStream_t)stream);
  IVF_CHECK_ARG(!act_d2s, "conv3d: bf16act depth-to-space needs the LDS-halo kernel (stride-1 form, k <= 4, Cin %% 8 == 0)");
  if (bf && !no_halo && !d->out2 && !a.gbo && conv_pix4_supported(a)) return conv_pix4_launch(a, d->math, IVF_CONV_PIX4, (hipStream_t)stream);
  return conv_launch(a, d->math, (hipStream_t)stream);
}

extern "C" int ivf_bn_fold(const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, float* scale, float* shift, int C,
                           ivf_stream_t stream) {
  IVF_CHECK_ARG(gamma && beta && mean && var && scale && shift && C > 0, "bn_fold: bad args");
  hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma,
                     beta, mean, var, eps, scale, shift, C);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" size_t ivf_conv3d_pack_fwd_elems(int Cout, int CinPad, int kT, int kH, int kW, int math) {
  return pack_floats((size_t)Cout, pack_ldw(kT * kH * kW * CinPad, math), math);
}

extern "C" int ivf_conv3d_pack_fwd_rows(const float* w_ref, float* w_packed, int Cout, int Cin, int CinPad,
                                        int kT, int kH, int kW, int row0, int rows_total, int math,
                                        ivf_stream_t stream) {
  IVF_CHECK_ARG(w_ref && w_packed && Cout > 0 && Cin > 0 && CinPad >= Cin && CinPad % 4 == 0 &&
                    (math >= 0 && math <= IVF_MATH_BF16ACT) && row0 >= 0 && row0 + Cout <= rows_total,
                "pack_fwd: bad args");
  int taps = kT * kH * kW;
  int ldw = pack_ldw(taps * CinPad, math);
  size_t total = (size_t)Cout * ldw;
  hipLaunchKernelGGL(pack_fwd_kernel, dim3(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_ref, w_packed, Cout, Cin, CinPad, taps, ldw, math, row0, rows_total);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_conv3d_pack_fwd(const float* w_ref, float* w_packed, int Cout, int Cin,
                                   int CinPad, int kT, int kH, int kW, int math, ivf_stream_t stream) {
  return ivf_conv3d_pack_fwd_rows(w_ref, w_packed, Cout, Cin, CinPad, kT, kH, kW, 0, Cout, math, stream);
}

extern "C" int ivf_conv3d_pack_bwd(const float* w_ref, const float* scale, float* w_packed,
                                   int Cout, int Cin, int CinPad, int kT, int kH, int kW, int sT,
                                   int sH, int sW, int pT, int pH, int pW, int math,
                                   ivf_conv3d_bwd_geom* geom, ivf_stream_t stream) {
  IVF_CHECK_ARG(w_ref && w_packed && geom && Cout > 0 && Cout % 4 == 0 && Cin > 0 && CinPad >= Cin &&
                    (math >= 0 && math <= IVF_MATH_BF16ACT),
                "pack_bwd: bad args (Cout must be a multiple of 4)");
  IVF_CHECK_ARG(sT >= 1 && sT <= 2 && sH >= 1 && sH <= 2 && sW >= 1 && sW <= 2,
                "pack_bwd: strides must be 1 or 2");
  hipStream_t s = (hipStream_t)stream;
  if (sT == 1 && sH == 1 && sW == 1) {
    geom->d2s = 0;
    geom->kT = kT; geom->kH = kH; geom->kW = kW;
    geom->pT = kT - 1 - pT; geom->pH = kH - 1 - pH; geom->pW = kW - 1 - pW;
    geom->rows = CinPad;
    int ldw = pack_ldw(kT * kH * kW * Cout, math);
    size_t total = (size_t)CinPad * ldw;
    hipLaunchKernelGGL(pack_bwd_s1_kernel, dim3(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256)),
                       dim3(256), 0, s, w_ref, scale, w_packed, Cout, Cin, CinPad, kT, kH, kW, ldw, math);
    IVF_CHECK_LAUNCH();
    return IVF_OK;
  }
  auto span = bwd_span;
  D2SPack p;
  p.kT = kT; p.kH = kH; p.kW = kW; p.sT = sT; p.sH = sH; p.sW = sW; p.pT = pT; p.pH = pH; p.pW = pW;
  span(kT, sT, pT, &p.KT, &p.oT);
  span(kH, sH, pH, &p.KH, &p.oH);
  span(kW, sW, pW, &p.KW, &p.oW);
  geom->d2s = 1;
  geom->kT = p.KT; geom->kH = p.KH; geom->kW = p.KW;
  geom->pT = p.oT; geom->pH = p.oH; geom->pW = p.oW;
  geom->rows = 8 * CinPad;
  int ldw = pack_ldw(p.KT * p.KH * p.KW * Cout, math);
  size_t total = (size_t)8 * CinPad * ldw;
  hipLaunchKernelGGL(pack_bwd_d2s_kernel, dim3(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256)),
                     dim3(256), 0, s, w_ref, scale, w_packed, Cout, Cin, CinPad, p, ldw, math);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" size_t ivf_conv3d_pack_bwd_elems(int Cout, int CinPad, int kT, int kH, int kW, int sT, int sH,
                                            int sW, int pT, int pH, int pW, int math) {
  if (sT == 1 && sH == 1 && sW == 1) return pack_floats((size_t)CinPad, pack_ldw(kT * kH * kW * Cout, math), math);
  int KT, KH, KW, o;
  bwd_span(kT, sT, pT, &KT, &o);
  bwd_span(kH, sH, pH, &KH, &o);
  bwd_span(kW, sW, pW, &KW, &o);
  return pack_floats((size_t)8 * CinPad, pack_ldw(KT * KH * KW * Cout, math), math);
}

// candidate kernel variants for a descriptor: fills ids[], returns the count
extern "C" int ivf_conv3d_variants(const ivf_conv3d_desc* d, int* ids, int max_ids) {
  if (!d || !ids) return 0;
  int n = 0;
  for (int v = 0; v < 3 && n < max_ids; ++v) ids[n++] = IVF_CONV_IGEMM_BASE + v;
  const bool bf = d->math != IVF_MATH_FP32;
  if (bf && d->kT * d->kH * d->kW == 1)
    for (int v = (d->math == IVF_MATH_BF16X6 ? 6 : 3); v < 8 && n < max_ids; ++v) ids[n++] = IVF_CONV_IGEMM_BASE + v;
  ConvKArgs a{};
  a.sT = d->sT; a.sH = d->sH; a.sW = d->sW; a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.Cin = d->Cin;
  a.in_ld = d->in_ld; a.in_coff = d->in_coff; a.d2s = d->d2s; a.in2 = d->in2;
  if (d->out2) return n;   // second output window: implicit-GEMM tiles only
  // bf16act: the two mixed-storage ends of the network have one kernel family each
  if (d->math == IVF_MATH_BF16ACT && conv_pix4_supported(a)) { ids[0] = IVF_CONV_PIX4; return 1; }
  if (d->math == IVF_MATH_BF16ACT && d->d2s) n = 0;
  if (bf && conv_halo_supported(a))
    for (int v = 0; v < conv_halo_num_variants() && n < max_ids; ++v) ids[n++] = IVF_CONV_HALO_BASE + v;
  if (bf && conv_pix4_supported(a) && !d->gate_out && n < max_ids) ids[n++] = IVF_CONV_PIX4;
  return n;
}

extern "C" size_t ivf_conv3d_pack_bwd_fused1x1_elems(int Ktotal, int CinPad, int math) {
  return pack_floats((size_t)CinPad, pack_ldw(Ktotal, math), math);
}

extern "C" int ivf_conv3d_pack_bwd_fused1x1(const float* w_ref, const float* scale, float* w_packed, int Cout,
                                            int Cin, int CinPad, int koff, int Ktotal, int math,
                                            ivf_stream_t stream) {
  IVF_CHECK_ARG(w_ref && w_packed && Cout > 0 && Cin > 0 && CinPad >= Cin && koff >= 0 && koff % 4 == 0 &&
                    koff + Cout <= Ktotal && (math >= 0 && math <= IVF_MATH_BF16ACT),
                "pack_bwd_fused1x1: bad args");
  int ldw = pack_ldw(Ktotal, math);
  size_t total = (size_t)CinPad * (Cout + ldw - Ktotal);
  hipLaunchKernelGGL(pack_bwd_fused1x1_kernel, dim3(cdiv(total, 256) > 4096 ? 4096 : cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, w_ref, scale, w_packed, Cout, Cin, CinPad, koff, Ktotal, ldw, math);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}
