// Strided k x k x k convolution over a 4-channel (RGB + pad) channels-last clip, split-bf16
// (3-pass) MFMA: the I3D stem (Conv3d_1a_7x7, models/I3D_doubled.py:266-270; 7x7x7, stride 2).
//
// With 4 input channels a tap is a single 16-byte pixel, so the plain implicit GEMM
// (conv3d.hip) spends its time on address arithmetic and 16-byte gathers: 343 of them per
// output row.  Here a workgroup owns a 4 x 8 x 8 box of outputs and stages the input
// neighbourhood it touches ONCE, pixel-major, as split hi/lo bf16 (8 + 8 bytes per pixel).
// In that image the 4 taps kw..kw+3 of one (kt, kh) are 4 adjacent pixels = one 16-element
// k-step, and a lane's A fragment (2 pixels) is one aligned ds_read_b128 whose address is
//     base(lane's output pixel, stride 2) + offset(kt, kh) + 32*kstep + 16*(lane >> 5).
// The packed weights are the ordinary forward pack ([Cout][tap*4 + c] bf16 hi/lo planes): a
// (kt, kh) step is the contiguous run of kW*4 values, zero-extended to 32 (the pad taps
// multiply staged, finite pixels).  Weight tiles (64 x 32 per step) stream through a register
// ring and a triple-buffered LDS tile, one barrier per (kt, kh) step; the fragments of the next
// step are read while the MFMAs of the current one run.
//
// The grid is persistent: one workgroup per CU (the pixel image fills most of the LDS) walks a
// contiguous run of boxes.  The next box's pixels and first weight tiles are loaded into
// registers BEFORE the current box's epilogue, so the output stores drain and the loads fly
// while neither blocks the other; without that every CU reaches its epilogue at the same
// moment and the whole chip waits on one burst of stores (measured: 21 % of the kernel).
//
// LDS row pitch is 24 pixels (192 B): two output rows (2 input rows apart) are then 128 B apart
// modulo the 256-B bank row, which makes every ds_read_b128 lane group (MI355X_MICROARCH.md,
// LDS) cover 16 distinct 16-byte slots: conflict-free A reads.
#include <type_traits>

#include "conv_common.h"

namespace ivf {

namespace {
constexpr int P4_TH = 8, P4_TW = 8;
constexpr int P4_BN = 64;
constexpr int P4_PW = 24;                  // LDS pixels per halo row
constexpr int P4_COLS = 2 * (P4_TW - 1) + 8;   // staged pixels per halo row (22)
constexpr int P4_ROWB = (32 + 8) * 2;      // weight tile row bytes (80: conflict-free 16-byte reads)
// MFMA passes of an operand mode as (activation plane, weight plane) pairs, smallest terms first; the kernel runs a
// pass over both column tiles before the next one (no two consecutive MFMAs on one accumulator)
template <int AM> struct P4Passes;
template <> struct P4Passes<AM_X3> { static constexpr int N = 3; static constexpr int pa[3] = {1, 0, 0}, pb[3] = {0, 1, 0}; };
template <> struct P4Passes<AM_X6> { static constexpr int N = 6; static constexpr int pa[6] = {2, 0, 1, 1, 0, 0}, pb[6] = {0, 2, 1, 0, 1, 0}; };
}  // namespace

// AM = operand mode (AM_X3: pixels and weights split hi/lo, AM_X6: three ways, six passes); P4_TT = frames of outputs per
// box (4; 2 where the third pixel image has to fit the LDS); OB = the epilogue stores bf16 (the bf16-activation mode,
// whose stem still reads fp32 pixels: its operands are AM_X3).
template <int AM, int P4_TT, bool OB>
__global__ __launch_bounds__(P4_TT * 2 * 64) void conv3d_pix4_kernel(ConvKArgs a, int tilesT, int tilesH, int tilesW) {
  constexpr int NPA = OpPlanes<AM>::A, NPB = OpPlanes<AM>::B;
  constexpr int P4_NT = P4_TT * 2 * 64;      // one wave per 32 output rows
  constexpr int P4_NSTG = (((P4_TT - 1) * 2 + 7) * (2 * (P4_TH - 1) + 7) * P4_COLS + P4_NT - 1) / P4_NT;   // pixels per thread of the largest box image
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int HT = (P4_TT - 1) * a.sT + a.kT, HH = 2 * (P4_TH - 1) + a.kH;
  const int plane = HT * HH * P4_PW * 8;   // bytes of one image plane
  unsigned char* a_hi = smem;              // NPA planes, `plane` bytes apart
  unsigned char* b_base = smem + NPA * (size_t)plane;   // [3 buffers][NPB planes][64 rows][P4_ROWB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wm = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  const int total = a.mtiles * a.ntiles;
  const int per = (total + gridDim.x - 1) / gridDim.x;
  int tile = blockIdx.x * per;
  const int tile_end = min(total, tile + per);
  if (tile >= tile_end) return;
  // box decode: n-tile fastest, then w, h, t, clip
  auto decode = [&](int id, int* b, int* t0, int* h0, int* w0, int* n0) {
    *n0 = (id % a.ntiles) * P4_BN;
    id /= a.ntiles;
    *w0 = (id % tilesW) * P4_TW;
    id /= tilesW;
    *h0 = (id % tilesH) * P4_TH;
    id /= tilesH;
    *t0 = (id % tilesT) * P4_TT;
    *b = id / tilesT;
  };
  int b, t0, h0, w0, n0;
  decode(tile, &b, &t0, &h0, &w0, &n0);
  int ld_n0 = n0;   // output-channel tile the weight ring is loading for

  // this lane's output pixel inside the box: row = wm*32 + li -> (t, h, w) = (row>>6, (row>>3)&7, row&7)
  const int abase = ((((wm >> 1) * a.sT) * HH + 2 * ((wm & 1) * 4 + (li >> 3))) * P4_PW + 2 * (li & 7)) * 8 + 16 * lh;

  const int kw4 = a.kW * 4;                // real k values per (kt, kh) step
  const int nks = (kw4 + 15) >> 4;
  const int nsteps = a.kT * a.kH;

  f32x16 acc[2];

  // weight ring: thread -> (plane, row, 8-value piece) of the 64 x 32 step tile
  constexpr int PF = 3;
  constexpr int NSL = NPB * 256 / P4_NT;   // (plane, row, piece) slots of the step tile per thread
  static_assert(NSL * P4_NT == NPB * 256, "weight-stream slots must divide over the threads");
  uint2 rb[PF][NSL][2];
  auto load_b = [&](int slot, int step) {
#pragma unroll
    for (int q = 0; q < NSL; ++q) {
      const int i = tid + q * P4_NT;
      const int bpl = i >> 8, brow = perm8((i & 255) >> 2), bg = i & 3;   // perm8: conflict-free 16-byte LDS writes
      const int n = ld_n0 + brow, k0 = 8 * bg;
      uint2 v0 = make_uint2(0u, 0u), v1 = v0;
      if (n < a.Cout) {
        const unsigned short* p = a.wbf + bpl * a.w_lo_off + (size_t)n * a.ldw + (size_t)step * kw4 + k0;
        if (k0 < kw4) v0 = *reinterpret_cast<const uint2*>(p);
        if (k0 + 4 < kw4) v1 = *reinterpret_cast<const uint2*>(p + 4);
      }
      rb[slot][q][0] = v0;
      rb[slot][q][1] = v1;
    }
  };
  auto store_b = [&](int slot, int buf) {
#pragma unroll
    for (int q = 0; q < NSL; ++q) {
      const int i = tid + q * P4_NT;
      const int bpl = i >> 8, brow = perm8((i & 255) >> 2), bg = i & 3;
      unsigned char* dst = b_base + (size_t)(buf * NPB + bpl) * P4_BN * P4_ROWB + brow * P4_ROWB + 16 * bg;
      *reinterpret_cast<uint4*>(dst) = make_uint4(rb[slot][q][0].x, rb[slot][q][0].y, rb[slot][q][1].x, rb[slot][q][1].y);
    }
  };
  // Fragment sets: while the MFMAs of step s run on set s&1, the fragments of step s+1 are
  // already being read into the other set.  The weight tile of step s+1 became visible at the
  // barrier that ended step s-1 (three LDS buffers: step s reads s%3, prefetches (s+1)%3 and
  // writes (s+2)%3), so no LDS latency sits between a barrier and the first MFMA after it.
  bf16x8 fa[2][2][NPA];      // [set][ks][plane]
  bf16x8 fb[2][2][2][NPB];   // [set][ks][j][plane]
  auto read_frags = [&](auto SET, int step, int buf) {
    constexpr int set = decltype(SET)::value;
    const int kt = step / a.kH, kh = step - kt * a.kH;
    const int soff = (kt * HH + kh) * P4_PW * 8;
    const unsigned char* bh = b_base + (size_t)(buf * NPB) * P4_BN * P4_ROWB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl)
          fa[set][ks][pl] = *reinterpret_cast<const bf16x8*>(a_hi + (size_t)pl * plane + abase + soff + 32 * ks);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int off = (j * 32 + li) * P4_ROWB + 32 * ks + 16 * lh;
#pragma unroll
          for (int pl = 0; pl < NPB; ++pl)
            fb[set][ks][j][pl] = *reinterpret_cast<const bf16x8*>(bh + (size_t)pl * P4_BN * P4_ROWB + off);
        }
      }
    }
  };
  auto mma_step = [&](auto SET) {
    constexpr int set = decltype(SET)::value;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < nks) {
#pragma unroll
        for (int ps = 0; ps < P4Passes<AM>::N; ++ps)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][ks][P4Passes<AM>::pa[ps]], fb[set][ks][j][P4Passes<AM>::pb[ps]],
                                                             acc[j], 0, 0, 0);
      }
    }
  };

  // the input box: one pixel (float4) per item, zero outside the clip (TF-'same' padding)
  const int npx = HT * HH * P4_COLS;
  float4 stg[P4_NSTG];
  auto halo_load = [&](int bb, int bt0, int bh0, int bw0) {
    const int it0 = bt0 * a.sT - a.pT, ih0 = 2 * bh0 - a.pH, iw0 = 2 * bw0 - a.pW;
    // (opaque copy of tid: keeps the per-pixel index arithmetic inside the call instead of
    // hoisted out of the box loop into ~60 long-lived registers)
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int u = 0; u < P4_NSTG; ++u) {
      const int idx = u * P4_NT + tv;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < npx) {
        const int col = idx % P4_COLS, row = idx / P4_COLS;
        const int hh = row % HH, ht = row / HH;
        const int ti = it0 + ht, hi = ih0 + hh, wi = iw0 + col;
        if ((unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
          v = *reinterpret_cast<const float4*>(a.in + ((size_t)((bb * a.Ti + ti) * a.Hi + hi) * a.Wi + wi) * 4);
      }
      stg[u] = v;
    }
  };
  auto halo_store = [&]() {
    int tv = tid;
    asm volatile("" : "+v"(tv));
#pragma unroll
    for (int u = 0; u < P4_NSTG; ++u) {
      const int idx = u * P4_NT + tv;
      if (idx < npx) {
        const int col = idx % P4_COLS, row = idx / P4_COLS;
        stage_planes<AM>(a_hi + (row * P4_PW + col) * 8, (size_t)plane, stg[u]);
      }
    }
  };
  // step s (u = s mod 6): ring slot (s+2)%3 (step s+2, loaded three steps ago) moves to LDS
  // buffer (s+2)%3 and is refilled with step s+5
  auto step_body = [&](auto U, int step0) {
    constexpr int u = decltype(U)::value;
    const int step = step0 + u;
    if (step < nsteps) {
      if (step + 1 < nsteps) read_frags(std::integral_constant<int, (u + 1) & 1>{}, step + 1, (u + 1) % 3);
      mma_step(std::integral_constant<int, u & 1>{});
      if (step + 2 < nsteps) store_b((u + 2) % PF, (u + 2) % 3);
      if (step + PF + 2 < nsteps) load_b((u + 2) % PF, step + PF + 2);
      __syncthreads();
    }
  };
  static_assert(PF == 3, "step loop is unrolled by hand for a 3-deep ring");

#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < nsteps) load_b(u, u);
  halo_load(b, t0, h0, w0);

  for (; tile < tile_end; ++tile) {
    // the LDS images are free here: the barrier that ended the previous box's last step is
    // behind every wave, and its fragments were read a step before that
    halo_store();
    store_b(0, 0);
    if (nsteps > 1) store_b(1, 1);
    if (3 < nsteps) load_b(0, 3);
    if (4 < nsteps) load_b(1, 4);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    __syncthreads();
    read_frags(std::integral_constant<int, 0>{}, 0, 0);
    for (int step0 = 0; step0 < nsteps; step0 += 6) {
      step_body(std::integral_constant<int, 0>{}, step0);
      step_body(std::integral_constant<int, 1>{}, step0);
      step_body(std::integral_constant<int, 2>{}, step0);
      step_body(std::integral_constant<int, 3>{}, step0);
      step_body(std::integral_constant<int, 4>{}, step0);
      step_body(std::integral_constant<int, 5>{}, step0);
    }

    // next box: its loads are issued before this box's stores
    const int cb = b, ct0 = t0, ch0 = h0, cw0 = w0, cn0 = n0;
    if (tile + 1 < tile_end) {
      decode(tile + 1, &b, &t0, &h0, &w0, &n0);
      ld_n0 = n0;
#pragma unroll
      for (int u = 0; u < PF; ++u)
        if (u < nsteps) load_b(u, u);
      halo_load(b, t0, h0, w0);
    }
    // epilogue (BN scale/shift + ReLU; no accumulate / gate forms: conv_pix4_supported).  Lane
    // holds column li of each 32-column tile and rows (r&3) + 8*(r>>2) + 4*lh of the wave's
    // 32 = 4 h x 8 w box rows: h = hb + (r>>2), w = cw0 + (r&3) + 4*lh, so all 16 addresses are
    // one lane base plus wave-uniform strides.
    {
      const int t = ct0 + (wm >> 1), hb = ch0 + (wm & 1) * 4;
      const int wl = cw0 + 4 * lh;
      const size_t rowstride = (size_t)a.Wo * a.out_ld;
      const size_t obase = ((size_t)((cb * a.To + t) * a.Ho + hb) * a.Wo + wl) * a.out_ld + a.out_coff + cn0 + li;   // (elements)
      const bool vec_ok = ((a.Cout | a.out_ld | a.out_coff) & 3) == 0;
      if (t < a.To && vec_ok) {
        // 16-byte stores: after a 4x4 transpose inside the lane quad (conv_common.h) lane q holds pixel
        // w = wl + q of each of the 4 h rows and the quad's 4 adjacent channels
        const int q = li & 3;
        const size_t qbase = ((size_t)((cb * a.To + t) * a.Ho + hb) * a.Wo + wl + q) * a.out_ld + a.out_coff;   // (elements)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = cn0 + j * 32 + li;
          const int nq = n & ~3;
          const bool nvalid = nq < a.Cout;
          const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
          const float sh = (a.shift && nvalid) ? a.shift[n] : 0.f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float x = acc[j][4 * g + k] * sc + sh;
              v[k] = (a.relu && !(x > 0.f)) ? 0.f : x;
            }
            quad_transpose4(v, q);
            if (!nvalid || hb + g >= a.Ho || wl + q >= a.Wo) continue;
            ep_st4<OB>(a.out, qbase + (size_t)g * rowstride + nq, v[0], v[1], v[2], v[3]);
          }
        }
      } else if (t < a.To) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = cn0 + j * 32 + li;
          const bool nvalid = n < a.Cout;
          const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
          const float sh = (a.shift && nvalid) ? a.shift[n] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (!nvalid || hb + (r >> 2) >= a.Ho || wl + (r & 3) >= a.Wo) continue;
            float v = acc[j][r] * sc + sh;
            if (a.relu) v = v > 0.f ? v : 0.f;
            ep_st1<OB>(a.out, obase + (r >> 2) * rowstride + (size_t)(r & 3) * a.out_ld + j * 32, v);
          }
        }
      }
    }
  }
}

int conv_pix4_supported(const ConvKArgs& a) {
  if (a.Cin != 4 || a.in_ld != 4 || a.in_coff != 0 || a.in2 || a.d2s || a.accumulate || a.mask) return 0;
  if (a.sH != 2 || a.sW != 2 || (a.sT != 1 && a.sT != 2)) return 0;
  if (a.kT < 1 || a.kT > 7 || a.kH < 1 || a.kH > 7 || a.kW < 1 || a.kW > 7) return 0;
  // the last staged column (2*7 + 7) must cover the pad taps of the second k-step
  return 1;
}

template <int AM, int P4_TT, bool OB>
static int launch_pix4(ConvKArgs& a, int variant_id, hipStream_t s) {
  constexpr int NPA = OpPlanes<AM>::A, NPB = OpPlanes<AM>::B;
  const int HT = (P4_TT - 1) * a.sT + a.kT, HH = 2 * (P4_TH - 1) + a.kH;
  const size_t shm = (size_t)NPA * HT * HH * P4_PW * 8 + (size_t)3 * NPB * P4_BN * P4_ROWB;
  if (shm > 160 * 1024) {
    set_error("conv3d_pix4: %zu bytes of LDS needed", shm);
    return IVF_ERR_UNSUPPORTED;
  }
  static LdsAttrOnce once;
  IVF_PROPAGATE(raise_lds_limit(reinterpret_cast<const void*>(&conv3d_pix4_kernel<AM, P4_TT, OB>), 160 * 1024, once));
  const int tilesT = cdiv(a.To, P4_TT), tilesH = cdiv(a.Ho, P4_TH), tilesW = cdiv(a.Wo, P4_TW);
  a.ntiles = cdiv(a.Cout, P4_BN);
  a.mtiles = a.B * tilesT * tilesH * tilesW;
  prof_name(variant_id, "conv3d_pix4_kernel<%d,%d,%s>", AM, P4_TT, OB ? "true" : "false");
  const bool timed = prof_begin(s, variant_id);
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    IVF_CHECK_HIP(hipGetDevice(&dev));
    IVF_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int total = a.mtiles * a.ntiles;
  hipLaunchKernelGGL((conv3d_pix4_kernel<AM, P4_TT, OB>), dim3(total < cus ? total : cus), dim3(P4_TT * 2 * 64), shm, s, a, tilesT,
                     tilesH, tilesW);
  if (timed) prof_end(s);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

int conv_pix4_launch(ConvKArgs& a, int math, int variant_id, hipStream_t s) {
  switch (math) {
    case IVF_MATH_BF16X3: return launch_pix4<AM_X3, 4, false>(a, variant_id, s);
    case IVF_MATH_BF16ACT: return launch_pix4<AM_X3, 4, true>(a, variant_id, s);    // fp32 pixels in, bf16 out
    case IVF_MATH_BF16X6: return launch_pix4<AM_X6, 2, false>(a, variant_id, s);   // 2-frame boxes: three pixel images in LDS
  }
  set_error("conv3d_pix4: arithmetic mode %d has no pix4 kernel", math);
  return IVF_ERR_UNSUPPORTED;
}

}  // namespace ivf
