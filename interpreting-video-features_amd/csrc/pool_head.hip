// Max-pool with TF-'same' zero padding (forward with arg-max record, backward as a
// deterministic gather), the I3D classification head (global average pool + 1x1x1
// logits + softmax) forward and backward, and the Grad-CAM reductions.
// All HBM-bound, channels-last, 16-byte vector accesses over channels.
//
// Reference: MaxPool3dSamePadding (models/I3D_doubled.py:8-40), head
// (I3D_doubled.py:360-380), GradCamVideo.__call__ (grad_cam_videos.py:85-140).
#include <cstdlib>

#include "ivf_common.h"

namespace ivf {

// Activation storage element: float, or bf16 (IVF_MATH_BF16ACT plans: 2 bytes per element, round-to-nearest-even on
// store; a max-pool only selects, so its forward is exact in either storage).  Offsets are in elements.
struct bf16s { unsigned short v; };
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16s* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned pk2_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16s* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pk2_bf16(v.x, v.y), pk2_bf16(v.z, v.w));
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16s* p) { return __uint_as_float((unsigned)p->v << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16s* p, float v) { p->v = (unsigned short)(pk2_bf16(v, 0.f) & 0xffffu); }

struct PoolArgs {
  int B, Ti, Hi, Wi, C, in_ld, in_coff;
  int To, Ho, Wo, out_ld, out_coff;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int dead;   // forward: arg-max 255 where the window maximum is not > 0
};

// forward: scan the window in (kt,kh,kw) order over the ZERO-padded input
// (I3D_doubled.py:36-39 pads with zeros, not -inf), first strict maximum wins
// (torch max_pool3d semantics).  idx = flat tap of the winner (may be a pad cell).
template <class T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   unsigned char* __restrict__ idx, PoolArgs a, int nbx) {
  // workgroup = a block of one output row's (w, 4-channel group) cells; rows, then clip frames: see
  // maxpool_bwd_fixed_kernel; numbered so that every XCD owns a contiguous run of rows (the input rows two output
  // rows share then meet in one L2: 0-11 % faster than the round-robin deal, most on the 3x3x3 / (2,2,2) pool)
  const unsigned C4 = (unsigned)a.C >> 2;
  unsigned blk = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const unsigned bx = blk % (unsigned)nbx; blk /= (unsigned)nbx;
  const unsigned r = bx * blockDim.x + threadIdx.x;
  if (r >= (unsigned)a.Wo * C4) return;
  {
    const int wo = (int)(r / C4);
    const int c4 = (int)(r - (unsigned)wo * C4);
    const int ho = (int)(blk % (unsigned)a.Ho); blk /= (unsigned)a.Ho;
    const int to = (int)(blk % (unsigned)a.To);
    const int b = (int)(blk / (unsigned)a.To);
    const size_t m = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
    float best[4];
    int bi[4];
    bool first = true;
    int tap = 0;
    for (int kt = 0; kt < a.kT; ++kt) {
      int ti = to * a.sT - a.pT + kt;
      for (int kh = 0; kh < a.kH; ++kh) {
        int hi = ho * a.sH - a.pH + kh;
        for (int kw = 0; kw < a.kW; ++kw, ++tap) {
          int wi = wo * a.sW - a.pW + kw;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if ((unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi &&
              (unsigned)wi < (unsigned)a.Wi) {
            v = ld4(x + ((size_t)((b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi) * a.in_ld + a.in_coff + 4 * c4);
          }
          float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (first || vv[q] > best[q] || vv[q] != vv[q]) {
              best[q] = vv[q];
              bi[q] = tap;
            }
          }
          first = false;
        }
      }
    }
    st4(y + m * a.out_ld + a.out_coff + 4 * c4, make_float4(best[0], best[1], best[2], best[3]));
    if (idx) {
      if (a.dead)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (!(best[q] > 0.f)) bi[q] = 255;
      uchar4 u = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
      *reinterpret_cast<uchar4*>(idx + m * a.C + 4 * c4) = u;
    }
  }
}

// The same forward with the window known at compile time and 16 bytes per lane (4 fp32 or 8 bf16 channels): every
// tap's raw bits are requested up front and scanned afterwards -- in the generic kernel each load is consumed inside
// its bounds branch, i.e. waited for at once, which the fp32 form hides behind occupancy (4.6 TB/s) and the bf16
// form, with half the bytes per instruction, does not (2.7 TB/s of its own bytes).
template <class T, int KT, int KH, int KW>
__global__ __launch_bounds__(256) void maxpool_fwd_fixed_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                unsigned char* __restrict__ idx, PoolArgs a, int nbx) {
  constexpr int NCH = 16 / (int)sizeof(T);
  const unsigned CG = (unsigned)a.C / NCH;
  unsigned blk = (unsigned)xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const unsigned bx = blk % (unsigned)nbx; blk /= (unsigned)nbx;
  const unsigned r = bx * blockDim.x + threadIdx.x;
  if (r >= (unsigned)a.Wo * CG) return;
  const int wo = (int)(r / CG);
  const int c = (int)(r - (unsigned)wo * CG) * NCH;
  const int ho = (int)(blk % (unsigned)a.Ho); blk /= (unsigned)a.Ho;
  const int to = (int)(blk % (unsigned)a.To);
  const int b = (int)(blk / (unsigned)a.To);
  const size_t m = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
  uint4 raw[KT * KH * KW];
#pragma unroll
  for (int kt = 0; kt < KT; ++kt) {
    const int ti = to * a.sT - a.pT + kt;
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) {
      const int hi = ho * a.sH - a.pH + kh;
#pragma unroll
      for (int kw = 0; kw < KW; ++kw) {
        const int wi = wo * a.sW - a.pW + kw;
        uint4 u = make_uint4(0u, 0u, 0u, 0u);   // the zero padding takes part in the scan
        if ((unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
          u = *reinterpret_cast<const uint4*>(x + ((size_t)((b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi) * a.in_ld + a.in_coff + c);
        raw[(kt * KH + kh) * KW + kw] = u;
      }
    }
  }
  float best[NCH];
  unsigned bi[NCH];
#pragma unroll
  for (int tap = 0; tap < KT * KH * KW; ++tap) {
    const unsigned w4[4] = {raw[tap].x, raw[tap].y, raw[tap].z, raw[tap].w};
    float vv[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      if constexpr (sizeof(T) == 2) vv[q] = __uint_as_float((q & 1) ? (w4[q >> 1] & 0xffff0000u) : (w4[q >> 1] << 16));
      else vv[q] = __uint_as_float(w4[q]);
    }
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
      if (tap == 0 || vv[q] > best[q] || vv[q] != vv[q]) {
        best[q] = vv[q];
        bi[q] = (unsigned)tap;
      }
    }
  }
  T* dst = y + m * a.out_ld + a.out_coff + c;
  st4(dst, make_float4(best[0], best[1], best[2], best[3]));
  if constexpr (sizeof(T) == 2) st4(dst + 4, make_float4(best[4], best[5], best[6], best[7]));
  if (idx) {
    if (a.dead)
#pragma unroll
      for (int q = 0; q < NCH; ++q)
        if (!(best[q] > 0.f)) bi[q] = 255u;
    unsigned pk[NCH / 4];
#pragma unroll
    for (int k = 0; k < NCH / 4; ++k) pk[k] = bi[4 * k] | (bi[4 * k + 1] << 8) | (bi[4 * k + 2] << 16) | (bi[4 * k + 3] << 24);
    if constexpr (NCH == 8) *reinterpret_cast<uint2*>(idx + m * a.C + c) = make_uint2(pk[0], pk[1]);
    else *reinterpret_cast<unsigned*>(idx + m * a.C + c) = pk[0];
  }
}

template <class T, int KT, int KH, int KW>
static bool launch_pool_fwd_fixed(const PoolArgs& a, const T* x, T* y, unsigned char* argmax, hipStream_t s) {
  constexpr int NCH = 16 / (int)sizeof(T);
  if (a.kT != KT || a.kH != KH || a.kW != KW) return false;
  if (a.C % NCH || a.in_ld % NCH || a.in_coff % NCH || a.out_ld % NCH || a.out_coff % NCH) return false;
  const long nbx = cdiv((long)a.Wo * (a.C / NCH), 256), nblk = nbx * a.Ho * a.B * a.To;
  if (nblk >= 0x7fffffffL) return false;
  hipLaunchKernelGGL((maxpool_fwd_fixed_kernel<T, KT, KH, KW>), dim3((unsigned)nblk), dim3(256), 0, s, x, y, argmax, a, (int)nbx);
  return true;
}

// backward: for every input cell sum dY of the windows whose recorded winner is
// this cell.  dy has (out_ld, out_coff) geometry, dx has (in_ld, in_coff).
template <class T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                   T* __restrict__ dx, const T* __restrict__ relu_mask,
                                   int accumulate, PoolArgs a) {
  const int C4 = a.C >> 2;
  size_t total = (size_t)a.B * a.Ti * a.Hi * a.Wi * C4;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int c4 = i % C4;
    size_t m = i / C4;
    int wi = m % a.Wi;
    size_t t1 = m / a.Wi;
    int hi = t1 % a.Hi;
    size_t t2 = t1 / a.Hi;
    int ti = t2 % a.Ti;
    int b = t2 / a.Ti;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    // windows `to` covering ti: to*s - p <= ti <= to*s - p + k - 1
    //   => ceil((ti + p - k + 1) / s) <= to <= floor((ti + p) / s), clipped to [0, To)
    const int nt = ti + a.pT, nh = hi + a.pH, nw = wi + a.pW;
    const int t_hi = min(nt / a.sT, a.To - 1), t_lo = max((nt - a.kT + a.sT) / a.sT, 0);
    const int h_hi = min(nh / a.sH, a.Ho - 1), h_lo = max((nh - a.kH + a.sH) / a.sH, 0);
    const int w_hi = min(nw / a.sW, a.Wo - 1), w_lo = max((nw - a.kW + a.sW) / a.sW, 0);
    for (int to = t_lo; to <= t_hi; ++to) {
      const int kt = nt - to * a.sT;
      for (int ho = h_lo; ho <= h_hi; ++ho) {
        const int kh = nh - ho * a.sH;
        const size_t rowo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
          const int kw = nw - wo * a.sW;
          const int tap = (kt * a.kH + kh) * a.kW + kw;
          const size_t mo = rowo + wo;
          uchar4 u = *reinterpret_cast<const uchar4*>(idx + mo * a.C + 4 * c4);
          float4 g = ld4(dy + mo * a.out_ld + a.out_coff + 4 * c4);
          if (u.x == tap) acc[0] += g.x;
          if (u.y == tap) acc[1] += g.y;
          if (u.z == tap) acc[2] += g.z;
          if (u.w == tap) acc[3] += g.w;
        }
      }
    }
    T* dst = dx + m * a.in_ld + a.in_coff + 4 * c4;
    if (accumulate) {
      float4 o = ld4(dst);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
    if (relu_mask) {
      float4 k = ld4(relu_mask + m * a.in_ld + a.in_coff + 4 * c4);
      if (!(k.x > 0.f)) acc[0] = 0.f;
      if (!(k.y > 0.f)) acc[1] = 0.f;
      if (!(k.z > 0.f)) acc[2] = 0.f;
      if (!(k.w > 0.f)) acc[3] = 0.f;
    }
    st4(dst, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
}

// backward for the strided pools with the window geometry known at compile time: at most
// ceil(k/s) outputs per dimension cover an input cell (2 x 2 for the 1x3x3 / (1,2,2) pools),
// so every candidate's (dY, arg-max) load is issued up front, unconditionally predicated --
// no dependent loops, neighbouring cells' re-reads come from L1.  One thread per input cell
// and 4 channels; contributions are added in ascending (to, ho, wo) order like everywhere else.
template <class T, int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool_bwd_fixed_kernel(const T* __restrict__ dy,
                                                                const unsigned char* __restrict__ idx,
                                                                T* __restrict__ dx,
                                                                const T* __restrict__ relu_mask, int accumulate,
                                                                PoolArgs a) {
  constexpr int NT = (KT + ST - 1) / ST, NH = (KH + SH - 1) / SH, NW = (KW + SW - 1) / SW;
  // grid (blocks of one image row's (w, 4-channel group) cells, h, clip * frame): the row coordinates come from the
  // block index on the scalar unit, one 32-bit division per thread is left (a flat 64-bit index cost five 64-bit
  // divisions per thread: this kernel ran at 2.6 TB/s beside a forward at 4.8 over the same bytes).
  // (Numbering the workgroups so that each XCD owns a contiguous run of rows -- as the forward does -- was measured in
  // round 3: 5 % SLOWER here; the kernel already moves its compulsory bytes at 4.5-4.7 TB/s.)
  const unsigned C4 = (unsigned)a.C >> 2;
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (unsigned)a.Wi * C4) return;
  const int wi = (int)(r / C4);
  const int c4 = (int)(r - (unsigned)wi * C4);
  const int hi = blockIdx.y;
  const int ti = (int)(blockIdx.z % (unsigned)a.Ti);
  const int b = (int)(blockIdx.z / (unsigned)a.Ti);
  const size_t m = ((size_t)(b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi;
  const int nt = ti + a.pT, nh = hi + a.pH, nw = wi + a.pW;
  float4 g[NT][NH][NW];
  unsigned u[NT][NH][NW];
  // candidate j along a dim: output o = n/s - (N-1-j) (ascending in j), tap k = n - o*s
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int to = nt / ST - (NT - 1 - jt), kt = nt - to * ST;
#pragma unroll
    for (int jh = 0; jh < NH; ++jh) {
      const int ho = nh / SH - (NH - 1 - jh), kh = nh - ho * SH;
#pragma unroll
      for (int jw = 0; jw < NW; ++jw) {
        const int wo = nw / SW - (NW - 1 - jw), kw = nw - wo * SW;
        const bool ok = to >= 0 && to < a.To && kt < KT && ho >= 0 && ho < a.Ho && kh < KH && wo >= 0 &&
                        wo < a.Wo && kw < KW;
        g[jt][jh][jw] = make_float4(0.f, 0.f, 0.f, 0.f);
        u[jt][jh][jw] = 0xffffffffu;
        if (ok) {
          const size_t mo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
          g[jt][jh][jw] = ld4(dy + mo * a.out_ld + a.out_coff + 4 * c4);
          // compare against this cell's tap: store (recorded tap XOR own tap), zero byte = match
          const unsigned tap = (unsigned)((kt * KH + kh) * KW + kw);
          u[jt][jh][jw] = *reinterpret_cast<const unsigned*>(idx + mo * a.C + 4 * c4) ^ (tap * 0x01010101u);
        }
      }
    }
  }
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int jh = 0; jh < NH; ++jh)
#pragma unroll
      for (int jw = 0; jw < NW; ++jw) {
        const unsigned x = u[jt][jh][jw];
        const float4 v = g[jt][jh][jw];
        if ((x & 0x000000ffu) == 0u) acc[0] += v.x;
        if ((x & 0x0000ff00u) == 0u) acc[1] += v.y;
        if ((x & 0x00ff0000u) == 0u) acc[2] += v.z;
        if ((x & 0xff000000u) == 0u) acc[3] += v.w;
      }
  T* dst = dx + m * a.in_ld + a.in_coff + 4 * c4;
  if (accumulate) {
    float4 o = ld4(dst);
    acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
  }
  if (relu_mask) {
    float4 k = ld4(relu_mask + m * a.in_ld + a.in_coff + 4 * c4);
    if (!(k.x > 0.f)) acc[0] = 0.f;
    if (!(k.y > 0.f)) acc[1] = 0.f;
    if (!(k.z > 0.f)) acc[2] = 0.f;
    if (!(k.w > 0.f)) acc[3] = 0.f;
  }
  st4(dst, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

template <class T, int KT, int KH, int KW, int ST, int SH, int SW>
static bool launch_pool_bwd_fixed(const PoolArgs& a, const T* dy, const unsigned char* idx, T* dx,
                                  const T* relu_mask, int accumulate, hipStream_t s) {
  if (a.kT != KT || a.kH != KH || a.kW != KW || a.sT != ST || a.sH != SH || a.sW != SW) return false;
  if (a.Hi > 65535 || (long)a.B * a.Ti > 65535) return false;
  hipLaunchKernelGGL((maxpool_bwd_fixed_kernel<T, KT, KH, KW, ST, SH, SW>),
                     dim3((unsigned)cdiv((long)a.Wi * (a.C / 4), 256), (unsigned)a.Hi, (unsigned)(a.B * a.Ti)), dim3(256), 0, s,
                     dy, idx, dx, relu_mask, accumulate, a);
  return true;
}

// The same backward with 16 bytes of channels per lane and every candidate's RAW bits requested first (bf16 storage:
// 8 channels per thread halve the instructions per byte; measured on fp32 the form above is faster, so it keeps it).
// backward for the strided pools with the window geometry known at compile time: at most
// ceil(k/s) outputs per dimension cover an input cell (2 x 2 for the 1x3x3 / (1,2,2) pools),
// so every candidate's (dY, arg-max) load is issued up front, unconditionally predicated --
// no dependent loops, neighbouring cells' re-reads come from L1.  One thread per input cell
// and 16 bytes of channels; contributions are added in ascending (to, ho, wo) order like everywhere else.
template <class T, int KT, int KH, int KW, int ST, int SH, int SW>
__global__ __launch_bounds__(256) void maxpool_bwd_fixed16_kernel(const T* __restrict__ dy,
                                                                const unsigned char* __restrict__ idx,
                                                                T* __restrict__ dx,
                                                                const T* __restrict__ relu_mask, int accumulate,
                                                                PoolArgs a) {
  constexpr int NT = (KT + ST - 1) / ST, NH = (KH + SH - 1) / SH, NW = (KW + SW - 1) / SW;
  constexpr int NCH = 16 / (int)sizeof(T);   // 16 bytes per lane: 4 fp32 or 8 bf16 channels
  // grid (blocks of one image row's (w, channel group) cells, h, clip * frame): the row coordinates come from the
  // block index on the scalar unit, one 32-bit division per thread is left (a flat 64-bit index cost five 64-bit
  // divisions per thread: this kernel ran at 2.6 TB/s beside a forward at 4.8 over the same bytes).
  // (Numbering the workgroups so that each XCD owns a contiguous run of rows -- as the forward does -- was measured in
  // round 3: 5 % SLOWER here.)
  const unsigned CG = (unsigned)a.C / NCH;
  const unsigned r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= (unsigned)a.Wi * CG) return;
  const int wi = (int)(r / CG);
  const int c = (int)(r - (unsigned)wi * CG) * NCH;
  const int hi = blockIdx.y;
  const int ti = (int)(blockIdx.z % (unsigned)a.Ti);
  const int b = (int)(blockIdx.z / (unsigned)a.Ti);
  const size_t m = ((size_t)(b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi;
  const int nt = ti + a.pT, nh = hi + a.pH, nw = wi + a.pW;
  // raw bits of every candidate first (dY: 16 bytes, arg-max: NCH bytes), compared and converted afterwards: a value
  // consumed inside its bounds branch is waited for at once
  uint4 g[NT][NH][NW];
  unsigned u[NT][NH][NW][NCH / 4];
  unsigned tp[NT][NH][NW];
  // candidate j along a dim: output o = n/s - (N-1-j) (ascending in j), tap k = n - o*s
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
    const int to = nt / ST - (NT - 1 - jt), kt = nt - to * ST;
#pragma unroll
    for (int jh = 0; jh < NH; ++jh) {
      const int ho = nh / SH - (NH - 1 - jh), kh = nh - ho * SH;
#pragma unroll
      for (int jw = 0; jw < NW; ++jw) {
        const int wo = nw / SW - (NW - 1 - jw), kw = nw - wo * SW;
        const bool ok = to >= 0 && to < a.To && kt < KT && ho >= 0 && ho < a.Ho && kh < KH && wo >= 0 &&
                        wo < a.Wo && kw < KW;
        g[jt][jh][jw] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int k = 0; k < NCH / 4; ++k) u[jt][jh][jw][k] = 0xffffffffu;
        tp[jt][jh][jw] = ok ? (unsigned)((kt * KH + kh) * KW + kw) * 0x01010101u : 0u;   // (0xff never equals a tap)
        if (ok) {
          const size_t mo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
          g[jt][jh][jw] = *reinterpret_cast<const uint4*>(dy + mo * a.out_ld + a.out_coff + c);
          if constexpr (NCH == 8) {
            const uint2 t2 = *reinterpret_cast<const uint2*>(idx + mo * a.C + c);
            u[jt][jh][jw][0] = t2.x;
            u[jt][jh][jw][1] = t2.y;
          } else {
            // (fp32: XOR-ed at once, i.e. waited for inside the branch -- measured 9 % FASTER than the all-up-front
            // form on this bandwidth-bound variant: 827 vs 903 us on MaxPool3d_2a at B=64)
            u[jt][jh][jw][0] = *reinterpret_cast<const unsigned*>(idx + mo * a.C + c) ^ tp[jt][jh][jw];
          }
        }
      }
    }
  }
  float acc[NCH];
#pragma unroll
  for (int q = 0; q < NCH; ++q) acc[q] = 0.f;
#pragma unroll
  for (int jt = 0; jt < NT; ++jt)
#pragma unroll
    for (int jh = 0; jh < NH; ++jh)
#pragma unroll
      for (int jw = 0; jw < NW; ++jw) {
        const unsigned w4[4] = {g[jt][jh][jw].x, g[jt][jh][jw].y, g[jt][jh][jw].z, g[jt][jh][jw].w};
        unsigned xr[NCH / 4];   // recorded taps XOR own tap: zero byte = this cell won that window
#pragma unroll
        for (int k = 0; k < NCH / 4; ++k) xr[k] = NCH == 4 ? u[jt][jh][jw][k] : (u[jt][jh][jw][k] ^ tp[jt][jh][jw]);
#pragma unroll
        for (int q = 0; q < NCH; ++q) {
          float v;
          if constexpr (sizeof(T) == 2) v = __uint_as_float((q & 1) ? (w4[q >> 1] & 0xffff0000u) : (w4[q >> 1] << 16));
          else v = __uint_as_float(w4[q]);
          if ((xr[q >> 2] & (0xffu << (8 * (q & 3)))) == 0u) acc[q] += v;
        }
      }
  T* dst = dx + m * a.in_ld + a.in_coff + c;
#pragma unroll
  for (int k = 0; k < NCH / 4; ++k) {
    float* o = acc + 4 * k;
    if (accumulate) {
      float4 old = ld4(dst + 4 * k);
      o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
    }
    if (relu_mask) {
      float4 kk = ld4(relu_mask + m * a.in_ld + a.in_coff + c + 4 * k);
      if (!(kk.x > 0.f)) o[0] = 0.f;
      if (!(kk.y > 0.f)) o[1] = 0.f;
      if (!(kk.z > 0.f)) o[2] = 0.f;
      if (!(kk.w > 0.f)) o[3] = 0.f;
    }
    st4(dst + 4 * k, make_float4(o[0], o[1], o[2], o[3]));
  }
}

template <class T, int KT, int KH, int KW, int ST, int SH, int SW>
static bool launch_pool_bwd_fixed16(const PoolArgs& a, const T* dy, const unsigned char* idx, T* dx,
                                  const T* relu_mask, int accumulate, hipStream_t s) {
  constexpr int NCH = 16 / (int)sizeof(T);
  if (a.kT != KT || a.kH != KH || a.kW != KW || a.sT != ST || a.sH != SH || a.sW != SW) return false;
  if (a.Hi > 65535 || (long)a.B * a.Ti > 65535) return false;
  if (a.C % NCH || a.in_ld % NCH || a.in_coff % NCH || a.out_ld % NCH || a.out_coff % NCH) return false;
  hipLaunchKernelGGL((maxpool_bwd_fixed16_kernel<T, KT, KH, KW, ST, SH, SW>),
                     dim3((unsigned)cdiv((long)a.Wi * (a.C / NCH), 256), (unsigned)a.Hi, (unsigned)(a.B * a.Ti)), dim3(256), 0, s,
                     dy, idx, dx, relu_mask, accumulate, a);
  return true;
}

// ---------------------------------------------------------------- LDS-tiled max-pool
// The direct kernels above re-read every input (forward) or every (dY, arg-max) pair
// (backward) once per window that covers it -- up to 27x through L1/L2.  The tiled forms
// stage the needed region of one 32-channel slab in LDS once (coalesced 128-byte rows) and
// run the same window scan / gather against LDS, so global traffic is ~1x the tensors.
constexpr int POOL_SLAB = 16;       // channels per workgroup
constexpr int POOL_ROW = POOL_SLAB + 4;   // floats per LDS row (pad: conflict-free 16-byte reads)

struct PoolTile {
  int tT, tH, tW;      // tile of outputs (forward) / inputs (backward) per workgroup
  int rT, rH, rW;      // staged region extents
  int nT, nH, nW;      // tiles per dim
  int slabs;
};

template <class T>
__global__ __launch_bounds__(256) void maxpool_fwd_tiled_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                unsigned char* __restrict__ idx, PoolArgs a,
                                                                PoolTile t) {
  extern __shared__ __attribute__((aligned(16))) float sx[];   // [rT*rH*rW][POOL_ROW]
  int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halo rows) on one XCD, i.e. one L2
  const int slab = blk % t.slabs; blk /= t.slabs;
  const int iw = blk % t.nW; blk /= t.nW;
  const int ih = blk % t.nH; blk /= t.nH;
  const int it = blk % t.nT;
  const int b = blk / t.nT;
  const int o_t0 = it * t.tT, o_h0 = ih * t.tH, o_w0 = iw * t.tW;
  const int i_t0 = o_t0 * a.sT - a.pT, i_h0 = o_h0 * a.sH - a.pH, i_w0 = o_w0 * a.sW - a.pW;
  const int c0 = slab * POOL_SLAB;
  const int nreg = t.rT * t.rH * t.rW;
  // stage: 8 float4 per position, zero fill outside the tensor (zero padding) and beyond C
  for (int i = threadIdx.x; i < nreg * (POOL_SLAB / 4); i += blockDim.x) {
    int g = i % (POOL_SLAB / 4), r = i / (POOL_SLAB / 4);
    int rw = r % t.rW;
    int r2 = r / t.rW;
    int rh = r2 % t.rH;
    int rt = r2 / t.rH;
    int ti = i_t0 + rt, hi = i_h0 + rh, wi = i_w0 + rw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((unsigned)ti < (unsigned)a.Ti && (unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi &&
        c0 + 4 * g < a.C)
      v = ld4(x + ((size_t)((b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi) * a.in_ld +
                                           a.in_coff + c0 + 4 * g);
    *reinterpret_cast<float4*>(&sx[r * POOL_ROW + 4 * g]) = v;
  }
  __syncthreads();
  const int nout = t.tT * t.tH * t.tW;
  for (int i = threadIdx.x; i < nout * (POOL_SLAB / 4); i += blockDim.x) {
    int g = i % (POOL_SLAB / 4), o = i / (POOL_SLAB / 4);
    int ow = o % t.tW;
    int o2 = o / t.tW;
    int oh = o2 % t.tH;
    int ot = o2 / t.tH;
    int to = o_t0 + ot, ho = o_h0 + oh, wo = o_w0 + ow;
    if (to >= a.To || ho >= a.Ho || wo >= a.Wo || c0 + 4 * g >= a.C) continue;
    float best[4];
    int bi[4];
    int tap = 0;
    for (int kt = 0; kt < a.kT; ++kt)
      for (int kh = 0; kh < a.kH; ++kh)
        for (int kw = 0; kw < a.kW; ++kw, ++tap) {
          int r = ((ot * a.sT + kt) * t.rH + oh * a.sH + kh) * t.rW + ow * a.sW + kw;
          float4 v = *reinterpret_cast<const float4*>(&sx[r * POOL_ROW + 4 * g]);
          float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (tap == 0 || vv[q] > best[q] || vv[q] != vv[q]) { best[q] = vv[q]; bi[q] = tap; }
        }
    size_t m = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
    st4(y + m * a.out_ld + a.out_coff + c0 + 4 * g, make_float4(best[0], best[1], best[2], best[3]));
    if (idx) {
      if (a.dead)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (!(best[q] > 0.f)) bi[q] = 255;
      *reinterpret_cast<uchar4*>(idx + m * a.C + c0 + 4 * g) = make_uchar4(bi[0], bi[1], bi[2], bi[3]);
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void maxpool_bwd_tiled_kernel(const T* __restrict__ dy,
                                                                const unsigned char* __restrict__ idx,
                                                                T* __restrict__ dx,
                                                                const T* __restrict__ relu_mask,
                                                                int accumulate, PoolArgs a, PoolTile t, int o_lo_t,
                                                                int o_lo_h, int o_lo_w) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // dY region [n][POOL_ROW] floats, then arg-max bytes
  int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halo rows) on one XCD, i.e. one L2
  const int slab = blk % t.slabs; blk /= t.slabs;
  const int iw = blk % t.nW; blk /= t.nW;
  const int ih = blk % t.nH; blk /= t.nH;
  const int it = blk % t.nT;
  const int b = blk / t.nT;
  const int i_t0 = it * t.tT, i_h0 = ih * t.tH, i_w0 = iw * t.tW;
  // first output index whose window can cover the tile's first input: ceil((i0 + p - k + 1)/s), clipped later
  auto first_out = [](int i0, int p, int k, int s) { int n = i0 + p - k + s; return n >= 0 ? n / s : -((-n + s - 1) / s); };
  const int r_t0 = first_out(i_t0, a.pT, a.kT, a.sT), r_h0 = first_out(i_h0, a.pH, a.kH, a.sH),
            r_w0 = first_out(i_w0, a.pW, a.kW, a.sW);
  (void)o_lo_t; (void)o_lo_h; (void)o_lo_w;
  const int c0 = slab * POOL_SLAB;
  const int nreg = t.rT * t.rH * t.rW;
  unsigned char* si = reinterpret_cast<unsigned char*>(sm + (size_t)nreg * POOL_ROW);   // [nreg][POOL_SLAB]
  for (int i = threadIdx.x; i < nreg * (POOL_SLAB / 4); i += blockDim.x) {
    int g = i % (POOL_SLAB / 4), r = i / (POOL_SLAB / 4);
    int rw = r % t.rW;
    int r2 = r / t.rW;
    int rh = r2 % t.rH;
    int rt = r2 / t.rH;
    int to = r_t0 + rt, ho = r_h0 + rh, wo = r_w0 + rw;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 u = make_uchar4(255, 255, 255, 255);   // matches no tap
    if ((unsigned)to < (unsigned)a.To && (unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo &&
        c0 + 4 * g < a.C) {
      size_t mo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
      v = ld4(dy + mo * a.out_ld + a.out_coff + c0 + 4 * g);
      u = *reinterpret_cast<const uchar4*>(idx + mo * a.C + c0 + 4 * g);
    }
    *reinterpret_cast<float4*>(&sm[r * POOL_ROW + 4 * g]) = v;
    *reinterpret_cast<uchar4*>(&si[r * POOL_SLAB + 4 * g]) = u;
  }
  __syncthreads();
  const int nin = t.tT * t.tH * t.tW;
  for (int i = threadIdx.x; i < nin * (POOL_SLAB / 4); i += blockDim.x) {
    int g = i % (POOL_SLAB / 4), p = i / (POOL_SLAB / 4);
    int pw = p % t.tW;
    int p2 = p / t.tW;
    int ph = p2 % t.tH;
    int pt = p2 / t.tH;
    int ti = i_t0 + pt, hi = i_h0 + ph, wi = i_w0 + pw;
    if (ti >= a.Ti || hi >= a.Hi || wi >= a.Wi || c0 + 4 * g >= a.C) continue;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nt = ti + a.pT, nh = hi + a.pH, nw = wi + a.pW;
    const int t_hi = min(nt / a.sT, a.To - 1), t_lo = max((nt - a.kT + a.sT) / a.sT, 0);
    const int h_hi = min(nh / a.sH, a.Ho - 1), h_lo = max((nh - a.kH + a.sH) / a.sH, 0);
    const int w_hi = min(nw / a.sW, a.Wo - 1), w_lo = max((nw - a.kW + a.sW) / a.sW, 0);
    for (int to = t_lo; to <= t_hi; ++to) {
      const int kt = nt - to * a.sT;
      for (int ho = h_lo; ho <= h_hi; ++ho) {
        const int kh = nh - ho * a.sH;
        for (int wo = w_lo; wo <= w_hi; ++wo) {
          const int kw = nw - wo * a.sW;
          const int tap = (kt * a.kH + kh) * a.kW + kw;
          const int r = ((to - r_t0) * t.rH + (ho - r_h0)) * t.rW + (wo - r_w0);
          uchar4 u = *reinterpret_cast<const uchar4*>(&si[r * POOL_SLAB + 4 * g]);
          float4 gq = *reinterpret_cast<const float4*>(&sm[r * POOL_ROW + 4 * g]);
          if (u.x == tap) acc[0] += gq.x;
          if (u.y == tap) acc[1] += gq.y;
          if (u.z == tap) acc[2] += gq.z;
          if (u.w == tap) acc[3] += gq.w;
        }
      }
    }
    size_t m = ((size_t)(b * a.Ti + ti) * a.Hi + hi) * a.Wi + wi;
    T* dst = dx + m * a.in_ld + a.in_coff + c0 + 4 * g;
    if (accumulate) {
      float4 o = ld4(dst);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
    if (relu_mask) {
      float4 k = ld4(relu_mask + m * a.in_ld + a.in_coff + c0 + 4 * g);
      if (!(k.x > 0.f)) acc[0] = 0.f;
      if (!(k.y > 0.f)) acc[1] = 0.f;
      if (!(k.z > 0.f)) acc[2] = 0.f;
      if (!(k.w > 0.f)) acc[3] = 0.f;
    }
    st4(dst, make_float4(acc[0], acc[1], acc[2], acc[3]));
  }
}

// ---------------------------------------------------------------- 3x3x3 stride-1 'same' pools
// The Inception branch pools (k 3, s 1, p 1).  One thread owns an image column (w, 4
// channels) of a TH-row tile and walks the planes of the clip.  The window maximum is
// separable and so is its first-occurrence arg-max (lowest kt, then kh, then kw with the
// maximum value): row maxima over kw (3 loads per staged row, neighbours served by L1), a
// rolling max over kh, and a rolling max over kt kept in registers -- 27/7 loads per output
// instead of 27, no LDS, no barriers.
constexpr int S1_TH = 7;

__device__ __forceinline__ bool pool_takes(float v, float best) { return v > best || v != v; }

template <class T>
__global__ __launch_bounds__(256) void maxpool3s1_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                             unsigned char* __restrict__ idx, PoolArgs a, int G,
                                                             int nH, int slabs) {
  int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halo rows) on one XCD, i.e. one L2
  const int slab = blk % slabs; blk /= slabs;
  const int ih = blk % nH;
  const int b = blk / nH;
  const int g = threadIdx.x % G, w = threadIdx.x / G;
  const int c = slab * G * 4 + 4 * g;
  if (w >= a.Wi || c >= a.C) return;
  const int h0 = ih * S1_TH;
  // (lane pairs share loads only when both lanes exist and the pair's 16 bytes are aligned: block-uniform)
  const bool pair_ok = sizeof(T) == 2 && (G & 1) == 0 && (a.C & 7) == 0 && (a.in_ld & 7) == 0 && (a.in_coff & 7) == 0;
  float pv1[S1_TH][4], pv2[S1_TH][4];
  unsigned pi1[S1_TH], pi2[S1_TH];   // 2-D taps (kh*3+kw), one byte per channel
#pragma unroll
  for (int h = 0; h < S1_TH; ++h) {
    pi1[h] = pi2[h] = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) pv1[h][q] = pv2[h][q] = 0.f;
  }
  for (int t = 0; t <= a.Ti; ++t) {
    float cv[S1_TH][4];
    unsigned ci[S1_TH];
    if (t < a.Ti) {
      float rv[S1_TH + 2][4];
      unsigned ri[S1_TH + 2];
      // bf16 storage.  (1) The raw bits of ALL rows are requested before any is converted: with the conversion next
      // to the load the compiler waited (`s_waitcnt vmcnt(0)`) inside every bounds branch, 42 serial round trips per
      // plane where the fp32 form has one, and the bf16 pools ran 1.7x SLOWER than the fp32 ones.  (2) An 8-byte load
      // per lane moves half the bytes per load instruction, so two lanes that hold neighbouring channel groups
      // (g even / odd: 16 contiguous bytes) share their loads: the even lane fetches both groups of row r, the odd
      // lane both groups of row r + 1, and they swap halves (two DPP moves).
      float4 vp[(S1_TH + 3) / 2 * 2][3];
      if constexpr (sizeof(T) == 2) {
        const bool odd = (g & 1) != 0;
        if (pair_ok) {
          const int cp = c & ~7;   // first channel of the pair
          uint4 raw[(S1_TH + 3) / 2][3];
#pragma unroll
          for (int r2 = 0; r2 < (S1_TH + 3) / 2; ++r2) {
            const int hi = h0 - 1 + 2 * r2 + (odd ? 1 : 0);   // the row THIS lane fetches
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int wi = w - 1 + j;
              raw[r2][j] = make_uint4(0u, 0u, 0u, 0u);
              if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi && 2 * r2 + (odd ? 1 : 0) < S1_TH + 2)
                raw[r2][j] = *reinterpret_cast<const uint4*>(x + ((size_t)((b * a.Ti + t) * a.Hi + hi) * a.Wi + wi) * a.in_ld + a.in_coff + cp);
            }
          }
#pragma unroll
          for (int r2 = 0; r2 < (S1_TH + 3) / 2; ++r2) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const uint4 u = raw[r2][j];
              // even lane keeps its low half (row 2 r2) and receives the partner's low half (row 2 r2 + 1);
              // odd lane receives the partner's high half (row 2 r2) and keeps its own high half (row 2 r2 + 1)
              const unsigned s0 = odd ? u.x : u.z, s1 = odd ? u.y : u.w;
              const unsigned g0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s0, 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
              const unsigned g1 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)s1, 0xb1, 0xf, 0xf, false);
              const unsigned a0 = odd ? g0 : u.x, a1 = odd ? g1 : u.y;     // row 2 r2, this lane's 4 channels
              const unsigned b0 = odd ? u.z : g0, b1 = odd ? u.w : g1;     // row 2 r2 + 1
              vp[2 * r2][j] = make_float4(__uint_as_float(a0 << 16), __uint_as_float(a0 & 0xffff0000u),
                                          __uint_as_float(a1 << 16), __uint_as_float(a1 & 0xffff0000u));
              vp[2 * r2 + 1][j] = make_float4(__uint_as_float(b0 << 16), __uint_as_float(b0 & 0xffff0000u),
                                              __uint_as_float(b1 << 16), __uint_as_float(b1 & 0xffff0000u));
            }
          }
        } else {
          uint2 raw[S1_TH + 2][3];
#pragma unroll
          for (int r = 0; r < S1_TH + 2; ++r) {
            const int hi = h0 - 1 + r;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              const int wi = w - 1 + j;
              raw[r][j] = make_uint2(0u, 0u);
              if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
                raw[r][j] = *reinterpret_cast<const uint2*>(x + ((size_t)((b * a.Ti + t) * a.Hi + hi) * a.Wi + wi) * a.in_ld + a.in_coff + c);
            }
          }
#pragma unroll
          for (int r = 0; r < S1_TH + 2; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j)
              vp[r][j] = make_float4(__uint_as_float(raw[r][j].x << 16), __uint_as_float(raw[r][j].x & 0xffff0000u),
                                     __uint_as_float(raw[r][j].y << 16), __uint_as_float(raw[r][j].y & 0xffff0000u));
        }
      }
#pragma unroll
      for (int r = 0; r < S1_TH + 2; ++r) {
        const int hi = h0 - 1 + r;
        float4 v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (sizeof(T) == 2) {
            v[j] = vp[r][j];
          } else {
            const int wi = w - 1 + j;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)hi < (unsigned)a.Hi && (unsigned)wi < (unsigned)a.Wi)
              v[j] = ld4(x + ((size_t)((b * a.Ti + t) * a.Hi + hi) * a.Wi + wi) * a.in_ld + a.in_coff + c);
          }
        }
        const float v0[4] = {v[0].x, v[0].y, v[0].z, v[0].w};
        const float v1[4] = {v[1].x, v[1].y, v[1].z, v[1].w};
        const float v2[4] = {v[2].x, v[2].y, v[2].z, v[2].w};
        unsigned pk = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float best = v0[q];
          unsigned k = 0u;
          if (pool_takes(v1[q], best)) { best = v1[q]; k = 1u; }
          if (pool_takes(v2[q], best)) { best = v2[q]; k = 2u; }
          rv[r][q] = best;
          pk |= k << (8 * q);
        }
        ri[r] = pk;
      }
#pragma unroll
      for (int h = 0; h < S1_TH; ++h) {
        unsigned pk = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float best = rv[h][q];
          unsigned k = (ri[h] >> (8 * q)) & 0xffu;
          if (pool_takes(rv[h + 1][q], best)) { best = rv[h + 1][q]; k = 3u + ((ri[h + 1] >> (8 * q)) & 0xffu); }
          if (pool_takes(rv[h + 2][q], best)) { best = rv[h + 2][q]; k = 6u + ((ri[h + 2] >> (8 * q)) & 0xffu); }
          cv[h][q] = best;
          pk |= k << (8 * q);
        }
        ci[h] = pk;
      }
    } else {
#pragma unroll
      for (int h = 0; h < S1_TH; ++h) {
        ci[h] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[h][q] = 0.f;
      }
    }
    if (t >= 1) {
      const int to = t - 1;
#pragma unroll
      for (int h = 0; h < S1_TH; ++h) {
        const int ho = h0 + h;
        if (ho >= a.Ho) continue;
        float o[4];
        unsigned pk = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float best = pv2[h][q];
          unsigned k = (pi2[h] >> (8 * q)) & 0xffu;
          if (pool_takes(pv1[h][q], best)) { best = pv1[h][q]; k = 9u + ((pi1[h] >> (8 * q)) & 0xffu); }
          if (pool_takes(cv[h][q], best)) { best = cv[h][q]; k = 18u + ((ci[h] >> (8 * q)) & 0xffu); }
          o[q] = best;
          pk |= k << (8 * q);
        }
        const size_t m = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + w;
        st4(y + m * a.out_ld + a.out_coff + c, make_float4(o[0], o[1], o[2], o[3]));
        if (idx) {
          if (a.dead)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (!(o[q] > 0.f)) pk |= 0xffu << (8 * q);
          *reinterpret_cast<unsigned*>(idx + m * a.C + c) = pk;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < S1_TH; ++h) {
      pi2[h] = pi1[h];
      pi1[h] = ci[h];
#pragma unroll
      for (int q = 0; q < 4; ++q) { pv2[h][q] = pv1[h][q]; pv1[h][q] = cv[h][q]; }
    }
  }
}

// backward of the same pools as a gather: the thread of input cell (h, w) reads the (dY,
// arg-max) of the 3x3 outputs around it plane by plane and adds dY where the recorded tap
// points back at its cell; three accumulators (input planes to-1, to, to+1) roll along t.
// The 27 compare-select-adds per element bound this kernel (VALU), so it is written for
// occupancy: ~60 registers, loads of neighbouring cells served by L1.  Contributions arrive
// in ascending (to, ho, wo) order, as in the other backward kernels.
template <class T>
__global__ __launch_bounds__(256) void maxpool3s1_bwd_kernel(const T* __restrict__ dy,
                                                             const unsigned char* __restrict__ idx,
                                                             T* __restrict__ dx,
                                                             const T* __restrict__ relu_mask, int accumulate,
                                                             PoolArgs a, int G, int slabs) {
  int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halo rows) on one XCD, i.e. one L2
  const int slab = blk % slabs; blk /= slabs;
  const int h = blk % a.Hi;
  const int b = blk / a.Hi;
  const int g = threadIdx.x % G, w = threadIdx.x / G;
  const int c = slab * G * 4 + 4 * g;
  if (w >= a.Wi || c >= a.C) return;
  float acc[3][4];
#pragma unroll
  for (int p = 0; p < 3; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[p][q] = 0.f;
  auto store_plane = [&](int ti) {
    const size_t m = ((size_t)(b * a.Ti + ti) * a.Hi + h) * a.Wi + w;
    T* dst = dx + m * a.in_ld + a.in_coff + c;
    float o[4] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3]};
    if (accumulate) {
      float4 old = ld4(dst);
      o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
    }
    if (relu_mask) {
      float4 k = ld4(relu_mask + m * a.in_ld + a.in_coff + c);
      if (!(k.x > 0.f)) o[0] = 0.f;
      if (!(k.y > 0.f)) o[1] = 0.f;
      if (!(k.z > 0.f)) o[2] = 0.f;
      if (!(k.w > 0.f)) o[3] = 0.f;
    }
    st4(dst, make_float4(o[0], o[1], o[2], o[3]));
  };
  for (int to = 0; to < a.To; ++to) {
    // bf16 storage: the raw bits of all nine neighbours are requested before any is converted (with the conversion
    // beside the load the compiler waits inside every bounds branch: nine serial round trips per plane)
    uint2 raw[3][3];
    unsigned ui[3][3];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int ho = h - 1 + r;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const int wo = w - 1 + j;
          raw[r][j] = make_uint2(0u, 0u);
          ui[r][j] = 0xffffffffu;
          if ((unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo) {
            const size_t mo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
            raw[r][j] = *reinterpret_cast<const uint2*>(dy + mo * a.out_ld + a.out_coff + c);
            ui[r][j] = *reinterpret_cast<const unsigned*>(idx + mo * a.C + c);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int ho = h - 1 + r;
      const int kh = 2 - r;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int wo = w - 1 + j;
        const int kw = 2 - j;
        float4 gq = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned u = 0xffffffffu;   // matches no tap
        if constexpr (sizeof(T) == 2) {
          gq = make_float4(__uint_as_float(raw[r][j].x << 16), __uint_as_float(raw[r][j].x & 0xffff0000u),
                           __uint_as_float(raw[r][j].y << 16), __uint_as_float(raw[r][j].y & 0xffff0000u));
          u = ui[r][j];
        } else if ((unsigned)ho < (unsigned)a.Ho && (unsigned)wo < (unsigned)a.Wo) {
          const size_t mo = ((size_t)(b * a.To + to) * a.Ho + ho) * a.Wo + wo;
          gq = ld4(dy + mo * a.out_ld + a.out_coff + c);
          u = *reinterpret_cast<const unsigned*>(idx + mo * a.C + c);
        }
        const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) {
          // output plane `to` reaches input plane to + kt - 1, kept in acc[kt]
          const unsigned tap = (unsigned)((kt * 3 + kh) * 3 + kw);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (((u >> (8 * q)) & 0xffu) == tap) acc[kt][q] += gv[q];
        }
      }
    }
    if (to >= 1) store_plane(to - 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[0][q] = acc[1][q];
      acc[1][q] = acc[2][q];
      acc[2][q] = 0.f;
    }
  }
  store_plane(a.Ti - 1);
}

static bool pool_is_3s1(const PoolArgs& a) {
  return a.kT == 3 && a.kH == 3 && a.kW == 3 && a.sT == 1 && a.sH == 1 && a.sW == 1 && a.pT == 1 && a.pH == 1 &&
         a.pW == 1 && a.To == a.Ti && a.Ho == a.Hi && a.Wo == a.Wi;
}
// channel groups (of 4) per workgroup: ~224 threads, at least a 128-byte run per position
static int pool_3s1_groups(const PoolArgs& a) {
  int G = 8;
  while (G < 64 && a.Wi * G * 2 <= 256 && G * 4 < a.C) G *= 2;
  while (G > 1 && a.Wi * G > 256) G /= 2;
  return G;
}

// tile choice: ~128-256 tile cells, staged region <= ~450 cells (<= 64 KB of LDS)
static void pool_fwd_tile(const PoolArgs& a, PoolTile* t) {
  t->tT = a.kT == 1 ? 1 : (a.sT == 1 ? 2 : 2);
  t->tH = a.sH == 1 ? 8 : 8;
  t->tW = a.sW == 1 ? 8 : 8;
  if (a.sH == 2 && a.kT > 1) { t->tH = 4; t->tW = 4; }
  t->rT = (t->tT - 1) * a.sT + a.kT;
  t->rH = (t->tH - 1) * a.sH + a.kH;
  t->rW = (t->tW - 1) * a.sW + a.kW;
  t->nT = cdiv(a.To, t->tT); t->nH = cdiv(a.Ho, t->tH); t->nW = cdiv(a.Wo, t->tW);
  t->slabs = cdiv(a.C, POOL_SLAB);
}
static void pool_bwd_tile(const PoolArgs& a, PoolTile* t) {
  t->tT = a.kT == 1 ? 1 : (a.sT == 1 ? 2 : 4);
  t->tH = a.sH == 1 ? 8 : 16;
  t->tW = a.sW == 1 ? 8 : 16;
  if (a.sH == 2 && a.kT > 1) { t->tH = 8; t->tW = 8; }
  auto span = [](int tile, int k, int s) { return (tile + k - 2) / s + 1; };   // max #outputs covering `tile` inputs
  t->rT = span(t->tT, a.kT, a.sT); t->rH = span(t->tH, a.kH, a.sH); t->rW = span(t->tW, a.kW, a.sW);
  t->nT = cdiv(a.Ti, t->tT); t->nH = cdiv(a.Hi, t->tH); t->nW = cdiv(a.Wi, t->tW);
  t->slabs = cdiv(a.C, POOL_SLAB);
}

// ---------------------------------------------------------------- head
// One block per clip.  pooled[c] = mean over the npos feature cells;
// logits[k] = bias[k] + sum_c pooled[c] W[k][c]; probs = softmax(logits).
template <class T>
__global__ __launch_bounds__(1024) void head_fwd_kernel(
    const T* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ pooled_out, float* __restrict__ logits, float* __restrict__ probs, int npos,
    int C, int K, int softmax) {
  extern __shared__ float sm[];  // C pooled + K logits + 8 scratch
  float* pooled = sm;
  float* lg = sm + C;
  float* red = lg + K;
  const int b = blockIdx.x;
  const T* f = feat + (size_t)b * npos * C;
  const float inv = 1.f / (float)npos;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float s = 0.f;
    for (int p = 0; p < npos; ++p) s += ld1(f + (size_t)p * C + c);
    pooled[c] = s * inv;
    if (pooled_out) pooled_out[(size_t)b * C + c] = s * inv;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int k = wave; k < K; k += nw) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += pooled[c] * w[(size_t)k * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) {
      s += bias ? bias[k] : 0.f;
      lg[k] = s;
      logits[(size_t)b * K + k] = s;
    }
  }
  __syncthreads();
  if (!probs) return;
  if (!softmax) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) probs[(size_t)b * K + k] = lg[k];
    return;
  }
  if (threadIdx.x == 0) {
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(lg[k] - mx);
    red[0] = mx;
    red[1] = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    probs[(size_t)b * K + k] = expf(lg[k] - red[0]) / red[1];
}

// backward to the feature map from an upstream gradient on the head output:
// dout[b,:] if given, else one-hot at target[b] (score[b] = out[b,target[b]]).
// softmax: dlogit[k] = p_k (dout_k - sum_j p_j dout_j); else dlogit = dout.
// dpooled[c] = sum_k dlogit[k] W[k][c]; dfeat[b,pos,c] = dpooled[c] / npos, optionally
// gated by (feat > 0) for the ReLU below.
template <class T>
__global__ __launch_bounds__(256) void head_bwd_kernel(
    const T* __restrict__ feat, const float* __restrict__ w, const float* __restrict__ probs,
    const int* __restrict__ target, const float* __restrict__ dout, float* __restrict__ score,
    float* __restrict__ dpooled_out, T* __restrict__ dfeat, int npos, int C, int K, int softmax,
    int gate, int cper) {
  // grid (clip, channel slice of `cper` channels): every block recomputes the K logit gradients of its clip (a few
  // hundred flops) and finishes its own channels -- one block per clip left 3/4 of the chip idle
  extern __shared__ float sm[];
  float* dl = sm;        // K
  float* dp = sm + K;    // cper
  __shared__ float dot;
  const int b = blockIdx.x;
  const int c_lo = blockIdx.y * cper, c_hi = min(C, c_lo + cper);
  const int t = target ? target[b] : -1;
  const float* pr = probs + (size_t)b * K;
  if (threadIdx.x == 0) {
    if (score && target && blockIdx.y == 0) score[b] = pr[t];
    float s = 0.f;
    if (softmax) {
      if (dout) {
        for (int k = 0; k < K; ++k) s += pr[k] * dout[(size_t)b * K + k];
      } else {
        s = pr[t];
      }
    }
    dot = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float d = dout ? dout[(size_t)b * K + k] : ((k == t) ? 1.f : 0.f);
    dl[k] = softmax ? pr[k] * (d - dot) : d;
  }
  __syncthreads();
  const float inv = 1.f / (float)npos;
  for (int c = c_lo + threadIdx.x; c < c_hi; c += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += dl[k] * w[(size_t)k * C + c];
    dp[c - c_lo] = s * inv;
    if (dpooled_out) dpooled_out[(size_t)b * C + c] = s;
  }
  __syncthreads();
  if (!dfeat) return;
  const T* f = feat + (size_t)b * npos * C;
  T* df = dfeat + (size_t)b * npos * C;
  const int cw = c_hi - c_lo;
  for (int i = threadIdx.x; i < npos * cw; i += blockDim.x) {
    const int p = i / cw, cc = i - p * cw;
    const size_t e = (size_t)p * C + c_lo + cc;
    float v = dp[cc];
    if (gate && !(ld1(f + e) > 0.f)) v = 0.f;
    st1(df + e, v);
  }
}

__global__ void argmax_kernel(const float* __restrict__ probs, int b, int K, int* __restrict__ target) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= b) return;
  const float* p = probs + (size_t)i * K;
  int best = 0;
  float bv = p[0];
  for (int k = 1; k < K; ++k)
    if (p[k] > bv) { bv = p[k]; best = k; }   // first maximum, as np.argmax
  target[i] = best;
}

// ---------------------------------------------------------------- Grad-CAM
// weights[b,k] = mean over positions of grad[b,pos,k]     (grad_cam_videos.py:98)
template <class T>
__global__ void gradcam_weights_kernel(const T* __restrict__ grad, float* __restrict__ wts, int npos,
                                       int C, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  int b = i / C, c = i % C;
  float s = 0.f;
  for (int p = 0; p < npos; ++p) s += ld1(grad + ((size_t)b * npos + p) * C + c);
  wts[i] = s / (float)npos;
}

// cam[b,pos] = max(sum_k w[b,k] feat[b,pos,k], 0)          (grad_cam_videos.py:101-110)
// one wave per position, lanes stride over channels
template <class T>
__global__ __launch_bounds__(256) void gradcam_cam_kernel(const T* __restrict__ feat,
                                                          const float* __restrict__ wts,
                                                          float* __restrict__ cam, int npos, int C,
                                                          int B) {
  int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (gw >= B * npos) return;
  int b = gw / npos;
  const T* f = feat + (size_t)gw * C;
  const float* w = wts + (size_t)b * C;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s += w[c] * ld1(f + c);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) cam[gw] = fmaxf(s, 0.f);
}

// OpenCV INTER_LINEAR sampling rule for float32 (cv2.resize call site
// grad_cam_videos.py:119): half-pixel centres, clamped source index.
__device__ __forceinline__ void lin_coeff(int d, float scale, int src, int* i0, int* i1, float* f) {
  float fx = ((float)d + 0.5f) * scale - 0.5f;
  int s = (int)floorf(fx);
  float fr = fx - (float)s;
  if (s < 0) { fr = 0.f; s = 0; }
  if (s >= src - 1) { fr = 0.f; s = src - 1; }
  *i0 = s;
  *i1 = min(s + 1, src - 1);
  *f = fr;
}

__device__ __forceinline__ float bilinear_at(const float* src, int sh, int sw, int y, int x, float sy,
                                             float sx) {
  int y0, y1, x0, x1;
  float fy, fx;
  lin_coeff(y, sy, sh, &y0, &y1, &fy);
  lin_coeff(x, sx, sw, &x0, &x1, &fx);
  float top = src[y0 * sw + x0] * (1.f - fx) + src[y0 * sw + x1] * fx;
  float bot = src[y1 * sw + x0] * (1.f - fx) + src[y1 * sw + x1] * fx;
  return top * (1.f - fy) + bot * fy;
}

// pass 1: per (b, slice) min / max of the resized map
__global__ __launch_bounds__(256) void cam_minmax_kernel(const float* __restrict__ cam,
                                                         float* __restrict__ mm, int sh, int sw, int H,
                                                         int W) {
  const float* src = cam + (size_t)blockIdx.x * sh * sw;
  float sy = (float)sh / (float)H, sx = (float)sw / (float)W;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
    float v = bilinear_at(src, sh, sw, i / W, i % W, sy, sx);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  __shared__ float smn[4], smx[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_down(mn, o, 64));
    mx = fmaxf(mx, __shfl_down(mx, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    mm[blockIdx.x * 2 + 0] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mm[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}

// pass 2: resize, subtract min, divide by max(x - min), repeat `step` frames
// (grad_cam_videos.py:121-138).  per_frame: statistics of the slice's own block,
// else of the whole clip.  0/0 -> NaN exactly as numpy does.
__global__ __launch_bounds__(256) void cam_resize_norm_kernel(const float* __restrict__ cam,
                                                              const float* __restrict__ mm,
                                                              float* __restrict__ out, int nslice, int sh,
                                                              int sw, int H, int W, int step,
                                                              int per_frame) {
  const int bs = blockIdx.x;  // (b, slice)
  const int b = bs / nslice, sl = bs % nslice;
  const float* src = cam + (size_t)bs * sh * sw;
  float mn, mx;
  if (per_frame) {
    mn = mm[bs * 2];
    mx = mm[bs * 2 + 1];
  } else {
    mn = INFINITY;
    mx = -INFINITY;
    for (int i = 0; i < nslice; ++i) {
      mn = fminf(mn, mm[(b * nslice + i) * 2]);
      mx = fmaxf(mx, mm[(b * nslice + i) * 2 + 1]);
    }
  }
  const float den = mx - mn;
  float sy = (float)sh / (float)H, sx = (float)sw / (float)W;
  float* dst = out + ((size_t)b * nslice + sl) * step * H * W;
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
    float v = (bilinear_at(src, sh, sw, i / W, i % W, sy, sx) - mn) / den;
    for (int r = 0; r < step; ++r) dst[(size_t)r * H * W + i] = v;
  }
}

static inline int grid_for(size_t total, int block = 256, int cap = 4096) {
  size_t g = (total + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

static int check_pool(const ivf_pool3d_desc* d) {
  IVF_CHECK_ARG(d, "pool: null descriptor");
  IVF_CHECK_ARG(d->B > 0 && d->Ti > 0 && d->Hi > 0 && d->Wi > 0 && d->To > 0 && d->Ho > 0 && d->Wo > 0,
                "pool: bad dims");
  IVF_CHECK_ARG(d->C > 0 && d->C % 4 == 0 && d->in_ld % 4 == 0 && d->in_coff % 4 == 0 &&
                    d->out_ld % 4 == 0 && d->out_coff % 4 == 0,
                "pool: channel counts/offsets must be multiples of 4");
  IVF_CHECK_ARG(d->in_coff + d->C <= d->in_ld && d->out_coff + d->C <= d->out_ld,
                "pool: channel window outside ld");
  IVF_CHECK_ARG(d->kT * d->kH * d->kW <= 255, "pool: window too large for uint8 arg-max");
  return IVF_OK;
}

static PoolArgs to_args(const ivf_pool3d_desc* d) {
  PoolArgs a;
  a.B = d->B; a.Ti = d->Ti; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.in_ld = d->in_ld;
  a.in_coff = d->in_coff; a.To = d->To; a.Ho = d->Ho; a.Wo = d->Wo; a.out_ld = d->out_ld;
  a.out_coff = d->out_coff; a.kT = d->kT; a.kH = d->kH; a.kW = d->kW; a.sT = d->sT; a.sH = d->sH;
  a.sW = d->sW; a.pT = d->pT; a.pH = d->pH; a.pW = d->pW; a.dead = d->gate_nonpos;
  return a;
}

}  // namespace ivf

using namespace ivf;

template <class T>
static int pool_fwd_impl(const PoolArgs& a, const T* x, T* y, unsigned char* argmax, hipStream_t stream) {
  static const bool direct = getenv("IVF_POOL_DIRECT") != nullptr;   // A/B switch for measurements
  PoolTile t;
  pool_fwd_tile(a, &t);
  size_t shm = (size_t)t.rT * t.rH * t.rW * POOL_ROW * sizeof(float);
  // measured on MI355X (16-channel slabs): the tiled form wins 1.4x for the stride-1 3x3x3
  // Inception pools (the direct kernel saturates L2 with its 27x re-reads) and loses for the
  // strided pools, whose windows barely overlap
  static const bool no_s1 = getenv("IVF_POOL_NO_S1") != nullptr;
  if (!direct && !no_s1 && pool_is_3s1(a) && a.Wi <= 256) {
    const int G = pool_3s1_groups(a), nH = cdiv(a.Hi, S1_TH), slabs = cdiv(a.C, 4 * G);
    const int threads = ((a.Wi * G + 63) / 64) * 64;
    hipLaunchKernelGGL((maxpool3s1_fwd_kernel<T>), dim3((unsigned)(a.B * nH * slabs)), dim3(threads), 0, stream, x, y,
                       argmax, a, G, nH, slabs);
    IVF_CHECK_LAUNCH();
    return IVF_OK;
  }
  const bool tiled_fwd = a.sT == 1 && a.sH == 1 && a.sW == 1;
  if (!direct && tiled_fwd && shm <= 64 * 1024) {
    long blocks = (long)a.B * t.nT * t.nH * t.nW * t.slabs;
    hipLaunchKernelGGL((maxpool_fwd_tiled_kernel<T>), dim3((unsigned)blocks), dim3(256), shm, stream, x, y, argmax, a, t);
    IVF_CHECK_LAUNCH();
    return IVF_OK;
  }
  // the strided pools of the I3D variants with their windows known at compile time
  static const bool no_fixed = getenv("IVF_POOL_NO_FIXED") != nullptr;
  if (!direct && !no_fixed && (a.sT > 1 || a.sH > 1 || a.sW > 1)) {
    if (launch_pool_fwd_fixed<T, 1, 3, 3>(a, x, y, argmax, stream) || launch_pool_fwd_fixed<T, 3, 3, 3>(a, x, y, argmax, stream) ||
        launch_pool_fwd_fixed<T, 2, 2, 2>(a, x, y, argmax, stream)) {
      IVF_CHECK_LAUNCH();
      return IVF_OK;
    }
  }
  const long nbx = cdiv((long)a.Wo * (a.C / 4), 256), nblk = nbx * a.Ho * a.B * a.To;
  IVF_CHECK_ARG(nblk < 0x7fffffffL, "maxpool_fwd: more than 2^31 workgroups");
  hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3((unsigned)nblk), dim3(256), 0, stream, x, y, argmax, a, (int)nbx);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_maxpool3d_fwd(const ivf_pool3d_desc* d, const void* x, void* y, unsigned char* argmax,
                                 ivf_stream_t stream) {
  IVF_PROPAGATE(check_pool(d));
  IVF_CHECK_ARG(x && y, "maxpool_fwd: null pointer");
  PoolArgs a = to_args(d);
  if (d->act_bf16) return pool_fwd_impl<bf16s>(a, (const bf16s*)x, (bf16s*)y, argmax, (hipStream_t)stream);
  return pool_fwd_impl<float>(a, (const float*)x, (float*)y, argmax, (hipStream_t)stream);
}

template <class T>
static int pool_bwd_impl(const PoolArgs& a, const T* dy, const unsigned char* argmax, T* dx, const T* relu_mask,
                         int accumulate, hipStream_t hs) {
  static const bool direct = getenv("IVF_POOL_DIRECT") != nullptr;
  PoolTile t;
  pool_bwd_tile(a, &t);
  size_t shm = (size_t)t.rT * t.rH * t.rW * (POOL_ROW * sizeof(float) + POOL_SLAB);
  static const bool no_s1 = getenv("IVF_POOL_NO_S1") != nullptr;
  if (!direct && !no_s1 && pool_is_3s1(a) && a.Wi <= 256) {
    const int G = pool_3s1_groups(a), slabs = cdiv(a.C, 4 * G);
    const int threads = ((a.Wi * G + 63) / 64) * 64;
    hipLaunchKernelGGL((maxpool3s1_bwd_kernel<T>), dim3((unsigned)(a.B * a.Hi * slabs)), dim3(threads), 0, hs, dy, argmax, dx,
                       relu_mask, accumulate, a, G, slabs);
    IVF_CHECK_LAUNCH();
    return IVF_OK;
  }
  // the strided pools of the I3D variants (I3D_doubled.py:272-300; temporal strides 1 or 2)
  static const bool no_fixed = getenv("IVF_POOL_NO_FIXED") != nullptr;
  if (!direct && !no_fixed && (size_t)a.B * a.Ti * a.Hi * a.Wi * (a.C / 4) / 256 < 0x7fffffffu) {
    if constexpr (sizeof(T) == 2) {   // 16-byte lanes, raw loads first
      if (launch_pool_bwd_fixed16<T, 1, 3, 3, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
          launch_pool_bwd_fixed16<T, 3, 3, 3, 2, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
          launch_pool_bwd_fixed16<T, 3, 3, 3, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
          launch_pool_bwd_fixed16<T, 2, 2, 2, 2, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
          launch_pool_bwd_fixed16<T, 2, 2, 2, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs)) {
        IVF_CHECK_LAUNCH();
        return IVF_OK;
      }
    }
    if (launch_pool_bwd_fixed<T, 1, 3, 3, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
        launch_pool_bwd_fixed<T, 3, 3, 3, 2, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
        launch_pool_bwd_fixed<T, 3, 3, 3, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
        launch_pool_bwd_fixed<T, 2, 2, 2, 2, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs) ||
        launch_pool_bwd_fixed<T, 2, 2, 2, 1, 2, 2>(a, dy, argmax, dx, relu_mask, accumulate, hs)) {
      IVF_CHECK_LAUNCH();
      return IVF_OK;
    }
  }
  // measured (16-channel slabs): the tiled gather wins 1.3-1.6x for every pool of the net
  if (!direct && shm <= 80 * 1024) {
    static LdsAttrOnce once;
    IVF_PROPAGATE(raise_lds_limit(reinterpret_cast<const void*>(&maxpool_bwd_tiled_kernel<T>), 80 * 1024, once));
    long blocks = (long)a.B * t.nT * t.nH * t.nW * t.slabs;
    hipLaunchKernelGGL((maxpool_bwd_tiled_kernel<T>), dim3((unsigned)blocks), dim3(256), shm, hs, dy, argmax, dx, relu_mask,
                       accumulate, a, t, 0, 0, 0);
    IVF_CHECK_LAUNCH();
    return IVF_OK;
  }
  size_t total = (size_t)a.B * a.Ti * a.Hi * a.Wi * (a.C / 4);
  hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(total)), dim3(256), 0, hs, dy, argmax, dx, relu_mask, accumulate, a);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_maxpool3d_bwd(const ivf_pool3d_desc* d, const void* dy, const unsigned char* argmax, void* dx,
                                 const void* relu_mask, int accumulate, ivf_stream_t stream) {
  IVF_PROPAGATE(check_pool(d));
  IVF_CHECK_ARG(dy && argmax && dx, "maxpool_bwd: null pointer");
  PoolArgs a = to_args(d);
  if (d->act_bf16)
    return pool_bwd_impl<bf16s>(a, (const bf16s*)dy, argmax, (bf16s*)dx, (const bf16s*)relu_mask, accumulate, (hipStream_t)stream);
  return pool_bwd_impl<float>(a, (const float*)dy, argmax, (float*)dx, (const float*)relu_mask, accumulate, (hipStream_t)stream);
}

template <class T>
static int head_fwd_impl(const T* feat, const float* w, const float* bias, float* pooled, float* logits, float* probs, int B,
                         int npos, int C, int K, int softmax, ivf_stream_t stream) {
  IVF_CHECK_ARG(feat && w && logits, "head_fwd: null pointer");
  IVF_CHECK_ARG(B > 0 && npos > 0 && C > 0 && K > 0 && (size_t)(C + K + 8) * 4 <= 64 * 1024,
                "head_fwd: bad dims");
  size_t shm = (size_t)(C + K + 8) * sizeof(float);
  hipLaunchKernelGGL((head_fwd_kernel<T>), dim3(B), dim3(1024), shm, (hipStream_t)stream, feat, w, bias, pooled,
                     logits, probs, npos, C, K, softmax);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_head_fwd(const float* feat, const float* w, const float* bias, float* pooled,
                            float* logits, float* probs, int B, int npos, int C, int K, int softmax,
                            ivf_stream_t stream) {
  return head_fwd_impl<float>(feat, w, bias, pooled, logits, probs, B, npos, C, K, softmax, stream);
}
extern "C" int ivf_head_fwd_bf16(const void* feat, const float* w, const float* bias, float* pooled,
                                 float* logits, float* probs, int B, int npos, int C, int K, int softmax,
                                 ivf_stream_t stream) {
  return head_fwd_impl<bf16s>((const bf16s*)feat, w, bias, pooled, logits, probs, B, npos, C, K, softmax, stream);
}

extern "C" int ivf_argmax(const float* probs, int b, int K, int* target, ivf_stream_t stream) {
  IVF_CHECK_ARG(probs && target && b > 0 && K > 0, "argmax: bad args");
  hipLaunchKernelGGL(argmax_kernel, dim3(cdiv(b, 64)), dim3(64), 0, (hipStream_t)stream, probs, b, K, target);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

template <class T>
static int head_bwd_impl(const T* feat, const float* w, const float* probs, const int* target, const float* dout,
                         float* score, float* dpooled, T* dfeat, int B, int npos, int C, int K, int softmax, int gate_relu,
                         ivf_stream_t stream) {
  IVF_CHECK_ARG(w && probs && (target || dout), "head_bwd: need probs, w and target or dout");
  IVF_CHECK_ARG(!dfeat || feat, "head_bwd: feat required with dfeat");
  IVF_CHECK_ARG(B > 0 && npos > 0 && C > 0 && K > 0 && (size_t)(C + K) * 4 <= 64 * 1024,
                "head_bwd: bad dims");
  // channel slices of 128 (at least 4 blocks per CU's worth of clips at small batch)
  const int cper = C >= 256 ? 128 : C;
  size_t shm = (size_t)(cper + K) * sizeof(float);
  hipLaunchKernelGGL((head_bwd_kernel<T>), dim3(B, cdiv(C, cper)), dim3(256), shm, (hipStream_t)stream, feat, w, probs,
                     target, dout, score, dpooled, dfeat, npos, C, K, softmax, gate_relu, cper);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_head_bwd(const float* feat, const float* w, const float* probs, const int* target,
                            const float* dout, float* score, float* dpooled, float* dfeat, int B,
                            int npos, int C, int K, int softmax, int gate_relu, ivf_stream_t stream) {
  return head_bwd_impl<float>(feat, w, probs, target, dout, score, dpooled, dfeat, B, npos, C, K, softmax, gate_relu, stream);
}
extern "C" int ivf_head_bwd_bf16(const void* feat, const float* w, const float* probs, const int* target,
                                 const float* dout, float* score, float* dpooled, void* dfeat, int B,
                                 int npos, int C, int K, int softmax, int gate_relu, ivf_stream_t stream) {
  return head_bwd_impl<bf16s>((const bf16s*)feat, w, probs, target, dout, score, dpooled, (bf16s*)dfeat, B, npos, C, K,
                              softmax, gate_relu, stream);
}

template <class T>
static int gradcam_reduce_impl(const T* feat, const T* grad, float* weights, float* cam, int B, int npos, int C,
                               ivf_stream_t stream) {
  IVF_CHECK_ARG(feat && grad && weights && cam && B > 0 && npos > 0 && C > 0, "gradcam_reduce: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((gradcam_weights_kernel<T>), dim3(cdiv(B * C, 256)), dim3(256), 0, s, grad, weights, npos,
                     C, B);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL((gradcam_cam_kernel<T>), dim3(cdiv((size_t)B * npos * 64, 256)), dim3(256), 0, s, feat,
                     weights, cam, npos, C, B);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_gradcam_reduce(const float* feat, const float* grad, float* weights, float* cam,
                                  int B, int npos, int C, ivf_stream_t stream) {
  return gradcam_reduce_impl<float>(feat, grad, weights, cam, B, npos, C, stream);
}
extern "C" int ivf_gradcam_reduce_bf16(const void* feat, const void* grad, float* weights, float* cam,
                                       int B, int npos, int C, ivf_stream_t stream) {
  return gradcam_reduce_impl<bf16s>((const bf16s*)feat, (const bf16s*)grad, weights, cam, B, npos, C, stream);
}

extern "C" int ivf_cam_resize_normalise(const float* cam, float* out, float* minmax_ws, int B, int nslice,
                                        int sh, int sw, int H, int W, int step, int per_frame,
                                        ivf_stream_t stream) {
  IVF_CHECK_ARG(cam && out && minmax_ws, "cam_resize: null pointer");
  IVF_CHECK_ARG(B > 0 && nslice > 0 && sh > 0 && sw > 0 && H > 0 && W > 0 && step > 0, "cam_resize: bad dims");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cam_minmax_kernel, dim3(B * nslice), dim3(256), 0, s, cam, minmax_ws, sh, sw, H, W);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(cam_resize_norm_kernel, dim3(B * nslice), dim3(256), 0, s, cam, minmax_ws, out,
                     nslice, sh, sw, H, W, step, per_frame);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}
