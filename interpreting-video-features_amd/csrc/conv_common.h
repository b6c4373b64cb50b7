// Shared pieces of the implicit-GEMM convolution kernels (conv3d.hip, conv3d_halo.hip).
#pragma once
#include "ivf_common.h"

namespace ivf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

struct ConvKArgs {
  const float* in;
  const float* w;
  float* out;
  const float* scale;  // [Cout] or null
  const float* shift;  // [Cout] or null
  const float* mask;   // ReLU mask source (post-ReLU activation), or null
  int B, Ti, Hi, Wi, Cin, in_ld, in_coff;
  int To, Ho, Wo, Cout, out_ld, out_coff;
  int mask_ld, mask_coff;
  int kT, kH, kW, sT, sH, sW, pT, pH, pW;
  int K, M;
  int relu, accumulate, d2s;
  // depth-to-space output (stride-2 backward-data as a stride-1 conv over 2x2x2
  // output blocks): real output dims
  int dT, dH, dW, dC;
  int bsT, bsH, bsW;  // block strides (the forward conv's strides, 1 or 2)
  int mtiles, ntiles;
  // split-bf16 modes: weights as bf16 planes [rows][ldw] (hi, then lo at +w_lo_off, then -- 3-way split -- the
  // last 8 bits at +2*w_lo_off)
  const unsigned short* wbf;
  int ldw;
  long w_lo_off;
  // second input for 1x1x1 convs: channels [K0, Cin) of the GEMM K dimension come from here
  // (same positions, own row length / channel offset) -- lets one backward GEMM consume the
  // gradients of several branches that live in different buffers
  const float* in2;
  int in2_ld, in2_coff, K0;
  // second output window: columns [N0, Cout) go to out2 (forward epilogues of the implicit GEMM only)
  float* out2;
  int out2_ld, out2_coff, N0;
  // 1-bit ReLU gates (ivf_conv3d_desc.gate_*): written by a forward epilogue, read by a backward one
  unsigned char* gbo;        // record for the `out` window (row = gbo_ld bytes, first channel gbo_coff)
  unsigned char* gbo2;       // record for the `out2` window (first channel 0 of its buffer + out2_coff)
  const unsigned char* gbi;  // gate of the gradient being written (instead of `mask`)
  int gbo_ld, gbo_coff, gbo2_ld, gbi_ld, gbi_coff;
};


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int LDS_ROW_BF = BK + 8;  // bf16 per LDS row: 80 B rows make the ds_read_b128 fragment reads conflict-free

__device__ __forceinline__ unsigned pk_bf16(float lo_elem, float hi_elem) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo_elem), "v"(hi_elem));
  return r;
}

__device__ __forceinline__ void split4(const float4& v, uint2* hi, uint2* lo) {
  unsigned h01 = pk_bf16(v.x, v.y), h23 = pk_bf16(v.z, v.w);
  float hx = __uint_as_float(h01 << 16), hy = __uint_as_float(h01 & 0xffff0000u);
  float hz = __uint_as_float(h23 << 16), hw = __uint_as_float(h23 & 0xffff0000u);
  *hi = make_uint2(h01, h23);
  *lo = make_uint2(pk_bf16(v.x - hx, v.y - hy), pk_bf16(v.z - hz, v.w - hw));
}

// 3-way split x = hi + mid + lo, each 8 significant bits (bf16, RNE): both residuals are exact in fp32 and the last
// one fits a bf16, so the three planes carry all 24 bits of an fp32 operand.
__device__ __forceinline__ void split4x3(const float4& v, uint2* hi, uint2* mid, uint2* lo) {
  unsigned h01 = pk_bf16(v.x, v.y), h23 = pk_bf16(v.z, v.w);
  const float rx = v.x - __uint_as_float(h01 << 16), ry = v.y - __uint_as_float(h01 & 0xffff0000u);
  const float rz = v.z - __uint_as_float(h23 << 16), rw = v.w - __uint_as_float(h23 & 0xffff0000u);
  unsigned m01 = pk_bf16(rx, ry), m23 = pk_bf16(rz, rw);
  *hi = make_uint2(h01, h23);
  *mid = make_uint2(m01, m23);
  *lo = make_uint2(pk_bf16(rx - __uint_as_float(m01 << 16), ry - __uint_as_float(m01 & 0xffff0000u)),
                   pk_bf16(rz - __uint_as_float(m23 << 16), rw - __uint_as_float(m23 & 0xffff0000u)));
}

// Operand modes of the bf16-MFMA kernels (template parameter AM):
//  AM_X3   activations fp32 in HBM, split hi/lo while staged; weights hi/lo: 3 MFMAs per k-step (lo*hi + hi*lo + hi*hi)
//  AM_BF16 activations (and what the epilogue stores) bf16 in HBM, ONE plane staged as it is; weights hi/lo:
//          2 MFMAs per k-step (a*lo + a*hi) -- the product a*w is as exact as in AM_X3, only storage rounds
//  AM_X6   activations fp32, split hi/mid/lo; weights hi/mid/lo: 6 MFMAs per k-step, every term down to 2^-16 of
//          the product (lh hl mm mh hm hh): 24-bit operands, i.e. an fp32 product to ~2^-23
enum { AM_X3 = 0, AM_BF16 = 1, AM_X6 = 2 };
template <int AM> struct OpPlanes {
  static constexpr int A = AM == AM_X6 ? 3 : (AM == AM_BF16 ? 1 : 2);
  static constexpr int B = AM == AM_X6 ? 3 : 2;
};
// acc += A * B over the planes of one 16-deep k-step, smallest terms first
template <int AM>
__device__ __forceinline__ void mma_planes(const bf16x8* fa, const bf16x8* fb, f32x16& acc) {
  if constexpr (AM == AM_X3) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
  } else if constexpr (AM == AM_BF16) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc, 0, 0, 0);
  }
}
// stage 4 channels into the A planes of one LDS row piece (8 bytes per plane, planes `pstride` bytes apart)
template <int AM>
__device__ __forceinline__ void stage_planes(unsigned char* dst, size_t pstride, const float4& v) {
  if constexpr (AM == AM_X6) {
    uint2 h, m, l;
    split4x3(v, &h, &m, &l);
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + pstride) = m;
    *reinterpret_cast<uint2*>(dst + 2 * pstride) = l;
  } else if constexpr (AM == AM_X3) {
    uint2 h, l;
    split4(v, &h, &l);
    *reinterpret_cast<uint2*>(dst) = h;
    *reinterpret_cast<uint2*>(dst + pstride) = l;
  } else {
    *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));   // (4 bf16 carried in v.x, v.y)
  }
}
// 4 channels of an activation row as the kernels stage them: a float4 of fp32 values, or (AM_BF16) the 8 bytes of 4
// bf16 values carried bit-for-bit in .x / .y
template <int AM>
__device__ __forceinline__ float4 load_act4(const float* base, size_t elem_off) {
  if constexpr (AM == AM_BF16) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem_off);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
  } else {
    return *reinterpret_cast<const float4*>(base + elem_off);
  }
}
// bf16 storage helpers of the AM_BF16 epilogues
__device__ __forceinline__ float4 bf16x4_to_f32(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 f32x4_to_bf16(float a, float b, float c, float d) { return make_uint2(pk_bf16(a, b), pk_bf16(c, d)); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16(float v) { return (unsigned short)(pk_bf16(v, 0.f) & 0xffffu); }

// Staging writes put 8 (A, 8-byte pieces) or 4 (weights, 16-byte pieces) lanes on one 80-byte LDS
// row; with consecutive rows on consecutive lane groups the rows of one LDS write group sit
// 80 B apart and overlap in banks (2-way: ~30 % of all LDS cycles of these kernels were
// conflict cycles).  Handing lane group i the row perm8(i) puts rows 4 apart (320 B = 64 mod
// 128) into each write group instead: disjoint banks.  Bijective on every aligned block of 8.
__device__ __forceinline__ int perm8(int r) { return (r & ~7) | ((r & 1) << 2) | ((r >> 1) & 3); }

// 4x4 transpose inside a lane quad: lane q holds v[k] = M[q][k] on entry and M[k][q] on exit (two DPP quad
// permutes per register pair).  All four lanes of the quad must be active.
__device__ __forceinline__ void quad_transpose4(float (&v)[4], int q) {
  // stage 1: exchange across lane bit 0 (registers k <-> k^1), stage 2: across lane bit 1 (k <-> k^2)
#pragma unroll
  for (int k = 0; k < 4; k += 2) {
    const float send = (q & 1) ? v[k] : v[k + 1];
    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xf, 0xf, false));
    if (q & 1) v[k] = recv; else v[k + 1] = recv;
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float send = (q & 2) ? v[k] : v[k + 2];
    const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xf, 0xf, false));
    if (q & 2) v[k] = recv; else v[k + 2] = recv;
  }
}

// Epilogue shared by both arithmetic variants.  Lane holds column n = li of each 32x32 tile,
// rows (r&3) + 8*(r>>2) + 4*lh.  Per tile all old-value / gate loads are issued before any
// store (the accumulate path reads and writes the same buffer, which would otherwise
// serialise every load behind the previous store).
// `rowmap(local_row)` returns the flattened global output position of a tile row, or -1.
// element accessors of the epilogues: fp32, or bf16 storage (offsets in ELEMENTS either way)
template <bool OB> __device__ __forceinline__ float4 ep_ld4(const float* base, size_t off) {
  if constexpr (OB) return bf16x4_to_f32(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
  else return *reinterpret_cast<const float4*>(base + off);
}
// the same load as raw bits (bf16 storage: 8 bytes in .x / .y) and its conversion: the epilogue requests the raw bits of a
// whole row tile first and converts at use -- a conversion beside the load, inside the `ok ? load : 0` branch, makes the
// compiler wait for every load in turn
template <bool OB> __device__ __forceinline__ float4 ep_ld4_raw(const float* base, size_t off) {
  if constexpr (OB) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
  } else {
    return *reinterpret_cast<const float4*>(base + off);
  }
}
template <bool OB> __device__ __forceinline__ float4 ep_cvt4(const float4& r) {
  if constexpr (OB) return bf16x4_to_f32(make_uint2(__float_as_uint(r.x), __float_as_uint(r.y)));
  else return r;
}
template <bool OB> __device__ __forceinline__ void ep_st4(float* base, size_t off, float a, float b, float c, float d) {
  if constexpr (OB) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = f32x4_to_bf16(a, b, c, d);
  else *reinterpret_cast<float4*>(base + off) = make_float4(a, b, c, d);
}
template <bool OB> __device__ __forceinline__ float ep_ld1(const float* base, size_t off) {
  if constexpr (OB) return bf16_to_f32(reinterpret_cast<const unsigned short*>(base)[off]);
  else return base[off];
}
template <bool OB> __device__ __forceinline__ void ep_st1(float* base, size_t off, float v) {
  if constexpr (OB) reinterpret_cast<unsigned short*>(base)[off] = f32_to_bf16(v);
  else base[off] = v;
}

template <int AM, int TM, int TN, class RowMap>
__device__ __forceinline__ void conv_epilogue(const ConvKArgs& a, f32x16 (&acc)[TM][TN], RowMap rowmap,
                                              int row_base, int col_base, int li, int lh) {
  constexpr bool OB = AM == AM_BF16;   // bf16 storage of out / out2 / relu_mask (the depth-to-space form always writes fp32)
  if (a.d2s) {
    // depth-to-space: n = ((pt*2+ph)*2+pw)*cpad + c with cpad == 4: the 4 channels of one
    // output pixel sit on 4 consecutive lanes; gather them with quad DPP moves and let the
    // c == 0 lane store one 16-byte vector.
    // (Cout = 8 * cpad <= 32 here -- the host checks it -- so only the first 32-column tile of a
    // wave can hold real columns; handling just j = 0 keeps this branch small enough to unroll.)
    const int cpad = a.Cout >> 3;
    {
      constexpr int j = 0;
      const int n = col_base + li;
      const int par = n / cpad, c = n - par * cpad;
      const int pt = par >> 2, ph = (par >> 1) & 1, pw = par & 1;
      const bool nvalid = n < a.Cout && pt < a.bsT && ph < a.bsH && pw < a.bsW;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = rowmap(row_base + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh);
          float v = acc[i][j][r];
          if (cpad == 4) {
            int vi = __float_as_int(v);
            float v0 = __int_as_float(__builtin_amdgcn_update_dpp(0, vi, 0x00, 0xf, 0xf, false));
            float v1 = __int_as_float(__builtin_amdgcn_update_dpp(0, vi, 0x55, 0xf, 0xf, false));
            float v2 = __int_as_float(__builtin_amdgcn_update_dpp(0, vi, 0xaa, 0xf, 0xf, false));
            float v3 = __int_as_float(__builtin_amdgcn_update_dpp(0, vi, 0xff, 0xf, 0xf, false));
            if (!nvalid || c != 0 || m < 0) continue;
            int wb = m % a.Wo;
            int t1 = m / a.Wo;
            int hb = t1 % a.Ho;
            int t2 = t1 / a.Ho;
            int tb = t2 % a.To;
            int b = t2 / a.To;
            int t = a.bsT * tb + pt, h = a.bsH * hb + ph, w = a.bsW * wb + pw;
            if (t >= a.dT || h >= a.dH || w >= a.dW) continue;
            size_t off = ((size_t)((b * a.dT + t) * a.dH + h) * a.dW + w) * a.out_ld + a.out_coff;
            float4 o = make_float4(v0, v1, v2, v3);
            if (a.accumulate) {
              float4 old = *reinterpret_cast<const float4*>(a.out + off);
              o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
            }
            if (a.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
            if (a.dC == 4 && (a.out_ld & 3) == 0 && (a.out_coff & 3) == 0) {
              *reinterpret_cast<float4*>(a.out + off) = o;
            } else {
              float ov[4] = {o.x, o.y, o.z, o.w};
              for (int q = 0; q < a.dC && q < 4; ++q) a.out[off + q] = ov[q];
            }
          } else {
            if (!nvalid || c >= a.dC || m < 0) continue;
            int wb = m % a.Wo;
            int t1 = m / a.Wo;
            int hb = t1 % a.Ho;
            int t2 = t1 / a.Ho;
            int tb = t2 % a.To;
            int b = t2 / a.To;
            int t = a.bsT * tb + pt, h = a.bsH * hb + ph, w = a.bsW * wb + pw;
            if (t >= a.dT || h >= a.dH || w >= a.dW) continue;
            size_t off = ((size_t)((b * a.dT + t) * a.dH + h) * a.dW + w) * a.out_ld + a.out_coff + c;
            if (a.accumulate) v += a.out[off];
            if (a.relu) v = v > 0.f ? v : 0.f;
            a.out[off] = v;
          }
        }
      }
    }
    return;
  }
  // 16-byte form: the 4 lanes of a quad hold 4 adjacent columns of the same 4 rows; a 4x4 transpose inside the quad
  // (DPP quad permutes) gives every lane ONE row and 4 adjacent columns, so a row group is stored by one
  // global_store_dwordx4 per lane instead of four dword stores (a wave instruction still covers whole 128-byte row
  // segments); the accumulate / gate reads become 16-byte loads the same way.  Same arithmetic per element.
  // (bf16 storage: the same with 8-byte accesses.)
  const bool vec_ok = (a.Cout & 3) == 0 && (a.out_ld & 3) == 0 && (a.out_coff & 3) == 0 && (col_base & 3) == 0 &&
                      (!a.mask || (((a.mask_ld | a.mask_coff) & 3) == 0)) &&
                      (!a.out2 || (((a.N0 | a.out2_ld | a.out2_coff) & 3) == 0));
  if (vec_ok) {
    const int q = li & 3;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = col_base + j * 32 + li;
      const int nq = n & ~3;                       // first column of this lane's quad (all or none of them valid)
      const bool nvalid = nq < a.Cout;
      const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
      const float sh = (a.shift && nvalid) ? a.shift[n] : 0.f;
      const bool second = a.out2 != nullptr && nq >= a.N0;
      float* const obase = second ? a.out2 : a.out;
      const size_t ocol = second ? (size_t)(a.out2_coff + (nq - a.N0)) : (size_t)(a.out_coff + nq);
      const size_t oldim = second ? (size_t)a.out2_ld : (size_t)a.out_ld;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int rbase = row_base + i * 32 + 4 * lh;
        float4 old4[4], gate4[4];
        unsigned gnib[4];     // 1-bit gates of this lane's 4 columns (all set when there is no bit record)
        int mrow[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = mrow[g] = rowmap(rbase + q + 8 * g);
          const bool ok = nvalid && m >= 0;
          // (raw bits; neutral elements in the raw domain: 0 = +0.0 either way, the gate's 1.0 is 0x3f80 per bf16)
          old4[g] = (a.accumulate && ok) ? ep_ld4_raw<OB>(obase, ocol + (size_t)m * oldim) : make_float4(0.f, 0.f, 0.f, 0.f);
          gate4[g] = (a.mask && ok) ? ep_ld4_raw<OB>(a.mask, (size_t)m * a.mask_ld + a.mask_coff + nq)
                                    : (OB ? make_float4(__uint_as_float(0x3f803f80u), __uint_as_float(0x3f803f80u), 0.f, 0.f)
                                          : make_float4(1.f, 1.f, 1.f, 1.f));
          // (the raw gate byte; its nibble is taken after all loads of the row tile are out: a value consumed inside
          // its bounds branch is waited for at once)
          unsigned gb = 0xffu;
          if (a.gbi && ok) gb = a.gbi[(size_t)m * a.gbi_ld + ((a.gbi_coff + nq) >> 3)];
          gnib[g] = gb;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) gnib[g] = (gnib[g] >> ((a.gbi_coff + nq) & 4)) & 15u;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = acc[i][j][4 * g + k] * sc + sh;
          quad_transpose4(v, q);     // v[k]: row rbase + q + 8g, column nq + k
          const int m = mrow[g];
          if (a.gbo || a.gbo2) {
            // forward record of (stored value > 0): this lane's 4 columns are a nibble, the lane 4 up (same row, next
            // 4 columns) supplies the other half of the byte; even quads write.  (No accumulate / gate here.)
            // (bf16 storage: a positive fp32 value never rounds to zero, so the record equals (stored value > 0).)
            unsigned nib = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float t = v[k];
              if (a.relu) t = t > 0.f ? t : 0.f;
              nib |= (t > 0.f ? 1u : 0u) << k;
            }
            const unsigned hi = (unsigned)__shfl_down((int)nib, 4, 64);
            if (nvalid && m >= 0 && ((li >> 2) & 1) == 0) {
              if (second) {
                if (a.gbo2) a.gbo2[(size_t)m * a.gbo2_ld + ((a.out2_coff + nq - a.N0) >> 3)] = (unsigned char)(nib | (hi << 4));
              } else if (a.gbo) {
                a.gbo[(size_t)m * a.gbo_ld + ((a.gbo_coff + nq) >> 3)] = (unsigned char)(nib | (hi << 4));
              }
            }
          }
          if (!nvalid || m < 0) continue;
          const float4 oc = ep_cvt4<OB>(old4[g]), gc = ep_cvt4<OB>(gate4[g]);
          const float o4[4] = {oc.x, oc.y, oc.z, oc.w};
          const float g4[4] = {gc.x, gc.y, gc.z, gc.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float t = v[k] + o4[k];
            if (a.relu) t = t > 0.f ? t : 0.f;
            if (!(g4[k] > 0.f) || !((gnib[g] >> k) & 1u)) t = 0.f;
            v[k] = t;
          }
          ep_st4<OB>(obase, ocol + (size_t)m * oldim, v[0], v[1], v[2], v[3]);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = col_base + j * 32 + li;
    const bool nvalid = n < a.Cout;
    const float sc = (a.scale && nvalid) ? a.scale[n] : 1.f;
    const float sh = (a.shift && nvalid) ? a.shift[n] : 0.f;
    // this column's destination: the main window, or the second one from column N0 on
    const bool second = a.out2 != nullptr && n >= a.N0;
    float* const obase = second ? a.out2 : a.out;
    const size_t ocol = second ? (size_t)(a.out2_coff + (n - a.N0)) : (size_t)(a.out_coff + n);
    const size_t oldim = second ? (size_t)a.out2_ld : (size_t)a.out_ld;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = row_base + i * 32 + 4 * lh;
      // two batches of 8 rows: enough loads in flight, half the live registers of one batch of 16
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float old[8], gate[8];
        int mrow[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int r = half * 8 + q;
          const int m = mrow[q] = rowmap(rbase + (r & 3) + 8 * (r >> 2));
          const bool ok = nvalid && m >= 0;
          old[q] = (a.accumulate && ok) ? ep_ld1<OB>(obase, ocol + (size_t)m * oldim) : 0.f;
          gate[q] = (a.mask && ok) ? ep_ld1<OB>(a.mask, (size_t)m * a.mask_ld + a.mask_coff + n) : 1.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int m = mrow[q];
          if (!nvalid || m < 0) continue;
          float v = acc[i][j][half * 8 + q] * sc + sh + old[q];
          if (a.relu) v = v > 0.f ? v : 0.f;
          if (!(gate[q] > 0.f)) v = 0.f;
          ep_st1<OB>(obase, ocol + (size_t)m * oldim, v);
        }
      }
    }
  }
}

int conv_launch(ConvKArgs& a, int math, hipStream_t s);
int conv_igemm_launch_variant(ConvKArgs& a, int math, int v, hipStream_t s);
int conv_halo_supported(const ConvKArgs& a);
int conv_halo_launch(ConvKArgs& a, int math, hipStream_t s);
int conv_halo_num_variants();
int conv_halo_launch_variant(ConvKArgs& a, int math, int v, hipStream_t s);
int conv_pix4_supported(const ConvKArgs& a);
int conv_pix4_launch(ConvKArgs& a, int math, int variant_id, hipStream_t s);
// bf16 planes of a weight pack in an arithmetic mode (0: fp32 pack)
static inline int math_planes(int math) { return math == IVF_MATH_BF16X6 ? 3 : (math == IVF_MATH_FP32 ? 0 : 2); }

}  // namespace ivf
