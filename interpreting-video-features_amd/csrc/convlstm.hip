// ConvLSTM classifier (reference models/convolution_lstm.py + models/CLSTM_4.py) as
// explicit forward and backward-DATA (BPTT) kernels, and its mask-search loop.
//
// The recurrence is latency-bound (T x layers dependent cell steps, hidden = 4), so
// the design removes everything that does not have to be sequential:
//   * the input convolutions Wx*x_t (+bias) of a layer are batched over ALL T steps
//     in one launch (layer i at step t only needs layer i-1's output at step t);
//   * one fused kernel per cell step does the 5x5 hidden convolution, the four gates
//     and the state update (convolution_lstm.py:38-48; the peephole terms are zeros,
//     :52-54); BatchNorm (ONE shared BN for all layers, :85,123) + MaxPool2d(2) are
//     batched over T;
//   * backward mirrors it: one fused kernel per step (Wh^T over dG[t+1] + gate
//     derivatives), the stride-2 transposed input convolution batched over T.
// Layout is planar per frame ([.., C, H, W], W fastest): channel counts are 1..16, so
// coalescing comes from W, and the reference's NCTHW clip is consumed in place.
// Weights (a few KB) are staged in LDS by every block.
#include <algorithm>
#include <cstring>
#include <vector>

#include <cstdlib>

#include "ivf_common.h"

namespace ivf {

constexpr int MAXK = 7;

// Gx[b,t,g*hid+j,yo,xo] = bias[g][j] + sum_{c,ky,kx} Wx[g][j][c][ky][kx] * x[b,c,t,yo*s-p+ky,xo*s-p+kx]
// x addressed by generic strides (elements): b*sB + c*sC + t*sT + y*W + x.
// Weight tables are pre-transposed to [c][ky][kx][16 gate-channels] (and [ky][kx][16][4] for the
// backward kernels): the index is wave-uniform, so the compiler fetches the 16 weights of a
// tap with one scalar load and the FMAs take them as SGPR operands -- no LDS traffic at all.
// The tap sums use fused multiply-adds (explicit fmaf: the file is built with -ffp-contract=off): these
// kernels are VALU-bound and the unfused form costs twice the instructions; parity gates are 1e-3.
__global__ __launch_bounds__(256) void clstm_xconv_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ wT, const float* __restrict__ bias,
    float* __restrict__ gx, int B, int T, int Cin, int H, int W, long sB, long sC, long sT, int hid, int k,
    int stride, int Ho, int Wo) {
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long total = (long)B * T * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xo = i % Wo;
    int yo = (i / Wo) % Ho;
    int t = (i / ((long)Wo * Ho)) % T;
    int b = i / ((long)Wo * Ho * T);
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = (bias && o < G) ? bias[o] : 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float* xp = x + b * sB + c * sC + t * sT;
      for (int ky = 0; ky < k; ++ky) {
        int y = yo * stride - pad + ky;
        for (int kx = 0; kx < k; ++kx) {
          int xx = xo * stride - pad + kx;
          float v = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) ? xp[(long)y * W + xx] : 0.f;
          const float* wp = wT + ((c * k + ky) * k + kx) * 16;
#pragma unroll
          for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wp[o], v, acc[o]);
        }
      }
    }
    float* gp = gx + (((long)b * T + t) * G) * Ho * Wo + (long)yo * Wo + xo;
#pragma unroll
    for (int o = 0; o < 16; ++o)
      if (o < G) gp[(long)o * Ho * Wo] = acc[o];
  }
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// One cell step for all clips: gates = Gx[t] + Wh * h[t-1]; state update; saves
// S[b,t,{i,f,g,o,c},j,y,x] and H[b,t,j,y,x].
__global__ __launch_bounds__(256) void clstm_step_fwd_kernel(
    const float* __restrict__ gx, const float* __restrict__ whT, float* __restrict__ S, float* __restrict__ Hs,
    int B, int T, int t, int hid, int k, int Ho, int Wo) {
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xo = i % Wo;
    int yo = (i / Wo) % Ho;
    int b = i / plane;
    const float* gp = gx + (((long)b * T + t) * G) * plane + (long)yo * Wo + xo;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = (o < G) ? gp[(long)o * plane] : 0.f;
    if (t > 0) {
      const float* hp = Hs + (((long)b * T + (t - 1)) * hid) * plane;
      for (int c = 0; c < hid; ++c)
        for (int ky = 0; ky < k; ++ky) {
          int y = yo - pad + ky;
          for (int kx = 0; kx < k; ++kx) {
            int xx = xo - pad + kx;
            float v = ((unsigned)y < (unsigned)Ho && (unsigned)xx < (unsigned)Wo)
                          ? hp[(long)c * plane + (long)y * Wo + xx] : 0.f;
            const float* wp = whT + ((c * k + ky) * k + kx) * 16;
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wp[o], v, acc[o]);
          }
        }
    }
    float* sp = S + (((long)b * T + t) * 5 * hid) * plane + (long)yo * Wo + xo;
    const float* cprev = (t > 0) ? S + ((((long)b * T + (t - 1)) * 5 + 4) * hid) * plane + (long)yo * Wo + xo : nullptr;
    float* hp2 = Hs + (((long)b * T + t) * hid) * plane + (long)yo * Wo + xo;
    for (int j = 0; j < hid; ++j) {
      float ci = sigmoidf_(acc[j]);
      float cf = sigmoidf_(acc[hid + j]);
      float cg = tanhf(acc[2 * hid + j]);
      float co = sigmoidf_(acc[3 * hid + j]);
      float cp = cprev ? cprev[(long)j * plane] : 0.f;
      float cc = cf * cp + ci * cg;
      float ch = co * tanhf(cc);
      sp[(long)(0 * hid + j) * plane] = ci;
      sp[(long)(1 * hid + j) * plane] = cf;
      sp[(long)(2 * hid + j) * plane] = cg;
      sp[(long)(3 * hid + j) * plane] = co;
      sp[(long)(4 * hid + j) * plane] = cc;
      hp2[(long)j * plane] = ch;
    }
  }
}

// The same cell step with the hidden-state convolution split over the 4 waves of a workgroup by
// input channel (hid <= 4): a workgroup owns 64 pixels, wave c accumulates the 25 taps of channel c
// into all 16 gates (its weights stay wave-uniform scalar operands), the partial sums meet in LDS
// and wave j finishes hidden channel j.  A quarter of the serial chain per thread: what the small
// maps of the second layer (75 workgroups in the one-thread-per-pixel form) are bound by.
__global__ __launch_bounds__(256) void clstm_step_fwd_split_kernel(
    const float* __restrict__ gx, const float* __restrict__ whT, float* __restrict__ S, float* __restrict__ Hs,
    int B, int T, int t, int hid, int k, int Ho, int Wo) {
  __shared__ float red[4][16][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  const long i = (long)blockIdx.x * 64 + lane;
  const bool live = i < total;
  const int xo = live ? (int)(i % Wo) : 0;
  const int yo = live ? (int)((i / Wo) % Ho) : 0;
  const int b = live ? (int)(i / plane) : 0;
  float acc[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc[o] = 0.f;
  if (w == 0 && live) {
    const float* gp = gx + (((long)b * T + t) * G) * plane + (long)yo * Wo + xo;
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = (o < G) ? gp[(long)o * plane] : 0.f;
  }
  if (t > 0 && w < hid && live) {
    const float* hp = Hs + ((((long)b * T + (t - 1)) * hid) + w) * plane;
    for (int ky = 0; ky < k; ++ky) {
      int y = yo - pad + ky;
      for (int kx = 0; kx < k; ++kx) {
        int xx = xo - pad + kx;
        float v = ((unsigned)y < (unsigned)Ho && (unsigned)xx < (unsigned)Wo) ? hp[(long)y * Wo + xx] : 0.f;
        const float* wp = whT + ((w * k + ky) * k + kx) * 16;
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wp[o], v, acc[o]);
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 16; ++o) red[w][o][lane] = acc[o];
  __syncthreads();
  const int j = w;
  if (j >= hid || !live) return;
  float a[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int o = g * hid + j;
    a[g] = ((red[0][o][lane] + red[1][o][lane]) + red[2][o][lane]) + red[3][o][lane];
  }
  const long px = (long)yo * Wo + xo;
  float* sp = S + (((long)b * T + t) * 5 * hid) * plane + px;
  float ci = sigmoidf_(a[0]), cf = sigmoidf_(a[1]), cg = tanhf(a[2]), co = sigmoidf_(a[3]);
  float cp = (t > 0) ? S[((((long)b * T + (t - 1)) * 5 + 4) * hid + j) * plane + px] : 0.f;
  float cc = cf * cp + ci * cg;
  float ch = co * tanhf(cc);
  sp[(long)(0 * hid + j) * plane] = ci;
  sp[(long)(1 * hid + j) * plane] = cf;
  sp[(long)(2 * hid + j) * plane] = cg;
  sp[(long)(3 * hid + j) * plane] = co;
  sp[(long)(4 * hid + j) * plane] = cc;
  Hs[(((long)b * T + t) * hid + j) * plane + px] = ch;
}

// ---------------------------------------------------------------- persistent recurrence (hid <= 4)
// The step kernels above pay, per time step, a launch, a trip of the hidden state through L2 and a bounds test per
// tap.  For full batches the recurrence runs instead as ONE launch per layer and direction: a workgroup owns a whole
// clip, keeps the hidden state (forward) / one gate group of dG (backward) as zero-bordered planes in LDS -- a 5x5 tap
// is an LDS read at a constant offset, no bounds test -- carries c (forward) / dC (backward) in registers across the T
// steps, and only streams what the other direction needs (gates, h, dG) through HBM.  Each thread owns the pixels
// p = tid + q * blockDim; the taps run outermost over batches of 3-4 of them, so one scalar load of a tap's 16 weights
// feeds 8 packed FMAs per pixel of the batch.  c / dC / dh live in the rows of the thread's own pixels (L1/L2-hot), so
// every pixel loop stays rolled (one copy of the code, < 128 registers: 16 waves per clip).
constexpr int SEQ_NT = 1024;   // threads per clip at most (16 waves: the rolled pixel loops need < 128 VGPRs)

// Gate non-linearities of the persistent kernels: v_exp_f32 / v_rcp_f32 forms (about 2 ulp; tanh(x) = 2 sigmoid(2x) - 1
// carries ~2e-7 absolute near zero).  libm's expf / tanhf expand to ~40-60 instructions each and 20 of them per pixel
// cost as much as the 25-tap convolution itself; parity gates of the ConvLSTM path are 1e-3.
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 2.f * sigmoid_fast(2.f * v) - 1.f; }

template <int HID>
__global__ __launch_bounds__(1024) void clstm_seq_fwd_kernel(
    const float* __restrict__ gx, const float* __restrict__ whT, float* __restrict__ S, float* __restrict__ Hs,
    int T, int k, int Ho, int Wo) {
  extern __shared__ float hl[];   // [HID][Ho + 2 pad][Wo + 2 pad], borders stay zero
  constexpr int G = 4 * HID;
  constexpr int PB = 5;           // pixels per batch (the 60 x 80 map over 960 threads: 5 each): one scalar load of a tap's weights feeds 5 x 8 packed FMAs
  const int pad = (k - 1) / 2;
  const int WP = Wo + 2 * pad, HP = Ho + 2 * pad;
  const int plane = Ho * Wo;
  const int b = blockIdx.x;
  const int NT = blockDim.x;
  for (int i = threadIdx.x; i < HID * HP * WP; i += NT) hl[i] = 0.f;
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const float* gp = gx + ((long)b * T + t) * G * plane;
    float* sp = S + ((long)b * T + t) * 5 * HID * plane;
    float* hp = Hs + ((long)b * T + t) * HID * plane;
    // c[t-1] comes back from the saved state rows of the thread's own pixels (S is written anyway for the backward):
    // no state registers, the batches are independent and the batch loop stays rolled (one copy of the code)
    const float* cprev = t > 0 ? S + (((long)b * T + (t - 1)) * 5 + 4) * HID * plane : nullptr;
#pragma unroll 1
    for (int p0 = threadIdx.x; p0 < plane; p0 += PB * NT) {
      int pix[PB], lds0[PB];
      float acc[PB][G];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int p = p0 + u * NT;
        pix[u] = p < plane ? p : -1;
        const int pc = p < plane ? p : 0;          // (lanes without a pixel compute on pixel 0 and store nothing)
        lds0[u] = (pc / Wo) * WP + pc % Wo;
#pragma unroll
        for (int o = 0; o < G; ++o) acc[u][o] = gp[(long)o * plane + pc];
      }
      for (int c = 0; c < HID; ++c)
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) {
            const float* wp = whT + ((c * k + ky) * k + kx) * 16;
            const int off = (c * HP + ky) * WP + kx;
            float v[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) v[u] = hl[lds0[u] + off];
#pragma unroll
            for (int u = 0; u < PB; ++u)
#pragma unroll
              for (int o = 0; o < G; ++o) acc[u][o] = __builtin_fmaf(wp[o], v[u], acc[u][o]);
          }
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int p = pix[u];
        if (p < 0) continue;
#pragma unroll
        for (int j = 0; j < HID; ++j) {
          const float ci = sigmoid_fast(acc[u][j]), cf = sigmoid_fast(acc[u][HID + j]);
          const float cg = tanh_fast(acc[u][2 * HID + j]), co = sigmoid_fast(acc[u][3 * HID + j]);
          const float cp = cprev ? cprev[(long)j * plane + p] : 0.f;
          const float cc = cf * cp + ci * cg;
          const float ch = co * tanh_fast(cc);
          sp[(long)(0 * HID + j) * plane + p] = ci;
          sp[(long)(1 * HID + j) * plane + p] = cf;
          sp[(long)(2 * HID + j) * plane + p] = cg;
          sp[(long)(3 * HID + j) * plane + p] = co;
          sp[(long)(4 * HID + j) * plane + p] = cc;
          hp[(long)j * plane + p] = ch;
        }
      }
    }
    __syncthreads();      // every read of h[t-1] is done
    // h[t] of the thread's OWN pixels back from the rows it has just written
    for (int p = threadIdx.x; p < plane; p += NT) {
      const int l = (p / Wo + pad) * WP + p % Wo + pad;
#pragma unroll
      for (int j = 0; j < HID; ++j) hl[j * HP * WP + l] = hp[(long)j * plane + p];
    }
    __syncthreads();
  }
}

// BPTT of the same: dG[t] for all T steps in one launch.  Per step the hidden-state convolution of dG[t+1] runs
// gate group by gate group (the hid planes of one gate type in LDS at a time: all 16 planes of the layer's first map
// would not fit); dh[t] accumulates in the dHpool rows of the thread's own pixels (L1/L2-hot, not read by anyone else),
// so every loop over pixels stays rolled; the gate derivatives follow from the saved gates; dC goes through the layer's
// dC rows (own pixels).
template <int HID>
__global__ __launch_bounds__(1024) void clstm_seq_bwd_kernel(
    float* __restrict__ dHpool, const float* __restrict__ whB, const float* __restrict__ S,
    float* __restrict__ dG, float* __restrict__ dC, int T, int k, int Ho, int Wo) {
  extern __shared__ float gl[];   // [HID][Ho + 2 pad][Wo + 2 pad]: one gate group of dG[t+1], zero borders
  constexpr int G = 4 * HID;
  constexpr int PB = 5;
  const int pad = (k - 1) / 2;
  const int WP = Wo + 2 * pad, HP = Ho + 2 * pad;
  const int plane = Ho * Wo;
  const int b = blockIdx.x;
  const int NT = blockDim.x;
  for (int i = threadIdx.x; i < HID * HP * WP; i += NT) gl[i] = 0.f;
  float* dcp = dC + (long)b * HID * plane;
  for (int t = T - 1; t >= 0; --t) {
    float* dhrow = dHpool + ((long)b * T + t) * HID * plane;
    if (t + 1 < T) {
      const float* gnext = dG + ((long)b * T + (t + 1)) * G * plane;
      for (int g = 0; g < 4; ++g) {
        __syncthreads();      // the previous group's taps are done
        // (this step's group rows of the thread's OWN pixels, written by itself one step ago: program order is enough)
        for (int p = threadIdx.x; p < plane; p += NT) {
          const int l = (p / Wo + pad) * WP + p % Wo + pad;
#pragma unroll
          for (int jj = 0; jj < HID; ++jj) gl[jj * HP * WP + l] = gnext[(long)(g * HID + jj) * plane + p];
        }
        __syncthreads();
        // dh[j] += sum_{ky,kx,jj} whB[ky][kx][g*hid+jj][j] * dG[t+1][g*hid+jj][y - ky + pad][x - kx + pad]
#pragma unroll 1
        for (int p0 = threadIdx.x; p0 < plane; p0 += PB * NT) {
          int pix[PB], lds0[PB];
          float dh[PB][HID];
#pragma unroll
          for (int u = 0; u < PB; ++u) {
            const int p = p0 + u * NT;
            pix[u] = p < plane ? p : -1;
            const int pc = p < plane ? p : 0;
            lds0[u] = (pc / Wo) * WP + pc % Wo;
#pragma unroll
            for (int j = 0; j < HID; ++j) dh[u][j] = dhrow[(long)j * plane + pc];
          }
          for (int ky = 0; ky < k; ++ky)
            for (int kx = 0; kx < k; ++kx) {
              const float* wp = whB + (ky * k + kx) * 64 + g * HID * 4;      // [jj][j], j padded to 4
              const int off = (2 * pad - ky) * WP + (2 * pad - kx);
#pragma unroll
              for (int jj = 0; jj < HID; ++jj) {
                float v[PB];
#pragma unroll
                for (int u = 0; u < PB; ++u) v[u] = gl[jj * HP * WP + lds0[u] + off];
#pragma unroll
                for (int u = 0; u < PB; ++u)
#pragma unroll
                  for (int j = 0; j < HID; ++j) dh[u][j] = __builtin_fmaf(wp[jj * 4 + j], v[u], dh[u][j]);
              }
            }
#pragma unroll
          for (int u = 0; u < PB; ++u)
            if (pix[u] >= 0)
#pragma unroll
              for (int j = 0; j < HID; ++j) dhrow[(long)j * plane + pix[u]] = dh[u][j];
        }
      }
    }
    const float* sp = S + ((long)b * T + t) * 5 * HID * plane;
    const float* cprev = (t > 0) ? S + (((long)b * T + (t - 1)) * 5 + 4) * HID * plane : nullptr;
    float* gout = dG + ((long)b * T + t) * G * plane;
#pragma unroll 1
    for (int p = threadIdx.x; p < plane; p += NT) {
#pragma unroll 1
      for (int j = 0; j < HID; ++j) {
        const float dhj = dhrow[(long)j * plane + p];
        const float ci = sp[(long)(0 * HID + j) * plane + p], cf = sp[(long)(1 * HID + j) * plane + p];
        const float cg = sp[(long)(2 * HID + j) * plane + p], co = sp[(long)(3 * HID + j) * plane + p];
        const float cc = sp[(long)(4 * HID + j) * plane + p];
        const float cp = cprev ? cprev[(long)j * plane + p] : 0.f;
        const float th = tanh_fast(cc);
        const float dco = dhj * th;
        const float dcc = dhj * co * (1.f - th * th) + ((t + 1 < T) ? dcp[(long)j * plane + p] : 0.f);
        gout[(long)(0 * HID + j) * plane + p] = dcc * cg * (ci * (1.f - ci));
        gout[(long)(1 * HID + j) * plane + p] = dcc * cp * (cf * (1.f - cf));
        gout[(long)(2 * HID + j) * plane + p] = dcc * ci * (1.f - cg * cg);
        gout[(long)(3 * HID + j) * plane + p] = dco * (co * (1.f - co));
        dcp[(long)j * plane + p] = dcc * cf;
      }
    }
  }
}

// X[b,t,j,yp,xp] = max over 2x2 of (scale[j]*H + shift[j]); first strict maximum wins
__global__ void clstm_bnpool_fwd_kernel(const float* __restrict__ Hs, const float* __restrict__ scale,
                                        const float* __restrict__ shift, float* __restrict__ X,
                                        unsigned char* __restrict__ arg, long n_frames, int hid, int Ho, int Wo,
                                        int Hp, int Wp) {
  const long total = n_frames * hid * Hp * Wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xp = i % Wp;
    int yp = (i / Wp) % Hp;
    int j = (i / ((long)Wp * Hp)) % hid;
    long f = i / ((long)Wp * Hp * hid);
    const float* hp = Hs + (f * hid + j) * (long)Ho * Wo;
    float sc = scale ? scale[j] : 1.f, sh = shift ? shift[j] : 0.f;
    float best = 0.f;
    int bi = 0;
    for (int q = 0; q < 4; ++q) {
      float v = hp[(long)(2 * yp + (q >> 1)) * Wo + 2 * xp + (q & 1)] * sc + sh;
      if (q == 0 || v > best || v != v) { best = v; bi = q; }
    }
    X[i] = best;
    arg[i] = (unsigned char)bi;
  }
}

// dHpool[b,t,j,y,x] = scale[j] * dX[pooled cell] if this cell won, else 0
__global__ void clstm_unpool_bwd_kernel(const float* __restrict__ dX, const unsigned char* __restrict__ arg,
                                        const float* __restrict__ scale, float* __restrict__ dH, long n_frames,
                                        int hid, int Ho, int Wo, int Hp, int Wp) {
  const long total = n_frames * hid * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int x = i % Wo;
    int y = (i / Wo) % Ho;
    long fj = i / ((long)Wo * Ho);
    int j = fj % hid;
    int yp = y >> 1, xp = x >> 1;
    float v = 0.f;
    if (yp < Hp && xp < Wp) {
      long pi = (fj * Hp + yp) * Wp + xp;
      int q = ((y & 1) << 1) | (x & 1);
      if (arg[pi] == q) v = dX[pi] * (scale ? scale[j] : 1.f);
    }
    dH[i] = v;
  }
}

// Backward cell step t (convolution_lstm.py:38-48 differentiated):
// dh = dHpool[t] + Wh^T (x) dG[t+1];  do = dh*tanh(c); dc = dh*o*(1-tanh^2 c) + dC;
// di = dc*g; df = dc*c_prev; dg = dc*i; dC <- dc*f;  dG = gate derivatives.
__global__ __launch_bounds__(256) void clstm_step_bwd_kernel(
    const float* __restrict__ dHpool, const float* __restrict__ whB, const float* __restrict__ S,
    float* __restrict__ dG, float* __restrict__ dC, int B, int T, int t, int hid, int k, int Ho, int Wo) {
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int x = i % Wo;
    int y = (i / Wo) % Ho;
    int b = i / plane;
    float dh[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < hid; ++j) dh[j] = dHpool[(((long)b * T + t) * hid + j) * plane + (long)y * Wo + x];
    if (t + 1 < T) {
      const float* gp = dG + (((long)b * T + (t + 1)) * G) * plane;
      for (int ky = 0; ky < k; ++ky) {
        int yy = y - ky + pad;
        for (int kx = 0; kx < k; ++kx) {
          int xx = x - kx + pad;
          const bool ok = (unsigned)yy < (unsigned)Ho && (unsigned)xx < (unsigned)Wo;
          const float* wp = whB + (ky * k + kx) * 64;     // [o][j], j padded to 4
#pragma unroll
          for (int o = 0; o < 16; ++o) {
            float g = (ok && o < G) ? gp[(long)o * plane + (long)yy * Wo + xx] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) dh[j] = __builtin_fmaf(wp[o * 4 + j], g, dh[j]);
          }
        }
      }
    }
    const float* sp = S + (((long)b * T + t) * 5 * hid) * plane + (long)y * Wo + x;
    const float* cprev = (t > 0) ? S + ((((long)b * T + (t - 1)) * 5 + 4) * hid) * plane + (long)y * Wo + x : nullptr;
    float* gout = dG + (((long)b * T + t) * G) * plane + (long)y * Wo + x;
    float* dcp = dC + ((long)b * hid) * plane + (long)y * Wo + x;
    for (int j = 0; j < hid; ++j) {
      float ci = sp[(long)(0 * hid + j) * plane], cf = sp[(long)(1 * hid + j) * plane];
      float cg = sp[(long)(2 * hid + j) * plane], co = sp[(long)(3 * hid + j) * plane];
      float cc = sp[(long)(4 * hid + j) * plane];
      float cp = cprev ? cprev[(long)j * plane] : 0.f;
      float th = tanhf(cc);
      float dco = dh[j] * th;
      float dcc = dh[j] * co * (1.f - th * th) + ((t + 1 < T) ? dcp[(long)j * plane] : 0.f);
      gout[(long)(0 * hid + j) * plane] = dcc * cg * (ci * (1.f - ci));
      gout[(long)(1 * hid + j) * plane] = dcc * cp * (cf * (1.f - cf));
      gout[(long)(2 * hid + j) * plane] = dcc * ci * (1.f - cg * cg);
      gout[(long)(3 * hid + j) * plane] = dco * (co * (1.f - co));
      dcp[(long)j * plane] = dcc * cf;
    }
  }
}

// Backward cell step split over the 4 waves of a workgroup by gate type (hid <= 4): wave g gathers
// the taps of gates [g*hid, (g+1)*hid) of step t+1 into a partial dH (weights stay scalar operands),
// the partials meet in LDS, wave j finishes hidden channel j.  Same reasoning as the forward split.
__global__ __launch_bounds__(256) void clstm_step_bwd_split_kernel(
    const float* __restrict__ dHpool, const float* __restrict__ whB, const float* __restrict__ S,
    float* __restrict__ dG, float* __restrict__ dC, int B, int T, int t, int hid, int k, int Ho, int Wo) {
  __shared__ float red[4][4][64];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  const long i = (long)blockIdx.x * 64 + lane;
  const bool live = i < total;
  const int x = live ? (int)(i % Wo) : 0;
  const int y = live ? (int)((i / Wo) % Ho) : 0;
  const int b = live ? (int)(i / plane) : 0;
  float dh[4] = {0.f, 0.f, 0.f, 0.f};
  if (t + 1 < T && live) {
    const float* gp = dG + (((long)b * T + (t + 1)) * G) * plane;
    for (int ky = 0; ky < k; ++ky) {
      int yy = y - ky + pad;
      for (int kx = 0; kx < k; ++kx) {
        int xx = x - kx + pad;
        const bool ok = (unsigned)yy < (unsigned)Ho && (unsigned)xx < (unsigned)Wo;
        const float* wp = whB + (ky * k + kx) * 64;     // [o][j], j padded to 4
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int o = w * hid + q;
          float g = (ok && q < hid) ? gp[(long)o * plane + (long)yy * Wo + xx] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) dh[j] = __builtin_fmaf(wp[o * 4 + j], g, dh[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[w][j][lane] = dh[j];
  __syncthreads();
  const int j = w;
  if (j >= hid || !live) return;
  const long px = (long)y * Wo + x;
  const float dhj = dHpool[(((long)b * T + t) * hid + j) * plane + px] +
                    (((red[0][j][lane] + red[1][j][lane]) + red[2][j][lane]) + red[3][j][lane]);
  const float* sp = S + (((long)b * T + t) * 5 * hid) * plane + px;
  float* gout = dG + (((long)b * T + t) * G) * plane + px;
  float* dcp = dC + ((long)b * hid) * plane + px;
  float ci = sp[(long)(0 * hid + j) * plane], cf = sp[(long)(1 * hid + j) * plane];
  float cg = sp[(long)(2 * hid + j) * plane], co = sp[(long)(3 * hid + j) * plane];
  float cc = sp[(long)(4 * hid + j) * plane];
  float cp = (t > 0) ? S[((((long)b * T + (t - 1)) * 5 + 4) * hid + j) * plane + px] : 0.f;
  float th = tanhf(cc);
  float dco = dhj * th;
  float dcc = dhj * co * (1.f - th * th) + ((t + 1 < T) ? dcp[(long)j * plane] : 0.f);
  gout[(long)(0 * hid + j) * plane] = dcc * cg * (ci * (1.f - ci));
  gout[(long)(1 * hid + j) * plane] = dcc * cp * (cf * (1.f - cf));
  gout[(long)(2 * hid + j) * plane] = dcc * ci * (1.f - cg * cg);
  gout[(long)(3 * hid + j) * plane] = dco * (co * (1.f - co));
  dcp[(long)j * plane] = dcc * cf;
}

// dx[b,c,t,y,x] = sum_{o,ky,kx} Wx[o][c][ky][kx] * dG[b,t,o,(y+p-ky)/s,(x+p-kx)/s]  (exact divisions only)
__global__ __launch_bounds__(256) void clstm_xconv_bwd_kernel(
    const float* __restrict__ dG, const float* __restrict__ wB, float* __restrict__ dx, int B, int T, int Cin,
    int H, int W, long sB, long sC, long sT, int hid, int k, int stride, int Ho, int Wo) {
  const int G = 4 * hid;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * T * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int x = i % W;
    int y = (i / W) % H;
    int t = (i / ((long)W * H)) % T;
    int b = i / ((long)W * H * T);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* gp = dG + (((long)b * T + t) * G) * plane;
    for (int ky = 0; ky < k; ++ky) {
      int ny = y + pad - ky;
      if (ny < 0 || ny % stride) continue;
      int yo = ny / stride;
      if (yo >= Ho) continue;
      for (int kx = 0; kx < k; ++kx) {
        int nx = x + pad - kx;
        if (nx < 0 || nx % stride) continue;
        int xo = nx / stride;
        if (xo >= Wo) continue;
        const float* wp = wB + (ky * k + kx) * 64;     // [o][c], c padded to 4
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          float g = (o < G) ? gp[(long)o * plane + (long)yo * Wo + xo] : 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_fmaf(wp[o * 4 + c], g, acc[c]);
        }
      }
    }
    for (int c = 0; c < Cin; ++c) dx[b * sB + c * sC + t * sT + (long)y * W + x] = acc[c];
  }
}

// The same sums for the reference geometry (k = 5, stride 2, pad 2, even H and W;
// convolution_lstm.py:23-31), one thread per 2 x 2 block of input pixels: a gate position
// reaches pixel parity (py, px) through tap (2j + py, 2i + px), so the 3 x 3 gate positions
// around the block are loaded once (16 gates each) and fan out to the four pixels with
// compile-time taps -- no parity branches (the per-pixel form above diverges on every tap) and
// 36 instead of 100 gate loads per pixel.  Sums run in the same (ky, kx, o) order per pixel.
template <int CIN>   // input channels (1..4) at compile time: only the live accumulators are computed
__global__ __launch_bounds__(256) void clstm_xconv_bwd_k5s2_kernel(
    const float* __restrict__ dG, const float* __restrict__ wB, float* __restrict__ dx, int B, int T, int,
    int H, int W, long sB, long sC, long sT, int hid, int Ho, int Wo) {
  const int G = 4 * hid;
  const long plane = (long)Ho * Wo;
  const int Hb = H >> 1, Wb = W >> 1;
  const long total = (long)B * T * Hb * Wb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int bx = i % Wb;
    const int by = (i / Wb) % Hb;
    const int t = (i / ((long)Wb * Hb)) % T;
    const int b = i / ((long)Wb * Hb * T);
    float acc[2][2][CIN];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int px = 0; px < 2; ++px)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[py][px][c] = 0.f;
    const float* gp = dG + (((long)b * T + t) * G) * plane;
    // tap order per pixel must stay ascending in (ky, kx): j (hence yo) descends as ky ascends
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int yo = by + 1 - j;
#pragma unroll
      for (int ii = 0; ii < 3; ++ii) {
        const int xo = bx + 1 - ii;
        float g[16];
        const bool ok = yo >= 0 && yo < Ho && xo >= 0 && xo < Wo;
#pragma unroll
        for (int o = 0; o < 16; ++o) g[o] = (ok && o < G) ? gp[(long)o * plane + (long)yo * Wo + xo] : 0.f;
#pragma unroll
        for (int py = 0; py < 2; ++py) {
          if (2 * j + py > 4) continue;
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            if (2 * ii + px > 4) continue;
            const float* wp = wB + ((2 * j + py) * 5 + (2 * ii + px)) * 64;
#pragma unroll
            for (int o = 0; o < 16; ++o)
#pragma unroll
              for (int c = 0; c < CIN; ++c) acc[py][px][c] = __builtin_fmaf(wp[o * 4 + c], g[o], acc[py][px][c]);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
      for (int py = 0; py < 2; ++py) {
        float* dst = dx + b * sB + c * sC + t * sT + (long)(2 * by + py) * W + 2 * bx;
        *reinterpret_cast<float2*>(dst) = make_float2(acc[py][0][c], acc[py][1][c]);
      }
  }
}

// raw [G][cin][k][k] -> fwd table [cin][k][k][16] and bwd table [k][k][16][4] (zero padded)
__global__ void clstm_weight_tables_kernel(const float* __restrict__ w, float* __restrict__ wT,
                                           float* __restrict__ wB, int G, int cin, int k) {
  int nT = cin * k * k * 16, nB = k * k * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nT + nB; i += gridDim.x * blockDim.x) {
    if (i < nT) {
      int o = i & 15, r = i >> 4;
      int kx = r % k, ky = (r / k) % k, c = r / (k * k);
      wT[i] = o < G ? w[((o * cin + c) * k + ky) * k + kx] : 0.f;
    } else {
      int q = i - nT;
      int c = q & 3, o = (q >> 2) & 15, r = q >> 6;
      int kx = r % k, ky = r / k;
      wB[q] = (o < G && c < cin) ? w[((o * cin + c) * k + ky) * k + kx] : 0.f;
    }
  }
}

// ---------------------------------------------------------------- hidden > 4 (up to 32, multiple of 4)
// The reference's default is nb_lstm_units = 32 (CLSTM_4.py:9); its KTH configs use 4, which is what the
// kernels above are shaped for (all 16 gate channels of a pixel in one thread).  Wider cells run the same
// arithmetic blocked by 16 gate channels: the gate index stays the global g*hid + j everywhere in memory;
// a thread handles one pixel and one block (blockIdx.y):
//   x-conv forward: gate block gb = gates [16 gb, 16 gb + 16);
//   cell steps: channel group q = hidden channels [4q, 4q+4) with their four gates (local o = g*4 + jj);
//   backward convs: 4 output channels, looping over all G gates.
// Weight tables (clstm_weight_tables_wide_kernel): wxW [gb][cin][k][k][16], whW [q][hid][k][k][16],
// whBW [q][k][k][G][4], wxBW [cb][k][k][G][4] -- wave-uniform indices, scalar loads, as above.
__global__ __launch_bounds__(256) void clstm_xconv_fwd_wide_kernel(
    const float* __restrict__ x, const float* __restrict__ wxW, const float* __restrict__ bias,
    float* __restrict__ gx, int B, int T, int Cin, int H, int W, long sB, long sC, long sT, int hid, int k,
    int stride, int Ho, int Wo) {
  const int G = 4 * hid;
  const int gb = blockIdx.y;
  const int pad = (k - 1) / 2;
  const float* wT = wxW + (size_t)gb * Cin * k * k * 16;
  const long total = (long)B * T * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xo = i % Wo;
    int yo = (i / Wo) % Ho;
    int t = (i / ((long)Wo * Ho)) % T;
    int b = i / ((long)Wo * Ho * T);
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = bias ? bias[gb * 16 + o] : 0.f;
    for (int c = 0; c < Cin; ++c) {
      const float* xp = x + b * sB + c * sC + t * sT;
      for (int ky = 0; ky < k; ++ky) {
        int y = yo * stride - pad + ky;
        for (int kx = 0; kx < k; ++kx) {
          int xx = xo * stride - pad + kx;
          float v = ((unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W) ? xp[(long)y * W + xx] : 0.f;
          const float* wp = wT + ((c * k + ky) * k + kx) * 16;
#pragma unroll
          for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wp[o], v, acc[o]);
        }
      }
    }
    float* gp = gx + (((long)b * T + t) * G + gb * 16) * Ho * Wo + (long)yo * Wo + xo;
#pragma unroll
    for (int o = 0; o < 16; ++o) gp[(long)o * Ho * Wo] = acc[o];
  }
}

__global__ __launch_bounds__(256) void clstm_step_fwd_wide_kernel(
    const float* __restrict__ gx, const float* __restrict__ whW, float* __restrict__ S, float* __restrict__ Hs,
    int B, int T, int t, int hid, int k, int Ho, int Wo) {
  const int G = 4 * hid;
  const int q = blockIdx.y;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  const float* whT = whW + (size_t)q * hid * k * k * 16;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int xo = i % Wo;
    int yo = (i / Wo) % Ho;
    int b = i / plane;
    const float* gp = gx + (((long)b * T + t) * G) * plane + (long)yo * Wo + xo;
    float acc[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = gp[(long)((o >> 2) * hid + 4 * q + (o & 3)) * plane];
    if (t > 0) {
      const float* hp = Hs + (((long)b * T + (t - 1)) * hid) * plane;
      for (int c = 0; c < hid; ++c)
        for (int ky = 0; ky < k; ++ky) {
          int y = yo - pad + ky;
          for (int kx = 0; kx < k; ++kx) {
            int xx = xo - pad + kx;
            float v = ((unsigned)y < (unsigned)Ho && (unsigned)xx < (unsigned)Wo)
                          ? hp[(long)c * plane + (long)y * Wo + xx] : 0.f;
            const float* wp = whT + ((c * k + ky) * k + kx) * 16;
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = __builtin_fmaf(wp[o], v, acc[o]);
          }
        }
    }
    float* sp = S + (((long)b * T + t) * 5 * hid) * plane + (long)yo * Wo + xo;
    const float* cprev = (t > 0) ? S + ((((long)b * T + (t - 1)) * 5 + 4) * hid) * plane + (long)yo * Wo + xo : nullptr;
    float* hp2 = Hs + (((long)b * T + t) * hid) * plane + (long)yo * Wo + xo;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * q + jj;
      float ci = sigmoidf_(acc[jj]);
      float cf = sigmoidf_(acc[4 + jj]);
      float cg = tanhf(acc[8 + jj]);
      float co = sigmoidf_(acc[12 + jj]);
      float cp = cprev ? cprev[(long)j * plane] : 0.f;
      float cc = cf * cp + ci * cg;
      float ch = co * tanhf(cc);
      sp[(long)(0 * hid + j) * plane] = ci;
      sp[(long)(1 * hid + j) * plane] = cf;
      sp[(long)(2 * hid + j) * plane] = cg;
      sp[(long)(3 * hid + j) * plane] = co;
      sp[(long)(4 * hid + j) * plane] = cc;
      hp2[(long)j * plane] = ch;
    }
  }
}

__global__ __launch_bounds__(256) void clstm_step_bwd_wide_kernel(
    const float* __restrict__ dHpool, const float* __restrict__ whBW, const float* __restrict__ S,
    float* __restrict__ dG, float* __restrict__ dC, int B, int T, int t, int hid, int k, int Ho, int Wo) {
  const int G = 4 * hid;
  const int q = blockIdx.y;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * plane;
  const float* whB = whBW + (size_t)q * k * k * G * 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int x = i % Wo;
    int y = (i / Wo) % Ho;
    int b = i / plane;
    float dh[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dh[jj] = dHpool[(((long)b * T + t) * hid + 4 * q + jj) * plane + (long)y * Wo + x];
    if (t + 1 < T) {
      const float* gp = dG + (((long)b * T + (t + 1)) * G) * plane;
      for (int ky = 0; ky < k; ++ky) {
        int yy = y - ky + pad;
        if ((unsigned)yy >= (unsigned)Ho) continue;
        for (int kx = 0; kx < k; ++kx) {
          int xx = x - kx + pad;
          if ((unsigned)xx >= (unsigned)Wo) continue;
          const float* wp = whB + (size_t)(ky * k + kx) * G * 4;
          for (int o = 0; o < G; ++o) {
            float g = gp[(long)o * plane + (long)yy * Wo + xx];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) dh[jj] = __builtin_fmaf(wp[o * 4 + jj], g, dh[jj]);
          }
        }
      }
    }
    const float* sp = S + (((long)b * T + t) * 5 * hid) * plane + (long)y * Wo + x;
    const float* cprev = (t > 0) ? S + ((((long)b * T + (t - 1)) * 5 + 4) * hid) * plane + (long)y * Wo + x : nullptr;
    float* gout = dG + (((long)b * T + t) * G) * plane + (long)y * Wo + x;
    float* dcp = dC + ((long)b * hid) * plane + (long)y * Wo + x;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = 4 * q + jj;
      float ci = sp[(long)(0 * hid + j) * plane], cf = sp[(long)(1 * hid + j) * plane];
      float cg = sp[(long)(2 * hid + j) * plane], co = sp[(long)(3 * hid + j) * plane];
      float cc = sp[(long)(4 * hid + j) * plane];
      float cp = cprev ? cprev[(long)j * plane] : 0.f;
      float th = tanhf(cc);
      float dco = dh[jj] * th;
      float dcc = dh[jj] * co * (1.f - th * th) + ((t + 1 < T) ? dcp[(long)j * plane] : 0.f);
      gout[(long)(0 * hid + j) * plane] = dcc * cg * (ci * (1.f - ci));
      gout[(long)(1 * hid + j) * plane] = dcc * cp * (cf * (1.f - cf));
      gout[(long)(2 * hid + j) * plane] = dcc * ci * (1.f - cg * cg);
      gout[(long)(3 * hid + j) * plane] = dco * (co * (1.f - co));
      dcp[(long)j * plane] = dcc * cf;
    }
  }
}

// NOTE on step order: the gate derivatives of step t for group q are written while other groups' launches
// of the SAME step may still read dG[t+1] (never dG[t]) -- the groups of one launch are independent.
__global__ __launch_bounds__(256) void clstm_xconv_bwd_wide_kernel(
    const float* __restrict__ dG, const float* __restrict__ wxBW, float* __restrict__ dx, int B, int T, int Cin,
    int H, int W, long sB, long sC, long sT, int hid, int k, int stride, int Ho, int Wo) {
  const int G = 4 * hid;
  const int cb = blockIdx.y;
  const int pad = (k - 1) / 2;
  const long plane = (long)Ho * Wo;
  const long total = (long)B * T * H * W;
  const float* wB = wxBW + (size_t)cb * k * k * G * 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int x = i % W;
    int y = (i / W) % H;
    int t = (i / ((long)W * H)) % T;
    int b = i / ((long)W * H * T);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* gp = dG + (((long)b * T + t) * G) * plane;
    for (int ky = 0; ky < k; ++ky) {
      int ny = y + pad - ky;
      if (ny < 0 || ny % stride) continue;
      int yo = ny / stride;
      if (yo >= Ho) continue;
      for (int kx = 0; kx < k; ++kx) {
        int nx = x + pad - kx;
        if (nx < 0 || nx % stride) continue;
        int xo = nx / stride;
        if (xo >= Wo) continue;
        const float* wp = wB + (size_t)(ky * k + kx) * G * 4;
        for (int o = 0; o < G; ++o) {
          float g = gp[(long)o * plane + (long)yo * Wo + xo];
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = __builtin_fmaf(wp[o * 4 + c], g, acc[c]);
        }
      }
    }
    for (int c = 0; c < 4; ++c)
      if (4 * cb + c < Cin) dx[b * sB + (4 * cb + c) * sC + t * sT + (long)y * W + x] = acc[c];
  }
}

// raw [G][cin][k][k] -> wide tables.  mode 0: forward gate blocks  wF[gb][cin][k][k][16]  (o = gb*16 + oo)
//                                     mode 1: forward channel groups wF[q][cin][k][k][16] (o = (oo>>2)*hid + 4q + (oo&3))
//                                     backward (both): wBk[cb][k][k][G][4] = w[o][4cb + cc][ky][kx]
__global__ void clstm_weight_tables_wide_kernel(const float* __restrict__ w, float* __restrict__ wF,
                                                float* __restrict__ wBk, int hid, int cin, int k, int mode) {
  const int G = 4 * hid;
  const int nblk = G / 16;                       // gate blocks == channel groups (hid / 4)
  const int ncb = (cin + 3) / 4;
  const long nF = (long)nblk * cin * k * k * 16, nB = (long)ncb * k * k * G * 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nF + nB; i += (long)gridDim.x * blockDim.x) {
    if (i < nF) {
      int oo = i & 15;
      long r = i >> 4;
      int kx = r % k, ky = (r / k) % k, c = (r / (k * k)) % cin, blk = r / ((long)k * k * cin);
      int o = mode == 0 ? blk * 16 + oo : (oo >> 2) * hid + 4 * blk + (oo & 3);
      wF[i] = w[(((long)o * cin + c) * k + ky) * k + kx];
    } else {
      long qi = i - nF;
      int cc = qi & 3;
      long r = qi >> 2;
      int o = r % G;
      r /= G;
      int kx = r % k, ky = (r / k) % k, cb = r / ((long)k * k);
      int c = 4 * cb + cc;
      wBk[qi] = c < cin ? w[(((long)o * cin + c) * k + ky) * k + kx] : 0.f;
    }
  }
}

__global__ void clstm_fill_kernel(float* p, long n, float v) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}

static inline int grid_for(long total, int block = 256, int cap = 4096) {
  long g = (total + block - 1) / block;
  return (int)(g > cap ? cap : (g ? g : 1));
}

struct LayerPlan {
  int cin, Hin, Win, Ho, Wo, Hp, Wp;
  size_t wx_off, bx_off, wh_off;                    // floats in weights arena (raw reference layout)
  size_t wxT_off, wxB_off, whT_off, whB_off;        // transposed tables for the scalar-load kernels
  size_t wxW_off = 0, wxBW_off = 0, whW_off = 0, whBW_off = 0;   // blocked tables (hidden > 4)
  size_t gx_off, S_off, H_off, X_off, dG_off, dHp_off, dC_off, dX_off;  // floats in workspace
  size_t arg_off;                                   // bytes
};

}  // namespace ivf

using namespace ivf;

struct ivf_clstm {
  ivf_clstm_config cfg;
  std::vector<LayerPlan> L;
  size_t bn_scale_off, bn_shift_off, fcw_off, fcb_off, weights_floats;
  int feat;      // hid * Hp * Wp of the top layer
  int fc_in;     // endFC inputs: feat * number of output steps
  size_t ws_bytes;
  size_t off_p, off_dp, off_flat, off_dflat, off_logits, off_probs, off_score, off_sig, off_terms, off_dreg,
      off_dsig, off_fbwd, off_pair;
  float* wa = nullptr;
  char* ws = nullptr;
  std::vector<bool> cell_loaded;
  bool head_loaded = false;
  float* wsf(size_t off) const { return (float*)ws + off; }
  template <class T>
  T* at(size_t off) const { return (T*)(ws + off); }
};

extern "C" int ivf_clstm_create(const ivf_clstm_config* c, ivf_clstm_t** out) {
  IVF_CHECK_ARG(c && out, "clstm_create: null pointer");
  IVF_CHECK_ARG(c->B > 0 && c->C > 0 && c->C <= 4 && c->T > 0 && c->T <= 64 && c->H > 0 && c->W > 0,
                "clstm_create: bad clip geometry (C<=4, T<=64)");
  IVF_CHECK_ARG(c->hidden > 0 && (c->hidden <= 4 || (c->hidden <= 32 && c->hidden % 4 == 0)) && c->layers > 0 &&
                    c->layers <= 8,
                "clstm_create: hidden must be 1..4 or a multiple of 4 up to 32 (CLSTM_4.py:9 default), layers 1..8");
  IVF_CHECK_ARG(c->kernel >= 1 && c->kernel <= MAXK && (c->kernel & 1) && c->stride >= 1 && c->num_classes > 0,
                "clstm_create: kernel must be odd <= %d", MAXK);
  IVF_CHECK_ARG(c->out_step >= 0 && c->out_step < c->T, "clstm_create: out_step outside [0,T)");
  IVF_CHECK_ARG(c->n_out_steps >= 0 && c->n_out_steps <= 16, "clstm_create: at most 16 output steps");
  for (int e = 0; e < c->n_out_steps; ++e)
    IVF_CHECK_ARG(c->out_steps[e] >= 0 && c->out_steps[e] < c->T && (e == 0 || c->out_steps[e] > c->out_steps[e - 1]),
                  "clstm_create: out_steps must be increasing steps inside [0,T)");
  ivf_clstm* n = new ivf_clstm();
  n->cfg = *c;
  if (c->n_out_steps == 0) {          // single step: the general code below sees a list of one
    n->cfg.n_out_steps = 1;
    n->cfg.out_steps[0] = c->out_step;
  }
  const int hid = c->hidden, k = c->kernel, G = 4 * hid;
  size_t w = 0;
  auto takew = [&](size_t e) { size_t o = w; w += (e + 63) / 64 * 64; return o; };
  size_t fl = 0;
  const size_t B = c->B, T = c->T;
  auto takef = [&](size_t e) { size_t o = fl; fl += (e + 63) / 64 * 64; return o; };
  int cin = c->C, H = c->H, W = c->W;
  for (int i = 0; i < c->layers; ++i) {
    LayerPlan p{};
    p.cin = cin; p.Hin = H; p.Win = W;
    p.Ho = H / c->stride; p.Wo = W / c->stride;      // convolution_lstm.py:58-59 (init_hidden shape)
    int pad = (k - 1) / 2;
    int conv_h = (H + 2 * pad - k) / c->stride + 1, conv_w = (W + 2 * pad - k) / c->stride + 1;
    if (conv_h != p.Ho || conv_w != p.Wo || p.Ho < 2 || p.Wo < 2) {
      set_error("clstm_create: layer %d: conv output %dx%d != hidden state %dx%d (the reference asserts the same)",
                i, conv_h, conv_w, p.Ho, p.Wo);
      delete n;
      return IVF_ERR_BAD_ARG;
    }
    p.Hp = p.Ho / 2; p.Wp = p.Wo / 2;
    p.wx_off = takew((size_t)G * cin * k * k);
    p.bx_off = takew(G);
    p.wh_off = takew((size_t)G * hid * k * k);
    p.wxT_off = takew((size_t)cin * k * k * 16);
    p.wxB_off = takew((size_t)k * k * 64);
    p.whT_off = takew((size_t)hid * k * k * 16);
    p.whB_off = takew((size_t)k * k * 64);
    if (hid > 4) {     // blocked tables of the wide path
      p.wxW_off = takew((size_t)(G / 16) * cin * k * k * 16);
      p.wxBW_off = takew((size_t)((cin + 3) / 4) * k * k * G * 4);
      p.whW_off = takew((size_t)(hid / 4) * hid * k * k * 16);
      p.whBW_off = takew((size_t)(hid / 4) * k * k * G * 4);
    }
    size_t plane = (size_t)p.Ho * p.Wo;
    p.gx_off = takef(B * T * G * plane);
    p.S_off = takef(B * T * 5 * hid * plane);
    p.H_off = takef(B * T * hid * plane);
    p.X_off = takef(B * T * hid * p.Hp * p.Wp);
    p.dG_off = takef(B * T * G * plane);
    p.dHp_off = takef(B * T * hid * plane);
    p.dC_off = takef(B * hid * plane);
    p.dX_off = takef(B * T * hid * p.Hp * p.Wp);
    n->L.push_back(p);
    cin = hid; H = p.Hp; W = p.Wp;
  }
  n->feat = hid * H * W;
  n->fc_in = n->feat * n->cfg.n_out_steps;
  n->bn_scale_off = takew(hid);
  n->bn_shift_off = takew(hid);
  n->fcw_off = takew((size_t)c->num_classes * n->fc_in);
  n->fcb_off = takew(c->num_classes);
  n->weights_floats = w;
  size_t clip = (size_t)c->C * c->T * c->H * c->W;
  n->off_p = takef(B * clip) * 4;
  n->off_dp = takef(B * clip) * 4;
  n->off_flat = takef(B * n->fc_in) * 4;
  n->off_dflat = takef(B * n->fc_in) * 4;
  size_t bytes = fl * 4;
  auto takeb = [&](size_t nb) { size_t o = bytes; bytes += align_up(nb, 256); return o; };
  for (auto& p : n->L) p.arg_off = takeb(B * T * hid * p.Hp * p.Wp);
  const int K = c->num_classes;
  n->off_logits = takeb(B * K * 4);
  n->off_probs = takeb(B * K * 4);
  n->off_score = takeb(B * 4);
  n->off_sig = takeb(B * T * 4);
  n->off_terms = takeb(B * 2 * 4);
  n->off_dreg = takeb(B * T * 4);
  n->off_dsig = takeb(B * T * 4);
  n->off_fbwd = takeb(ivf_freeze_bwd_workspace_bytes((int)B, (int)T));
  n->off_pair = takeb(B * T * 8);
  n->ws_bytes = bytes;
  n->cell_loaded.assign(c->layers, false);
  *out = n;
  return IVF_OK;
}

extern "C" void ivf_clstm_destroy(ivf_clstm_t* n) { delete n; }
extern "C" size_t ivf_clstm_weights_bytes(const ivf_clstm_t* n) { return n ? n->weights_floats * 4 : 0; }
extern "C" size_t ivf_clstm_workspace_bytes(const ivf_clstm_t* n) { return n ? n->ws_bytes : 0; }

extern "C" int ivf_clstm_bind(ivf_clstm_t* n, void* weights_arena, void* workspace) {
  IVF_CHECK_ARG(n && weights_arena && workspace, "clstm_bind: null pointer");
  IVF_CHECK_ARG(((uintptr_t)weights_arena & 255) == 0 && ((uintptr_t)workspace & 255) == 0,
                "clstm_bind: arenas must be 256-byte aligned");
  n->wa = (float*)weights_arena;
  n->ws = (char*)workspace;
  return IVF_OK;
}

extern "C" int ivf_clstm_load_cell(ivf_clstm_t* n, int layer, const float* wxi, const float* wxf,
                                   const float* wxc, const float* wxo, const float* bxi, const float* bxf,
                                   const float* bxc, const float* bxo, const float* whi, const float* whf,
                                   const float* whc, const float* who, ivf_stream_t stream) {
  IVF_CHECK_ARG(n && n->wa, "clstm_load_cell: bind first");
  IVF_CHECK_ARG(layer >= 0 && layer < (int)n->L.size(), "clstm_load_cell: bad layer");
  IVF_CHECK_ARG(wxi && wxf && wxc && wxo && bxi && bxf && bxc && bxo && whi && whf && whc && who,
                "clstm_load_cell: null tensor");
  const LayerPlan& p = n->L[layer];
  const int hid = n->cfg.hidden, k = n->cfg.kernel;
  hipStream_t s = (hipStream_t)stream;
  const float* wx[4] = {wxi, wxf, wxc, wxo};   // gate order i, f, c(g), o -- convolution_lstm.py:22-29
  const float* bx[4] = {bxi, bxf, bxc, bxo};
  const float* wh[4] = {whi, whf, whc, who};
  size_t ex = (size_t)hid * p.cin * k * k, eh = (size_t)hid * hid * k * k;
  for (int g = 0; g < 4; ++g) {
    IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.wx_off + g * ex, wx[g], ex * 4, hipMemcpyDeviceToDevice, s));
    IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.bx_off + g * hid, bx[g], (size_t)hid * 4, hipMemcpyDeviceToDevice, s));
    IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.wh_off + g * eh, wh[g], eh * 4, hipMemcpyDeviceToDevice, s));
  }
  if (hid > 4) {
    hipLaunchKernelGGL(clstm_weight_tables_wide_kernel, dim3(64), dim3(256), 0, s, n->wa + p.wx_off, n->wa + p.wxW_off,
                       n->wa + p.wxBW_off, hid, p.cin, k, 0);
    IVF_CHECK_LAUNCH();
    hipLaunchKernelGGL(clstm_weight_tables_wide_kernel, dim3(64), dim3(256), 0, s, n->wa + p.wh_off, n->wa + p.whW_off,
                       n->wa + p.whBW_off, hid, hid, k, 1);
    IVF_CHECK_LAUNCH();
  } else {
    hipLaunchKernelGGL(clstm_weight_tables_kernel, dim3(8), dim3(256), 0, s, n->wa + p.wx_off, n->wa + p.wxT_off,
                       n->wa + p.wxB_off, 4 * hid, p.cin, k);
    IVF_CHECK_LAUNCH();
    hipLaunchKernelGGL(clstm_weight_tables_kernel, dim3(8), dim3(256), 0, s, n->wa + p.wh_off, n->wa + p.whT_off,
                       n->wa + p.whB_off, 4 * hid, hid, k);
    IVF_CHECK_LAUNCH();
  }
  n->cell_loaded[layer] = true;
  return IVF_OK;
}

extern "C" int ivf_clstm_load_head(ivf_clstm_t* n, const float* bn_gamma, const float* bn_beta,
                                   const float* bn_mean, const float* bn_var, const float* fc_w,
                                   const float* fc_b, float bn_eps, ivf_stream_t stream) {
  IVF_CHECK_ARG(n && n->wa, "clstm_load_head: bind first");
  IVF_CHECK_ARG(fc_w && fc_b, "clstm_load_head: null FC tensors");
  hipStream_t s = (hipStream_t)stream;
  if (n->cfg.batch_norm) {
    IVF_CHECK_ARG(bn_gamma && bn_beta && bn_mean && bn_var, "clstm_load_head: BatchNorm tensors required");
    IVF_PROPAGATE(ivf_bn_fold(bn_gamma, bn_beta, bn_mean, bn_var, bn_eps, n->wa + n->bn_scale_off,
                              n->wa + n->bn_shift_off, n->cfg.hidden, s));
  }
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + n->fcw_off, fc_w, (size_t)n->cfg.num_classes * n->fc_in * 4,
                               hipMemcpyDeviceToDevice, s));
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + n->fcb_off, fc_b, (size_t)n->cfg.num_classes * 4, hipMemcpyDeviceToDevice, s));
  n->head_loaded = true;
  return IVF_OK;
}

namespace ivf {

// The persistent recurrence (clstm_seq_*_kernel): hid <= 4, the clip's map within SEQ_MAXP pixels per thread and its
// padded planes within the LDS; chosen for batches that fill the chip with one workgroup per clip
// (IVF_CLSTM_PERSIST=1 always / 0 never: tests and A/B measurements).
static size_t seq_lds_bytes(const LayerPlan& p, const ivf_clstm_config& c) {
  const int pad = (c.kernel - 1) / 2;
  return (size_t)c.hidden * (p.Ho + 2 * pad) * (p.Wo + 2 * pad) * sizeof(float);
}
static int seq_threads(const LayerPlan& p) {
  const int plane = p.Ho * p.Wo;
  const int per = (plane + SEQ_NT - 1) / SEQ_NT;         // pixels per thread
  return std::min(SEQ_NT, ((plane + per - 1) / per + 63) / 64 * 64);
}
static bool seq_ok(const LayerPlan& p, const ivf_clstm_config& c, int b) {
  static const int mode = getenv("IVF_CLSTM_PERSIST") ? atoi(getenv("IVF_CLSTM_PERSIST")) : -1;
  if (mode == 0 || c.hidden > 4) return false;
  if (seq_lds_bytes(p, c) > 160 * 1024) return false;
  return mode == 1 || b >= 64;
}

static int clstm_ready(const ivf_clstm* n, int b) {
  IVF_CHECK_ARG(n && n->wa && n->ws, "clstm: not bound");
  IVF_CHECK_ARG(b > 0 && b <= n->cfg.B, "clstm: batch %d outside [1,%d]", b, n->cfg.B);
  for (bool l : n->cell_loaded) IVF_CHECK_ARG(l, "clstm: a cell's weights are not loaded");
  IVF_CHECK_ARG(n->head_loaded, "clstm: head weights not loaded");
  return IVF_OK;
}

// forward of the clip held at `x` (NCTHW); leaves activations in the workspace
static int clstm_run_forward(ivf_clstm* n, const float* x, int b, float* logits, float* probs, hipStream_t s) {
  const ivf_clstm_config& c = n->cfg;
  const int hid = c.hidden, k = c.kernel, T = c.T;
  const float* sc = c.batch_norm ? n->wa + n->bn_scale_off : nullptr;
  const float* sh = c.batch_norm ? n->wa + n->bn_shift_off : nullptr;
  for (size_t i = 0; i < n->L.size(); ++i) {
    const LayerPlan& p = n->L[i];
    const float* in;
    long sB, sC, sT;
    if (i == 0) {
      in = x;
      sC = (long)T * c.H * c.W; sB = sC * c.C; sT = (long)c.H * c.W;
    } else {
      const LayerPlan& q = n->L[i - 1];
      in = n->wsf(q.X_off);
      sC = (long)q.Hp * q.Wp; sT = sC * hid; sB = sT * T;
    }
    if (hid > 4)
      hipLaunchKernelGGL(clstm_xconv_fwd_wide_kernel, dim3(grid_for((long)b * T * p.Ho * p.Wo, 256, 16384), hid / 4),
                         dim3(256), 0, s, in, n->wa + p.wxW_off, n->wa + p.bx_off, n->wsf(p.gx_off), b, T, p.cin, p.Hin,
                         p.Win, sB, sC, sT, hid, k, c.stride, p.Ho, p.Wo);
    else
      hipLaunchKernelGGL(clstm_xconv_fwd_kernel, dim3(grid_for((long)b * T * p.Ho * p.Wo, 256, 16384)), dim3(256), 0, s, in,
                         n->wa + p.wxT_off, n->wa + p.bx_off, n->wsf(p.gx_off), b, T, p.cin, p.Hin, p.Win, sB, sC, sT,
                         hid, k, c.stride, p.Ho, p.Wo);
    IVF_CHECK_LAUNCH();
    if (seq_ok(p, c, b)) {       // the whole recurrence of the layer in one launch, one workgroup per clip
#define IVF_SEQ_F(HH)                                                                                                   \
  {                                                                                                                      \
    static LdsAttrOnce once;                                                                                             \
    IVF_PROPAGATE(raise_lds_limit(reinterpret_cast<const void*>(&clstm_seq_fwd_kernel<HH>), 160 * 1024, once));          \
    hipLaunchKernelGGL(clstm_seq_fwd_kernel<HH>, dim3(b), dim3(seq_threads(p)), seq_lds_bytes(p, c), s, n->wsf(p.gx_off), \
                       n->wa + p.whT_off, n->wsf(p.S_off), n->wsf(p.H_off), T, k, p.Ho, p.Wo);                          \
  }
      switch (hid) {
        case 1: IVF_SEQ_F(1) break;
        case 2: IVF_SEQ_F(2) break;
        case 3: IVF_SEQ_F(3) break;
        default: IVF_SEQ_F(4) break;
      }
#undef IVF_SEQ_F
      IVF_CHECK_LAUNCH();
    } else
    for (int t = 0; t < T; ++t) {
      if (hid > 4) {
        hipLaunchKernelGGL(clstm_step_fwd_wide_kernel, dim3(grid_for((long)b * p.Ho * p.Wo, 256, 16384), hid / 4),
                           dim3(256), 0, s, n->wsf(p.gx_off), n->wa + p.whW_off, n->wsf(p.S_off), n->wsf(p.H_off), b, T,
                           t, hid, k, p.Ho, p.Wo);
        IVF_CHECK_LAUNCH();
        continue;
      }
      static const int split_below = getenv("IVF_CLSTM_SPLIT") ? atoi(getenv("IVF_CLSTM_SPLIT")) : (1 << 30);
      if (hid <= 4 && (long)b * p.Ho * p.Wo <= split_below)
        hipLaunchKernelGGL(clstm_step_fwd_split_kernel, dim3((unsigned)(((long)b * p.Ho * p.Wo + 63) / 64)), dim3(256), 0,
                           s, n->wsf(p.gx_off), n->wa + p.whT_off, n->wsf(p.S_off), n->wsf(p.H_off), b, T, t, hid, k,
                           p.Ho, p.Wo);
      else
        hipLaunchKernelGGL(clstm_step_fwd_kernel, dim3(grid_for((long)b * p.Ho * p.Wo, 256, 16384)), dim3(256), 0,
                           s, n->wsf(p.gx_off), n->wa + p.whT_off, n->wsf(p.S_off), n->wsf(p.H_off), b, T, t, hid, k,
                           p.Ho, p.Wo);
      IVF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(clstm_bnpool_fwd_kernel, dim3(grid_for((long)b * T * hid * p.Hp * p.Wp)), dim3(256), 0, s,
                       n->wsf(p.H_off), sc, sh, n->wsf(p.X_off), n->at<unsigned char>(p.arg_off), (long)b * T, hid,
                       p.Ho, p.Wo, p.Hp, p.Wp);
    IVF_CHECK_LAUNCH();
  }
  // CLSTM_4.py:73-80: endFC on output[-1] (last effective step) flattened [hid*Hp*Wp], or on the
  // effective steps' outputs concatenated in step order (use_entire_seq)
  const LayerPlan& q = n->L.back();
  float* flat = n->at<float>(n->off_flat);
  for (int e = 0; e < c.n_out_steps; ++e)
    IVF_CHECK_HIP(hipMemcpy2DAsync(flat + (size_t)e * n->feat, (size_t)n->fc_in * 4,
                                   n->wsf(q.X_off) + (size_t)c.out_steps[e] * n->feat, (size_t)T * n->feat * 4,
                                   (size_t)n->feat * 4, b, hipMemcpyDeviceToDevice, s));
  float* lg = n->at<float>(n->off_logits);
  float* pr = n->at<float>(n->off_probs);
  IVF_PROPAGATE(ivf_head_fwd(flat, n->wa + n->fcw_off, n->wa + n->fcb_off, nullptr, lg, pr, b, 1, n->fc_in,
                             c.num_classes, c.softmax, s));
  size_t nb = (size_t)b * c.num_classes * 4;
  if (logits) IVF_CHECK_HIP(hipMemcpyAsync(logits, lg, nb, hipMemcpyDeviceToDevice, s));
  if (probs) IVF_CHECK_HIP(hipMemcpyAsync(probs, pr, nb, hipMemcpyDeviceToDevice, s));
  return IVF_OK;
}

// BPTT to the input clip gradient dx (NCTHW)
static int clstm_run_backward(ivf_clstm* n, int b, const int* target, const float* dout, float* score, float* dx,
                              hipStream_t s) {
  const ivf_clstm_config& c = n->cfg;
  const int hid = c.hidden, k = c.kernel, T = c.T;
  const float* sc = c.batch_norm ? n->wa + n->bn_scale_off : nullptr;
  float* flat = n->at<float>(n->off_flat);
  float* dflat = n->at<float>(n->off_dflat);
  IVF_PROPAGATE(ivf_head_bwd(flat, n->wa + n->fcw_off, n->at<float>(n->off_probs), target, dout, score, nullptr,
                             dflat, b, 1, n->fc_in, c.num_classes, c.softmax, 0, s));
  // gradient of the pooled top-layer outputs: zero except at the output step(s)
  const LayerPlan& top = n->L.back();
  long ntop = (long)b * T * n->feat;
  hipLaunchKernelGGL(clstm_fill_kernel, dim3(grid_for(ntop)), dim3(256), 0, s, n->wsf(top.dX_off), ntop, 0.f);
  IVF_CHECK_LAUNCH();
  for (int e = 0; e < c.n_out_steps; ++e)
    IVF_CHECK_HIP(hipMemcpy2DAsync(n->wsf(top.dX_off) + (size_t)c.out_steps[e] * n->feat, (size_t)T * n->feat * 4,
                                   dflat + (size_t)e * n->feat, (size_t)n->fc_in * 4, (size_t)n->feat * 4, b,
                                   hipMemcpyDeviceToDevice, s));
  for (int i = (int)n->L.size() - 1; i >= 0; --i) {
    const LayerPlan& p = n->L[i];
    hipLaunchKernelGGL(clstm_unpool_bwd_kernel, dim3(grid_for((long)b * T * hid * p.Ho * p.Wo)), dim3(256), 0, s,
                       n->wsf(p.dX_off), n->at<unsigned char>(p.arg_off), sc, n->wsf(p.dHp_off), (long)b * T, hid,
                       p.Ho, p.Wo, p.Hp, p.Wp);
    IVF_CHECK_LAUNCH();
    if (seq_ok(p, c, b)) {
#define IVF_SEQ_B(HH)                                                                                                     \
  {                                                                                                                        \
    static LdsAttrOnce once;                                                                                               \
    IVF_PROPAGATE(raise_lds_limit(reinterpret_cast<const void*>(&clstm_seq_bwd_kernel<HH>), 160 * 1024, once));            \
    hipLaunchKernelGGL(clstm_seq_bwd_kernel<HH>, dim3(b), dim3(seq_threads(p)), seq_lds_bytes(p, c), s, n->wsf(p.dHp_off), \
                       n->wa + p.whB_off, n->wsf(p.S_off), n->wsf(p.dG_off), n->wsf(p.dC_off), T, k, p.Ho, p.Wo);        \
  }
      switch (hid) {
        case 1: IVF_SEQ_B(1) break;
        case 2: IVF_SEQ_B(2) break;
        case 3: IVF_SEQ_B(3) break;
        default: IVF_SEQ_B(4) break;
      }
#undef IVF_SEQ_B
      IVF_CHECK_LAUNCH();
    } else
    for (int t = T - 1; t >= 0; --t) {
      static const int split_below = getenv("IVF_CLSTM_SPLIT") ? atoi(getenv("IVF_CLSTM_SPLIT")) : (1 << 30);
      if (hid > 4) {
        hipLaunchKernelGGL(clstm_step_bwd_wide_kernel, dim3(grid_for((long)b * p.Ho * p.Wo, 256, 16384), hid / 4),
                           dim3(256), 0, s, n->wsf(p.dHp_off), n->wa + p.whBW_off, n->wsf(p.S_off), n->wsf(p.dG_off),
                           n->wsf(p.dC_off), b, T, t, hid, k, p.Ho, p.Wo);
        IVF_CHECK_LAUNCH();
        continue;
      }
      if (hid <= 4 && (long)b * p.Ho * p.Wo <= split_below)
        hipLaunchKernelGGL(clstm_step_bwd_split_kernel, dim3((unsigned)(((long)b * p.Ho * p.Wo + 63) / 64)), dim3(256), 0,
                           s, n->wsf(p.dHp_off), n->wa + p.whB_off, n->wsf(p.S_off), n->wsf(p.dG_off),
                           n->wsf(p.dC_off), b, T, t, hid, k, p.Ho, p.Wo);
      else
        hipLaunchKernelGGL(clstm_step_bwd_kernel, dim3(grid_for((long)b * p.Ho * p.Wo, 256, 16384)), dim3(256), 0,
                           s, n->wsf(p.dHp_off), n->wa + p.whB_off, n->wsf(p.S_off), n->wsf(p.dG_off),
                           n->wsf(p.dC_off), b, T, t, hid, k, p.Ho, p.Wo);
      IVF_CHECK_LAUNCH();
    }
    float* out;
    long sB, sC, sT;
    if (i == 0) {
      out = dx;
      sC = (long)T * c.H * c.W; sB = sC * c.C; sT = (long)c.H * c.W;
    } else {
      const LayerPlan& q = n->L[i - 1];
      out = n->wsf(q.dX_off);
      sC = (long)q.Hp * q.Wp; sT = sC * hid; sB = sT * T;
    }
    if (k == 5 && c.stride == 2 && p.Hin % 2 == 0 && p.Win % 2 == 0 && hid <= 4 && p.cin <= 4 && sC % 2 == 0 &&
        sT % 2 == 0 && sB % 2 == 0)
    {
      const dim3 grid(grid_for((long)b * T * (p.Hin / 2) * (p.Win / 2), 256, 16384));
#define IVF_XB(CI)                                                                                              \
  hipLaunchKernelGGL(clstm_xconv_bwd_k5s2_kernel<CI>, grid, dim3(256), 0, s, n->wsf(p.dG_off), n->wa + p.wxB_off, \
                     out, b, T, p.cin, p.Hin, p.Win, sB, sC, sT, hid, p.Ho, p.Wo)
      switch (p.cin) {
        case 1: IVF_XB(1); break;
        case 2: IVF_XB(2); break;
        case 3: IVF_XB(3); break;
        default: IVF_XB(4); break;
      }
#undef IVF_XB
    }
    else if (hid > 4)
      hipLaunchKernelGGL(clstm_xconv_bwd_wide_kernel,
                         dim3(grid_for((long)b * T * p.Hin * p.Win, 256, 16384), (p.cin + 3) / 4), dim3(256), 0, s,
                         n->wsf(p.dG_off), n->wa + p.wxBW_off, out, b, T, p.cin, p.Hin, p.Win, sB, sC, sT, hid, k,
                         c.stride, p.Ho, p.Wo);
    else
      hipLaunchKernelGGL(clstm_xconv_bwd_kernel, dim3(grid_for((long)b * T * p.Hin * p.Win, 256, 16384)), dim3(256), 0,
                         s, n->wsf(p.dG_off), n->wa + p.wxB_off, out, b, T, p.cin, p.Hin, p.Win, sB, sC, sT, hid, k,
                         c.stride, p.Ho, p.Wo);
    IVF_CHECK_LAUNCH();
  }
  return IVF_OK;
}

}  // namespace ivf

extern "C" int ivf_clstm_forward(ivf_clstm_t* n, const float* x, int b, float* logits, float* probs,
                                 ivf_stream_t stream) {
  IVF_PROPAGATE(clstm_ready(n, b));
  IVF_CHECK_ARG(x, "clstm_forward: null clip");
  return clstm_run_forward(n, x, b, logits, probs, (hipStream_t)stream);
}

extern "C" int ivf_clstm_backward(ivf_clstm_t* n, int b, const int* target, const float* dout, float* score,
                                  float* dx, ivf_stream_t stream) {
  IVF_PROPAGATE(clstm_ready(n, b));
  IVF_CHECK_ARG((target || dout) && dx, "clstm_backward: need target or dout, and dx");
  return clstm_run_backward(n, b, target, dout, score, dx, (hipStream_t)stream);
}

extern "C" int ivf_clstm_search(ivf_clstm_t* n, const float* x, int b, const int* target, float* raw_mask,
                                float* exp_avg, float* exp_avg_sq, float lam1, float lam2, float lr, float beta1,
                                float beta2, float eps, int N, int first_step, int mode, float* traj,
                                ivf_stream_t stream) {
  IVF_PROPAGATE(clstm_ready(n, b));
  IVF_CHECK_ARG(x && target && raw_mask && exp_avg && exp_avg_sq, "clstm_search: null pointer");
  IVF_CHECK_ARG(N >= 0 && first_step >= 1, "clstm_search: bad iteration counts");
  IVF_CHECK_ARG(mode == 0 || mode == 1, "clstm_search: mode must be 0 (freeze) or 1 (reverse)");
  const ivf_clstm_config& c = n->cfg;
  hipStream_t s = (hipStream_t)stream;
  const int T = c.T, HW = c.H * c.W;
  float* sig = n->at<float>(n->off_sig);
  float* terms = n->at<float>(n->off_terms);
  float* dreg = n->at<float>(n->off_dreg);
  float* dsig = n->at<float>(n->off_dsig);
  float* score = n->at<float>(n->off_score);
  float* P = n->at<float>(n->off_p);
  float* dP = n->at<float>(n->off_dp);
  int* partner = n->at<int>(n->off_pair);
  float* weight = (float*)(partner + (size_t)c.B * c.T);
  for (int it = 0; it < N; ++it) {
    IVF_PROPAGATE(ivf_mask_reg(raw_mask, b, T, lam1, lam2, sig, terms, dreg, s));
    if (mode == 0) {
      IVF_PROPAGATE(ivf_freeze_fwd(x, sig, P, b, c.C, T, HW, 1, 0, s));
    } else {
      IVF_PROPAGATE(ivf_submask_pairs_batched(sig, b, T, 0.1f, partner, weight, s));
      IVF_PROPAGATE(ivf_reverse_fwd_batched(x, partner, weight, P, b, c.C, T, HW, 0, s));
    }
    IVF_PROPAGATE(clstm_run_forward(n, P, b, nullptr, nullptr, s));
    IVF_PROPAGATE(clstm_run_backward(n, b, target, nullptr, score, dP, s));
    if (mode == 0)
      IVF_PROPAGATE(ivf_freeze_bwd(x, sig, dP, dsig, nullptr, b, c.C, T, HW, 1, 0, n->at<void>(n->off_fbwd), s));
    else
      IVF_PROPAGATE(ivf_reverse_bwd(x, partner, dP, dsig, b, c.C, T, HW, 0, n->at<void>(n->off_fbwd), s));
    IVF_PROPAGATE(ivf_search_step(raw_mask, sig, dsig, dreg, terms, score, exp_avg, exp_avg_sq,
                                  traj ? traj + (size_t)it * b * 4 : nullptr, b, T, first_step + it, lr, beta1,
                                  beta2, eps, s));
  }
  return IVF_OK;
}

extern "C" int ivf_clstm_perturbed_forward(ivf_clstm_t* n, const float* x, int b, const float* mask, int mode,
                                           float* probs, ivf_stream_t stream) {
  IVF_PROPAGATE(clstm_ready(n, b));
  IVF_CHECK_ARG(x && mask && (mode == 0 || mode == 1), "clstm_perturbed_forward: bad args");
  const ivf_clstm_config& c = n->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* P = n->at<float>(n->off_p);
  if (mode == 0) {
    IVF_PROPAGATE(ivf_freeze_fwd(x, mask, P, b, c.C, c.T, c.H * c.W, 1, 0, s));
  } else {
    int* partner = n->at<int>(n->off_pair);
    float* weight = (float*)(partner + (size_t)c.B * c.T);
    IVF_PROPAGATE(ivf_submask_pairs_batched(mask, b, c.T, 0.1f, partner, weight, s));
    IVF_PROPAGATE(ivf_reverse_fwd_batched(x, partner, weight, P, b, c.C, c.T, c.H * c.W, 0, s));
  }
  return clstm_run_forward(n, P, b, nullptr, probs, s);
}
