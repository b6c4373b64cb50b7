// Temporal perturbation-mask kernels: freeze scan + reverse-scan gradient, reverse
// (sub-clip swap) perturbation, sub-mask run detection, TV/L1 regulariser with its
// gradient, Adam.  All HBM-bound: one coalesced pass over [B,C,T,H*W].
//
// Reference: video_features_pytorch/mask.py:4-100 and the loop body of
// FindMasksComparison_I3D_smth.py:198-214.
#include "ivf_common.h"

namespace ivf {

constexpr int MAX_T = 64;

// ---------------------------------------------------------------- freeze forward
// P[0] = X[0]; P[u] = (1-m[u]) X[u] + m[u] P[u-1]      (mask.py:11-22)
// One thread per (b, c, pixel); the T-long recurrence lives in registers; adjacent
// threads touch adjacent pixels so every per-frame access is a coalesced row.
// out_cpad == 0 : NCTHW output; else channels-last [B,T,HW,out_cpad] (pad lanes
// beyond C written as zero by the c == 0 thread's neighbours, see below).
__global__ void freeze_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                  float* __restrict__ p, int B, int C, int T, int HW,
                                  int mask_per_clip, int out_cpad) {
  size_t total = (size_t)B * C * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int c = (i / HW) % C;
    int b = i / ((size_t)HW * C);
    const float* xp = x + ((size_t)(b * C + c) * T) * HW + px;
    const float* mp = mask + (mask_per_clip ? (size_t)b * T : 0);
    float prev = 0.f;
    for (int u = 0; u < T; ++u) {
      float xv = xp[(size_t)u * HW];
      float v;
      if (u == 0) {
        v = xv;
      } else {
        float m = mp[u];
        v = (1.f - m) * xv + m * prev;
      }
      prev = v;
      if (out_cpad == 0) {
        p[((size_t)(b * C + c) * T + u) * HW + px] = v;
      } else {
        p[((size_t)(b * T + u) * HW + px) * out_cpad + c] = v;
      }
    }
  }
}

// channels-last variant: one thread per (b, pixel) handles all C channels and
// writes one 16-byte vector per frame (C <= 4, cpad == 4)
__global__ void freeze_fwd_cl4_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                      float* __restrict__ p, int B, int C, int T, int HW,
                                      int mask_per_clip) {
  size_t total = (size_t)B * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int b = i / HW;
    const float* mp = mask + (mask_per_clip ? (size_t)b * T : 0);
    float prev[4] = {0.f, 0.f, 0.f, 0.f};
    for (int u = 0; u < T; ++u) {
      float m = u ? mp[u] : 0.f;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < C; ++c) {
        float xv = x[((size_t)(b * C + c) * T + u) * HW + px];
        v[c] = u ? (1.f - m) * xv + m * prev[c] : xv;
        prev[c] = v[c];
      }
      *reinterpret_cast<float4*>(p + ((size_t)(b * T + u) * HW + px) * 4) =
          make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---------------------------------------------------------------- freeze backward
// G[u] = g[u] + m[u+1] G[u+1];  dm[u] = sum_{c,px} (P[u-1] - X[u]) G[u], u >= 1
// dX[0] = G[0]... in full: dX[u] = (1-m[u]) G[u] (u>=1), dX[0] = G[0].
// Each thread recomputes P for its pixel (registers), runs the reverse scan and
// accumulates T partial sums; block partials go to a workspace and a second kernel
// sums them in a fixed order (bitwise reproducible, no float atomics).
template <int TT>
__global__ __launch_bounds__(256) void freeze_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ mask, const float* __restrict__ g,
    float* __restrict__ dx, float* __restrict__ partial, int B, int C, int T, int HW,
    int mask_per_clip, int g_cpad, int blocks_per_clip) {
  const int b = blockIdx.x / blocks_per_clip;
  const int blk = blockIdx.x % blocks_per_clip;
  const float* mp = mask + (mask_per_clip ? (size_t)b * T : 0);
  float acc[TT];
#pragma unroll
  for (int u = 0; u < TT; ++u) acc[u] = 0.f;
  const int per_clip = C * HW;
  for (int i = blk * blockDim.x + threadIdx.x; i < per_clip; i += blocks_per_clip * blockDim.x) {
    int px = i % HW;
    int c = i / HW;
    const float* xp = x + ((size_t)(b * C + c) * T) * HW + px;
    float xv[TT], pv[TT], gq[TT];
    // every frame of the pixel (and of its gradient) is requested before the scans start: a load consumed inside
    // the `u < T` branch is waited for at once, T serial round trips per thread
#pragma unroll
    for (int u = 0; u < TT; ++u)
      if (u < T) xv[u] = xp[(size_t)u * HW];
#pragma unroll
    for (int u = 0; u < TT; ++u)
      if (u < T)
        gq[u] = g_cpad ? g[((size_t)(b * T + u) * HW + px) * g_cpad + c] : g[((size_t)(b * C + c) * T + u) * HW + px];
#pragma unroll
    for (int u = 0; u < TT; ++u)
      if (u < T) pv[u] = u ? (1.f - mp[u]) * xv[u] + mp[u] * pv[u - 1] : xv[u];
    float G = 0.f;
#pragma unroll
    for (int u = TT - 1; u >= 0; --u) {
      if (u < T) {
        float gv = gq[u];
        float mnext = (u + 1 < T) ? mp[u + 1] : 0.f;
        G = gv + mnext * G;
        if (u > 0) acc[u] += (pv[u - 1] - xv[u]) * G;
        if (dx) dx[((size_t)(b * C + c) * T + u) * HW + px] = u ? (1.f - mp[u]) * G : G;
      }
    }
  }
  // block reduce each of T sums: wave shuffle then LDS across the 4 waves
  __shared__ float red[4][TT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < TT; ++u) {
    float v = acc[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][u] = v;
  }
  __syncthreads();
  if (threadIdx.x < T) {
    int u = threadIdx.x;
    float v = red[0][u] + red[1][u] + red[2][u] + red[3][u];
    partial[((size_t)b * blocks_per_clip + blk) * T + u] = v;
  }
}

// channels-last gradient (row of 4 floats per pixel, C <= 4) and no dX: one thread per
// pixel loads each frame's 16-byte gradient once and reuses it for all channels
template <int TT>
__global__ __launch_bounds__(256) void freeze_bwd_cl4_kernel(
    const float* __restrict__ x, const float* __restrict__ mask, const float* __restrict__ g,
    float* __restrict__ partial, int B, int C, int T, int HW, int mask_per_clip, int blocks_per_clip) {
  const int b = blockIdx.x / blocks_per_clip;
  const int blk = blockIdx.x % blocks_per_clip;
  const float* mp = mask + (mask_per_clip ? (size_t)b * T : 0);
  float acc[TT];
#pragma unroll
  for (int u = 0; u < TT; ++u) acc[u] = 0.f;
  for (int px = blk * blockDim.x + threadIdx.x; px < HW; px += blocks_per_clip * blockDim.x) {
    float4 gv[TT];
#pragma unroll
    for (int u = 0; u < TT; ++u)
      if (u < T) gv[u] = *reinterpret_cast<const float4*>(g + ((size_t)(b * T + u) * HW + px) * 4);
    for (int c = 0; c < C; ++c) {
      const float* xp = x + ((size_t)(b * C + c) * T) * HW + px;
      float xv[TT], pv[TT];
      // (all frames of the pixel requested before the scan: see freeze_bwd_kernel)
#pragma unroll
      for (int u = 0; u < TT; ++u)
        if (u < T) xv[u] = xp[(size_t)u * HW];
#pragma unroll
      for (int u = 0; u < TT; ++u)
        if (u < T) pv[u] = u ? (1.f - mp[u]) * xv[u] + mp[u] * pv[u - 1] : xv[u];
      float G = 0.f;
#pragma unroll
      for (int u = TT - 1; u >= 0; --u) {
        if (u < T) {
          float gc = c == 0 ? gv[u].x : (c == 1 ? gv[u].y : (c == 2 ? gv[u].z : gv[u].w));
          float mnext = (u + 1 < T) ? mp[u + 1] : 0.f;
          G = gc + mnext * G;
          if (u > 0) acc[u] += (pv[u - 1] - xv[u]) * G;
        }
      }
    }
  }
  __shared__ float red[4][TT];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int u = 0; u < TT; ++u) {
    float v = acc[u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[wave][u] = v;
  }
  __syncthreads();
  if (threadIdx.x < T) {
    int u = threadIdx.x;
    partial[((size_t)b * blocks_per_clip + blk) * T + u] = red[0][u] + red[1][u] + red[2][u] + red[3][u];
  }
}

__global__ void freeze_bwd_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dmask,
                                         int B, int T, int blocks_per_clip) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  int b = i / T, u = i % T;
  float s = 0.f;
  for (int k = 0; k < blocks_per_clip; ++k) s += partial[((size_t)b * blocks_per_clip + k) * T + u];
  dmask[i] = s;
}

// ---------------------------------------------------------------- sub-mask runs + reverse
// find_submasks_from_mask (mask.py:60-85) and the pairing of reverse (mask.py:40-56):
// for every frame t: partner[t] (= t when copied), weight[t] (m[a] of the pair's
// first-half member a), run[t] = index of its run or -1.  T is tiny: one thread.
__global__ void submask_pairs_kernel(const float* __restrict__ mask, int T, float thresh,
                                     int* __restrict__ run, int* __restrict__ partner,
                                     float* __restrict__ weight) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  int nrun = 0;
  int start = -1;
  for (int j = 0; j <= T; ++j) {
    bool on = (j < T) && (mask[j] > thresh);
    if (j < T) {
      partner[j] = j;
      weight[j] = 0.f;
      run[j] = -1;
    }
    if (on && start < 0) start = j;
    if (!on && start >= 0) {
      int len = j - start;
      for (int u = 0; u < len; ++u) run[start + u] = nrun;
      for (int u = 0; u < len / 2; ++u) {
        int a = start + u, bb = start + len - 1 - u;
        float ma = mask[a];
        partner[a] = bb;
        partner[bb] = a;
        weight[a] = ma;
        weight[bb] = ma;  // mask.py:54-56 uses mask[mask_on_inds[u]] for both
      }
      ++nrun;
      start = -1;
    }
  }
}

__global__ void reverse_fwd_kernel(const float* __restrict__ x, const int* __restrict__ partner,
                                   const float* __restrict__ weight, float* __restrict__ p, int B,
                                   int C, int T, int HW, int out_cpad) {
  size_t total = (size_t)B * C * T * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int t = (i / HW) % T;
    size_t bc = i / ((size_t)HW * T);
    int pt = partner[t];
    float v = x[i];
    if (pt != t) {
      float w = weight[t];
      v = (1.f - w) * v + w * x[(bc * T + pt) * HW + px];
    }
    if (out_cpad == 0) {
      p[i] = v;
    } else {
      int c = bc % C;
      int b = bc / C;
      p[((size_t)(b * T + t) * HW + px) * out_cpad + c] = v;
    }
  }
}

// ---------------------------------------------------------------- reverse, per-clip masks
// The batched forms used by the search loops: clip b has its own mask row, hence its own
// pairing (partner/weight rows [B,T]).
__global__ void submask_pairs_batched_kernel(const float* __restrict__ mask, int B, int T, float thresh,
                                             int* __restrict__ partner, float* __restrict__ weight) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float* m = mask + (size_t)i * T;
  int* pr = partner + (size_t)i * T;
  float* wt = weight + (size_t)i * T;
  int start = -1;
  for (int j = 0; j <= T; ++j) {
    bool on = (j < T) && (m[j] > thresh);
    if (j < T) { pr[j] = j; wt[j] = 0.f; }
    if (on && start < 0) start = j;
    if (!on && start >= 0) {
      int len = j - start;
      for (int u = 0; u < len / 2; ++u) {
        int a = start + u, bb = start + len - 1 - u;
        pr[a] = bb; pr[bb] = a; wt[a] = m[a]; wt[bb] = m[a];   // mask.py:50-56: m[a] for both
      }
      start = -1;
    }
  }
}

// one thread per (b, t, pixel): all C channels, NCTHW or 16-byte channels-last output
__global__ void reverse_fwd_batched_kernel(const float* __restrict__ x, const int* __restrict__ partner,
                                           const float* __restrict__ weight, float* __restrict__ p, int B,
                                           int C, int T, int HW, int out_cpad) {
  size_t total = (size_t)B * T * HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    int px = i % HW;
    int t = (i / HW) % T;
    int b = i / ((size_t)HW * T);
    int pt = partner[b * T + t];
    float w = weight[b * T + t];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) {
      float xv = x[((size_t)(b * C + c) * T + t) * HW + px];
      if (pt != t) xv = (1.f - w) * xv + w * x[((size_t)(b * C + c) * T + pt) * HW + px];
      if (out_cpad == 0) p[((size_t)(b * C + c) * T + t) * HW + px] = xv;
      v[c & 3] = xv;
    }
    if (out_cpad == 4) *reinterpret_cast<float4*>(p + i * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// d(loss)/d(mask) of the reverse perturbation (autograd of mask.py:49-56).  For a pair (a, b'),
// a in the first half of its run:  P[a] = (1-m[a]) X[a] + m[a] X[b'],  P[b'] = (1-m[a]) X[b'] + m[a] X[a]
//   => dL/dm[a] = sum_{c,px} (X[b'] - X[a]) * (G[a] - G[b']);  every other entry is 0
// (the run detection `mask > 0.1` is not differentiable).  Block partials + fixed-order
// reduction, as in the freeze backward.
__global__ __launch_bounds__(256) void reverse_bwd_kernel(
    const float* __restrict__ x, const int* __restrict__ partner, const float* __restrict__ g,
    float* __restrict__ partial, int B, int C, int T, int HW, int g_cpad, int blocks_per_clip) {
  const int b = blockIdx.x / blocks_per_clip;
  const int blk = blockIdx.x % blocks_per_clip;
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int t = 0; t < T; ++t) {
    const int pt = partner[b * T + t];
    float acc = 0.f;
    if (pt > t) {   // block-uniform
      for (int i = blk * blockDim.x + threadIdx.x; i < C * HW; i += blocks_per_clip * blockDim.x) {
        int px = i % HW, c = i / HW;
        float xa = x[((size_t)(b * C + c) * T + t) * HW + px];
        float xb = x[((size_t)(b * C + c) * T + pt) * HW + px];
        float ga = g_cpad ? g[((size_t)(b * T + t) * HW + px) * g_cpad + c] : g[((size_t)(b * C + c) * T + t) * HW + px];
        float gb = g_cpad ? g[((size_t)(b * T + pt) * HW + px) * g_cpad + c] : g[((size_t)(b * C + c) * T + pt) * HW + px];
        acc += (xb - xa) * (ga - gb);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __syncthreads();
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
      partial[((size_t)b * blocks_per_clip + blk) * T + t] = red[0] + red[1] + red[2] + red[3];
  }
}

__global__ void zero_kernel(float* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = 0.f;
}

// ---------------------------------------------------------------- TV / L1
__device__ __forceinline__ float pow_int_or_f(float v, float p) {
  // torch evaluates x**3 as x*x*x and x**2 as x*x on CPU; other exponents via powf
  if (p == 3.f) return v * v * v;
  if (p == 2.f) return v * v;
  if (p == 1.f) return v;
  return powf(v, p);
}

// calc_tv_norm (mask.py:88-100) value and gradient for one [T] vector.
// Autograd chain restated in fp32 with the same NaN behaviour at val == 0:
//   val = sum_{u=1}^{T-2} |m[u-1]-m[u]|^p + |m[u+1]-m[u]|^p ; y = val^(1/p) ; tv = y^q
//   dtv/dy = q y^(q-1) ; dy/dval = (1/p) val^(1/p - 1) ; d|d|^p = p |d|^(p-1) sign(d)
__device__ void tv_norm_dev(const float* m, int T, float p, float q, float* val_out, float* grad,
                            float gscale) {
  float val = 0.f;
  for (int u = 1; u < T - 1; ++u) {
    val += pow_int_or_f(fabsf(m[u - 1] - m[u]), p);
    val += pow_int_or_f(fabsf(m[u + 1] - m[u]), p);
  }
  float invp = (float)(1.0 / (double)p);
  float y = powf(val, invp);
  float tv = pow_int_or_f(y, q);
  *val_out = tv;
  if (!grad) return;
  float dy = q * pow_int_or_f(y, q - 1.f) * gscale;
  float dval = dy * (invp * powf(val, invp - 1.f));
  for (int u = 0; u < T; ++u) grad[u] = 0.f;
  for (int u = 1; u < T - 1; ++u) {
    {
      float d = m[u - 1] - m[u];
      float sg = (d > 0.f) - (d < 0.f);
      float gd = dval * (p * pow_int_or_f(fabsf(d), p - 1.f)) * sg;
      grad[u - 1] += gd;
      grad[u] -= gd;
    }
    {
      float d = m[u + 1] - m[u];
      float sg = (d > 0.f) - (d < 0.f);
      float gd = dval * (p * pow_int_or_f(fabsf(d), p - 1.f)) * sg;
      grad[u + 1] += gd;
      grad[u] -= gd;
    }
  }
}

__global__ void tv_norm_kernel(const float* __restrict__ mask, int B, int T, float p, float q,
                               float* __restrict__ val, float* __restrict__ grad) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float g[MAX_T];
  float v;
  tv_norm_dev(mask + (size_t)b * T, T, p, q, &v, grad ? g : nullptr, 1.f);
  val[b] = v;
  if (grad)
    for (int u = 0; u < T; ++u) grad[(size_t)b * T + u] = g[u];
}

// Regulariser of the search loop (FindMasksComparison_I3D_smth.py:198-200):
// sig = sigmoid(raw); l1 = lam1 * sum|sig|; tv = lam2 * TV_{3,3}(sig);
// dreg = d(l1+tv)/d sig.   One thread per clip.
__global__ void mask_reg_kernel(const float* __restrict__ raw, int B, int T, float lam1, float lam2,
                                float* __restrict__ sig, float* __restrict__ terms,
                                float* __restrict__ dreg) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s[MAX_T], g[MAX_T];
  float l1 = 0.f;
  for (int u = 0; u < T; ++u) {
    float r = raw[(size_t)b * T + u];
    s[u] = 1.f / (1.f + expf(-r));
    sig[(size_t)b * T + u] = s[u];
    l1 += fabsf(s[u]);
  }
  float tv;
  tv_norm_dev(s, T, 3.f, 3.f, &tv, g, lam2);
  terms[b * 2 + 0] = lam1 * l1;
  terms[b * 2 + 1] = lam2 * tv;
  for (int u = 0; u < T; ++u) {
    float sg = (s[u] > 0.f) - (s[u] < 0.f);
    dreg[(size_t)b * T + u] = g[u] + lam1 * sg;
  }
}

// ---------------------------------------------------------------- Adam
// torch.optim.Adam single-tensor path, defaults of FindMasksComparison_I3D_smth.py:191:
// m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, int n, float step_size, float inv_sqrt_bc2,
                            float b1, float b2, float eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  float mi = m[i] * b1 + (1.f - b1) * gi;
  float vi = v[i] * b2 + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

// Search-loop tail (FindMasksComparison_I3D_smth.py:207-214): total gradient w.r.t.
// the raw mask = (dscore/dsig + dreg/dsig) * sig*(1-sig); record (loss, l1, tv, score)
// and take the Adam step.  (For 'freeze' m[0] never enters the recurrence, mask.py:16-18:
// ivf_freeze_bwd writes an exact 0 there; for 'reverse' frame 0 may belong to a run.)
__global__ void search_step_kernel(float* __restrict__ raw, const float* __restrict__ sig,
                                   const float* __restrict__ dscore_dsig,
                                   const float* __restrict__ dreg, const float* __restrict__ terms,
                                   const float* __restrict__ score, float* __restrict__ am,
                                   float* __restrict__ av, float* __restrict__ traj, int B, int T,
                                   float step_size, float inv_sqrt_bc2, float b1, float b2,
                                   float eps) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  int b = i / T, u = i % T;
  float s = sig[i];
  float gs = dreg[i] + dscore_dsig[i];
  float gi = gs * (s * (1.f - s));
  float mi = am[i] * b1 + (1.f - b1) * gi;
  float vi = av[i] * b2 + (1.f - b2) * gi * gi;
  am[i] = mi;
  av[i] = vi;
  raw[i] = raw[i] - step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
  if (u == 0 && traj) {
    float l1 = terms[b * 2], tv = terms[b * 2 + 1], sc = score[b];
    traj[b * 4 + 0] = l1 + tv + sc;
    traj[b * 4 + 1] = l1;
    traj[b * 4 + 2] = tv;
    traj[b * 4 + 3] = sc;
  }
}

__global__ void sigmoid_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 1.f / (1.f + expf(-x[i]));
}

static inline int grid_for(size_t total, int block = 256, int cap = 2048) {
  size_t g = (total + block - 1) / block;
  return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

constexpr int FREEZE_BWD_BLOCKS_PER_CLIP = 64;

}  // namespace ivf

using namespace ivf;

extern "C" int ivf_freeze_fwd(const float* x, const float* mask, float* p, int B, int C, int T, int HW,
                              int mask_per_clip, int out_cpad, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && mask && p, "freeze_fwd: null pointer");
  IVF_CHECK_ARG(B > 0 && C > 0 && T > 0 && HW > 0, "freeze_fwd: bad dims");
  IVF_CHECK_ARG(out_cpad == 0 || out_cpad >= C, "freeze_fwd: out_cpad (%d) < C (%d)", out_cpad, C);
  hipStream_t s = (hipStream_t)stream;
  if (out_cpad == 4 && C <= 4) {
    hipLaunchKernelGGL(freeze_fwd_cl4_kernel, dim3(grid_for((size_t)B * HW)), dim3(256), 0, s, x, mask,
                       p, B, C, T, HW, mask_per_clip);
  } else {
    if (out_cpad > C) {
      // pad channels must read as zero for the conv that consumes them
      size_t n = (size_t)B * T * HW * out_cpad;
      hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n);
    }
    hipLaunchKernelGGL(freeze_fwd_kernel, dim3(grid_for((size_t)B * C * HW)), dim3(256), 0, s, x, mask,
                       p, B, C, T, HW, mask_per_clip, out_cpad);
  }
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" size_t ivf_freeze_bwd_workspace_bytes(int B, int T) {
  return (size_t)B * FREEZE_BWD_BLOCKS_PER_CLIP * T * sizeof(float);
}

extern "C" int ivf_freeze_bwd(const float* x, const float* mask, const float* g, float* dmask,
                              float* dx, int B, int C, int T, int HW, int mask_per_clip, int g_cpad,
                              void* workspace, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && mask && g && dmask && workspace, "freeze_bwd: null pointer");
  IVF_CHECK_ARG(B > 0 && C > 0 && T > 0 && T <= MAX_T && HW > 0, "freeze_bwd: bad dims (T <= %d)", MAX_T);
  IVF_CHECK_ARG(g_cpad == 0 || g_cpad >= C, "freeze_bwd: g_cpad < C");
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  const int bpc = FREEZE_BWD_BLOCKS_PER_CLIP;
  dim3 grid(B * bpc), block(256);
  if (g_cpad == 4 && C <= 4 && !dx && T <= 32) {
    if (T <= 16)
      hipLaunchKernelGGL(freeze_bwd_cl4_kernel<16>, grid, block, 0, s, x, mask, g, partial, B, C, T, HW,
                         mask_per_clip, bpc);
    else
      hipLaunchKernelGGL(freeze_bwd_cl4_kernel<32>, grid, block, 0, s, x, mask, g, partial, B, C, T, HW,
                         mask_per_clip, bpc);
  } else if (T <= 16)
    hipLaunchKernelGGL(freeze_bwd_kernel<16>, grid, block, 0, s, x, mask, g, dx, partial, B, C, T, HW,
                       mask_per_clip, g_cpad, bpc);
  else if (T <= 32)
    hipLaunchKernelGGL(freeze_bwd_kernel<32>, grid, block, 0, s, x, mask, g, dx, partial, B, C, T, HW,
                       mask_per_clip, g_cpad, bpc);
  else
    hipLaunchKernelGGL(freeze_bwd_kernel<64>, grid, block, 0, s, x, mask, g, dx, partial, B, C, T, HW,
                       mask_per_clip, g_cpad, bpc);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(freeze_bwd_reduce_kernel, dim3(cdiv(B * T, 64)), dim3(64), 0, s, partial, dmask, B,
                     T, bpc);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_submask_pairs(const float* mask, int T, float thresh, int* run, int* partner,
                                 float* weight, ivf_stream_t stream) {
  IVF_CHECK_ARG(mask && run && partner && weight && T > 0, "submask_pairs: bad args");
  hipLaunchKernelGGL(submask_pairs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mask, T, thresh,
                     run, partner, weight);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_reverse_fwd(const float* x, const int* partner, const float* weight, float* p, int B,
                               int C, int T, int HW, int out_cpad, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && partner && weight && p, "reverse_fwd: null pointer");
  IVF_CHECK_ARG(B > 0 && C > 0 && T > 0 && HW > 0, "reverse_fwd: bad dims");
  IVF_CHECK_ARG(out_cpad == 0 || out_cpad >= C, "reverse_fwd: out_cpad < C");
  hipStream_t s = (hipStream_t)stream;
  if (out_cpad > C) {
    size_t n = (size_t)B * T * HW * out_cpad;
    hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n);
  }
  hipLaunchKernelGGL(reverse_fwd_kernel, dim3(grid_for((size_t)B * C * T * HW)), dim3(256), 0, s, x,
                     partner, weight, p, B, C, T, HW, out_cpad);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_submask_pairs_batched(const float* mask, int B, int T, float thresh, int* partner,
                                         float* weight, ivf_stream_t stream) {
  IVF_CHECK_ARG(mask && partner && weight && B > 0 && T > 0, "submask_pairs_batched: bad args");
  hipLaunchKernelGGL(submask_pairs_batched_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, mask, B,
                     T, thresh, partner, weight);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_reverse_fwd_batched(const float* x, const int* partner, const float* weight, float* p,
                                       int B, int C, int T, int HW, int out_cpad, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && partner && weight && p, "reverse_fwd_batched: null pointer");
  IVF_CHECK_ARG(B > 0 && C > 0 && T > 0 && HW > 0, "reverse_fwd_batched: bad dims");
  IVF_CHECK_ARG(out_cpad == 0 || (out_cpad == 4 && C <= 4), "reverse_fwd_batched: out_cpad must be 0 or 4 (C <= 4)");
  hipLaunchKernelGGL(reverse_fwd_batched_kernel, dim3(grid_for((size_t)B * T * HW, 256, 4096)), dim3(256), 0,
                     (hipStream_t)stream, x, partner, weight, p, B, C, T, HW, out_cpad);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_reverse_bwd(const float* x, const int* partner, const float* g, float* dmask, int B, int C,
                               int T, int HW, int g_cpad, void* workspace, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && partner && g && dmask && workspace, "reverse_bwd: null pointer");
  IVF_CHECK_ARG(B > 0 && C > 0 && T > 0 && T <= MAX_T && HW > 0, "reverse_bwd: bad dims (T <= %d)", MAX_T);
  IVF_CHECK_ARG(g_cpad == 0 || g_cpad >= C, "reverse_bwd: g_cpad < C");
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;   // ivf_freeze_bwd_workspace_bytes(B, T)
  const int bpc = FREEZE_BWD_BLOCKS_PER_CLIP;
  hipLaunchKernelGGL(reverse_bwd_kernel, dim3(B * bpc), dim3(256), 0, s, x, partner, g, partial, B, C, T, HW,
                     g_cpad, bpc);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(freeze_bwd_reduce_kernel, dim3(cdiv(B * T, 64)), dim3(64), 0, s, partial, dmask, B, T, bpc);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_tv_norm(const float* mask, int B, int T, float p, float q, float* val, float* grad,
                           ivf_stream_t stream) {
  IVF_CHECK_ARG(mask && val && B > 0 && T > 0 && T <= MAX_T, "tv_norm: bad args (T <= %d)", MAX_T);
  hipLaunchKernelGGL(tv_norm_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, mask, B, T, p,
                     q, val, grad);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_mask_reg(const float* raw_mask, int B, int T, float lam1, float lam2, float* sig,
                            float* terms, float* dreg_dsig, ivf_stream_t stream) {
  IVF_CHECK_ARG(raw_mask && sig && terms && dreg_dsig && B > 0 && T > 0 && T <= MAX_T,
                "mask_reg: bad args (T <= %d)", MAX_T);
  hipLaunchKernelGGL(mask_reg_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, raw_mask, B,
                     T, lam1, lam2, sig, terms, dreg_dsig);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

static void adam_coeffs(int step, float lr, float b1, float b2, float* step_size, float* inv_sqrt_bc2) {
  double bc1 = 1.0 - pow((double)b1, step);
  double bc2 = 1.0 - pow((double)b2, step);
  *step_size = (float)((double)lr / bc1);
  *inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}

extern "C" int ivf_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int n,
                             int step, float lr, float beta1, float beta2, float eps,
                             ivf_stream_t stream) {
  IVF_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adam_step: bad args");
  float ss, isb;
  adam_coeffs(step, lr, beta1, beta2, &ss, &isb);
  hipLaunchKernelGGL(adam_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, n, ss, isb, beta1, beta2, eps);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_search_step(float* raw_mask, const float* sig, const float* dscore_dsig,
                               const float* dreg_dsig, const float* terms, const float* score,
                               float* exp_avg, float* exp_avg_sq, float* traj_row, int B, int T,
                               int step, float lr, float beta1, float beta2, float eps,
                               ivf_stream_t stream) {
  IVF_CHECK_ARG(raw_mask && sig && dscore_dsig && dreg_dsig && terms && score && exp_avg && exp_avg_sq,
                "search_step: null pointer");
  IVF_CHECK_ARG(B > 0 && T > 0 && step >= 1, "search_step: bad dims");
  float ss, isb;
  adam_coeffs(step, lr, beta1, beta2, &ss, &isb);
  hipLaunchKernelGGL(search_step_kernel, dim3(cdiv(B * T, 256)), dim3(256), 0, (hipStream_t)stream,
                     raw_mask, sig, dscore_dsig, dreg_dsig, terms, score, exp_avg, exp_avg_sq, traj_row,
                     B, T, ss, isb, beta1, beta2, eps);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

// uint8 [B][T][H][W][C] -> fp32 NCTHW or channels-last (cpad).  One thread per 4 consecutive
// pixels of a row: 4*C bytes in (dword loads when aligned), 16-byte stores out.
__global__ void clip_ingest_kernel(const unsigned char* __restrict__ f, float* __restrict__ out, int B, int T,
                                   int HW, int C, int layout, int cpad) {
  const size_t quads_per_frame = (size_t)(HW + 3) / 4;
  const size_t total = (size_t)B * T * quads_per_frame;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t fr = i / quads_per_frame;          // b*T + t
    const int px = (int)(i - fr * quads_per_frame) * 4;
    const int np = min(4, HW - px);
    const int b = (int)(fr / T), t = (int)(fr - (size_t)b * T);
    const unsigned char* src = f + (fr * HW + px) * C;
    float v[4][4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) v[p][c] = 0.f;
    if (C == 3 && np == 4 && (reinterpret_cast<size_t>(src) & 3) == 0) {
      const uint3 w = *reinterpret_cast<const uint3*>(src);
      const unsigned wd[3] = {w.x, w.y, w.z};
#pragma unroll
      for (int k = 0; k < 12; ++k) v[k / 3][k % 3] = (float)((wd[k >> 2] >> (8 * (k & 3))) & 0xffu);
    } else {
      for (int p = 0; p < np; ++p)
        for (int c = 0; c < C && c < 4; ++c) v[p][c] = (float)src[p * C + c];
    }
    if (layout == IVF_INGEST_NCTHW) {
      for (int c = 0; c < C; ++c) {
        float* dst = out + (((size_t)b * C + c) * T + t) * HW + px;
        if (np == 4 && (reinterpret_cast<size_t>(dst) & 15) == 0)
          *reinterpret_cast<float4*>(dst) = make_float4(v[0][c], v[1][c], v[2][c], v[3][c]);
        else
          for (int p = 0; p < np; ++p) dst[p] = v[p][c];
      }
    } else {
      for (int p = 0; p < np; ++p) {
        float* dst = out + (fr * HW + px + p) * cpad;
        if (cpad == 4) {
          *reinterpret_cast<float4*>(dst) = make_float4(v[p][0], v[p][1], v[p][2], v[p][3]);
        } else {
          for (int c = 0; c < cpad; ++c) dst[c] = c < C ? v[p][c] : 0.f;
        }
      }
    }
  }
}

extern "C" int ivf_clip_ingest_u8(const unsigned char* frames, float* out, int B, int T, int H, int W, int C,
                                  int layout, int cpad, ivf_stream_t stream) {
  IVF_CHECK_ARG(frames && out, "clip_ingest: null pointer");
  IVF_CHECK_ARG(B > 0 && T > 0 && H > 0 && W > 0 && C >= 1 && C <= 4, "clip_ingest: bad dims (1 <= C <= 4)");
  IVF_CHECK_ARG(layout == IVF_INGEST_NCTHW || (layout == IVF_INGEST_CL && cpad >= C),
                "clip_ingest: layout must be NCTHW or channels-last with cpad >= C");
  const size_t total = (size_t)B * T * (((size_t)H * W + 3) / 4);
  hipLaunchKernelGGL(clip_ingest_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, frames, out, B, T,
                     H * W, C, layout, cpad);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

// Frame-importance ranking: order[b][r] = the frame with the r-th largest mask value, ties by frame index (a stable
// descending sort; NaN values last, in frame order -- what torch.argsort(-mask, stable=True) returns).  One thread
// per frame counts the frames that come before it.
__global__ void rank_desc_kernel(const float* __restrict__ mask, int* __restrict__ order, int T) {
  const float* m = mask + (size_t)blockIdx.x * T;
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const float v = m[t];
    const bool vnan = v != v;
    int r = 0;
    for (int s = 0; s < T; ++s) {
      const float u = m[s];
      const bool unan = u != u;
      const bool before = vnan ? (!unan || s < t) : (!unan && (u > v || (u == v && s < t)));
      r += before ? 1 : 0;
    }
    order[(size_t)blockIdx.x * T + r] = t;
  }
}

extern "C" int ivf_rank_frames(const float* mask, int B, int T, int* order, ivf_stream_t stream) {
  IVF_CHECK_ARG(mask && order && B > 0 && T > 0, "rank_frames: bad args");
  hipLaunchKernelGGL(rank_desc_kernel, dim3(B), dim3(T < 256 ? ((T + 63) / 64) * 64 : 256), 0, (hipStream_t)stream, mask,
                     order, T);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

// init_mask('central') selection (mask.py:134-154) for b clips: candidate i (1-based) = ones with i zeros at each
// end; ratio_i = (orig - central_i) / (orig - full); the first i whose ratio is below the threshold (NaN compares
// false, as in the reference), else the last candidate; raw mask = -5 on its zeros, +5 on its ones.
__global__ void central_select_kernel(const float* __restrict__ orig, const float* __restrict__ full,
                                      const float* __restrict__ central, int B, int n, int T, float threshold,
                                      float* __restrict__ raw, int* __restrict__ chosen_i, float* __restrict__ ratio) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int pick = n;   // 1-based
  bool found = false;
  for (int i = 0; i < n; ++i) {
    const float r = (orig[b] - central[(size_t)b * n + i]) / (orig[b] - full[b]);
    if (ratio) ratio[(size_t)b * n + i] = r;
    if (!found && r < threshold) { pick = i + 1; found = true; }
  }
  if (chosen_i) chosen_i[b] = pick;
  for (int t = 0; t < T; ++t) raw[(size_t)b * T + t] = (t < pick || t >= T - pick) ? -5.f : 5.f;
}

extern "C" int ivf_init_central_select(const float* orig, const float* full, const float* central, int B, int n, int T,
                                       float threshold, float* raw_mask, int* chosen_i, float* ratio,
                                       ivf_stream_t stream) {
  IVF_CHECK_ARG(orig && full && central && raw_mask, "init_central_select: null pointer");
  IVF_CHECK_ARG(B > 0 && n > 0 && T >= 2 * n, "init_central_select: need B > 0 and 0 < n <= T / 2 candidates");
  hipLaunchKernelGGL(central_select_kernel, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, orig, full, central, B, n,
                     T, threshold, raw_mask, chosen_i, ratio);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_sigmoid(const float* x, float* y, int n, ivf_stream_t stream) {
  IVF_CHECK_ARG(x && y && n > 0, "sigmoid: bad args");
  hipLaunchKernelGGL(sigmoid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}
