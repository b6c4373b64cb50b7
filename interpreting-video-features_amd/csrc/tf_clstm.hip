// TF-style ConvLSTM classifier and its mask search: a DOCUMENTED EXTENSION, parity unpinned.
//
// The reference's TensorFlow half (video_features_tf/models/clstm.py:9-52,87-126, mask/find_mask_kth.py:300-372,
// mask/gradcam.py:28-111) runs the same temporal-mask search over a Keras ConvLSTM2D stack.  TensorFlow 1.12 /
// Keras are not installable in the build container, so nothing here can be checked against the reference's own
// outputs; the arithmetic follows the published Keras ConvLSTM2D definition (keras/layers/convolutional_recurrent.py
// of TF 1.12) as the reference's call site configures it:
//   x-part   : conv2d(x_t, kernel[kh,kw,Cin,4F], strides (s,s), padding 'valid' | 'same') + bias[4F]
//   h-part   : conv2d(h_{t-1}, recurrent_kernel[kh,kw,F,4F], stride 1, padding 'same')
//   gates    : i, f, o = recurrent_activation(.) (hard_sigmoid = clip(0.2 z + 0.5, 0, 1), the TF 1.12 default;
//              sigmoid selectable), c = f*c + i*tanh(.), h = o*tanh(c); gate order i, f, c, o along the 4F axis
//   block    : ConvLSTM2D -> MaxPooling2D(2x2, valid), TimeDistributed (clstm.py:23-40); batch norm is OFF in the
//              search graph (find_mask_kth.py:331: bn=False)
//   head     : flatten (NHWC order) of the last element or of the whole sequence -> dense (clstm.py:112-120)
//   Grad-CAM : per frame on the LAST ConvLSTM2D's output sequence: d(logit of the class)/d(output) as seen from
//              the layers above (tf.gradients w.r.t. the layer output, not through the recurrence),
//              cam_t = relu(sum_k mean_yx(grad_t,k) * out_t,k), normalised per frame or per sequence
//              (gradcam.py:42-99, 104-111).
// These are plain direct-convolution kernels (one thread per output element and unit, any kh x kw, fp32 FMAs): the
// extension is about semantics, not speed; the tuned ConvLSTM path of convlstm.hip is untouched.
// Layout: planar [b][t][channel][y][x]; the clip is the reference PyTorch layout NCTHW like everywhere else here
// (the Python wrapper accepts the TF layout [B,T,H,W,C] and permutes).
#include <algorithm>
#include <cstring>
#include <vector>

#include "ivf_common.h"

namespace ivf {

__device__ __forceinline__ float tf_rec_act(float z, int hard) {
  return hard ? fminf(fmaxf(0.2f * z + 0.5f, 0.f), 1.f) : 1.f / (1.f + expf(-z));
}
// derivative of the recurrent activation in terms of its input z (hard) or output y (sigmoid)
__device__ __forceinline__ float tf_rec_act_grad(float z, float y, int hard) {
  return hard ? ((z > -2.5f && z < 2.5f) ? 0.2f : 0.f) : y * (1.f - y);
}

struct TfGeom {
  int Cin, F, Hin, Win, Ho, Wo, Hp, Wp, kh, kw, s, pt, pl, ph, pw;   // pt/pl: front pads of the x conv; ph/pw: of the h conv
};

// Zx[b,t,g*F+j,y,x] = bias + sum wx[ky][kx][c][g*F+j] * X[b,t,c,y*s-pt+ky,x*s-pl+kx]; X by strides (elements)
__global__ __launch_bounds__(256) void tf_xconv_fwd_kernel(const float* __restrict__ X, const float* __restrict__ wx,
                                                           const float* __restrict__ bias, float* __restrict__ Zx, int B,
                                                           int T, long sB, long sC, long sT, TfGeom g) {
  const long plane = (long)g.Ho * g.Wo;
  const long total = (long)B * T * g.F * plane;
  const int G = 4 * g.F;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = i % g.Wo, y = (i / g.Wo) % g.Ho;
    const int j = (i / plane) % g.F;
    const long bt = i / (plane * g.F);
    const int t = bt % T, b = bt / T;
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = bias ? bias[q * g.F + j] : 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int yi = y * g.s - g.pt + ky;
      if ((unsigned)yi >= (unsigned)g.Hin) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int xi = x * g.s - g.pl + kx;
        if ((unsigned)xi >= (unsigned)g.Win) continue;
        for (int c = 0; c < g.Cin; ++c) {
          const float v = X[b * sB + c * sC + t * sT + (long)yi * g.Win + xi];
          const float* w = wx + ((long)(ky * g.kw + kx) * g.Cin + c) * G + j;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(w[q * g.F], v, acc[q]);
        }
      }
    }
    float* z = Zx + ((long)bt * G + j) * plane + (long)y * g.Wo + x;
#pragma unroll
    for (int q = 0; q < 4; ++q) z[(long)q * g.F * plane] = acc[q];
  }
}

// one cell step: z = Zx[t] + conv_same(h[t-1], wh); S[b,t,{zi,zf,zo (pre-activations), i,f,g,o,c}] and H
// S planes per unit: 0 i, 1 f, 2 g, 3 o, 4 c, 5 zi, 6 zf, 7 zo  (pre-activations kept for the hard-sigmoid derivative)
__global__ __launch_bounds__(256) void tf_step_fwd_kernel(const float* __restrict__ Zx, const float* __restrict__ wh,
                                                          float* __restrict__ S, float* __restrict__ Hs, int B, int T, int t,
                                                          int hard, TfGeom g) {
  const long plane = (long)g.Ho * g.Wo;
  const long total = (long)B * g.F * plane;
  const int G = 4 * g.F;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = i % g.Wo, y = (i / g.Wo) % g.Ho;
    const int j = (i / plane) % g.F;
    const int b = i / (plane * g.F);
    const long px = (long)y * g.Wo + x;
    float acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = Zx[(((long)b * T + t) * G + q * g.F + j) * plane + px];
    if (t > 0) {
      const float* hp = Hs + ((long)b * T + (t - 1)) * g.F * plane;
      for (int ky = 0; ky < g.kh; ++ky) {
        const int yy = y - g.ph + ky;
        if ((unsigned)yy >= (unsigned)g.Ho) continue;
        for (int kx = 0; kx < g.kw; ++kx) {
          const int xx = x - g.pw + kx;
          if ((unsigned)xx >= (unsigned)g.Wo) continue;
          for (int c = 0; c < g.F; ++c) {
            const float v = hp[(long)c * plane + (long)yy * g.Wo + xx];
            const float* w = wh + ((long)(ky * g.kw + kx) * g.F + c) * G + j;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_fmaf(w[q * g.F], v, acc[q]);
          }
        }
      }
    }
    const float ci = tf_rec_act(acc[0], hard), cf = tf_rec_act(acc[1], hard), co = tf_rec_act(acc[3], hard);
    const float cg = tanhf(acc[2]);
    const float cp = t > 0 ? S[((((long)b * T + (t - 1)) * 8 + 4) * g.F + j) * plane + px] : 0.f;
    const float cc = cf * cp + ci * cg;
    float* sp = S + (((long)b * T + t) * 8 * g.F + j) * plane + px;
    const long ps = (long)g.F * plane;
    sp[0 * ps] = ci; sp[1 * ps] = cf; sp[2 * ps] = cg; sp[3 * ps] = co; sp[4 * ps] = cc;
    sp[5 * ps] = acc[0]; sp[6 * ps] = acc[1]; sp[7 * ps] = acc[3];
    Hs[(((long)b * T + t) * g.F + j) * plane + px] = co * tanhf(cc);
  }
}

// MaxPooling2D(2x2, strides 2, valid), first maximum wins (scan order (0,0),(0,1),(1,0),(1,1))
__global__ void tf_pool_fwd_kernel(const float* __restrict__ Hs, float* __restrict__ Xp, unsigned char* __restrict__ arg,
                                   long frames_units, int Ho, int Wo, int Hp, int Wp) {
  const long total = frames_units * Hp * Wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xp = i % Wp, yp = (i / Wp) % Hp;
    const long fu = i / ((long)Wp * Hp);
    const float* hp = Hs + fu * (long)Ho * Wo;
    float best = 0.f;
    int bi = 0;
    for (int q = 0; q < 4; ++q) {
      const float v = hp[(long)(2 * yp + (q >> 1)) * Wo + 2 * xp + (q & 1)];
      if (q == 0 || v > best || v != v) { best = v; bi = q; }
    }
    Xp[i] = best;
    arg[i] = (unsigned char)bi;
  }
}

__global__ void tf_unpool_bwd_kernel(const float* __restrict__ dXp, const unsigned char* __restrict__ arg,
                                     float* __restrict__ dH, long frames_units, int Ho, int Wo, int Hp, int Wp) {
  const long total = frames_units * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = i % Wo, y = (i / Wo) % Ho;
    const long fu = i / ((long)Wo * Ho);
    const int yp = y >> 1, xp = x >> 1;
    float v = 0.f;
    if (yp < Hp && xp < Wp) {
      const long pi = (fu * Hp + yp) * Wp + xp;
      if (arg[pi] == (((y & 1) << 1) | (x & 1))) v = dXp[pi];
    }
    dH[i] = v;
  }
}

// flat[b][e*feat + (y*Wp + x)*F + j] = Xp[b, step_e, j, y, x]   (tf.layers.flatten of NHWC maps)
__global__ void tf_flatten_kernel(const float* __restrict__ Xp, float* __restrict__ flat, int B, int T, int F, int Hp, int Wp,
                                  int first_step, int nsteps, int to_flat) {
  const long feat = (long)F * Hp * Wp;
  const long total = (long)B * nsteps * feat;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = i % F;
    const long r = i / F;
    const int x = r % Wp, y = (r / Wp) % Hp;
    const int e = (r / ((long)Wp * Hp)) % nsteps;
    const int b = r / ((long)Wp * Hp * nsteps);
    const long src = ((((long)b * T + first_step + e) * F + j) * Hp + y) * Wp + x;
    const long dst = (long)b * nsteps * feat + (long)e * feat + ((long)y * Wp + x) * F + j;
    if (to_flat) flat[dst] = Xp[src];
    else const_cast<float*>(Xp)[src] = flat[dst];      // backward: scatter d(flat) into d(Xp)
  }
}

// logits[b][k] = bias[k] + sum_i flat[b][i] * W[i][k]  (tf.layers.dense kernel [in][out]); probs = softmax
__global__ __launch_bounds__(256) void tf_dense_fwd_kernel(const float* __restrict__ flat, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ logits,
                                                           float* __restrict__ probs, int n_in, int K) {
  extern __shared__ float red[];     // [256] partial sums, then K logits
  const int b = blockIdx.x;
  float* lg = red + 256;
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n_in; i += blockDim.x) s = __builtin_fmaf(flat[(long)b * n_in + i], W[(long)i * K + k], s);
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) lg[k] = red[0] + (bias ? bias[k] : 0.f);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float mx = -INFINITY;
    for (int k = 0; k < K; ++k) mx = fmaxf(mx, lg[k]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(lg[k] - mx);
    for (int k = 0; k < K; ++k) {
      logits[(long)b * K + k] = lg[k];
      probs[(long)b * K + k] = expf(lg[k] - mx) / s;
    }
  }
}

// upstream gradient on the head: of probs[target] (wrt_logit = 0: softmax backward) or of logits[target] (1: Grad-CAM's
// y_c, gradcam.py:42-47); dflat[b][i] = sum_k dl[k] W[i][k]
__global__ __launch_bounds__(256) void tf_dense_bwd_kernel(const float* __restrict__ W, const float* __restrict__ probs,
                                                           const int* __restrict__ target, float* __restrict__ score,
                                                           float* __restrict__ dflat, int n_in, int K, int wrt_logit) {
  extern __shared__ float dl[];
  const int b = blockIdx.x;
  const int tg = target[b];
  const float* pr = probs + (long)b * K;
  if (threadIdx.x == 0 && score) score[b] = pr[tg];
  for (int k = threadIdx.x; k < K; k += blockDim.x)
    dl[k] = wrt_logit ? (k == tg ? 1.f : 0.f) : pr[k] * ((k == tg ? 1.f : 0.f) - pr[tg]);
  __syncthreads();
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = __builtin_fmaf(dl[k], W[(long)i * K + k], s);
    dflat[(long)b * n_in + i] = s;
  }
}

// backward cell step t: dh = dHd[t] + conv_same^T(dZ[t+1], wh); gate derivatives -> dZ[t]; dC carried
__global__ __launch_bounds__(256) void tf_step_bwd_kernel(const float* __restrict__ dHd, const float* __restrict__ wh,
                                                          const float* __restrict__ S, float* __restrict__ dZ,
                                                          float* __restrict__ dC, int B, int T, int t, int hard, TfGeom g) {
  const long plane = (long)g.Ho * g.Wo;
  const long total = (long)B * g.F * plane;
  const int G = 4 * g.F;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = i % g.Wo, y = (i / g.Wo) % g.Ho;
    const int j = (i / plane) % g.F;          // hidden channel whose dh this thread owns
    const int b = i / (plane * g.F);
    const long px = (long)y * g.Wo + x;
    float dh = dHd[(((long)b * T + t) * g.F + j) * plane + px];
    if (t + 1 < T) {
      // forward: z[t+1][o][yo][xo] += wh[ky][kx][j][o] * h[t][j][yo-ph+ky][xo-pw+kx]  =>  yo = y + ph - ky
      const float* gz = dZ + ((long)b * T + (t + 1)) * G * plane;
      for (int ky = 0; ky < g.kh; ++ky) {
        const int yo = y + g.ph - ky;
        if ((unsigned)yo >= (unsigned)g.Ho) continue;
        for (int kx = 0; kx < g.kw; ++kx) {
          const int xo = x + g.pw - kx;
          if ((unsigned)xo >= (unsigned)g.Wo) continue;
          const float* w = wh + ((long)(ky * g.kw + kx) * g.F + j) * G;
          const float* gp = gz + (long)yo * g.Wo + xo;
          for (int o = 0; o < G; ++o) dh = __builtin_fmaf(w[o], gp[(long)o * plane], dh);
        }
      }
    }
    const float* sp = S + (((long)b * T + t) * 8 * g.F + j) * plane + px;
    const long ps = (long)g.F * plane;
    const float ci = sp[0], cf = sp[ps], cg = sp[2 * ps], co = sp[3 * ps], cc = sp[4 * ps];
    const float zi = sp[5 * ps], zf = sp[6 * ps], zo = sp[7 * ps];
    const float cp = t > 0 ? S[((((long)b * T + (t - 1)) * 8 + 4) * g.F + j) * plane + px] : 0.f;
    const float th = tanhf(cc);
    float* dcp = dC + ((long)b * g.F + j) * plane + px;
    const float dcc = dh * co * (1.f - th * th) + (t + 1 < T ? *dcp : 0.f);
    float* go = dZ + (((long)b * T + t) * G + j) * plane + px;
    go[0] = dcc * cg * tf_rec_act_grad(zi, ci, hard);
    go[ps] = dcc * cp * tf_rec_act_grad(zf, cf, hard);
    go[2 * ps] = dcc * ci * (1.f - cg * cg);
    go[3 * ps] = dh * th * tf_rec_act_grad(zo, co, hard);
    *dcp = dcc * cf;
  }
}

// dX[b,t,c,yi,xi] = sum wx[ky][kx][c][o] * dZ[b,t,o,yo,xo] over yo*s - pt + ky == yi, xo*s - pl + kx == xi
__global__ __launch_bounds__(256) void tf_xconv_bwd_kernel(const float* __restrict__ dZ, const float* __restrict__ wx,
                                                           float* __restrict__ dX, int B, int T, long sB, long sC, long sT,
                                                           TfGeom g) {
  const long plane = (long)g.Ho * g.Wo;
  const long total = (long)B * T * g.Cin * g.Hin * g.Win;
  const int G = 4 * g.F;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xi = i % g.Win, yi = (i / g.Win) % g.Hin;
    const int c = (i / ((long)g.Win * g.Hin)) % g.Cin;
    const long bt = i / ((long)g.Win * g.Hin * g.Cin);
    const int t = bt % T, b = bt / T;
    float acc = 0.f;
    for (int ky = 0; ky < g.kh; ++ky) {
      const int ny = yi + g.pt - ky;
      if (ny < 0 || ny % g.s) continue;
      const int yo = ny / g.s;
      if (yo >= g.Ho) continue;
      for (int kx = 0; kx < g.kw; ++kx) {
        const int nx = xi + g.pl - kx;
        if (nx < 0 || nx % g.s) continue;
        const int xo = nx / g.s;
        if (xo >= g.Wo) continue;
        const float* w = wx + ((long)(ky * g.kw + kx) * g.Cin + c) * G;
        const float* gp = dZ + (long)bt * G * plane + (long)yo * g.Wo + xo;
        for (int o = 0; o < G; ++o) acc = __builtin_fmaf(w[o], gp[(long)o * plane], acc);
      }
    }
    dX[b * sB + c * sC + t * sT + (long)yi * g.Win + xi] = acc;
  }
}

// per-frame Grad-CAM (gradcam.py:104-111): cam[b,t,y,x] = relu(sum_j mean_yx(grad[b,t,j]) * out[b,t,j,y,x]); one block
// per (b,t); also the frame maximum
__global__ __launch_bounds__(256) void tf_gradcam_kernel(const float* __restrict__ out, const float* __restrict__ grad,
                                                         float* __restrict__ cam, float* __restrict__ fmax, int F, int plane) {
  extern __shared__ float sm[];      // [F] weights + [256] reduction
  float* wts = sm;
  float* red = sm + F;
  const long bt = blockIdx.x;
  const float* gp = grad + bt * (long)F * plane;
  const float* op = out + bt * (long)F * plane;
  for (int j = 0; j < F; ++j) {
    float s = 0.f;
    for (int p = threadIdx.x; p < plane; p += blockDim.x) s += gp[(long)j * plane + p];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) wts[j] = red[0] / (float)plane;
    __syncthreads();
  }
  float mx = 0.f;
  for (int p = threadIdx.x; p < plane; p += blockDim.x) {
    float s = 0.f;
    for (int j = 0; j < F; ++j) s += wts[j] * op[(long)j * plane + p];     // (sequential fp32 sum, as the numpy loop)
    s = fmaxf(s, 0.f);
    cam[bt * plane + p] = s;
    mx = fmaxf(mx, s);
  }
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) fmax[bt] = red[0];
}

// cam / max (frame | sequence), then bilinear resize to (H, W), half-pixel centres, clamped source index
// (skimage.transform.resize order 1 restated: PARITY UNPINNED)
__global__ __launch_bounds__(256) void tf_cam_resize_kernel(const float* __restrict__ cam, const float* __restrict__ fmax,
                                                            float* __restrict__ outp, int T, int sh, int sw, int H, int W,
                                                            int per_frame) {
  const long bt = blockIdx.x;
  const int b = bt / T;
  float mx = fmax[bt];
  if (!per_frame) {
    mx = 0.f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, fmax[(long)b * T + t]);
  }
  const float* src = cam + bt * (long)sh * sw;
  const float sy = (float)sh / (float)H, sx = (float)sw / (float)W;
  for (int i = threadIdx.x; i < H * W; i += blockDim.x) {
    const int y = i / W, x = i % W;
    float fy = ((float)y + 0.5f) * sy - 0.5f, fx = ((float)x + 0.5f) * sx - 0.5f;
    fy = fminf(fmaxf(fy, 0.f), (float)(sh - 1));
    fx = fminf(fmaxf(fx, 0.f), (float)(sw - 1));
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = min(y0 + 1, sh - 1), x1 = min(x0 + 1, sw - 1);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const float top = src[y0 * sw + x0] * (1.f - wx) + src[y0 * sw + x1] * wx;
    const float bot = src[y1 * sw + x0] * (1.f - wx) + src[y1 * sw + x1] * wx;
    outp[bt * (long)H * W + i] = (top * (1.f - wy) + bot * wy) / mx;      // 0/0 -> NaN as numpy
  }
}

static inline int tf_grid(long total, int cap = 16384) {
  long gsz = (total + 255) / 256;
  return (int)(gsz > cap ? cap : (gsz ? gsz : 1));
}

struct TfLayerPlan {
  TfGeom g;
  size_t wx_off, bx_off, wh_off;                                  // floats in the weights arena (Keras layouts as given)
  size_t zx_off, S_off, H_off, X_off, dZ_off, dHd_off, dC_off, dX_off;   // floats in the workspace
  size_t arg_off;                                                  // bytes
  bool loaded = false;
};

}  // namespace ivf

using namespace ivf;

struct ivf_tfclstm {
  ivf_tfclstm_config cfg;
  std::vector<TfLayerPlan> L;
  int feat = 0, fc_in = 0, fc_steps = 0, fc_first = 0;
  size_t fcw_off = 0, fcb_off = 0, weights_floats = 0, ws_bytes = 0;
  size_t off_p, off_dp, off_flat, off_dflat, off_logits, off_probs, off_score, off_sig, off_terms, off_dreg, off_dsig,
      off_fbwd, off_cam, off_fmax;
  float* wa = nullptr;
  char* ws = nullptr;
  bool head_loaded = false;
  float* wsf(size_t off) const { return (float*)ws + off; }
  template <class T>
  T* at(size_t off) const { return (T*)(ws + off); }
};

extern "C" int ivf_tfclstm_create(const ivf_tfclstm_config* c, ivf_tfclstm_t** out) {
  IVF_CHECK_ARG(c && out, "tfclstm_create: null pointer");
  IVF_CHECK_ARG(c->B > 0 && c->C > 0 && c->T > 0 && c->T <= 64 && c->H > 0 && c->W > 0 && c->num_classes > 0,
                "tfclstm_create: bad clip geometry");
  IVF_CHECK_ARG(c->layers >= 1 && c->layers <= 8 && c->kh >= 1 && c->kw >= 1 && c->kh <= 9 && c->kw <= 9 && c->stride >= 1,
                "tfclstm_create: 1..8 layers, kernel up to 9 x 9, stride >= 1");
  IVF_CHECK_ARG(c->padding == 0 || c->padding == 1, "tfclstm_create: padding 0 ('valid') or 1 ('same')");
  ivf_tfclstm* n = new ivf_tfclstm();
  n->cfg = *c;
  size_t w = 0, fl = 0;
  auto takew = [&](size_t e) { size_t o = w; w += (e + 63) / 64 * 64; return o; };
  auto takef = [&](size_t e) { size_t o = fl; fl += (e + 63) / 64 * 64; return o; };
  const size_t B = c->B, T = c->T;
  int cin = c->C, H = c->H, W = c->W;
  for (int i = 0; i < c->layers; ++i) {
    TfLayerPlan p{};
    TfGeom& g = p.g;
    g.Cin = cin; g.F = c->units[i]; g.Hin = H; g.Win = W; g.kh = c->kh; g.kw = c->kw; g.s = c->stride;
    if (g.F <= 0) { delete n; set_error("tfclstm_create: layer %d has no units", i); return IVF_ERR_BAD_ARG; }
    if (c->padding == 0) {           // 'valid'
      g.Ho = (H - g.kh) / g.s + 1; g.Wo = (W - g.kw) / g.s + 1; g.pt = g.pl = 0;
    } else {                         // 'same' (TensorFlow rule: out = ceil(in / s), the extra pad cell goes to the back)
      g.Ho = (H + g.s - 1) / g.s; g.Wo = (W + g.s - 1) / g.s;
      g.pt = std::max((g.Ho - 1) * g.s + g.kh - H, 0) / 2;
      g.pl = std::max((g.Wo - 1) * g.s + g.kw - W, 0) / 2;
    }
    g.ph = (g.kh - 1) / 2; g.pw = (g.kw - 1) / 2;      // recurrent conv: 'same', stride 1
    g.Hp = g.Ho / 2; g.Wp = g.Wo / 2;
    if (H < g.kh || W < g.kw || g.Ho < 2 || g.Wo < 2) {
      delete n;
      set_error("tfclstm_create: layer %d: map %dx%d too small for the %dx%d kernel and the 2x2 pool", i, H, W, g.kh, g.kw);
      return IVF_ERR_BAD_ARG;
    }
    const size_t G = 4 * (size_t)g.F, plane = (size_t)g.Ho * g.Wo;
    p.wx_off = takew((size_t)g.kh * g.kw * cin * G);
    p.bx_off = takew(G);
    p.wh_off = takew((size_t)g.kh * g.kw * g.F * G);
    p.zx_off = takef(B * T * G * plane);
    p.S_off = takef(B * T * 8 * g.F * plane);
    p.H_off = takef(B * T * g.F * plane);
    p.X_off = takef(B * T * g.F * g.Hp * g.Wp);
    p.dZ_off = takef(B * T * G * plane);
    p.dHd_off = takef(B * T * g.F * plane);
    p.dC_off = takef(B * g.F * plane);
    p.dX_off = takef(B * T * g.F * g.Hp * g.Wp);
    n->L.push_back(p);
    cin = g.F; H = g.Hp; W = g.Wp;
  }
  n->feat = cin * H * W;
  n->fc_steps = c->only_last ? 1 : c->T;
  n->fc_first = c->only_last ? c->T - 1 : 0;
  n->fc_in = n->feat * n->fc_steps;
  n->fcw_off = takew((size_t)n->fc_in * c->num_classes);
  n->fcb_off = takew(c->num_classes);
  n->weights_floats = w;
  const size_t clip = (size_t)c->C * c->T * c->H * c->W;
  n->off_p = takef(B * clip) * 4;
  n->off_dp = takef(B * clip) * 4;
  n->off_flat = takef(B * n->fc_in) * 4;
  n->off_dflat = takef(B * n->fc_in) * 4;
  const TfGeom& top = n->L.back().g;
  n->off_cam = takef(B * T * (size_t)top.Ho * top.Wo) * 4;
  n->off_fmax = takef(B * T) * 4;
  size_t bytes = fl * 4;
  auto takeb = [&](size_t nb) { size_t o = bytes; bytes += align_up(nb, 256); return o; };
  for (auto& p : n->L) p.arg_off = takeb(B * T * p.g.F * p.g.Hp * p.g.Wp);
  const int K = c->num_classes;
  n->off_logits = takeb(B * K * 4);
  n->off_probs = takeb(B * K * 4);
  n->off_score = takeb(B * 4);
  n->off_sig = takeb(B * T * 4);
  n->off_terms = takeb(B * 2 * 4);
  n->off_dreg = takeb(B * T * 4);
  n->off_dsig = takeb(B * T * 4);
  n->off_fbwd = takeb(ivf_freeze_bwd_workspace_bytes((int)B, (int)T));
  n->ws_bytes = bytes;
  *out = n;
  return IVF_OK;
}

extern "C" void ivf_tfclstm_destroy(ivf_tfclstm_t* n) { delete n; }
extern "C" size_t ivf_tfclstm_weights_bytes(const ivf_tfclstm_t* n) { return n ? n->weights_floats * 4 : 0; }
extern "C" size_t ivf_tfclstm_workspace_bytes(const ivf_tfclstm_t* n) { return n ? n->ws_bytes : 0; }

extern "C" int ivf_tfclstm_bind(ivf_tfclstm_t* n, void* weights_arena, void* workspace) {
  IVF_CHECK_ARG(n && weights_arena && workspace, "tfclstm_bind: null pointer");
  IVF_CHECK_ARG(((uintptr_t)weights_arena & 255) == 0 && ((uintptr_t)workspace & 255) == 0,
                "tfclstm_bind: arenas must be 256-byte aligned");
  n->wa = (float*)weights_arena;
  n->ws = (char*)workspace;
  return IVF_OK;
}

extern "C" int ivf_tfclstm_layer_dims(const ivf_tfclstm_t* n, int layer, int* Ho, int* Wo, int* Hp, int* Wp, int* units) {
  IVF_CHECK_ARG(n && layer >= 0 && layer < (int)n->L.size(), "tfclstm_layer_dims: bad layer");
  const TfGeom& g = n->L[layer].g;
  if (Ho) *Ho = g.Ho;
  if (Wo) *Wo = g.Wo;
  if (Hp) *Hp = g.Hp;
  if (Wp) *Wp = g.Wp;
  if (units) *units = g.F;
  return IVF_OK;
}

extern "C" int ivf_tfclstm_load_layer(ivf_tfclstm_t* n, int layer, const float* kernel, const float* recurrent_kernel,
                                      const float* bias, ivf_stream_t stream) {
  IVF_CHECK_ARG(n && n->wa, "tfclstm_load_layer: bind first");
  IVF_CHECK_ARG(layer >= 0 && layer < (int)n->L.size() && kernel && recurrent_kernel, "tfclstm_load_layer: bad layer / null tensor");
  TfLayerPlan& p = n->L[layer];
  const TfGeom& g = p.g;
  hipStream_t s = (hipStream_t)stream;
  const size_t G = 4 * (size_t)g.F;
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.wx_off, kernel, (size_t)g.kh * g.kw * g.Cin * G * 4, hipMemcpyDeviceToDevice, s));
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.wh_off, recurrent_kernel, (size_t)g.kh * g.kw * g.F * G * 4, hipMemcpyDeviceToDevice, s));
  if (bias) IVF_CHECK_HIP(hipMemcpyAsync(n->wa + p.bx_off, bias, G * 4, hipMemcpyDeviceToDevice, s));
  else IVF_CHECK_HIP(hipMemsetAsync(n->wa + p.bx_off, 0, G * 4, s));
  p.loaded = true;
  return IVF_OK;
}

extern "C" int ivf_tfclstm_load_head(ivf_tfclstm_t* n, const float* dense_kernel, const float* dense_bias, ivf_stream_t stream) {
  IVF_CHECK_ARG(n && n->wa && dense_kernel && dense_bias, "tfclstm_load_head: bind first / null tensor");
  hipStream_t s = (hipStream_t)stream;
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + n->fcw_off, dense_kernel, (size_t)n->fc_in * n->cfg.num_classes * 4, hipMemcpyDeviceToDevice, s));
  IVF_CHECK_HIP(hipMemcpyAsync(n->wa + n->fcb_off, dense_bias, (size_t)n->cfg.num_classes * 4, hipMemcpyDeviceToDevice, s));
  n->head_loaded = true;
  return IVF_OK;
}

extern "C" int ivf_tfclstm_fc_inputs(const ivf_tfclstm_t* n) { return n ? n->fc_in : 0; }

namespace ivf {

static int tf_ready(const ivf_tfclstm* n, int b) {
  IVF_CHECK_ARG(n && n->wa && n->ws, "tfclstm: not bound");
  IVF_CHECK_ARG(b > 0 && b <= n->cfg.B, "tfclstm: batch %d outside [1,%d]", b, n->cfg.B);
  for (const auto& p : n->L) IVF_CHECK_ARG(p.loaded, "tfclstm: a layer's weights are not loaded");
  IVF_CHECK_ARG(n->head_loaded, "tfclstm: head weights not loaded");
  return IVF_OK;
}

static void tf_in_strides(const ivf_tfclstm* n, size_t i, long* sB, long* sC, long* sT) {
  const ivf_tfclstm_config& c = n->cfg;
  if (i == 0) {                                   // the clip, NCTHW
    *sC = (long)c.T * c.H * c.W; *sB = *sC * c.C; *sT = (long)c.H * c.W;
  } else {                                        // pooled maps of the layer below, [b][t][j][y][x]
    const TfGeom& q = n->L[i - 1].g;
    *sC = (long)q.Hp * q.Wp; *sT = *sC * q.F; *sB = *sT * c.T;
  }
}

static int tf_run_forward(ivf_tfclstm* n, const float* x, int b, float* logits, float* probs, hipStream_t s) {
  const ivf_tfclstm_config& c = n->cfg;
  const int T = c.T;
  for (size_t i = 0; i < n->L.size(); ++i) {
    const TfLayerPlan& p = n->L[i];
    const TfGeom& g = p.g;
    const float* in = i == 0 ? x : n->wsf(n->L[i - 1].X_off);
    long sB, sC, sT;
    tf_in_strides(n, i, &sB, &sC, &sT);
    const long plane = (long)g.Ho * g.Wo;
    hipLaunchKernelGGL(tf_xconv_fwd_kernel, dim3(tf_grid((long)b * T * g.F * plane)), dim3(256), 0, s, in, n->wa + p.wx_off,
                       n->wa + p.bx_off, n->wsf(p.zx_off), b, T, sB, sC, sT, g);
    IVF_CHECK_LAUNCH();
    for (int t = 0; t < T; ++t) {
      hipLaunchKernelGGL(tf_step_fwd_kernel, dim3(tf_grid((long)b * g.F * plane)), dim3(256), 0, s, n->wsf(p.zx_off),
                         n->wa + p.wh_off, n->wsf(p.S_off), n->wsf(p.H_off), b, T, t, c.recurrent_hard_sigmoid, g);
      IVF_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(tf_pool_fwd_kernel, dim3(tf_grid((long)b * T * g.F * g.Hp * g.Wp)), dim3(256), 0, s, n->wsf(p.H_off),
                       n->wsf(p.X_off), n->at<unsigned char>(p.arg_off), (long)b * T * g.F, g.Ho, g.Wo, g.Hp, g.Wp);
    IVF_CHECK_LAUNCH();
  }
  const TfGeom& top = n->L.back().g;
  float* flat = n->at<float>(n->off_flat);
  hipLaunchKernelGGL(tf_flatten_kernel, dim3(tf_grid((long)b * n->fc_in)), dim3(256), 0, s, n->wsf(n->L.back().X_off), flat, b,
                     T, top.F, top.Hp, top.Wp, n->fc_first, n->fc_steps, 1);
  IVF_CHECK_LAUNCH();
  float* lg = n->at<float>(n->off_logits);
  float* pr = n->at<float>(n->off_probs);
  hipLaunchKernelGGL(tf_dense_fwd_kernel, dim3(b), dim3(256), (256 + c.num_classes) * sizeof(float), s, flat,
                     n->wa + n->fcw_off, n->wa + n->fcb_off, lg, pr, n->fc_in, c.num_classes);
  IVF_CHECK_LAUNCH();
  const size_t nb = (size_t)b * c.num_classes * 4;
  if (logits) IVF_CHECK_HIP(hipMemcpyAsync(logits, lg, nb, hipMemcpyDeviceToDevice, s));
  if (probs) IVF_CHECK_HIP(hipMemcpyAsync(probs, pr, nb, hipMemcpyDeviceToDevice, s));
  return IVF_OK;
}

// head backward + un-pooling of the top layer: d(score)/d(output sequence of the last ConvLSTM2D) as seen from above
static int tf_head_backward(ivf_tfclstm* n, int b, const int* target, float* score, int wrt_logit, hipStream_t s) {
  const ivf_tfclstm_config& c = n->cfg;
  const TfLayerPlan& top = n->L.back();
  const TfGeom& g = top.g;
  float* dflat = n->at<float>(n->off_dflat);
  hipLaunchKernelGGL(tf_dense_bwd_kernel, dim3(b), dim3(256), c.num_classes * sizeof(float), s, n->wa + n->fcw_off,
                     n->at<float>(n->off_probs), target, score, dflat, n->fc_in, c.num_classes, wrt_logit);
  IVF_CHECK_LAUNCH();
  IVF_CHECK_HIP(hipMemsetAsync(n->wsf(top.dX_off), 0, (size_t)b * c.T * g.F * g.Hp * g.Wp * 4, s));
  hipLaunchKernelGGL(tf_flatten_kernel, dim3(tf_grid((long)b * n->fc_in)), dim3(256), 0, s, n->wsf(top.dX_off), dflat, b, c.T,
                     g.F, g.Hp, g.Wp, n->fc_first, n->fc_steps, 0);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(tf_unpool_bwd_kernel, dim3(tf_grid((long)b * c.T * g.F * g.Ho * g.Wo)), dim3(256), 0, s, n->wsf(top.dX_off),
                     n->at<unsigned char>(top.arg_off), n->wsf(top.dHd_off), (long)b * c.T * g.F, g.Ho, g.Wo, g.Hp, g.Wp);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

static int tf_run_backward(ivf_tfclstm* n, int b, const int* target, float* score, float* dx, hipStream_t s) {
  const ivf_tfclstm_config& c = n->cfg;
  const int T = c.T;
  IVF_PROPAGATE(tf_head_backward(n, b, target, score, 0, s));
  for (int i = (int)n->L.size() - 1; i >= 0; --i) {
    const TfLayerPlan& p = n->L[i];
    const TfGeom& g = p.g;
    const long plane = (long)g.Ho * g.Wo;
    if (i != (int)n->L.size() - 1) {
      hipLaunchKernelGGL(tf_unpool_bwd_kernel, dim3(tf_grid((long)b * T * g.F * plane)), dim3(256), 0, s, n->wsf(p.dX_off),
                         n->at<unsigned char>(p.arg_off), n->wsf(p.dHd_off), (long)b * T * g.F, g.Ho, g.Wo, g.Hp, g.Wp);
      IVF_CHECK_LAUNCH();
    }
    for (int t = T - 1; t >= 0; --t) {
      hipLaunchKernelGGL(tf_step_bwd_kernel, dim3(tf_grid((long)b * g.F * plane)), dim3(256), 0, s, n->wsf(p.dHd_off),
                         n->wa + p.wh_off, n->wsf(p.S_off), n->wsf(p.dZ_off), n->wsf(p.dC_off), b, T, t,
                         c.recurrent_hard_sigmoid, g);
      IVF_CHECK_LAUNCH();
    }
    float* out = i == 0 ? dx : n->wsf(n->L[i - 1].dX_off);
    long sB, sC, sT;
    tf_in_strides(n, i, &sB, &sC, &sT);
    hipLaunchKernelGGL(tf_xconv_bwd_kernel, dim3(tf_grid((long)b * T * g.Cin * g.Hin * g.Win)), dim3(256), 0, s, n->wsf(p.dZ_off),
                       n->wa + p.wx_off, out, b, T, sB, sC, sT, g);
    IVF_CHECK_LAUNCH();
  }
  return IVF_OK;
}

}  // namespace ivf

extern "C" int ivf_tfclstm_forward(ivf_tfclstm_t* n, const float* x, int b, float* logits, float* probs, ivf_stream_t stream) {
  IVF_PROPAGATE(tf_ready(n, b));
  IVF_CHECK_ARG(x, "tfclstm_forward: null clip");
  return tf_run_forward(n, x, b, logits, probs, (hipStream_t)stream);
}

extern "C" int ivf_tfclstm_backward(ivf_tfclstm_t* n, int b, const int* target, float* score, float* dx, ivf_stream_t stream) {
  IVF_PROPAGATE(tf_ready(n, b));
  IVF_CHECK_ARG(target && dx, "tfclstm_backward: need target and dx");
  return tf_run_backward(n, b, target, score, dx, (hipStream_t)stream);
}

extern "C" int ivf_tfclstm_perturbed_forward(ivf_tfclstm_t* n, const float* x, int b, const float* mask, float* probs,
                                             ivf_stream_t stream) {
  IVF_PROPAGATE(tf_ready(n, b));
  IVF_CHECK_ARG(x && mask, "tfclstm_perturbed_forward: bad args");
  const ivf_tfclstm_config& c = n->cfg;
  hipStream_t s = (hipStream_t)stream;
  float* P = n->at<float>(n->off_p);
  IVF_PROPAGATE(ivf_freeze_fwd(x, mask, P, b, c.C, c.T, c.H * c.W, 1, 0, s));       // the tf.scan recurrence, find_mask_kth.py:318-327
  return tf_run_forward(n, P, b, nullptr, probs, s);
}

// the loop of find_mask_kth.py:356-372,431-452: sigmoid, L1 + TV, tf.scan freeze, model, softmax score, tf.train.Adam
// (its epsilon sits beside sqrt(v) BEFORE the bias correction: eps_hat = eps / sqrt(1 - beta2^t) in torch's form)
extern "C" int ivf_tfclstm_search(ivf_tfclstm_t* n, const float* x, int b, const int* target, float* raw_mask, float* exp_avg,
                                  float* exp_avg_sq, float lam1, float lam2, float lr, float beta1, float beta2, float eps,
                                  int N, int first_step, float* traj, ivf_stream_t stream) {
  IVF_PROPAGATE(tf_ready(n, b));
  IVF_CHECK_ARG(x && target && raw_mask && exp_avg && exp_avg_sq && N >= 0 && first_step >= 1, "tfclstm_search: bad args");
  const ivf_tfclstm_config& c = n->cfg;
  hipStream_t s = (hipStream_t)stream;
  const int T = c.T, HW = c.H * c.W;
  float* sig = n->at<float>(n->off_sig);
  float* terms = n->at<float>(n->off_terms);
  float* dreg = n->at<float>(n->off_dreg);
  float* dsig = n->at<float>(n->off_dsig);
  float* score = n->at<float>(n->off_score);
  float* P = n->at<float>(n->off_p);
  float* dP = n->at<float>(n->off_dp);
  for (int it = 0; it < N; ++it) {
    IVF_PROPAGATE(ivf_mask_reg(raw_mask, b, T, lam1, lam2, sig, terms, dreg, s));
    IVF_PROPAGATE(ivf_freeze_fwd(x, sig, P, b, c.C, T, HW, 1, 0, s));
    IVF_PROPAGATE(tf_run_forward(n, P, b, nullptr, nullptr, s));
    IVF_PROPAGATE(tf_run_backward(n, b, target, score, dP, s));
    IVF_PROPAGATE(ivf_freeze_bwd(x, sig, dP, dsig, nullptr, b, c.C, T, HW, 1, 0, n->at<void>(n->off_fbwd), s));
    const int step = first_step + it;
    const float eps_hat = eps / sqrtf(1.f - powf(beta2, (float)step));
    IVF_PROPAGATE(ivf_search_step(raw_mask, sig, dsig, dreg, terms, score, exp_avg, exp_avg_sq,
                                  traj ? traj + (size_t)it * b * 4 : nullptr, b, T, step, lr, beta1, beta2, eps_hat, s));
  }
  return IVF_OK;
}

// gradcam.py:28-99 for b clips: forward of the UNPERTURBED clip as the graph sees it with mask_var = 0 (the caller passes
// the mask values; find_mask_kth feeds zeros, i.e. sigmoid(0) = 0.5 through the freeze recurrence), gradient of the class
// LOGIT w.r.t. the last ConvLSTM2D's output sequence, per-frame maps normalised per frame (1) or per sequence (0),
// resized to (out_h, out_w): cam [b,T,out_h,out_w]; probs [b,K] optional.
extern "C" int ivf_tfclstm_gradcam(ivf_tfclstm_t* n, const float* x, int b, const float* mask, const int* target,
                                   int per_frame, int out_h, int out_w, float* cam, float* probs, ivf_stream_t stream) {
  IVF_PROPAGATE(tf_ready(n, b));
  IVF_CHECK_ARG(x && target && cam && out_h > 0 && out_w > 0, "tfclstm_gradcam: bad args");
  const ivf_tfclstm_config& c = n->cfg;
  hipStream_t s = (hipStream_t)stream;
  if (mask) IVF_PROPAGATE(ivf_tfclstm_perturbed_forward(n, x, b, mask, probs, stream));
  else IVF_PROPAGATE(tf_run_forward(n, x, b, nullptr, probs, s));
  IVF_PROPAGATE(tf_head_backward(n, b, target, nullptr, 1, s));
  const TfLayerPlan& top = n->L.back();
  const TfGeom& g = top.g;
  const int plane = g.Ho * g.Wo;
  float* cm = n->at<float>(n->off_cam);
  float* fm = n->at<float>(n->off_fmax);
  hipLaunchKernelGGL(tf_gradcam_kernel, dim3(b * c.T), dim3(256), (g.F + 256) * sizeof(float), s, n->wsf(top.H_off),
                     n->wsf(top.dHd_off), cm, fm, g.F, plane);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(tf_cam_resize_kernel, dim3(b * c.T), dim3(256), 0, s, cm, fm, cam, c.T, g.Ho, g.Wo, out_h, out_w, per_frame);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}
