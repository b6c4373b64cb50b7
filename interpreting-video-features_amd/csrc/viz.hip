// Visualisation blend: the reference's create_image_arrays /
// vizualize_results_on_gradcam (video_features_pytorch/visualisation.py:96-130, :35-64) as two
// HBM-bound kernels: per frame  original | JET(Grad-CAM) + original, normalised by its maximum |
// perturbed clip,  as one BGR uint8 strip [T][H][3W][3], then the red/green mask dots on the third
// panel.  Conversions follow numpy's float -> uint8 casts (truncation toward zero).
#include "ivf_common.h"

namespace ivf {

__device__ __forceinline__ unsigned char u8_trunc(float v) {
  // np.uint8(float32): C cast semantics for in-range values; NaN -> 0 as on the reference's x86 host
  if (!(v >= 0.f)) return 0;
  if (v >= 255.f) return 255;
  return (unsigned char)(int)v;
}

// per-frame maximum of LUT[u8(255 cam)][c] + img_bgr[c]  (visualisation.py:104-110); values are >= 0, so
// an integer atomicMax on the float bits is exact and order independent
__global__ void viz_frame_max_kernel(const float* __restrict__ clip, const float* __restrict__ cam,
                                     const unsigned char* __restrict__ lut, float* __restrict__ fmax, int T,
                                     int HW) {
  const int t = blockIdx.y;
  float m = 0.f;
  for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < HW; px += gridDim.x * blockDim.x) {
    const unsigned char idx = u8_trunc(255.f * cam[(size_t)t * HW + px]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // BGR channel c = clip plane 2 - c
      const float v = (float)lut[idx * 3 + c] + clip[((size_t)(2 - c) * T + t) * HW + px];
      m = fmaxf(m, v);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(fmax) + t, __float_as_int(m));
}

__global__ void viz_compose_kernel(const float* __restrict__ clip, const float* __restrict__ cam,
                                   const float* __restrict__ pert, const unsigned char* __restrict__ lut,
                                   const float* __restrict__ fmax, unsigned char* __restrict__ out, int T, int H,
                                   int W) {
  const size_t total = (size_t)T * H * 3 * W;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = i % (3 * W);
    const int h = (i / (3 * W)) % H;
    const int t = i / ((size_t)3 * W * H);
    const int panel = col / W, w = col - panel * W;
    const size_t px = (size_t)h * W + w;
    const size_t HW = (size_t)H * W;
    unsigned char bgr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float img = clip[((size_t)(2 - c) * T + t) * HW + px];
      if (panel == 0) {
        bgr[c] = u8_trunc(img);                                                      // np.uint8(input_data_img)
      } else if (panel == 1) {
        const unsigned char idx = u8_trunc(255.f * cam[(size_t)t * HW + px]);
        const float v = ((float)lut[idx * 3 + c] + img) / fmax[t];                   // cam / np.max(cam)
        bgr[c] = u8_trunc(255.f * v);
      } else {
        bgr[c] = u8_trunc(pert[((size_t)(2 - c) * T + t) * HW + px]);                // np.uint8(perturbed)[:, :, ::-1]
      }
    }
    unsigned char* o = out + i * 3;
    o[0] = bgr[0]; o[1] = bgr[1]; o[2] = bgr[2];
  }
}

// vizualize_results_on_gradcam :44-58: for frame i and dot j the last dot_h rows of columns
// [off + xs_j, off + xe_j) become 0 except channel ch_j = intensity (255 if i == j else 150).
__global__ void viz_dots_kernel(unsigned char* __restrict__ img, const float* __restrict__ mask, int T, int H,
                                int W3, int n, int dot_w, int dot_pad, int dot_h, int off) {
  const int rows = dot_h < H ? dot_h : H;
  const size_t total = (size_t)T * n * rows * dot_w;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int dx = i % dot_w;
    const int r = (i / dot_w) % rows;
    const int j = (i / ((size_t)dot_w * rows)) % n;
    const int t = i / ((size_t)dot_w * rows * n);
    const int col = off + j * (dot_w + dot_pad) + dx;
    if (col >= W3) continue;                              // numpy clips slices that run past the image
    const int h = H - rows + r;
    const int ch = mask[j] == 0.f ? 1 : 2;                // :86-89 (BGR: 1 = green, 2 = red)
    unsigned char* o = img + (((size_t)t * H + h) * W3 + col) * 3;
    const unsigned char inten = t == j ? 255 : 150;
    o[0] = ch == 0 ? inten : 0; o[1] = ch == 1 ? inten : 0; o[2] = ch == 2 ? inten : 0;
  }
}

static inline int vgrid(size_t total, int cap = 4096) {
  size_t g = (total + 255) / 256;
  return (int)(g > (size_t)cap ? cap : (g ? g : 1));
}

}  // namespace ivf

using namespace ivf;

extern "C" int ivf_viz_blend(const float* clip, const float* cam, const float* perturbed,
                             const unsigned char* lut_bgr, float* frame_max, unsigned char* out, int T, int H,
                             int W, ivf_stream_t stream) {
  IVF_CHECK_ARG(clip && cam && perturbed && lut_bgr && frame_max && out, "viz_blend: null pointer");
  IVF_CHECK_ARG(T > 0 && H > 0 && W > 0, "viz_blend: bad dims");
  hipStream_t s = (hipStream_t)stream;
  IVF_CHECK_HIP(hipMemsetAsync(frame_max, 0, (size_t)T * sizeof(float), s));
  hipLaunchKernelGGL(viz_frame_max_kernel, dim3(vgrid((size_t)H * W, 64), T), dim3(256), 0, s, clip, cam, lut_bgr,
                     frame_max, T, H * W);
  IVF_CHECK_LAUNCH();
  hipLaunchKernelGGL(viz_compose_kernel, dim3(vgrid((size_t)T * H * 3 * W)), dim3(256), 0, s, clip, cam, perturbed,
                     lut_bgr, frame_max, out, T, H, W);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}

extern "C" int ivf_viz_dots(unsigned char* img, const float* mask_snapped, int T, int H, int W3, int image_width,
                            int image_height, ivf_stream_t stream) {
  IVF_CHECK_ARG(img && mask_snapped && T > 0 && H > 0 && W3 > 0 && image_width > 0 && image_height > 0,
                "viz_dots: bad args");
  const int n = T;                                            // one dot per mask entry (:69)
  const int dot_w = image_width / (n + 4);                    // :70
  IVF_CHECK_ARG(dot_w > 0, "viz_dots: image_width (%d) too small for %d dots", image_width, n);
  const int dot_pad = (image_width - dot_w * n) / n;          // :71
  const int dot_h = image_height / 20;                        // :72
  if (dot_h <= 0) return IVF_OK;
  hipLaunchKernelGGL(viz_dots_kernel, dim3(vgrid((size_t)T * n * dot_h * dot_w)), dim3(256), 0, (hipStream_t)stream,
                     img, mask_snapped, T, H, W3, n, dot_w, dot_pad, dot_h, image_width * 2);
  IVF_CHECK_LAUNCH();
  return IVF_OK;
}
