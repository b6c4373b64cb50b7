// LDS-halo convolution (conv3d_halo_impl.h): the AM_X3 instantiations (fp32 activations split hi/lo, 3 MFMA passes)
// and the mode-independent host side: applicability, variant dispatch by arithmetic mode, built-in variant choice.
#include "conv3d_halo_impl.h"

namespace ivf {

template int conv_halo_launch_variant_am<AM_X3>(ConvKArgs& a, int v, hipStream_t s);
// the other two modes are instantiated in their own translation units (conv3d_halo_x6.hip, conv3d_halo_bf16.hip)
extern template int conv_halo_launch_variant_am<AM_X6>(ConvKArgs& a, int v, hipStream_t s);
extern template int conv_halo_launch_variant_am<AM_BF16>(ConvKArgs& a, int v, hipStream_t s);

int conv_halo_supported(const ConvKArgs& a) {
  if (a.sT != 1 || a.sH != 1 || a.sW != 1) return 0;
  if (a.kT * a.kH * a.kW <= 1 || a.kT > 4 || a.kH > 4 || a.kW > 4) return 0;
  if (a.Cin % 8) return 0;
  return 1;
}

int conv_halo_num_variants() { return HALO_NUM_VARIANTS; }

int conv_halo_launch_variant(ConvKArgs& a, int math, int v, hipStream_t s) {
  switch (math) {
    case IVF_MATH_BF16X3: return conv_halo_launch_variant_am<AM_X3>(a, v, s);
    case IVF_MATH_BF16X6: return conv_halo_launch_variant_am<AM_X6>(a, v, s);
    case IVF_MATH_BF16ACT: return conv_halo_launch_variant_am<AM_BF16>(a, v, s);
  }
  set_error("conv3d_halo: arithmetic mode %d has no LDS-halo kernel", math);
  return IVF_ERR_UNSUPPORTED;
}

// default choice when the plan has not been tuned
int conv_halo_launch(ConvKArgs& a, int math, hipStream_t s) {
  // Output-channel tile width: every tile re-stages the halo, so weigh padded columns against
  // the number of tiles (a staging pass costs about as much as ~40 columns of MFMA work).
  static const int widths[5] = {192, 128, 96, 64, 32};
  int best = 32, best_cost = 1 << 30;
  for (int w : widths) {
    int cost = cdiv(a.Cout, w) * (w + 40);
    if (cost < best_cost) { best_cost = cost; best = w; }
  }
  const bool deep = a.To >= 4;   // 4-frame boxes when the map has them, else 2-frame boxes
  if (math == IVF_MATH_BF16X6) {
    // three activation planes: 16-channel chunks (4-frame boxes) or 2-frame boxes with at most 128 columns
    switch (best) {
      // (68 = the 8-wave form of tile 60: as fast or faster on data, and not at the register budget's edge)
      case 192: return conv_halo_launch_variant(a, math, deep ? 68 : 18, s);
      case 128: return conv_halo_launch_variant(a, math, deep ? 61 : 9, s);
      case 96: return conv_halo_launch_variant(a, math, deep ? 20 : 10, s);
      case 64: return conv_halo_launch_variant(a, math, deep ? 21 : 11, s);
      default: return conv_halo_launch_variant(a, math, deep ? 37 : 12, s);
    }
  }
  switch (best) {
    case 192: return conv_halo_launch_variant(a, math, deep ? 0 : 8, s);
    case 128: return conv_halo_launch_variant(a, math, deep ? 1 : 9, s);
    case 96: return conv_halo_launch_variant(a, math, deep ? 3 : 10, s);
    case 64: return conv_halo_launch_variant(a, math, deep ? 5 : 11, s);
    default: return conv_halo_launch_variant(a, math, deep ? 7 : 12, s);
  }
}

}  // namespace ivf
