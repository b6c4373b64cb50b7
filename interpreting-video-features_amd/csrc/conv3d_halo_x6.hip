// LDS-halo convolution (conv3d_halo_impl.h): the AM_X6 instantiations -- fp32 activations and weights split three
// ways (hi/mid/lo bf16), six MFMA passes per k-step: fp32-class products on the bf16 matrix cores.
#include "conv3d_halo_impl.h"

namespace ivf {
template int conv_halo_launch_variant_am<AM_X6>(ConvKArgs& a, int v, hipStream_t s);
}
