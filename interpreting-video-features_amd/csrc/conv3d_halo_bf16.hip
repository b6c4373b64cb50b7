// LDS-halo convolution (conv3d_halo_impl.h): the AM_BF16 instantiations -- bf16 activations / gradients in HBM staged
// as they are (one plane), weights hi/lo, two MFMA passes per k-step, bf16 stores.
#include "conv3d_halo_impl.h"

namespace ivf {
template int conv_halo_launch_variant_am<AM_BF16>(ConvKArgs& a, int v, hipStream_t s);
}
