// SDP passes (c4_sdp_wave.h) of the affine family: reverse and forward sweep
#include "../c4_sdp_launch.h"
namespace c4sdp {
C4SDP_DEFINE_KERNELS(sdp_kernels_affine, c4k::AffineDesc, false)
}
