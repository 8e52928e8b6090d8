// SDP passes (c4_sdp_wave.h) of the protein2genome family: reverse and forward sweep
#include "../c4_sdp_launch.h"
namespace c4sdp {
C4SDP_DEFINE_KERNELS(sdp_kernels_protein2genome, c4k::Protein2GenomeDesc, true)
}
