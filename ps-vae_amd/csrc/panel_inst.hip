// One instantiation of the panel launcher per object file: -DPSVAE_TN=<accumulator tiles per wave>
// -DPSVAE_BK=<k-tile depth> (see the Makefile).
#include "gemm_panel.h"

namespace psvae_gemm {
template void launch_panel<PSVAE_TN, PSVAE_BK>(const GemmArgs &, bool, bool, dim3, hipStream_t);
}
