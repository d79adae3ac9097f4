// The latency form of the panel kernel (32 x 32 tile per workgroup, its four waves splitting K): every product
// of a small batch.
#include "gemm_panel.h"

namespace psvae_gemm {

void launch_tile32(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    const bool mask = g.amask != nullptr, ones = g.rowsum != nullptr;
    const dim3 blk(THREADS);
#define PSVAE_DIRECT(AKC, BKC, MASK, ONES) \
    hipLaunchKernelGGL((gemm_panel_kernel<1, TILE64_BK, AKC, BKC, MASK, ONES, 1, false, TILE64_KG>), grid, blk, 0, s, g)
    if (a_kc && b_kc) {
        if (mask) PSVAE_DIRECT(true, true, true, false); else PSVAE_DIRECT(true, true, false, false);
    } else if (b_kc) {
        if (mask) PSVAE_DIRECT(false, true, true, false); else PSVAE_DIRECT(false, true, false, false);
    } else if (a_kc) {
        if (mask && ones) PSVAE_DIRECT(true, false, true, true);
        else if (mask) PSVAE_DIRECT(true, false, true, false);
        else if (ones) PSVAE_DIRECT(true, false, false, true);
        else PSVAE_DIRECT(true, false, false, false);
    } else {
        if (mask && ones) PSVAE_DIRECT(false, false, true, true);
        else if (mask) PSVAE_DIRECT(false, false, true, false);
        else if (ones) PSVAE_DIRECT(false, false, false, true);
        else PSVAE_DIRECT(false, false, false, false);
    }
#undef PSVAE_DIRECT
}

}  // namespace psvae_gemm
