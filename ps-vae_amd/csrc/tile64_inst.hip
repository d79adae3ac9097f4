// 64 x 64 tiles (2 x 2 waves, one accumulator tile each, 64-deep k-tiles in two LDS stages) of the
// panel kernel, for products whose 128-row panels would occupy a fraction of the chip: the
// per-molecule-row layers of a large batch (4096 rows) and every product of a small one.
#include "gemm_panel.h"

namespace psvae_gemm {

void launch_tile64(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    const bool mask = g.amask != nullptr, ones = g.rowsum != nullptr;
    const dim3 blk(THREADS);
#define PSVAE_TILE64(AKC, BKC, MASK, ONES) \
    hipLaunchKernelGGL((gemm_panel_kernel<1, TILE64_BK, AKC, BKC, MASK, ONES, 2>), grid, blk, 0, s, g)
    if (a_kc && b_kc) {
        if (mask) PSVAE_TILE64(true, true, true, false); else PSVAE_TILE64(true, true, false, false);
    } else if (b_kc) {
        if (mask) PSVAE_TILE64(false, true, true, false); else PSVAE_TILE64(false, true, false, false);
    } else if (a_kc) {
        if (mask && ones) PSVAE_TILE64(true, false, true, true);
        else if (mask) PSVAE_TILE64(true, false, true, false);
        else if (ones) PSVAE_TILE64(true, false, false, true);
        else PSVAE_TILE64(true, false, false, false);
    } else {
        if (mask && ones) PSVAE_TILE64(false, false, true, true);
        else if (mask) PSVAE_TILE64(false, false, true, false);
        else if (ones) PSVAE_TILE64(false, false, false, true);
        else PSVAE_TILE64(false, false, false, false);
    }
#undef PSVAE_TILE64
}

}  // namespace psvae_gemm
