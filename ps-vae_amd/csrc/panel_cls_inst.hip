// The panel kernel's forward form with the class layer in its epilogue (gemm_panel.h, CLS), for the column-tile
// widths 5 and 6 x 32: an object of its own so that it compiles beside the other panel variants.
#include "gemm_panel.h"

namespace psvae_gemm {
void launch_panel_classes(const GemmArgs &g, int tn, dim3 grid, hipStream_t s) {
    if (tn == 6)
        launch_panel_classes_tn<6>(g, grid, s);
    else
        launch_panel_classes_tn<5>(g, grid, s);
}
}  // namespace psvae_gemm
