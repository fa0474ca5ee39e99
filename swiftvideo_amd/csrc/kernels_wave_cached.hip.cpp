// kernels_wave_cached.hip.cpp — the instantiations of tick_bgra_wave that take their per-layer geometry from a batch's tables (geom_cache.h;
// WaveStrip::setup_cached), compiled on their own so that the build's longest translation unit does not double.  Everything is in
// kernels_wave.hip.cpp.
#define CHV_WAVE_TU 1
#include "kernels_wave.hip.cpp"
