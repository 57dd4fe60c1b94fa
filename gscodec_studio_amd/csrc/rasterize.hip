// TEMPORARY stub: wave-per-tile kernels not written yet
#include "rasterize_common.h"
int32_t raster_wave_fwd(const RasterArgs &a, hipStream_t st) { return raster_ref_fwd(a, st); }
int32_t raster_wave_bwd(const RasterArgs &a, const RasterGradArgs &ga, hipStream_t st) { return raster_ref_bwd(a, ga, st); }
