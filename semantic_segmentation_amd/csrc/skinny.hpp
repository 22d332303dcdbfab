// weight-streaming form of gs_conv_igemm / gs_conv_igemm_batch for launches with <= 128 output pixels (skinny.hip)
#pragma once
#include <hip/hip_runtime.h>

#include "gsseg.h"

// GS_OK: the launch was handled; GS_EUNSUPPORTED: not covered (nothing launched, no error string set)
int gs_skinny_try(int n, const GsConvGeom* const* g, const void* x, const void* const* w, void* y, const float* bias,
                  float* const* bn_partials, int act, int dtype, float* ws, int64_t ws_floats, hipStream_t s);
// BatchNorm partial rows ([rows][2][Cout]) such a launch writes (one per 16 output rows); 0: the geometry is not covered
int gs_skinny_stat_rows(const GsConvGeom* g);
