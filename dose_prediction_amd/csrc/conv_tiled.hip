// LDS-tiled convolution kernels for the hot stride-1 layers (placeholder dispatch: not applicable -> generic path).
#include "common.h"
int dp_conv3d_tiled_try(const void*, int, const void*, const float*, void*, int, int, int, int, int, int, int, int, int, int, int, int, int, int,
                        int, int, void*) { return -1; }
int dp_wgrad_tiled_try(const void*, int, const void*, int, float*, int, int, int, int, int, int, int, int, int, int, int, int, int, int, int,
                       int64_t, int64_t, int64_t, int, void*) { return -1; }
