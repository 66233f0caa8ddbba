// The 8-wave shape of mlp_h2w.hip (16 x 16 x 32 MFMAs, two waves per SIMD): a second translation unit of the same source.
#define PNY_HW_NW 8
#include "mlp_h2w.hip"
