// The f16x2 fused MLP kernel in its SPLIT shape: 4 waves per workgroup, a wave owns 128 features x 32 samples (4 x 1 accumulator
// tiles), 32-sample tiles, 65 KiB of LDS -- two workgroups per CU, so that one's gather / epilogue / prologue phases run
// underneath the other's GEMMs (in the 8-wave shape of mlp_h2.hip all waves of a CU are in the same phase and the matrix pipe
// idles 31 % of the time).  Price: the weight stream per sample doubles (1 MiB per 512 x 512 layer per 32 samples).  Same source
// as mlp_h2.hip, compiled with another tile shape (mlp_h2_core.h); render launches only.
#define PNY_H2_SPLIT 1
#define PNY_H2_NT 4
#define PNY_H2_MT 1
#include "mlp_h2.hip"
