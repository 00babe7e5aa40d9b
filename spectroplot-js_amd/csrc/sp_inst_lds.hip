// One translation unit per FFT size of k_lds_r16 (compiled with -DSP_INST_LDS_LOG2N=6..13): the per-n launcher and its 12 variants.
#include "sp_kernel_lds.h"
