// One translation unit per FFT size of k_frames (compiled with -DSP_INST_FRAMES_LOG2N=6..13): the per-n launcher and its 12 variants.
#include "sp_kernel_frames.h"
