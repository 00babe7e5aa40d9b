// One translation unit per FFT size of k_frames (compiled with -DSP_INST_FRAMES_LOG2N=6..13): the per-n launcher and its 12 variants
// (I/Q or L/R split x six loaders: 1-, 2-, 3-, 4-, 8-byte samples one frame ahead, or the checked generic loader).
#include "sp_kernel_frames.h"
