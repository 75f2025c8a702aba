// Profiling build of the baseline variant's fused kernel.
#define FZ_BASE 1
#define FZ_PROF 1
#include "fused_step.hip"
