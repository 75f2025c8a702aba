// Profiling build of the fused frame-step kernel (workgroup 0 stamps wall_clock64() at every op boundary): the same
// source as fused_step.hip, compiled as its own translation unit so that the two builds run in parallel.
#define FZ_PROF 1
#include "fused_step.hip"
