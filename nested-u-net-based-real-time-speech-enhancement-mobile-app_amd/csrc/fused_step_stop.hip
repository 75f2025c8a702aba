// Stop twin of the one-stream frame-step kernel: the same source as fused_step.hip -- no stamps, the production instruction stream -- with one
// scalar compare per op that ends the launch in front of op FzTa::skew.  nutls_profile_production times it for every op index; the differences
// are the un-instrumented kernel's time per op (the profiling build's stamps sit on the critical wave of every op and its workgroup 0 never
// sees the memory system loaded).  Its own translation unit, so that it compiles beside the others.
#define FZ_STOPAT 1
#define FZ_STOP_TWIN 1
#include "fused_step.hip"
