// Profiling build of the four-stream packed kernel (op-boundary stamps of workgroup 0).  NOT part of the default build (12 more minutes of
// compile time): NUTLS_BUILD_G4_PROF=1 python -m nunet_amd.build adds it; fused_step_g4.hip reaches it through a weak reference.
#define FZ_STREAMS 4
#define FZ_PROF 1
#include "fused_step.hip"
