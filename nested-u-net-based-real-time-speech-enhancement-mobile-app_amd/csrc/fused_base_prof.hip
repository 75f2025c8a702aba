// Profiling build of the baseline variant's fused kernel.
#define FZ_BASE 1
#define FZ_PROF 1
#include "ddb_fused.hpp"       // (only the baseline build needs the block: fused_step.hip does not include it)
#include "fused_step.hip"
