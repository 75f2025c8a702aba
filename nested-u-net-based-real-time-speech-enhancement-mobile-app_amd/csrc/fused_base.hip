// The baseline variant's build of the fused frame-step kernel (dilated-dense bottlenecks, plan fused_plan_base.inc):
// the same source as fused_step.hip, its own translation unit.
#define FZ_BASE 1
#include "ddb_fused.hpp"       // (only the baseline build needs the block: fused_step.hip does not include it)
#include "fused_step.hip"
