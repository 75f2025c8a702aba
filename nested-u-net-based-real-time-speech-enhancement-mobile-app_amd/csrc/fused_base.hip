// The baseline variant's build of the fused frame-step kernel (dilated-dense bottlenecks, plan fused_plan_base.inc):
// the same source as fused_step.hip, its own translation unit.
#define FZ_BASE 1
#include "fused_step.hip"
