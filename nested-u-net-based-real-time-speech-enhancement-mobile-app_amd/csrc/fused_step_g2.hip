// Packed build of the fused frame-step kernel: TWO streams per workgroup (plan fused_plan_lstm_g2.inc) -- the same source as
// fused_step.hip, its own translation unit.  Used by handles of B >= 512 streams (engine.cpp fused_setup).
#define FZ_STREAMS 2
#include "fused_step.hip"
