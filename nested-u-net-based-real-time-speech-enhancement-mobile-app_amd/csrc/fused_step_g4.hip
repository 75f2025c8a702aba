// Packed build of the fused frame-step kernel: FOUR streams per workgroup (plan fused_plan_lstm_g4.inc) -- the same source as
// fused_step.hip, its own translation unit.  Used by handles of B >= 1024 streams (engine.cpp fused_setup).
#define FZ_STREAMS 4
#include "fused_step.hip"
