// Packed build of the fused frame-step kernel: FOUR streams per workgroup (plan fused_plan_lstm_g4.inc) -- the same source as
// fused_step.hip, its own translation unit.  Chosen by the cost model of engine.cpp fused_setup (e.g. 768, 1024, 2048 streams on 256 CUs) or by nutls_create_plan.
#define FZ_STREAMS 4
#include "fused_step.hip"
