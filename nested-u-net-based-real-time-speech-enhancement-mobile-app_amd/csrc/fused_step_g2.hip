// Packed build of the fused frame-step kernel: TWO streams per workgroup (plan fused_plan_lstm_g2.inc) -- the same source as
// fused_step.hip, its own translation unit.  Chosen by the cost model of engine.cpp fused_setup (e.g. 300 .. 512 and 1536 streams on 256 CUs) or by nutls_create_plan.
#define FZ_STREAMS 2
#include "fused_step.hip"
