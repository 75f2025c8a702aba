// Host side of the fused frame-step kernel (fused_step.hip / fused_base.hip): packs the weight blob in the order of the
// variant's static plan (fused_plan_lstm.inc / fused_plan_base.inc) and reports the arena layout the plan addresses,
// which engine.cpp checks against its own.
#include <cstring>

#include "../../include/nutls.h"
#include "nutls_internal.hpp"
#include "fused_plan.hpp"

namespace nutls {
namespace {

const HostTensor* get(const WeightMap& w, const std::string& k, std::string* err) {
  auto it = w.find(k);
  if (it == w.end()) {
    if (err->empty()) *err = "weight tensor missing: " + k;
    return nullptr;
  }
  return &it->second;
}

namespace lstm_plan {
namespace fz {
using namespace ::nutls::fz;
#include "fused_plan_lstm.inc"
}  // namespace fz
#include "fused_host_impl.inc"
}  // namespace lstm_plan

// packed plan of the LSTM variant: two streams per workgroup (fused_step_g2.hip)
namespace lstm_g2_plan {
namespace fz {
using namespace ::nutls::fz;
#include "fused_plan_lstm_g2.inc"
}  // namespace fz
#include "fused_host_impl.inc"
}  // namespace lstm_g2_plan

// ... and four (fused_step_g4.hip)
namespace lstm_g4_plan {
namespace fz {
using namespace ::nutls::fz;
#include "fused_plan_lstm_g4.inc"
}  // namespace fz
#include "fused_host_impl.inc"
}  // namespace lstm_g4_plan

namespace base_plan {
namespace fz {
using namespace ::nutls::fz;
#include "fused_plan_base.inc"
}  // namespace fz
#include "fused_host_impl.inc"
}  // namespace base_plan

}  // namespace

#define FZ_BY_VARIANT(call) (variant == NUTLS_VARIANT_BASELINE ? base_plan::call : lstm_plan::call)
int fused_blob_floats(int variant) { return FZ_BY_VARIANT(fused_blob_floats()); }
int fused_num_ops(int variant) { return FZ_BY_VARIANT(fused_num_ops()); }
const char* fused_op_name(int variant, int i) { return FZ_BY_VARIANT(fused_op_name(i)); }
double fused_op_flops(int variant, int i) { return FZ_BY_VARIANT(fused_op_flops(i)); }
int fused_parity_stride(int variant) { return FZ_BY_VARIANT(fused_parity_stride()); }
int fused_arena_floats(int variant) { return FZ_BY_VARIANT(fused_arena_floats()); }
int fused_num_states(int variant) { return FZ_BY_VARIANT(fused_num_states()); }
int fused_num_pingpong(int variant) { return FZ_BY_VARIANT(fused_num_pingpong()); }
const char* fused_state_name(int variant, int i) { return FZ_BY_VARIANT(fused_state_name(i)); }
int fused_state_off(int variant, int i) { return FZ_BY_VARIANT(fused_state_off(i)); }
int fused_num_scratch(int variant) { return FZ_BY_VARIANT(fused_num_scratch()); }
const char* fused_scratch_name(int variant, int i) { return FZ_BY_VARIANT(fused_scratch_name(i)); }
int fused_scratch_off(int variant, int i) { return FZ_BY_VARIANT(fused_scratch_off(i)); }
int fused_ys_block(int variant) { return FZ_BY_VARIANT(fused_ys_block()); }
int fused_ys_off(int variant) { return FZ_BY_VARIANT(fused_ys_off()); }
void fused_lazy_table(int variant, std::vector<LazyCopy>* tab) { FZ_BY_VARIANT(fused_lazy_table(tab)); }
// (packed plans -- streams per workgroup > 1, LSTM variant -- share the arena layout of the one-stream plan; what differs is the tiling,
//  hence the blob and the layout of the carried partial sums)
#define FZ_BY_PLAN(call) \
  ((streams == 2 && variant == NUTLS_VARIANT_LSTM) ? lstm_g2_plan::call : ((streams == 4 && variant == NUTLS_VARIANT_LSTM) ? lstm_g4_plan::call : FZ_BY_VARIANT(call)))
bool fused_has_plan(int variant, int streams) { return streams == 1 || ((streams == 2 || streams == 4) && variant == NUTLS_VARIANT_LSTM); }
bool fused_ys_table(int variant, const WeightMap& wm, std::vector<YsOp>* ops, std::vector<float>* w, std::string* err, int streams) {
  return FZ_BY_PLAN(fused_ys_table(wm, ops, w, err));
}
int fused_pack_blob(int variant, const WeightMap& wm, std::vector<float>* out, std::string* err, int streams) {
  return FZ_BY_PLAN(fused_pack_blob(wm, out, err));
}
int fused_plan_blob_floats(int variant, int streams) { return FZ_BY_PLAN(fused_blob_floats()); }
int fused_plan_arena_floats(int variant, int streams) { return FZ_BY_PLAN(fused_arena_floats()); }
int fused_plan_parity_stride(int variant, int streams) { return FZ_BY_PLAN(fused_parity_stride()); }
int fused_plan_ys_off(int variant, int streams) { return FZ_BY_PLAN(fused_ys_off()); }
int fused_plan_ys_block(int variant, int streams) { return FZ_BY_PLAN(fused_ys_block()); }
int fused_plan_num_ops(int variant, int streams) { return FZ_BY_PLAN(fused_num_ops()); }
const char* fused_plan_op_name(int variant, int streams, int i) { return FZ_BY_PLAN(fused_op_name(i)); }
double fused_plan_op_flops(int variant, int streams, int i) { return FZ_BY_PLAN(fused_op_flops(i)); }

}  // namespace nutls
