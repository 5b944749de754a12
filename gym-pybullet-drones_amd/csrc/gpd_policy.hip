// gpd_policy.hip -- second translation unit of libgpd.so: gpd_rollout_policy and its kernel (gpd.hip, section "policy in the
// loop"), compiled WITHOUT -amdgpu-sched-strategy=max-ilp.  The physics device functions it inlines are the same source as in
// gpd.hip and every fused multiply-add in them is explicit (-ffp-contract=off), so the instruction ORDER differs between the two
// units but no result bit does (tests/test_gpu_policy.py::test_policy_rollout_is_bitwise_stepping_its_actions).
#define GPD_POLICY_TU 1
#include "gpd.hip"
