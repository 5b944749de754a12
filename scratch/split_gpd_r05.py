#!/usr/bin/env python3
"""One-off (round 5, VERDICT r04 "next" #8): split csrc/gpd.hip (3 800 lines, compiled twice under a macro) into

    gpd_common.inc        includes, constants, math, the per-drone physics (substep / dslpid / map_action / env_step), load / store of the carry
    policy_kernel.inc     the policy rollout kernel template (instantiated by policy.hip; its DSLPID variants by step_rollout.hip)
    step_rollout.hip      gpd_step_kernel, gpd_rollout_kernel, gpd_rollout1_kernel, launch + argument checks, gpd_step / gpd_rollout / gpd_rollout_history
    policy.hip            gpd_rollout_policy
    swarm.hip             dwg_* (binning, force), gpd_swarm_* kernels, gpd_downwash_global / gpd_swarm_*
    abi.hip               reset / history / pid / state-vector / clock-probe kernels, RCCL, version / error / struct sizes / debug status

and write the timeline instrumentation (the GPD_EXP_TS* blocks) out of the product sources into a patch under scratch/.
Reads gpd.hip by line numbers of the commit it was written for; verifies anchors before cutting."""
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C = os.path.join(R, "gym-pybullet-drones_amd", "csrc")
src = open(os.path.join(C, "gpd.hip")).read().split("\n")
L = [None] + src           # 1-indexed


def find(pat, start=1, exact=False):
    for i in range(start, len(L)):
        if (L[i] == pat) if exact else L[i].startswith(pat):
            return i
    raise SystemExit(f"anchor not found: {pat!r} from {start}")


def block(a, b):           # lines a .. b-1
    return L[a:b]


a_includes = find("#include <hip/hip_runtime.h>")
a_lasterr_def = find("#ifndef GPD_POLICY_TU", a_includes)                   # std::string& gpd_detail_last_error() { ... } #endif
assert L[a_lasterr_def + 1].startswith("std::string& gpd_detail_last_error() {")
a_lasterr_end = find("#endif", a_lasterr_def) + 1
a_step_hdr = find("// gpd_step: ONE env step per launch.") - 1          # the dashed line above it
a_rollout_hdr = find("// gpd_rollout: K env steps per launch.") - 1
a_policy_hdr = find("// gpd_rollout_policy: K env steps per launch with an MLP policy IN the loop") - 1
a_main_only = find("#ifndef GPD_POLICY_TU", a_policy_hdr)                    # reset / hist kernels follow
a_state20 = find("__device__ __forceinline__ void state20_row(")
# state20_row's leading comment block
s20 = a_state20
while L[s20 - 1].startswith("//"):
    s20 -= 1
a_state20_end = find("}", a_state20, exact=True) + 1
a_dw_hdr = find("// Downwash inside ONE aviary of any size") - 1
a_pid_hdr = find("// standalone batched DSLPIDControl.computeControl") - 1
a_swarm_hdr = find("// GpdSwarm: the physics sub-step of ONE world's drones") - 1
a_clock_hdr = find("// Shader-clock probe") - 1
a_rccl_hdr = find("// RCCL, resolved at run time") - 1
a_launch = find("template <bool PID, bool EXT, int AW, int ACT>", a_rccl_hdr)
assert L[a_launch + 1].startswith("hipError_t launch_step(")
a_main_end = find("#endif  // !GPD_POLICY_TU", a_launch)
a_ns_end = find("}  // namespace", a_main_end)
a_polpid = find("#if defined(GPD_POLICY_TU) == defined(GPD_PID_POLICY_IN_POLICY_TU)")
a_polpid_end = find("#endif", a_polpid) + 1
a_extc = find('extern "C" {', a_polpid_end)
e_abi0 = find("int gpd_abi_version(void)", a_extc)
e_step = find("int gpd_step(const GpdParams* params", a_extc)
e_main_end1 = find("#endif  // !GPD_POLICY_TU", e_step)
e_pol = find("#ifdef GPD_POLICY_TU", e_main_end1)
e_pol_end = find("#endif  // GPD_POLICY_TU", e_pol)
e_hist = find("static int hist_args(", e_pol_end)
e_dw = find("int gpd_downwash_global(", e_hist)
e_reset = find("int gpd_reset(", e_dw)
e_ts = find("#ifdef GPD_EXP_TS", e_reset)
e_ts_end = find("#endif", e_ts) + 1
e_dbg = find("int gpd_debug_status(", e_ts_end)
e_end = find("#endif  // !GPD_POLICY_TU", e_dbg)
a_sizeof_swarm = find("int gpd_sizeof_swarm(void)", a_extc)

HDR = {
    "gpd_common.inc": "// gpd_common.inc -- what every translation unit of libgpd.so shares: constants, the fp32 math, the per-drone physics\n"
                      "// (substep<> / dslpid() / map_action<> / env_step<>), loading and storing a drone's carry.  Included, not compiled on its own.\n",
}

common = []
common += block(1, a_includes)                                   # the file comment
common += ["#pragma once"]
common += block(a_includes, a_lasterr_def)
common += block(a_lasterr_end, a_step_hdr)
common += ["// (BaseAviary._getDroneStateVector rows: shared by gpd_state_vectors and the one-world kernels)"] + block(s20, a_state20_end) + [""]
common += ["}  // namespace", ""]

step = ['// step_rollout.hip -- gpd_step / gpd_rollout / gpd_rollout_history: one env step per launch, K env steps per launch (DESIGN.md sections 3.1, 3.2)',
        '#include "gpd_common.inc"', '#include "policy_kernel.inc"', "", "namespace {", ""]
step += block(a_step_hdr, a_policy_hdr)
step += block(a_launch, a_main_end)
step += ["}  // namespace", ""]
polpid = block(a_polpid + 1, a_polpid_end - 1)
step += ["// The DSLPID variants of the policy kernel are instantiated HERE, under this unit's scheduler (gpd_common.inc says why);",
         "// GPD_PID_POLICY_IN_POLICY_TU (experiment / regression switch, tests/test_kernel_isa.py) moves them to policy.hip",
         "#ifndef GPD_PID_POLICY_IN_POLICY_TU"] + polpid + ["#endif", ""]
step += ['extern "C" {', ""] + block(e_step, e_main_end1) + ['}  // extern "C"', ""]

polk = ["// policy_kernel.inc -- the policy-in-the-loop rollout kernel (DESIGN.md section 3.7), a template: policy.hip instantiates the RPM variants",
        "// under the default scheduler, step_rollout.hip the DSLPID variants under max-ilp.  Included after gpd_common.inc.",
        "#pragma once", "namespace {", ""]
polk += block(a_policy_hdr, a_main_only)
polk += ["}  // namespace", ""]

pol = ['// policy.hip -- gpd_rollout_policy (DESIGN.md section 3.7)', '#include "gpd_common.inc"', '#include "policy_kernel.inc"', "",
       "#ifdef GPD_PID_POLICY_IN_POLICY_TU"] + polpid + ["#endif", "", 'extern "C" {', ""]
pol += block(e_pol + 1, e_pol_end)
pol += ['}  // extern "C"', ""]

swarm = ['// swarm.hip -- ONE aviary of any size: binning, the downwash force kernel with its wake lists, the one-world step (DESIGN.md section 3.4)',
         '#include "gpd_common.inc"', "", "namespace {", ""]
swarm += block(a_dw_hdr, a_pid_hdr)
swarm += block(a_swarm_hdr, a_clock_hdr)
swarm += ["}  // namespace", "", 'extern "C" {', ""]
swarm += block(a_sizeof_swarm, a_sizeof_swarm + 2)
swarm += block(e_dw, e_reset)
swarm += ['}  // extern "C"', ""]

abi = ['// abi.hip -- the small kernels (masked reset, action-history rows, batched DSLPID, state vectors, clock probe), RCCL, and the library-level',
       '// entries of the C ABI (version, last error, struct sizes, debug status)', '#include "gpd_common.inc"', ""]
abi += block(a_lasterr_def + 1, a_lasterr_end - 1) + ["", "namespace {", ""]
abi += block(a_main_only + 1, s20)
abi += block(a_state20_end, a_dw_hdr)
abi += block(a_pid_hdr, a_swarm_hdr)
abi += block(a_clock_hdr, a_launch)
abi += ["}  // namespace", "", 'extern "C" {', ""]
abi += block(e_abi0, a_sizeof_swarm)
abi += block(e_hist, e_dw)
abi += block(e_reset, e_ts)
abi += block(e_ts_end, e_end)
abi += ['}  // extern "C"', ""]

outs = {"gpd_common.inc": common, "policy_kernel.inc": polk, "step_rollout.hip": step, "policy.hip": pol, "swarm.hip": swarm, "abi.hip": abi}


def strip_ts(lines):
    """drop `#if(def) ...GPD_EXP_TS... / #endif` blocks (incl. nested #ifdef GPD_EXP_HWID) and GPD_POLICY_TU conditionals that are now moot"""
    out, skip, depth = [], False, 0
    for l in lines:
        if not skip and re.match(r"#\s*if", l) and "GPD_EXP_TS" in l:
            skip, depth = True, 1
            continue
        if skip:
            if re.match(r"#\s*if", l):
                depth += 1
            elif re.match(r"#\s*endif", l):
                depth -= 1
                if depth == 0:
                    skip = False
            continue
        out.append(l)
    return out


os.makedirs(os.path.join(R, "scratch", "exp_r05"), exist_ok=True)
tmp_with = os.path.join("/tmp", "gpd_split_with_ts")
tmp_without = os.path.join("/tmp", "gpd_split_without_ts")
for d in (tmp_with, tmp_without):
    os.makedirs(d, exist_ok=True)
for name, lines in outs.items():
    open(os.path.join(tmp_with, name), "w").write("\n".join(lines) + "\n")
    open(os.path.join(tmp_without, name), "w").write("\n".join(strip_ts(lines)) + "\n")
    open(os.path.join(C, name), "w").write("\n".join(strip_ts(lines)) + "\n")
patch = subprocess.run(["diff", "-ruN", "--label", "a", "--label", "b", tmp_without, tmp_with], capture_output=True, text=True).stdout
patch = patch.replace(tmp_without + "/", "a/gym-pybullet-drones_amd/csrc/").replace(tmp_with + "/", "b/gym-pybullet-drones_amd/csrc/")
open(os.path.join(R, "scratch", "exp_r05", "timeline_instrumentation.patch"), "w").write(patch)
print({k: len(v) for k, v in outs.items()}, "patch lines", patch.count("\n"))
