#!/bin/bash
# Reproducer of the hipcc / clang 20 (ROCm 7.2) register-allocation defect behind the "VEL policy kernel reports truncated = 1 for
# every aviary" symptom (gpd.hip, comment above GpdPolicyLaunch; VERDICT r03 "next" #6).
#   1. the source as of commit 7333ea5, with the DSLPID policy kernels instantiated in the POLICY unit (default scheduler);
#   2. hipcc -S of that unit;
#   3. tests/isa_spill_check.py over its kernels: exactly one is flagged -- gpd_rollout_policy_kernel<true, 4, VEL, 1, false> --
#      and the three instructions that make the defect are printed.
# Needs only hipcc (no GPU).  usage: bash scratch/exp_r04/repro_vel_miscompile.sh [workdir]
set -e
REPO=$(cd "$(dirname "$0")/../.." && pwd)
W=${1:-/tmp/gpd_vel_repro}
rm -rf "$W" && mkdir -p "$W"
git -C "$REPO" archive 7333ea5 gym-pybullet-drones_amd/csrc include | tar -x -C "$W"
SRC=$W/gym-pybullet-drones_amd/csrc
# (the launcher of the DSLPID variants moves to the policy unit: the switch today's source carries as GPD_PID_POLICY_IN_POLICY_TU)
python3 - "$SRC/gpd.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
a = "#ifndef GPD_POLICY_TU\nvoid gpd_detail_launch_policy_pid("
assert a in s
open(p, "w").write(s.replace(a, "#ifdef GPD_POLICY_TU\nvoid gpd_detail_launch_policy_pid("))
PY
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -I "$W/include" -S --cuda-device-only \
    "$SRC/gpd_policy.hip" -o "$W/policy_default_sched.s" 2>/dev/null
/opt/rocm/bin/hipcc --version | head -2
python3 "$REPO/tests/isa_spill_check.py" "$W/policy_default_sched.s"
K='_ZN12_GLOBAL__N_125gpd_rollout_policy_kernelILb1ELi4ELi2ELi1ELb0E'
echo "---- the three instructions, in $W/policy_default_sched.s (kernel <PID, 4, VEL, 1, tanh>):"
awk -v k="$K" 'index($0, k) == 1 && /:/ {p = 1} p && /s_load_dwordx8 s\[48:55\], s\[0:1\], 0x19c|s_load_dwordx16 s\[36:51\], s\[0:1\], 0x18|v_writelane_b32 v164, s4[89], (29|30)$/ {print NR": "$0} p && /^.Lfunc_end/ {exit}' "$W/policy_default_sched.s"
