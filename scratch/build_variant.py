#!/usr/bin/env python3
"""Build a variant of libgpd.so for an A/B run:

    scratch/build_variant.py OUT.so [--define D...] [--flags-UNIT FLAGS...] [--patch FILE.patch]

--define       -D macros for every unit
--flags-UNIT   replace the extra flags of one unit (UNIT = step_rollout | policy | swarm | abi), e.g. --flags-policy -mllvm -amdgpu-sched-strategy=max-ilp
--patch        apply a patch (`patch -p1`, paths relative to the repo) to a COPY of csrc/ and include/ first: experiment scaffolding lives
               outside the product sources (the per-workgroup timeline instrumentation of rounds 3 / 4 -- `GPD_EXP_TS*`, read by
               scratch/exp_r04/force_timeline.py / step_timeline.py -- left them in round 5; `git show fd66d7c:gym_pybullet_drones_amd/csrc/gpd.hip`
               has its last form)"""
import os
import shutil
import subprocess
import sys
import tempfile

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gym_pybullet_drones_amd import _native  # noqa: E402

out = os.path.abspath(sys.argv[1])
extra = {"--define": [], "--patch": []}
cur = None
for a in sys.argv[2:]:
    if a.startswith("--flags-") or a in extra:
        cur = a
        extra.setdefault(cur, [])
    else:
        extra[cur].append(a)
csrc, include = _native.CSRC, _native.INCLUDE
tmp = None
if extra["--patch"]:
    tmp = tempfile.mkdtemp(prefix="gpd_variant_")
    os.makedirs(os.path.join(tmp, "gym_pybullet_drones_amd"))
    shutil.copytree(_native.CSRC, os.path.join(tmp, "gym_pybullet_drones_amd", "csrc"), ignore=shutil.ignore_patterns("*.so", "*.o"))
    shutil.copytree(_native.INCLUDE, os.path.join(tmp, "include"))
    for pf in extra["--patch"]:
        subprocess.run(["patch", "-p1", "-i", os.path.abspath(pf)], cwd=tmp, check=True)
    csrc, include = os.path.join(tmp, "gym_pybullet_drones_amd", "csrc"), os.path.join(tmp, "include")
objs, procs = [], []
for unit, flags in _native.UNITS:
    key = "--flags-" + unit.replace(".hip", "")
    obj = out + "." + unit + ".o"
    cmd = ["/opt/rocm/bin/hipcc"] + _native.COMMON_FLAGS + (extra[key] if key in extra else flags) + ["-D" + d for d in extra["--define"]] + \
        ["-I", include, "-c", os.path.join(csrc, unit), "-o", obj]
    procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs.append(obj)
for cmd, p in procs:
    o, _ = p.communicate()
    if p.returncode != 0:
        raise SystemExit("FAILED: " + " ".join(cmd) + "\n" + o[-4000:])
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
for o in objs:
    os.remove(o)
if tmp:
    shutil.rmtree(tmp, ignore_errors=True)
print(out)
