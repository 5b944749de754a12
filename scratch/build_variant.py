#!/usr/bin/env python3
"""Build a variant of libgpd.so for an A/B run: scratch/build_variant.py OUT.so [--main FLAGS...] [--policy FLAGS...] [--define D...]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from gym_pybullet_drones_amd import _native
out = sys.argv[1]
extra = {"--main": [], "--policy": [], "--define": []}
cur = None
for a in sys.argv[2:]:
    if a in extra:
        cur = a
    else:
        extra[cur].append(a)
objs = []
procs = []
for (unit, flags), key in zip(_native.UNITS, ("--main", "--policy")):
    obj = out + "." + unit + ".o"
    cmd = ["/opt/rocm/bin/hipcc"] + _native.COMMON_FLAGS + (extra[key] if extra[key] else flags) + ["-D" + d for d in extra["--define"]] + \
        ["-I", _native.INCLUDE, "-c", os.path.join(_native.CSRC, unit), "-o", obj]
    procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    objs.append(obj)
assert all(p.wait() == 0 for p in procs)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out], check=True)
for o in objs:
    os.remove(o)
print(out)
