""""-fsanitize=address host build".  The HOST side of libgpd -- argument checks, error strings, struct plumbing,
launch arithmetic -- compiled from the four product units with AddressSanitizer + UndefinedBehaviorSanitizer (`hipcc --cuda-host-only`:
no device code, seconds), linked against a HIP runtime that launches nothing (tests/stubs/hip_stub.c) and driven through every entry of
include/gpd.h by a plain C program (tests/c/asan_host.c).  No GPU needed."""
import os
import re
import subprocess

import pytest

from conftest import REPO


def test_host_side_of_the_c_abi_under_asan_and_ubsan(tmp_path):
    from gym_pybullet_drones_amd import _native
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    clang = "/opt/rocm/lib/llvm/bin/clang"
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip("no hipcc / clang")
    san = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
    objs, procs = [], []
    for unit, _ in _native.UNITS:
        obj = str(tmp_path / unit.replace(".hip", ".host.o"))
        cmd = [hipcc, "-std=c++17", "--offload-arch=gfx950", "--cuda-host-only", "-fPIC"] + san + ["-I", _native.INCLUDE, "-c", os.path.join(_native.CSRC, unit), "-o", obj]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        objs.append(obj)
    for p in procs:
        out, _ = p.communicate()
        assert p.returncode == 0, out[-3000:]
    # the device images the host objects expect to be linked against: empty stand-ins (nothing is ever launched)
    undefined = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True, check=True).stdout
    fatbins = sorted(set(re.findall(r"__hip_fatbin_\w+", undefined)))
    assert len(fatbins) == len(_native.UNITS), fatbins
    stub_c = str(tmp_path / "fatbin_stubs.c")
    open(stub_c, "w").write("".join(f"const char {s}[16] = {{0}};\n" for s in fatbins))
    lib = str(tmp_path / "libgpd_asan.so")
    link = [clang + "++", "-shared", "-fPIC"] + san + objs + ["-x", "c", stub_c, os.path.join(REPO, "tests", "stubs", "hip_stub.c"), "-o", lib, "-ldl"]
    res = subprocess.run(link, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    exe = str(tmp_path / "asan_host")
    res = subprocess.run([clang] + san + ["-std=c11", "-I", _native.INCLUDE, os.path.join(REPO, "tests", "c", "asan_host.c"), lib, f"-Wl,-rpath,{tmp_path}", "-o", exe],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    print(run.stdout[-3000:])
    assert "AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0 and " 0 checks failed" in run.stdout, run.stdout[-3000:] + run.stderr[-2000:]
    assert run.stdout.count("\nok ") + run.stdout.startswith("ok ") >= 50
