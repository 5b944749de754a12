"""The drop-in boundary driven by a host that is not Python: tests/c/hover_host.c (plain C11 over include/gpd.h and the HIP
runtime's C API: its own hipMalloc'd buffers, gpd_reset, K x gpd_step, then the same K steps as one gpd_rollout) against the
Python mirror on the same library -- bit for bit -- and against the float64 oracle stepped K times (< 1e-4)."""
import os
import subprocess

import numpy as np
import pytest
import torch

from conftest import REPO, urdf
from test_gpu_parity import _actions, _core, _oracle_kin, _random_scene

SRC = os.path.join(REPO, "tests", "c", "hover_host.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(tmp):
    from gym_pybullet_drones_amd import _native
    csrc = os.path.dirname(_native.LIB_PATH)
    exe = os.path.join(str(tmp), "hover_host")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(REPO, "include"), "-I", os.path.join(ROCM, "include"),
           SRC, "-L", csrc, "-lgpd", "-L", os.path.join(ROCM, "lib"), "-lamdhip64", "-Wl,-rpath," + csrc, "-Wl,-rpath," + os.path.join(ROCM, "lib"),
           "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_the_c_host_builds_as_plain_c11(tmp_path):
    """No GPU needed: the program compiles warning-free with gcc -std=c11 against the public header and links the library."""
    from gym_pybullet_drones_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build(verbose=False)
    assert os.path.exists(_build(tmp_path))


def _blocks(buf, E, D):
    """the two result blocks of hover_host's output file"""
    N = E * D
    out, off = [], 0
    for _ in range(2):
        b = {}
        for name, dt, n in (("kin", np.float32, 13 * N), ("obs12", np.float32, 12 * N), ("reward", np.float32, E), ("terminated", np.uint8, E),
                            ("truncated", np.uint8, E), ("step_counter", np.int32, E)):
            b[name] = np.frombuffer(buf, dtype=dt, count=n, offset=off)
            off += n * np.dtype(dt).itemsize
        # (the host's state block has ld = N; GpdState.kin is four planes since ABI 9: back to the logical [13][N] rows)
        from gym_pybullet_drones_amd.engine import kin_rows_from_planes
        b["kin"], b["obs12"] = kin_rows_from_planes(b["kin"], N), b["obs12"].reshape(N, 12)
        out.append(b)
    assert off == len(buf)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("act,flags,D,S,model,E,K", [("rpm", 0, 1, 1, "cf2x", 1000, 12), ("rpm", 7, 4, 2, "cf2x", 300, 8),
                                                      ("one_d_rpm", 2, 1, 8, "cf2p", 257, 6), ("pid", 0, 2, 1, "cf2x", 129, 10)])
def test_a_c_host_gets_the_python_mirrors_bits_and_the_oracles_trajectory(gpu_device, tmp_path, act, flags, D, S, model, E, K):
    from oracle.batched_oracle import BatchedAviary
    rng = np.random.default_rng(E + K)
    task = "hover" if D == 1 else "multihover"
    xyz, rpy = _random_scene(rng, E, D)
    if D > 1:   # a well-conditioned downwash scene (tests/test_gpu_rollout.py::test_rollout_against_oracle)
        xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.12, 0.0, 0.3]) + np.array([0, 0, 0.8])
        rpy = rng.uniform(-0.05, 0.05, size=(E, D, 3))
    xyz, rpy = xyz.astype(np.float32).astype(np.float64), rpy.astype(np.float32).astype(np.float64)
    orc = BatchedAviary(urdf(model), model, num_envs=E, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=flags, pyb_freq=240,
                        ctrl_freq=240 // S, act=act, task=task, pid_urdf_path=urdf(model))
    core = _core(model, E, D, flags, S, act, task, xyz, rpy, gpu_device, auto_reset=False, target=orc.TARGET_POS)
    if act == "pid":
        acts = (xyz + np.array([0, 0, 0.2]) + 0.05 * rng.uniform(-1, 1, size=(K, E, D, 3))).astype(np.float32)
    else:
        acts = _actions(rng, act, (K, E, D), core.P.HOVER_RPM).astype(np.float32)
        acts = (0.02 * acts if act == "rpm" else acts).astype(np.float32)
    A = acts.shape[-1]
    # what the Python mirror would pass, as bytes
    head = np.array([E, D, A, K, core.init_pose.shape[0], core.target.shape[0], int(core.last_rpm is not None), int(core.pid is not None)], dtype=np.int32)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(head.tobytes()); f.write(bytes(core._params)); f.write(bytes(core._cfg))
        f.write(core.init_pose.cpu().numpy().astype(np.float32).tobytes()); f.write(core.target.cpu().numpy().astype(np.float32).tobytes())
        f.write(np.ascontiguousarray(acts).tobytes())
    res = subprocess.run([_build(tmp_path), str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    stepped, rolled = _blocks(open(fout, "rb").read(), E, D)
    # the Python mirror, same library, same calls
    core.reset(reset_pid=True)
    dev_acts = torch.as_tensor(acts, device=gpu_device)
    for k in range(K):
        core.step(dev_acts[k])
    for blk in (stepped, rolled):
        for name in ("kin", "obs12", "reward", "terminated", "truncated", "step_counter"):
            mine = getattr(core, name)
            mine = (mine[:, :E * D] if name == "kin" else mine).cpu().numpy()
            assert np.array_equal(blk[name].reshape(mine.shape), mine.astype(blk[name].dtype)), name
    # the oracle, stepped K times from the same poses
    for k in range(K):
        orc.step(acts[k].astype(np.float64))
    ref = _oracle_kin(orc)
    err = np.abs(stepped["kin"].astype(np.float64) - ref) / np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
    assert err.max() < 1e-4, err.max(axis=1)
    np.testing.assert_array_equal(stepped["step_counter"], orc.step_counter)
