"""The C restatement of the oracle must reproduce the numpy oracle (which is pinned to the reference)."""
import zlib

import numpy as np
import pytest

from conftest import golden, urdf
from oracle.aviary_oracle import ACT_DIM
from oracle.batched_oracle import BatchedAviary
from oracle.c_oracle import CAviary


@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
@pytest.mark.parametrize("act", ["rpm", "one_d_rpm", "pid", "vel", "one_d_pid", "raw_rpm"])
@pytest.mark.parametrize("flags,D,S", [(0, 1, 1), (7, 3, 2), (2, 2, 8), (5, 4, 1), (16, 1, 2), (31, 2, 4)])
def test_c_equals_batched(model, act, flags, D, S):
    if model == "racer" and act in ("pid", "vel", "one_d_pid"):
        pytest.skip("no DSLPID controller for the racer")
    rng = np.random.default_rng(zlib.crc32(repr((model, act, flags, D, S)).encode()))
    E, steps = 5, 25
    task = "none" if act == "raw_rpm" else ("hover" if D == 1 else "multihover")
    xyz = rng.uniform(-1, 1, size=(E, D, 3)) * np.array([0.15, 0.15, 0.03]) + \
        np.arange(D)[None, :, None] * np.array([0.12, 0.0, 0.3]) + np.array([0, 0, 0.3])
    rpy = rng.uniform(-0.2, 0.2, size=(E, D, 3))
    kw = dict(num_envs=E, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=flags, pyb_freq=240,
              ctrl_freq=240 // S, act=act, task=task, pid_urdf_path=urdf("cf2x"), auto_reset=True)
    a_, b_ = BatchedAviary(urdf(model), model, **kw), CAviary(urdf(model), model, **kw)
    A = ACT_DIM[act]
    for k in range(steps):
        if act == "raw_rpm":
            a = a_.C.HOVER_RPM * (1 + 0.05 * rng.uniform(-1, 1, size=(E, D, A)))
        elif act == "pid":
            a = xyz + 0.2 * rng.uniform(-1, 1, size=(E, D, A))
        elif act == "rpm":
            a = 0.1 * rng.uniform(-1, 1, size=(E, D, A))
        else:
            a = rng.uniform(-1, 1, size=(E, D, A))
        o1, r1, t1, u1, to1 = a_.step(a)
        o2, r2, t2, u2, to2 = b_.step(a)
        np.testing.assert_allclose(b_.state20(), a_.state20(), rtol=1e-7, atol=1e-9, err_msg=f"step {k}")
        np.testing.assert_allclose(o2, o1, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(r2, r1, rtol=1e-9, atol=1e-12)
        np.testing.assert_array_equal(t2, t1)
        np.testing.assert_array_equal(u2, u1)
        np.testing.assert_array_equal(b_.step_counter, a_.step_counter)


def test_c_against_reference_fixture_hover_240():
    g = golden("hover_240")
    c = CAviary(urdf("cf2x"), "cf2x", 1, 1, initial_xyzs=g["init_xyz"], initial_rpys=g["init_rpy"], pyb_freq=240,
                ctrl_freq=240, act="rpm", task="hover")
    for k, a in enumerate(g["actions"]):
        obs, rew, term, trunc, _ = c.step(a[None])
        np.testing.assert_allclose(c.state20()[0], g["state20"][k], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(rew[0], g["reward"][k], rtol=1e-9)
        assert bool(trunc[0]) == bool(g["truncated"][k])


def test_c_auto_reset():
    c = CAviary(urdf("cf2x"), "cf2x", 3, 1, pyb_freq=240, ctrl_freq=30, act="one_d_rpm", task="hover", auto_reset=True)
    for k in range(245):
        obs, rew, term, trunc, tobs = c.step(np.zeros((3, 1, 1)))
        if k == 241:
            assert trunc.all() and (c.step_counter == 0).all()
            np.testing.assert_allclose(obs[:, 0, :3], c.INIT_XYZS[:, 0])


def test_all_pairs_downwash_for_a_sample_of_receivers_is_the_full_loop_restricted():
    """`orc_downwash_some` (every drone a source, some of them receivers: bench.py's bounded check of a world of 10^6 drones)
    returns the entries of `orc_downwash_all_pairs` bit for bit, on any thread count; bad indices are refused."""
    import ctypes
    from conftest import urdf
    from oracle import c_oracle
    rng = np.random.default_rng(5)
    xyz = rng.uniform([-20, -20, 0.5], [20, 20, 6.0], size=(1500, 3))
    full = c_oracle.downwash_all_pairs(urdf("cf2x"), xyz, threads=2)
    assert (np.abs(full) > 1e-6).mean() > 0.3
    recv = rng.choice(1500, 300, replace=False)
    for th in (1, 3):
        np.testing.assert_array_equal(c_oracle.downwash_some(urdf("cf2x"), xyz, recv, threads=th), full[recv])
    p = c_oracle.make_params(c_oracle.UrdfConstants(urdf("cf2x"), "cf2x"))
    pos = np.ascontiguousarray(xyz)
    bad = np.array([0, 1500], dtype=np.int32)
    out = np.zeros(2)
    assert c_oracle.lib().orc_downwash_some(ctypes.byref(p), 1500, c_oracle._ptr(pos), 2, c_oracle._ptr(bad), c_oracle._ptr(out)) == -1
