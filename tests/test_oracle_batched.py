"""The vectorised float64 oracle must reproduce the loop oracle (which is pinned to the reference)."""
import zlib

import numpy as np
import pytest

from conftest import golden, urdf
from oracle.aviary_oracle import ACT_DIM, OracleAviary
from oracle.batched_oracle import BatchedAviary


def _actions(rng, act, steps, E, D, hover_rpm, max_rpm):
    A = ACT_DIM[act]
    if act == "raw_rpm":
        return hover_rpm * (1 + 0.05 * rng.uniform(-1, 1, size=(steps, E, D, A)))
    if act == "pid":
        return np.array([0, 0, 0.5]) + 0.3 * rng.uniform(-1, 1, size=(steps, E, D, A))
    if act == "rpm":
        return 0.1 * rng.uniform(-1, 1, size=(steps, E, D, A))
    return rng.uniform(-1, 1, size=(steps, E, D, A))


@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
@pytest.mark.parametrize("act", ["rpm", "one_d_rpm", "pid", "vel", "one_d_pid", "raw_rpm"])
@pytest.mark.parametrize("flags,D,S", [(0, 1, 1), (7, 3, 2), (2, 2, 8), (5, 4, 1), (16, 1, 2), (31, 2, 4)])
def test_batched_equals_loop(model, act, flags, D, S):
    if model == "racer" and act in ("pid", "vel", "one_d_pid"):
        pytest.skip("DSLPID has no racer controller (BaseRLAviary.py:75-78)")
    rng = np.random.default_rng(zlib.crc32(repr((model, act, flags, D, S)).encode()))
    E, steps = 3, 25
    task = "none" if act == "raw_rpm" else ("hover" if D == 1 else "multihover")
    init_xyz = rng.uniform(-0.3, 0.3, size=(E, D, 3)) + np.array([0, 0, 0.5]) + \
        np.arange(D)[None, :, None] * np.array([0.05, 0.0, 0.25])
    init_rpy = rng.uniform(-0.2, 0.2, size=(E, D, 3))
    kw = dict(physics_flags=flags, pyb_freq=240, ctrl_freq=240 // S, act=act, task=task, pid_urdf_path=urdf("cf2x"))
    loops = [OracleAviary(urdf(model), model, num_drones=D, initial_xyzs=init_xyz[e], initial_rpys=init_rpy[e], **kw)
             for e in range(E)]
    bat = BatchedAviary(urdf(model), model, num_envs=E, num_drones=D, initial_xyzs=init_xyz, initial_rpys=init_rpy, **kw)
    if task == "multihover":
        np.testing.assert_allclose(bat.TARGET_POS[1], loops[1].TARGET_POS, rtol=0, atol=1e-15)
    acts = _actions(rng, act, steps, E, D, bat.C.HOVER_RPM, bat.C.MAX_RPM)
    for k in range(steps):
        obs, rew, term, trunc, _ = bat.step(acts[k])
        for e in range(E):
            o, r, te, tr = loops[e].step(acts[k, e])
            sv = np.array([loops[e]._getDroneStateVector(i) for i in range(D)])
            np.testing.assert_allclose(bat.state20()[e], sv, rtol=1e-7, atol=1e-9, err_msg=f"step {k} env {e}")
            np.testing.assert_allclose(bat.rpy_rates[e], loops[e].rpy_rates, rtol=1e-7, atol=1e-9)
            np.testing.assert_allclose(rew[e], r, rtol=1e-9, atol=1e-12)
            assert bool(term[e]) == te and bool(trunc[e]) == tr
            assert int(bat.step_counter[e]) == loops[e].step_counter
            if act in ("pid", "vel", "one_d_pid"):
                np.testing.assert_allclose(bat.pid.integral_rpy_e[e], np.array([c.integral_rpy_e for c in loops[e].ctrl]),
                                           rtol=1e-7, atol=1e-9)


def test_batched_against_golden_hover_pid():
    """Direct check of the vectorised oracle against the reference fixture (not only via the loop oracle).

    At 30 Hz the DSLPID attitude loop saturates its +-3200 torque clip and chatters, so the closed loop
    amplifies rounding differences (here: R* used directly instead of scipy's matrix->Euler->quat->matrix
    round trip, 3e-16 apart) by ~10x every 5 steps; only the first 30 steps are comparable at 1e-8.
    """
    g = golden("hover_pid")
    bat = BatchedAviary(urdf("cf2x"), "cf2x", 1, 1, pyb_freq=240, ctrl_freq=30, act="pid", task="hover")
    for k, a in enumerate(g["actions"][:30]):
        obs, rew, term, trunc, _ = bat.step(a[None])
        np.testing.assert_allclose(bat.state20()[0], g["state20"][k], rtol=1e-8, atol=1e-8)
        np.testing.assert_allclose(rew[0], g["reward"][k], rtol=1e-9)
        assert bool(trunc[0]) == bool(g["truncated"][k])


def test_auto_reset_semantics():
    """Same-step reset: obs returned is the reset obs, counters restart, PID state survives."""
    E = 4
    bat = BatchedAviary(urdf("cf2x"), "cf2x", E, 1, pyb_freq=240, ctrl_freq=30, act="one_d_pid", task="hover",
                        auto_reset=True)
    a = np.zeros((E, 1, 1)); a[0] = 1.0
    first_done = None
    for k in range(260):
        obs, rew, term, trunc, term_obs = bat.step(a)
        done = term | trunc
        if done.any() and first_done is None:
            first_done = k
            e = int(np.argmax(done))
            assert bat.step_counter[e] == 0
            np.testing.assert_allclose(obs[e, 0, :3], bat.INIT_XYZS[e, 0])
            assert not np.allclose(term_obs[e, 0, :3], bat.INIT_XYZS[e, 0])
            assert np.any(bat.pid.integral_pos_e[e] != 0)
    assert first_done is not None
