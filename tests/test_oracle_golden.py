"""Pin the CPU oracle against fixtures produced by the REFERENCE'S OWN PYTHON
(tests/golden/make_golden.py ran the unmodified reference classes over oracle/pybullet_shim).

float64 on both sides, identical operation order up to numpy-vs-scalar association, so the
tolerance is a few ulps amplified by the (sometimes tumbling) dynamics: rtol 1e-9 / atol 1e-11.
"""
import numpy as np
import pytest

from conftest import golden, urdf
from oracle.aviary_oracle import OracleAviary, OracleDSLPID, UrdfConstants

RTOL, ATOL = 1e-9, 1e-11


def _run_rl(g, act, task, num_drones=1, ctrl_freq=30, check_ctrl=False):
    init_xyz = g["init_xyz"] if "init_xyz" in g.files else None
    init_rpy = g["init_rpy"] if "init_rpy" in g.files else None
    env = OracleAviary(urdf("cf2x"), "cf2x", num_drones=num_drones, initial_xyzs=init_xyz, initial_rpys=init_rpy,
                       pyb_freq=240, ctrl_freq=ctrl_freq, act=act, task=task)
    obs0 = env.reset()
    np.testing.assert_allclose(obs0[:, :12].astype(np.float32), g["obs0"][:, :12], rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(obs0[:, 12:], g["obs0"][:, 12:])
    for k, a in enumerate(g["actions"]):
        obs, rew, term, trunc = env.step(a)
        sv = np.array([env._getDroneStateVector(i) for i in range(num_drones)])
        np.testing.assert_allclose(sv, g["state20"][k], rtol=RTOL, atol=ATOL, err_msg=f"state20 step {k}")
        np.testing.assert_allclose(env.rpy_rates, g["rpy_rates"][k], rtol=RTOL, atol=ATOL)
        # the reference casts the 12 kinematic floats to float32 before stacking the action tail
        ref_obs = g["obs"][k]
        np.testing.assert_allclose(obs[:, :12].astype(np.float32), ref_obs[:, :12], rtol=1e-6, atol=1e-7)
        np.testing.assert_array_equal(obs[:, 12:], ref_obs[:, 12:])
        np.testing.assert_allclose(rew, g["reward"][k], rtol=1e-9, atol=1e-12)
        assert term == bool(g["terminated"][k]), f"terminated step {k}"
        assert trunc == bool(g["truncated"][k]), f"truncated step {k}"
        assert env.step_counter == int(g["step_counter"][k])
        if check_ctrl:
            np.testing.assert_allclose(np.array([c.integral_pos_e for c in env.ctrl]), g["integral_pos_e"][k], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.array([c.last_rpy for c in env.ctrl]), g["last_rpy"][k], rtol=RTOL, atol=ATOL)
            np.testing.assert_allclose(np.array([c.integral_rpy_e for c in env.ctrl]), g["integral_rpy_e"][k], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("name,act,ctrl_freq,ctrl", [
    ("hover_rpm", "rpm", 30, False),
    ("hover_rpm_tilt", "rpm", 30, False),
    ("hover_one_d_rpm", "one_d_rpm", 30, False),
    ("hover_time_trunc", "one_d_rpm", 30, False),
    ("hover_pid", "pid", 30, True),
    ("hover_vel", "vel", 30, True),
    ("hover_one_d_pid", "one_d_pid", 30, True),
    ("hover_240", "rpm", 240, False),
])
def test_hover_matches_reference(name, act, ctrl_freq, ctrl):
    _run_rl(golden(name), act, "hover", 1, ctrl_freq, ctrl)


def test_time_truncation_on_step_242():
    g = golden("hover_time_trunc")
    assert int(np.argmax(g["truncated"])) + 1 == 242


@pytest.mark.parametrize("name,act,n,ctrl", [("multihover_rpm", "rpm", 2, False), ("multihover_pid", "pid", 3, True)])
def test_multihover_matches_reference(name, act, n, ctrl):
    g = golden(name)
    env = OracleAviary(urdf("cf2x"), "cf2x", num_drones=n, act=act, task="multihover", ctrl_freq=30)
    np.testing.assert_allclose(env.TARGET_POS, g["target_pos"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(env.INIT_XYZS, g["init_xyzs"], rtol=0, atol=1e-15)
    _run_rl(g, act, "multihover", n, 30, ctrl)


@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
def test_ctrl_tumble_matches_reference(model):
    g = golden("ctrl_tumble_" + model)
    n = g["init_xyzs"].shape[0]
    env = OracleAviary(urdf(model), model, num_drones=n, initial_xyzs=g["init_xyzs"], initial_rpys=g["init_rpys"],
                       pyb_freq=240, ctrl_freq=240, act="raw_rpm", task="none")
    assert env.C.MAX_RPM == pytest.approx(float(g["max_rpm"]), rel=1e-15)
    assert env.C.HOVER_RPM == pytest.approx(float(g["hover_rpm"]), rel=1e-15)
    for k in range(g["rpm"].shape[0]):
        obs, _, _, _ = env.step(g["rpm"][k])
        np.testing.assert_allclose(obs, g["obs"][k], rtol=RTOL, atol=ATOL, err_msg=f"step {k}")
        np.testing.assert_allclose(env.rpy_rates, g["rpy_rates"][k], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("model", ["cf2x", "cf2p"])
def test_pid_circle_matches_reference(model):
    """examples/pid.py: CtrlAviary(DYN) + external DSLPIDControl, 48 Hz control / 240 Hz physics."""
    g = golden("ctrl_pid_circle_" + model)
    n = g["init_xyzs"].shape[0]
    hz = int(g["ctrl_hz"])
    env = OracleAviary(urdf(model), model, num_drones=n, initial_xyzs=g["init_xyzs"], initial_rpys=g["init_rpys"],
                       pyb_freq=240, ctrl_freq=hz, act="raw_rpm", task="none")
    ctrl = [OracleDSLPID(UrdfConstants(urdf(model), model)) for _ in range(n)]
    action = np.zeros((n, 4))
    for k in range(g["obs"].shape[0]):
        obs, _, _, _ = env.step(action)
        np.testing.assert_allclose(obs, g["obs"][k], rtol=1e-8, atol=1e-10, err_msg=f"step {k}")
        for j in range(n):
            s = obs[j]
            action[j], _, _ = ctrl[j].computeControl(env.CTRL_TIMESTEP, s[0:3], s[3:7], s[10:13], s[13:16],
                                                     g["target"][k, j], g["init_rpys"][j])
        np.testing.assert_allclose(action, g["rpm"][k], rtol=1e-8, atol=1e-7)


@pytest.mark.parametrize("model", ["cf2x", "cf2p"])
def test_dslpid_calls_match_reference(model):
    g = golden("dslpid_calls_" + model)
    calls, n = g["pos"].shape[:2]
    ctrls = [OracleDSLPID(UrdfConstants(urdf(model), model)) for _ in range(n)]
    dt = float(g["dt"])
    for c in range(calls):
        for i in range(n):
            rpm, pe, ye = ctrls[i].computeControl(dt, g["pos"][c, i], g["quat"][c, i], g["vel"][c, i], np.zeros(3),
                                                  g["tpos"][c, i], g["trpy"][c, i], g["tvel"][c, i], g["trates"][c, i])
            np.testing.assert_allclose(rpm, g["rpm"][c, i], rtol=1e-12, atol=1e-8)
            np.testing.assert_allclose(pe, g["pos_e"][c, i], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(ye, g["yaw_e"][c, i], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(np.array([k.integral_rpy_e for k in ctrls]), g["integral_rpy_e"][c], rtol=1e-10, atol=1e-13)


@pytest.mark.parametrize("model", ["cf2x", "racer"])
def test_force_models_match_reference(model):
    """The forces _groundEffect/_drag/_downwash hand to PyBullet (captured through the shim)."""
    g = golden("force_models_" + model)
    trials, n = g["pos"].shape[:2]
    for t in range(trials):
        env = OracleAviary(urdf(model), model, num_drones=n, act="raw_rpm")
        env.pos[:], env.quat[:], env.rpy[:], env.vel[:] = g["pos"][t], g["quat"][t], g["rpy"][t], g["vel"][t]
        for i in range(n):
            np.testing.assert_allclose(env.ground_effect_forces(g["rpm"][t, i], i), g["gnd"][t, i], rtol=1e-12, atol=1e-18)
            np.testing.assert_allclose(env.drag_force_body(g["last_rpm"][t, i], i), g["drag_body"][t, i], rtol=1e-12, atol=1e-18)
            np.testing.assert_allclose(env.downwash_force(i), g["dw"][t, i], rtol=1e-12, atol=1e-18)
    assert g["gnd_on"][0, 0] == 0 and g["gnd_on"][0, 1:].all()      # |roll| > pi/2 switches it off


def test_velocity_aviary_matches_reference():
    """examples/pid_velocity.py: the reference's VelocityAviary(DYN), 4 drones, 48 Hz control / 240 Hz physics.
    Tight for the first 20 steps; at 48 Hz the DSLPID attitude loop rides its torque clip and amplifies
    rounding-level differences (see test_batched_against_golden_hover_pid), so later steps are held to 1e-6."""
    g = golden("velocity_aviary_cf2x")
    n = g["init_xyzs"].shape[0]
    env = OracleAviary(urdf("cf2x"), "cf2x", num_drones=n, initial_xyzs=g["init_xyzs"], initial_rpys=g["init_rpys"],
                       pyb_freq=240, ctrl_freq=int(g["ctrl_hz"]), act="vel", task="none")
    assert env.C.SPEED_LIMIT == pytest.approx(float(g["speed_limit"]), rel=1e-15)
    for k in range(g["obs"].shape[0]):
        env.step(g["actions"][k])
        sv = np.array([env._getDroneStateVector(i) for i in range(n)])
        tol = dict(rtol=1e-9, atol=1e-10) if k < 20 else dict(rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(sv[:, :16], g["obs"][k][:, :16], err_msg=f"step {k}", **tol)
        np.testing.assert_allclose(sv[:, 16:], g["obs"][k][:, 16:], rtol=tol["rtol"], atol=1e-3, err_msg=f"rpm, step {k}")


def test_bullet_damping_known_answer():
    """GPD_PHYS_DAMP (an extension: Bullet's default multibody damping, d = 0.04 -- third-party origin, parity unpinned, see
    oracle/aviary_oracle.py): at exactly the hover RPM thrust cancels weight and nothing else acts, so ONE sub-step from a known
    (v, w) is  v' = v - h d (1 + |v|) v,  w' = w - h [J^-1 (w x J w) + d (1 + |w|) w],  x' = x + h v'.  Without the flag the
    same state keeps its velocity (the reference's Physics.DYN has no damping, envs/BaseAviary.py:831-877)."""
    from oracle.aviary_oracle import BULLET_DAMPING, PHYS_DAMP
    h, d = 1 / 240, BULLET_DAMPING
    assert d == 0.04
    v0, w0, x0 = np.array([3.0, -4.0, 12.0]), np.array([0.3, -0.4, 1.2]), np.array([0.5, 0.5, 2.0])
    out = {}
    for flags in (0, PHYS_DAMP):
        env = OracleAviary(urdf("cf2x"), "cf2x", num_drones=1, initial_xyzs=x0[None], physics_flags=flags, pyb_freq=240,
                           ctrl_freq=240, act="raw_rpm", task="none")
        env.vel[0], env.rpy_rates[0] = v0, w0
        env.step(np.full((1, 4), env.C.HOVER_RPM))
        out[flags] = (env.pos[0].copy(), env.vel[0].copy(), env.rpy_rates[0].copy())
        C = env.C
    gyro = np.diag(C.J_INV) * np.cross(w0, np.diag(C.J) * w0)
    np.testing.assert_allclose(out[0][1], v0, rtol=0, atol=1e-12)
    np.testing.assert_allclose(out[0][2], w0 - h * gyro, rtol=1e-12)
    v1 = v0 - h * d * (1 + 13.0) * v0                     # |v0| = 13, |w0| = 1.3
    w1 = w0 - h * (gyro + d * (1 + 1.3) * w0)
    np.testing.assert_allclose(out[PHYS_DAMP][1], v1, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(out[PHYS_DAMP][2], w1, rtol=1e-12)
    np.testing.assert_allclose(out[PHYS_DAMP][0], x0 + h * v1, rtol=1e-12)
