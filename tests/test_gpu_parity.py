"""Parity of the HIP path (through the C-ABI) with the float64 oracle and the reference fixtures.

All tests here need a real MI355X (`-m gpu`).  Tolerances (fp32 kernel vs fp64 oracle):
  * ONE-STEP parity from identical (fp32-rounded) states: every output within 2e-5 of the
    field group's scale (positions ~1 m, velocities ~1 m/s, rates ~1 rad/s, RPM ~1e4 ...);
  * open-loop TRAJECTORIES: the north-star metric, per field group g and time t
        err_g(t) = ||x32_g - x64_g||_inf / max(||x64_g||_inf over batch and time, floor_g)
    with floors 1 m, 1, 1 m/s, 1 rad/s, 1 rad; bound 1e-4 after 1920 physics steps.
"""
import zlib

import numpy as np
import pytest
import torch

from conftest import golden, urdf
from oracle.aviary_oracle import ACT_DIM
from oracle.batched_oracle import BatchedAviary

pytestmark = pytest.mark.gpu

ACT_CODE = {"rpm": 0, "pid": 1, "vel": 2, "one_d_rpm": 3, "one_d_pid": 4, "raw_rpm": 5}
TASK_CODE = {"none": 0, "hover": 1, "multihover": 2}
GROUPS = {"pos": (slice(0, 3), 1.0), "quat": (slice(3, 7), 1.0), "vel": (slice(7, 10), 1.0), "rates": (slice(10, 13), 1.0)}


def _core(model, E, D, flags, S, act, task, init_xyz, init_rpy, device, auto_reset=False, target=None, keep_term=False):
    from gym_pybullet_drones_amd import engine
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    return engine.SimCore(drone_model=DroneModel(model), num_envs=E, drones_per_env=D, physics=flags, pyb_freq=240,
                          ctrl_freq=240 // S, act_code=ACT_CODE[act], task=TASK_CODE[task], initial_xyzs=init_xyz,
                          initial_rpys=init_rpy, target_pos=target, xy_bound=1.5 if task == "hover" else 2.0,
                          auto_reset=auto_reset, track_rpm=True, keep_terminal_obs=keep_term, device=device)


def _oracle_kin(b):
    """[13, N] float64 kinematic block of a BatchedAviary."""
    N = b.E * b.D
    return np.concatenate([b.pos.reshape(N, 3), b.quat.reshape(N, 4), b.vel.reshape(N, 3), b.rpy_rates.reshape(N, 3)],
                          axis=1).T


def _sync_from_oracle(core, b):
    """Copy the oracle's state (rounded to fp32) into the device state, and the rounded values back."""
    N = b.E * b.D
    kin = _oracle_kin(b).astype(np.float32)
    core.set_state(kin=kin, last_rpm=b.last_rpm.reshape(N, 4).T.astype(np.float32),
                   step_counter=b.step_counter.astype(np.int32))
    k64 = kin.astype(np.float64).T
    b.pos, b.quat = k64[:, 0:3].reshape(b.E, b.D, 3).copy(), k64[:, 3:7].reshape(b.E, b.D, 4).copy()
    b.vel, b.rpy_rates = k64[:, 7:10].reshape(b.E, b.D, 3).copy(), k64[:, 10:13].reshape(b.E, b.D, 3).copy()
    b.last_rpm = b.last_rpm.astype(np.float32).astype(np.float64)
    if core.pid is not None:
        pid = np.concatenate([b.pid.integral_pos_e.reshape(N, 3), b.pid.last_rpy.reshape(N, 3),
                              b.pid.integral_rpy_e.reshape(N, 3)], axis=1).T.astype(np.float32)
        core.set_state(pid=pid)
        p64 = pid.astype(np.float64).T
        b.pid.integral_pos_e = p64[:, 0:3].reshape(b.E, b.D, 3).copy()
        b.pid.last_rpy = p64[:, 3:6].reshape(b.E, b.D, 3).copy()
        b.pid.integral_rpy_e = p64[:, 6:9].reshape(b.E, b.D, 3).copy()
    from oracle import bullet_math as bm
    b.rpy = bm.euler_from_quaternion_b(b.quat)


def _random_scene(rng, E, D):
    """Random poses.  With several drones per aviary the heights are staggered 0.3 m +- 0.03 m: the
    reference's downwash model is singular for dz -> 0+ (alpha ~ 1/dz^2) and for dz = 0.6875 m (beta = 0),
    where no finite-precision evaluation is meaningful."""
    if D == 1:
        xyz = rng.uniform(-0.6, 0.6, size=(E, D, 3)) + np.array([0, 0, 0.8])
    else:
        xyz = rng.uniform(-1, 1, size=(E, D, 3)) * np.array([0.15, 0.15, 0.03]) + \
            np.arange(D)[None, :, None] * np.array([0.02, 0.0, 0.3]) + np.array([0, 0, 0.1])
    rpy = rng.uniform(-0.3, 0.3, size=(E, D, 3))
    return xyz, rpy


def _actions(rng, act, shape, hover_rpm):
    A = ACT_DIM[act]
    if act == "raw_rpm":
        a = hover_rpm * (1 + 0.1 * rng.uniform(-1, 1, size=shape + (A,)))
        a[..., 0][rng.uniform(size=shape) < 0.02] = -50.0      # exercises the clip at 0
        a[..., 1][rng.uniform(size=shape) < 0.02] = 1e5        # ... and at MAX_RPM
        return a
    if act == "pid":
        return np.array([0, 0, 0.8]) + 0.6 * rng.uniform(-1, 1, size=shape + (A,))
    return rng.uniform(-1, 1, size=shape + (A,))


@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
@pytest.mark.parametrize("act", ["rpm", "one_d_rpm", "pid", "vel", "one_d_pid", "raw_rpm"])
@pytest.mark.parametrize("flags,D,S", [(0, 1, 1), (0, 1, 8), (7, 1, 2), (7, 3, 2), (2, 2, 8), (5, 8, 1), (1, 5, 4),
                                       (7, 1, 3), (2, 1, 5),    # (odd sub-step counts: the remainder of the 2x unrolled loop)
                                       # the pair path of the wave-local downwash exchange (D == 2: `mate(0); mate(1)`),
                                       # BASELINE config 5's shape, alone and with every term, and config 3(i)'s shape
                                       (4, 2, 1), (7, 2, 8), (7, 1, 1),
                                       # GPD_PHYS_GROUND (8), the plane at z = 0 (an extension, see include/gpd.h): alone, with
                                       # the ground effect it belongs with, and with every term in a multi-drone aviary
                                       (8, 1, 1), (9, 1, 8), (15, 2, 2),
                                       # GPD_PHYS_DAMP (16), Bullet's default multibody damping (an extension as well): alone, as
                                       # Physics.PYB resolves (ground + damping), and with everything in a multi-drone aviary
                                       (16, 1, 1), (24, 1, 8), (31, 2, 2)])
def test_one_step_parity(gpu_device, model, act, flags, D, S):
    if model == "racer" and act in ("pid", "vel", "one_d_pid"):
        pytest.skip("no DSLPID controller for the racer")
    rng = np.random.default_rng(zlib.crc32(repr((model, act, flags, D, S)).encode()))
    b, core = _oracle_and_core(rng, model, act, flags, D, S, gpu_device)
    _check_single_steps(rng, b, core, act, flags, gpu_device)


def _oracle_and_core(rng, model, act, flags, D, S, gpu_device):
    E = 2048 // D
    task = "none" if act == "raw_rpm" else ("hover" if D == 1 else "multihover")
    xyz, rpy = _random_scene(rng, E, D)
    if flags & 8:      # a third of the aviaries start within centimetres of the plane: contact happens within the 4 passes
        xyz[::3, :, 2] = rng.uniform(0.0, 0.04, size=xyz[::3, :, 2].shape) + (0.3 * np.arange(D) if D > 1 else 0.0)
    kw = dict(physics_flags=flags, pyb_freq=240, ctrl_freq=240 // S, act=act, task=task, pid_urdf_path=urdf("cf2x"))
    b = BatchedAviary(urdf(model), model, num_envs=E, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy, **kw)
    core = _core(model, E, D, flags, S, act, task, xyz, rpy, gpu_device,
                 target=None if task == "none" else b.TARGET_POS)
    return b, core


def _check_single_steps(rng, b, core, act, flags, gpu_device):
    E, D, S = b.E, b.D, b.S
    # a few free-running steps to get non-trivial velocities, rates and PID memories, re-syncing each time
    most_touched = 0.0
    for k in range(4):
        # random velocities / rates on the first pass so every term of the integrator is exercised
        if k == 0:
            b.vel = rng.uniform(-1, 1, size=b.vel.shape)
            b.rpy_rates = rng.uniform(-2, 2, size=b.rpy_rates.shape)
            b.last_rpm = b.C.HOVER_RPM * (1 + 0.1 * rng.uniform(-1, 1, size=b.last_rpm.shape))
            b.step_counter[:] = rng.integers(0, 1936, size=E) // S * S
        _sync_from_oracle(core, b)
        a = _actions(rng, act, (E, D), b.C.HOVER_RPM).astype(np.float32)
        obs, rew, term, trunc, _ = b.step(a.astype(np.float64))
        core.step(torch.as_tensor(a, device=gpu_device))
        torch.cuda.synchronize()
        kin = core.kin[:, :E * D].cpu().numpy().astype(np.float64)
        ref = _oracle_kin(b)
        scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
        err = np.abs(kin - ref) / scale
        if flags & 8:
            # contact is a threshold: a drone that touches down within rounding of the plane may do so on one side only (its
            # velocity is then zeroed on that side only) -- at most a handful among 2048, excluded like the flag mismatches
            touched32, touched64 = kin[2] == np.float32(b.C.COLLISION_H / 2 - b.C.COLLISION_Z_OFFSET), ref[2] == b.C.COLLISION_H / 2 - b.C.COLLISION_Z_OFFSET
            assert (touched32 != touched64).mean() <= 0.002
            err[:, touched32 != touched64] = 0.0
            most_touched = max(most_touched, float(touched64.mean()))
            if k == 3:
                # (drones ON the plane at the end of a step, i.e. those that touched down and were not lifted off again by
                # their thrust -- with the ground effect a drone on the plane has ~3x its hover thrust -- a few per mille)
                assert most_touched > 0.001, "the scene must exercise the contact"
        assert err.max() < 2e-5, f"kin rows max err {err.max(axis=1)} at pass {k}"
        o = core.obs12.cpu().numpy().astype(np.float64).reshape(E, D, 12)
        oscale = np.maximum(np.abs(obs).reshape(-1, 12).max(axis=0), 1.0)
        oerr = np.abs(o - obs) / oscale
        # yaw/roll near +-pi may wrap: compare angles modulo 2*pi
        ang = np.abs((o[..., 3:6] - obs[..., 3:6] + np.pi) % (2 * np.pi) - np.pi)
        oerr[..., 3:6] = ang
        if flags & 8:
            oerr[(touched32 != touched64).reshape(E, D)] = 0.0
        assert oerr.max() < 2e-5, f"obs12 col max err {oerr.reshape(-1, 12).max(axis=0)}"
        np.testing.assert_allclose(core.last_rpm[:, :E * D].cpu().numpy().T, b.last_rpm.reshape(-1, 4),
                                   rtol=2e-5, atol=0.5)
        np.testing.assert_allclose(core.reward.cpu().numpy(), rew, rtol=1e-4, atol=1e-4)
        # flags may legitimately differ only where a quantity sits within fp32 rounding of its threshold
        kt, ktr = core.terminated.cpu().numpy().astype(bool), core.truncated.cpu().numpy().astype(bool)
        assert (kt != term).mean() <= 0.002 and (ktr != trunc).mean() <= 0.002
        np.testing.assert_array_equal(core.step_counter.cpu().numpy(), b.step_counter)
        if core.pid is not None:
            pid = core.pid[:, :E * D].cpu().numpy().astype(np.float64).T
            ref_pid = np.concatenate([b.pid.integral_pos_e.reshape(-1, 3), b.pid.last_rpy.reshape(-1, 3),
                                      b.pid.integral_rpy_e.reshape(-1, 3)], axis=1)
            np.testing.assert_allclose(pid, ref_pid, rtol=1e-5, atol=2e-6)


def _perturb_every_constant(rng, b, core):
    """Give every airframe and controller constant its own value (x 0.85 .. 1.2, element by element) on both sides."""
    from gym_pybullet_drones_amd.params import DroneParams, PIDGains
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    f = lambda *shape: rng.uniform(0.85, 1.2, size=shape) if shape else float(rng.uniform(0.85, 1.2))  # noqa: E731
    C, P = b.C, core.P

    def both(name, value):
        setattr(C, name, value)
        setattr(P, name, np.array(value) if isinstance(value, np.ndarray) else value)

    for name in ("M", "L", "KF", "KM", "GND_EFF_COEFF", "PROP_RADIUS", "GND_EFF_H_CLIP", "DW_COEFF_1", "DW_COEFF_2", "DW_COEFF_3",
                 "SPEED_LIMIT", "COLLISION_Z_OFFSET"):
        both(name, getattr(C, name) * f())
    J = np.diag(np.diag(C.J) * f(3))
    both("J", J)
    both("J_INV", np.linalg.inv(J))
    both("DRAG_COEFF", C.DRAG_COEFF * f(3))
    off = C.PROP_OFFSETS.copy()
    off[:, :2] *= f(4, 2)
    both("PROP_OFFSETS", off)
    both("GRAVITY", C.G * C.M)
    both("HOVER_RPM", np.sqrt(C.GRAVITY / (4 * C.KF)))
    both("MAX_RPM", C.MAX_RPM * f())
    pp = gains = None
    pid = getattr(b, "pid", None)
    if pid is not None:
        pp = DroneParams(DroneModel.CF2X)
        pp.M, pp.KF = pp.M * f(), pp.KF * f()
        b.pid.GRAVITY, b.pid.KF = 9.8 * pp.M, pp.KF
        gains = PIDGains(P_COEFF_FOR=b.pid.P_FOR * f(3), I_COEFF_FOR=b.pid.I_FOR * f(3), D_COEFF_FOR=b.pid.D_FOR * f(3),
                         P_COEFF_TOR=b.pid.P_TOR * f(3), I_COEFF_TOR=np.array([300., 200., 500.]) * f(3), D_COEFF_TOR=b.pid.D_TOR * f(3),
                         PWM2RPM_SCALE=b.pid.SCALE * f(), PWM2RPM_CONST=b.pid.CONST * f(), MIN_PWM=b.pid.MIN_PWM * f(),
                         MAX_PWM=b.pid.MAX_PWM * f())
        b.pid.P_FOR, b.pid.I_FOR, b.pid.D_FOR = gains.P_COEFF_FOR, gains.I_COEFF_FOR, gains.D_COEFF_FOR
        b.pid.P_TOR, b.pid.I_TOR, b.pid.D_TOR = gains.P_COEFF_TOR, gains.I_COEFF_TOR, gains.D_COEFF_TOR
        b.pid.SCALE, b.pid.CONST, b.pid.MIN_PWM, b.pid.MAX_PWM = gains.PWM2RPM_SCALE, gains.PWM2RPM_CONST, gains.MIN_PWM, gains.MAX_PWM
        b.pid.MIXER = b.pid.MIXER * f(4, 3)
    core._params = P.to_struct(pid_model=DroneModel.CF2X, gains=gains, pid_params=pp)
    if pid is not None:
        for k, v in enumerate(b.pid.MIXER.reshape(-1)):
            core._params.mixer[k] = v


@pytest.mark.parametrize("model,act,flags,D,S", [("cf2x", "pid", 7, 3, 2), ("cf2p", "vel", 15, 1, 1), ("cf2x", "rpm", 15, 2, 8),
                                                 ("cf2x", "one_d_pid", 2, 1, 4), ("racer", "one_d_rpm", 9, 1, 1), ("cf2p", "raw_rpm", 7, 5, 2),
                                                 ("cf2x", "rpm", 0, 1, 1), ("cf2x", "pid", 0, 1, 1), ("cf2p", "pid", 5, 8, 1)])
def test_one_step_parity_with_every_constant_perturbed(gpu_device, model, act, flags, D, S):
    """The shipped airframes and the DSLPID gains are full of coincidences -- P_COEFF_FOR x == y, I_COEFF_FOR all equal,
    I_COEFF_TOR x == y == 0, Ixx ~ Iyy, drag x == y, a +-0.5/+-1 mixer -- behind which a swapped index or a kernel-argument
    word delivered to the wrong register (DESIGN.md section 3.7 has a compiler doing exactly that) would be invisible.  Here
    every element of every constant gets a value of its own before the same single-step comparison is made."""
    rng = np.random.default_rng(zlib.crc32(repr(("perturbed", model, act, flags, D, S)).encode()))
    b, core = _oracle_and_core(rng, model, act, flags, D, S, gpu_device)
    _perturb_every_constant(rng, b, core)
    _check_single_steps(rng, b, core, act, flags, gpu_device)


def _traj_errors(core, b, acts, gpu_device, checkpoints):
    """Run both sides open loop; return {t: {group: err}} with the north-star normalisation."""
    N = b.E * b.D
    hist32, hist64 = {}, {}
    maxima = {g: fl for g, (_, fl) in GROUPS.items()}
    S = b.S
    for k in range(acts.shape[0]):
        a = acts[k]
        b.step(a.astype(np.float64))
        core.step(torch.as_tensor(a, device=gpu_device))
        ref = _oracle_kin(b)
        for g, (sl, _) in GROUPS.items():
            maxima[g] = max(maxima[g], float(np.abs(ref[sl]).max()))
        t = (k + 1) * S
        if t in checkpoints:
            hist32[t] = core.kin[:, :N].cpu().numpy().astype(np.float64)
            hist64[t] = ref.copy()
    out = {}
    for t in hist32:
        out[t] = {g: float(np.abs(hist32[t][sl] - hist64[t][sl]).max() / maxima[g]) for g, (sl, _) in GROUPS.items()}
    return out


@pytest.mark.parametrize("S", [1, 8])
def test_open_loop_trajectory_1920_steps(gpu_device, S):
    """N=4096 drones, DYN, act RPM, identical inputs, 1920 physics steps: <= 1e-4 norm-relative."""
    rng = np.random.default_rng(7 + S)
    E, D = 4096, 1
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    b = BatchedAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240, ctrl_freq=240 // S,
                      act="rpm", task="none")
    core = _core("cf2x", E, D, 0, S, "rpm", "none", xyz, rpy, gpu_device)
    _sync_from_oracle(core, b)
    steps = 1920 // S
    # smooth, small actions: the open-loop attitude dynamics are a double integrator, large random torques
    # would tumble every drone within a second
    acts = (0.2 * rng.uniform(-1, 1, size=(1, E, D, 4)) * np.ones((steps, 1, 1, 1)) * 0.05 +
            0.01 * rng.uniform(-1, 1, size=(steps, E, D, 4))).astype(np.float32)
    errs = _traj_errors(core, b, acts, gpu_device, {8, 16, 104, 240, 1920})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        for g, v in e.items():
            assert v < 1e-4, f"group {g} at t={t}: {v}"


def test_open_loop_with_all_force_terms(gpu_device):
    """8 drones stacked 0.3 m apart per aviary (lowest at 0.8 m), GND|DRAG|DW on, 120 physics steps (0.5 s).

    Horizon and start height are chosen on purpose: every drone but the top one is pushed down by the
    downwash of its neighbour above (~half its weight), so the stack sinks; once the lowest drone reaches
    the ground-effect zone (z < ~0.1 m, after ~0.6 s from 0.8 m) it is held up while the next one keeps
    falling, their heights cross and alpha ~ 1/dz^2 diverges — the reference's downwash model is singular
    at dz -> 0+ and no two precisions (nor two float64 runs) agree past such an encounter.  Ground effect at
    small heights is covered by the one-step tests above."""
    rng = np.random.default_rng(11)
    E, D, S = 512, 8, 2
    # 0.12 m lateral offset per level keeps dxy >> |beta| so the downwash Gaussian stays well-conditioned
    # while the stack stretches through dz = 0.6875 m (beta = 0)
    xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.12, 0.0, 0.3]) + \
        np.array([0, 0, 0.8])
    rpy = rng.uniform(-0.05, 0.05, size=(E, D, 3))
    b = BatchedAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=7, pyb_freq=240,
                      ctrl_freq=120, act="rpm", task="multihover")
    core = _core("cf2x", E, D, 7, S, "rpm", "multihover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    _sync_from_oracle(core, b)
    acts = (0.2 + 0.02 * rng.uniform(-1, 1, size=(60, E, D, 4))).astype(np.float32)
    errs = _traj_errors(core, b, acts, gpu_device, {2, 60, 120})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        for g, v in e.items():
            assert v < 1e-4, f"group {g} at t={t}: {v}"


def test_closed_loop_pid_240hz(gpu_device):
    """ActionType.PID at 240 Hz control (no torque-clip chatter): 480 steps towards random waypoints."""
    rng = np.random.default_rng(5)
    E, D, S = 2048, 1, 1
    xyz, rpy = _random_scene(rng, E, D)
    b = BatchedAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy * 0.3, pyb_freq=240, ctrl_freq=240,
                      act="pid", task="hover")
    core = _core("cf2x", E, D, 0, S, "pid", "hover", xyz, rpy * 0.3, gpu_device, target=b.TARGET_POS)
    _sync_from_oracle(core, b)
    wp = (xyz + rng.uniform(-0.3, 0.3, size=(E, D, 3))).astype(np.float32)
    acts = np.broadcast_to(wp, (480, E, D, 3)).copy()
    errs = _traj_errors(core, b, acts, gpu_device, {1, 24, 240, 480})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
    # closed loop with gains up to 7e4; bounds = ~3x what the MI355X measures (pos 5.8e-7, quat 2.9e-6, vel 2.4e-6,
    # rates 1.2e-6 at t = 480): a 10x regression fails
    for t, e in errs.items():
        assert e["pos"] < 2e-6 and e["vel"] < 8e-6 and e["quat"] < 1e-5 and e["rates"] < 5e-6, (t, e)


def _closed_loop_actions(rng, act, xyz):
    """A constant command per drone for the DSLPID action types: a waypoint within 0.3 m (PID), a velocity and the fraction of the
    speed limit it is flown at (VEL), a climb / descent rate (ONE_D_PID: target z = current z + 0.1 a, every control step)."""
    E, D, _ = xyz.shape
    if act == "pid":
        return (xyz + rng.uniform(-0.3, 0.3, size=(E, D, 3))).astype(np.float32)
    if act == "vel":
        return np.concatenate([rng.uniform(-1, 1, size=(E, D, 3)), rng.uniform(0.2, 1.0, size=(E, D, 1))], axis=-1).astype(np.float32)
    return rng.uniform(-1, 1, size=(E, D, 1)).astype(np.float32)


@pytest.mark.parametrize("ctrl", [30, 48])
@pytest.mark.parametrize("act,D", [("pid", 1), ("vel", 1), ("one_d_pid", 1), ("pid", 2), ("pid", 3)])
def test_closed_loop_dslpid_low_rate_stays_inside_the_float64_envelope(gpu_device, ctrl, act, D):
    """DSLPID at the reference's default 30 Hz (HoverAviary / MultiHoverAviary, envs/HoverAviary.py:16-17) and at 48 Hz
    (examples/pid.py:101-113), through EVERY action type that closes the loop in the kernel -- PID, VEL, ONE_D_PID -- and for
    MultiHover's 2 and 3 drones: the attitude loop (control/DSLPIDControl.py:212-259) rides its +-3200 torque clip and
    chatters, so ANY rounding-level difference grows ~10x per 4 control steps until it saturates at the chatter amplitude --
    two float64 runs do that too.  This test quantifies it: run the float64 oracle twice, the second time with its state
    nudged after every control step by what the fp32 state array's rounding amounts to over that step -- the array is rounded
    once per physics sub-step, half an ulp (relative 2^-24) with a random sign each time, S = pyb_freq / ctrl_freq of them per
    control step adding in quadrature: relative 2^-24 sqrt(S) -- and demand that the fp32 HIP run stays within a small factor
    of that envelope at every checkpoint, per field group, on the 95th percentile and the median over 1024 drones.  (Measured
    on the MI355X, round 6, worst checkpoint, median / p95 of fp32 over the envelope: PID 1.05 / 1.03 at 30 Hz and 1.04 / 0.96 at
    48 Hz, VEL 2.01 / 1.94 and 1.28 / 1.07, ONE_D_PID 2.56 / 1.26 and 1.39 / 1.12, MultiHover PID with 2 drones 1.31 / 1.32 and
    1.32 / 0.99, with 3 drones 1.27 / 1.18 and 1.25 / 1.02 -- profiles/r06_parity_measured.log; with ONE half-ulp nudge per control
    step, the round-2 form of this test, the same runs read 2.3 - 5.8: the fp32 array is rounded S times per control step.)"""
    rng = np.random.default_rng(100 + ctrl + 1000 * D + zlib.crc32(act.encode()) % 997)
    E, S, T = 1024 // D, 240 // ctrl, 48
    task = "hover" if D == 1 else "multihover"
    xyz, rpy = _random_scene(rng, E, D)
    mk = lambda: BatchedAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy * 0.3, pyb_freq=240,  # noqa: E731
                               ctrl_freq=ctrl, act=act, task=task)
    b, bp = mk(), mk()
    core = _core("cf2x", E, D, 0, S, act, task, xyz, rpy * 0.3, gpu_device, target=b.TARGET_POS)
    _sync_from_oracle(core, b)
    for name in ("pos", "quat", "vel", "rpy_rates", "rpy", "last_rpm"):
        setattr(bp, name, getattr(b, name).copy())
    bp.pid.integral_pos_e, bp.pid.last_rpy, bp.pid.integral_rpy_e = (b.pid.integral_pos_e.copy(), b.pid.last_rpy.copy(),
                                                                    b.pid.integral_rpy_e.copy())
    cmd = _closed_loop_actions(rng, act, xyz)
    eps = 2.0 ** -24 * np.sqrt(S)
    from oracle import bullet_math as bm
    rows = []
    for k in range(T):
        b.step(cmd.astype(np.float64))
        bp.step(cmd.astype(np.float64))
        for name in ("pos", "quat", "vel", "rpy_rates"):
            arr = getattr(bp, name)
            arr *= 1.0 + eps * rng.choice([-1.0, 1.0], size=arr.shape)
        bp.rpy = bm.euler_from_quaternion_b(bp.quat)
        core.step(torch.as_tensor(cmd, device=gpu_device))
        if (k + 1) in (2, 4, 8, 12, 16, 24, 32, 48):
            ref, per = _oracle_kin(b), _oracle_kin(bp)
            k32 = core.kin[:, :E * D].cpu().numpy().astype(np.float64)
            for g, (sl, _) in GROUPS.items():
                e32 = np.abs(k32[sl] - ref[sl]).max(axis=0)
                env = np.abs(per[sl] - ref[sl]).max(axis=0)
                rows.append((k + 1, g, np.percentile(e32, 50), np.percentile(env, 50), np.percentile(e32, 95),
                             np.percentile(env, 95), e32.max(), env.max()))
    for r in rows:
        print("t=%3d %-5s median fp32 %.2e envelope %.2e | p95 fp32 %.2e envelope %.2e | max fp32 %.2e envelope %.2e" % r)
    print("ENVELOPE %s D=%d %d Hz: worst ratio median %.2f p95 %.2f" % (
        act, D, ctrl, max(r[2] / (r[3] + 1.25e-7) for r in rows), max(r[4] / (r[5] + 1.25e-7) for r in rows)))
    for t, g, m32, menv, p32, penv, x32, xenv in rows:
        floor = 5e-7                     # one-step fp32 rounding of O(1) quantities
        assert m32 <= 4.0 * menv + floor, (t, g, "median", m32, menv)
        assert p32 <= 4.0 * penv + floor, (t, g, "p95", p32, penv)
    # the divergence saturates at the chatter amplitude on both sides (bounded, not a blow-up)
    assert rows[-4][6] < 0.1 and rows[-4][7] < 0.1


# ---- against the reference's own fixtures -------------------------------------------------------------

def test_reference_fixture_hover_240(gpu_device):
    """tests/golden/hover_240.npz: the reference's HoverAviary(DYN), 240 Hz, open-loop RPM actions."""
    from gym_pybullet_drones_amd.envs import HoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    g = golden("hover_240")
    env = HoverAviary(initial_xyzs=g["init_xyz"], initial_rpys=g["init_rpy"], physics=Physics.DYN, ctrl_freq=240,
                      act=ActionType.RPM, device=gpu_device)
    obs, info = env.reset(seed=0)
    assert obs.shape == (1, 12 + 120 * 4) and obs.dtype == np.float32 and info == {"answer": 42}
    np.testing.assert_allclose(obs[:, :12], g["obs0"][:, :12], rtol=1e-6, atol=1e-6)
    worst = 0.0
    for k, a in enumerate(g["actions"]):
        obs, rew, term, trunc, info = env.step(a.astype(np.float32))
        sv = env._getDroneStateVector(0)
        ref = g["state20"][k, 0]
        err = np.abs(sv[:16] - ref[:16])
        err[7:10] = np.abs((sv[7:10] - ref[7:10] + np.pi) % (2 * np.pi) - np.pi)
        worst = max(worst, float((err / np.maximum(np.abs(ref[:16]), 1.0)).max()))
        np.testing.assert_allclose(sv[16:20], ref[16:20], rtol=1e-6)
        assert isinstance(rew, float) and isinstance(term, bool) and isinstance(trunc, bool)
        assert rew == pytest.approx(float(g["reward"][k]), rel=1e-3, abs=1e-4)
        if k < 230:   # later the tumbling drone sits near the truncation thresholds
            assert trunc == bool(g["truncated"][k])
    print("worst relative state error over 300 steps:", worst)
    assert worst < 1e-5      # measured 2.9e-6 (this fixture tumbles, |rpy| up to 0.6 rad, open loop)


def test_reference_fixture_time_truncation(gpu_device):
    """tests/golden/hover_time_trunc.npz: ONE_D_RPM near hover; truncation on the 242nd step (App. B.7)."""
    from gym_pybullet_drones_amd.envs import HoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    g = golden("hover_time_trunc")
    env = HoverAviary(physics=Physics.DYN, act=ActionType.ONE_D_RPM, device=gpu_device)
    obs, _ = env.reset()
    assert obs.shape == (1, 12 + 15)
    first_trunc = None
    for k, a in enumerate(g["actions"]):
        obs, rew, term, trunc, _ = env.step(a)
        np.testing.assert_allclose(obs[0, :12], g["obs"][k, 0, :12], rtol=1e-4, atol=2e-5)
        np.testing.assert_array_equal(obs[0, 12:], g["obs"][k, 0, 12:].astype(np.float32))
        assert rew == pytest.approx(float(g["reward"][k]), rel=1e-4)
        assert trunc == bool(g["truncated"][k]) and term == bool(g["terminated"][k])
        assert env.step_counter == int(g["step_counter"][k])
        if trunc and first_trunc is None:
            first_trunc = k + 1
    assert first_trunc == 242


@pytest.mark.parametrize("name,act,n", [("multihover_rpm", "rpm", 2), ("multihover_pid", "pid", 3)])
def test_reference_fixture_multihover(gpu_device, name, act, n):
    from gym_pybullet_drones_amd.envs import MultiHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    g = golden(name)
    env = MultiHoverAviary(num_drones=n, physics=Physics.DYN, act=ActionType(act), device=gpu_device)
    np.testing.assert_allclose(env.TARGET_POS, g["target_pos"], atol=1e-12)
    env.reset()
    # RPM: open loop, errors grow polynomially.  PID at 30 Hz: the attitude loop rides its +-3200 torque clip
    # and chatters, any rounding difference grows ~10x per 4 control steps (fp64-vs-fp64 with a 3e-16
    # perturbation decorrelates by step 60 too, tests/test_oracle_batched.py) until it saturates at the
    # chatter amplitude (~2 cm): tight comparison on the first 12 steps, boundedness afterwards.
    horizon = 40 if act == "rpm" else 12
    worst_pos = worst_rew = 0.0
    for k, a in enumerate(g["actions"][:60]):
        obs, rew, term, trunc, _ = env.step(a)
        assert obs.shape == (n, 12 + 15 * ACT_DIM[act])
        if k < horizon:
            worst_pos = max(worst_pos, float(np.abs(obs[:, :3] - g["obs"][k, :, :3]).max()))
            worst_rew = max(worst_rew, abs(rew - float(g["reward"][k])) / max(abs(float(g["reward"][k])), 1.0))
            assert trunc == bool(g["truncated"][k])
        elif act == "pid":
            assert np.abs(obs[:, :3] - g["obs"][k, :, :3]).max() < 0.06
    print(f"MEASURED multihover fixture {act}: max |pos - reference| over the first {horizon} steps {worst_pos:.2e} m, reward {worst_rew:.2e} (relative)")
    # bounds: 3x what the MI355X measures (round 6, gpurun_out/r06a: RPM 1.34e-7 m / 2.1e-7 over 40 steps, PID 3.3e-7 m / 1.2e-7 over 12) --
    # round 5 allowed 2e-4 m and 2e-3 here, which a 100x regression would have passed
    assert worst_pos < POS_BOUND[act] and worst_rew < REW_BOUND[act]


POS_BOUND, REW_BOUND = {"rpm": 4e-7, "pid": 1e-6}, {"rpm": 6e-7, "pid": 4e-7}


@pytest.mark.parametrize("model", ["cf2x", "cf2p"])
def test_reference_fixture_dslpid_calls(gpu_device, model):
    """DSLPIDControl.computeControl vs the reference's own outputs (cf2x and cf2p mixers)."""
    from gym_pybullet_drones_amd.control import DSLPIDControl, DSLPIDControlBatch
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    g = golden("dslpid_calls_" + model)
    calls, n = g["pos"].shape[:2]
    ctrl = DSLPIDControlBatch(n, DroneModel(model), device=gpu_device)
    single = DSLPIDControl(DroneModel(model), device=gpu_device)
    dt = float(g["dt"])
    for c in range(calls):
        rpm, pos_e, yaw_e = ctrl.computeControl(dt, g["pos"][c], g["quat"][c], g["vel"][c], None, g["tpos"][c],
                                                g["trpy"][c], g["tvel"][c], g["trates"][c])
        if c == 0:
            # fresh controller state on both sides: outputs must agree to fp32 rounding of the PWM
            np.testing.assert_allclose(rpm.cpu().numpy(), g["rpm"][c], rtol=0, atol=0.2)
            r1, pe1, ye1 = single.computeControl(dt, g["pos"][c, 0], g["quat"][c, 0], g["vel"][c, 0], np.zeros(3),
                                                 g["tpos"][c, 0], g["trpy"][c, 0], g["tvel"][c, 0], g["trates"][c, 0])
            assert r1.shape == (4,) and pe1.shape == (3,) and isinstance(ye1, float)
            np.testing.assert_allclose(r1, g["rpm"][c, 0], atol=0.2)
        np.testing.assert_allclose(pos_e.cpu().numpy(), g["pos_e"][c], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(yaw_e.cpu().numpy(), g["yaw_e"][c], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(ctrl.integral_pos_e, g["integral_pos_e"][c], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ctrl.last_rpy, g["last_rpy"][c], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ctrl.integral_rpy_e, g["integral_rpy_e"][c], rtol=1e-4, atol=1e-5)
        # torques saturate at +-3200 for most random samples; compare RPMs where the PWM is not on a clip edge
        ref = g["rpm"][c]
        ok = np.abs(rpm.cpu().numpy() - ref) < 1.0
        assert ok.mean() > 0.97


def test_reference_fixture_pid_circle(gpu_device):
    """examples/pid.py scenario: CtrlAviary(DYN) + external DSLPIDControl at 48 Hz, 3 drones, 1 s."""
    from gym_pybullet_drones_amd.control import DSLPIDControl
    from gym_pybullet_drones_amd.envs import CtrlAviary
    from gym_pybullet_drones_amd.utils.enums import DroneModel, Physics
    g = golden("ctrl_pid_circle_cf2x")
    n, hz = g["init_xyzs"].shape[0], int(g["ctrl_hz"])
    env = CtrlAviary(drone_model=DroneModel.CF2X, num_drones=n, initial_xyzs=g["init_xyzs"], initial_rpys=g["init_rpys"],
                     physics=Physics.DYN, pyb_freq=240, ctrl_freq=hz, device=gpu_device)
    ctrl = [DSLPIDControl(DroneModel.CF2X, device=gpu_device) for _ in range(n)]
    action = np.zeros((n, 4))
    for k in range(144):
        obs, rew, term, trunc, info = env.step(action)
        assert obs.shape == (n, 20) and rew == -1 and term is False and trunc is False
        err = np.abs(obs[:, :3] - g["obs"][k, :, :3]).max()
        # 48 Hz DSLPID chatters on its torque clip (see test_reference_fixture_multihover): rounding-level
        # differences grow ~10x per 4 steps, then stay bounded by the chatter amplitude
        assert err < (2e-6 if k < 20 else 0.05), f"step {k}: {err}"
        for j in range(n):
            action[j], _, _ = ctrl[j].computeControlFromState(env.CTRL_TIMESTEP, obs[j], g["target"][k, j], g["init_rpys"][j])
        if k < 12:
            np.testing.assert_allclose(action, g["rpm"][k], rtol=0, atol=0.5)


# ---- batch semantics -----------------------------------------------------------------------------------

def test_auto_reset_matches_oracle(gpu_device):
    from gym_pybullet_drones_amd.envs import VectorHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    E = 1024
    rng = np.random.default_rng(3)
    env = VectorHoverAviary(E, act=ActionType.ONE_D_RPM, ctrl_freq=30, auto_reset=True, keep_terminal_obs=True,
                            device=gpu_device)
    b = BatchedAviary(urdf("cf2x"), "cf2x", E, 1, pyb_freq=240, ctrl_freq=30, act="one_d_rpm", task="hover",
                      auto_reset=True)
    obs, _ = env.reset()
    assert obs.shape == (E, 1, 12)
    bias = rng.uniform(-1, 1, size=(E, 1, 1))
    n_done = 0
    worst = {"obs": 0.0, "terminal_obs": 0.0}

    def rel(x32, x64):
        """SURVEY section 8(d)'s normalisation per observation group: max |x32 - x64| / max(max |x64|, 1)"""
        out = 0.0
        for sl in (slice(0, 3), slice(3, 6), slice(6, 9), slice(9, 12)):
            if x64.size:
                out = max(out, float(np.abs(x32[..., sl] - x64[..., sl]).max() / max(float(np.abs(x64[..., sl]).max()), 1.0)))
        return out

    for k in range(260):
        a = np.clip(bias + 0.3 * rng.uniform(-1, 1, size=(E, 1, 1)), -1, 1).astype(np.float32)
        o64, r64, te64, tr64, tobs64 = b.step(a.astype(np.float64))
        obs, rew, term, trunc, info = env.step(torch.as_tensor(a, device=gpu_device))
        done64 = te64 | tr64
        done = (term | trunc).cpu().numpy()
        assert (done != done64).mean() < 0.003
        same = done == done64
        n_done += int(done.sum())
        worst["obs"] = max(worst["obs"], rel(obs.cpu().numpy()[same].astype(np.float64), o64[same]))
        tob = info["terminal_observation"].cpu().numpy()
        both = done & done64
        if both.any():
            worst["terminal_obs"] = max(worst["terminal_obs"], rel(tob[both].astype(np.float64), tobs64[both]))
        np.testing.assert_array_equal(env.core.step_counter.cpu().numpy()[same], b.step_counter[same])
        # keep the two sides in lock-step where a threshold was crossed on one side only
        if (~same).any():
            b.pos, b.quat = b.pos.copy(), b.quat.copy()
            _sync_from_oracle_inverse(env.core, b)
    print("MEASURED auto-reset run, 260 steps of 1024 aviaries at 30 Hz: " + " ".join(f"{n}={v:.2e}" for n, v in worst.items()))
    assert n_done > E        # every aviary finished at least one episode (time truncation at step 242)
    # 3x what the MI355X measures (round 6, gpurun_out/r06a: obs 9.1e-6, terminal_obs 6.3e-6 of the group's scale over a full 8 s open-loop
    # episode at 30 Hz); round 5 allowed rtol 1e-3 / atol 3e-4 here
    assert worst["obs"] < AUTO_RESET_BOUND and worst["terminal_obs"] < AUTO_RESET_BOUND, worst


AUTO_RESET_BOUND = 3e-5


def _sync_from_oracle_inverse(core, b):
    """Overwrite the oracle's state with the device state (used to re-align after a threshold disagreement)."""
    N = b.E * b.D
    kin = core.kin[:, :N].cpu().numpy().astype(np.float64).T
    b.pos, b.quat = kin[:, 0:3].reshape(b.E, b.D, 3).copy(), kin[:, 3:7].reshape(b.E, b.D, 4).copy()
    b.vel, b.rpy_rates = kin[:, 7:10].reshape(b.E, b.D, 3).copy(), kin[:, 10:13].reshape(b.E, b.D, 3).copy()
    b.step_counter = core.step_counter.cpu().numpy().astype(np.int64)
    from oracle import bullet_math as bm
    b.rpy = bm.euler_from_quaternion_b(b.quat)


def test_full_obs_history_layout(gpu_device):
    """`full_obs=True` reproduces the reference's (12 + H*A) row: kinematics, then the last H actions oldest first."""
    from gym_pybullet_drones_amd.envs import VectorMultiHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    E, D = 8, 2
    env = VectorMultiHoverAviary(E, D, act=ActionType.RPM, ctrl_freq=30, full_obs=True, auto_reset=False, device=gpu_device)
    H = 15
    sent = []
    for k in range(40):
        a = torch.full((E, D, 4), float(k + 1), device=gpu_device) * 0.001
        sent.append(a.cpu().numpy())
        obs, *_ = env.step(a)
        assert obs.shape == (E, D, 12 + H * 4)
        tail = obs.cpu().numpy()[..., 12:].reshape(E, D, H, 4)
        for h in range(H):
            idx = k - (H - 1) + h
            want = sent[idx] if idx >= 0 else np.zeros((E, D, 4), dtype=np.float32)
            np.testing.assert_array_equal(tail[:, :, h, :], want)


def test_argument_errors_are_reported(gpu_device):
    """The C-ABI returns codes + messages instead of crashing."""
    import ctypes
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.params import GpdParams
    L = _native.lib()
    rc = L.gpd_step(None, None, None, None, None, None, None, None, None, None, None, None)
    assert rc == -1 and b"NULL" in L.gpd_last_error()
    core = _core("cf2x", 4, 1, 0, 1, "rpm", "none", None, None, gpu_device)
    with pytest.raises(ValueError):
        core.step(torch.zeros((4, 3), device=gpu_device))
    from gym_pybullet_drones_amd.envs import HoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, DroneModel
    with pytest.raises(ValueError):
        HoverAviary(drone_model=DroneModel.RACE, act=ActionType.PID, device=gpu_device)
    with pytest.raises(ValueError):
        HoverAviary(pyb_freq=240, ctrl_freq=7, device=gpu_device)


@pytest.mark.parametrize("D,flags", [(256, 7), (200, 5), (129, 2)])
def test_largest_aviaries_one_workgroup_per_aviary(gpu_device, D, flags):
    """drones_per_env up to the kernel's maximum (256 = one workgroup): every drone sees the downwash of up to 255
    others, the MultiHover reductions run over the whole workgroup; D = 200 / 129 leave idle lanes in it."""
    rng = np.random.default_rng(D)
    E, S = 3, 2
    # a tall, slightly tilted column: 0.3 m vertical spacing (away from the dz -> 0+ and dz = 0.6875 m singularities
    # of the downwash model for the nearest neighbours), lateral jitter
    xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.013, 0.007, 0.3]) + \
        np.array([0, 0, 0.5])
    rpy = rng.uniform(-0.05, 0.05, size=(E, D, 3))
    b = BatchedAviary(urdf("cf2x"), "cf2x", num_envs=E, num_drones=D, initial_xyzs=xyz, initial_rpys=rpy,
                      physics_flags=flags, pyb_freq=240, ctrl_freq=240 // S, act="rpm", task="multihover",
                      pid_urdf_path=urdf("cf2x"))
    core = _core("cf2x", E, D, flags, S, "rpm", "multihover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    _sync_from_oracle(core, b)
    for k in range(3):
        a = (0.05 * rng.uniform(-1, 1, size=(E, D, 4))).astype(np.float32)
        obs, rew, term, trunc, _ = b.step(a.astype(np.float64))
        core.step(torch.as_tensor(a, device=gpu_device))
        kin = core.kin[:, :E * D].cpu().numpy().astype(np.float64)
        ref = _oracle_kin(b)
        err = np.abs(kin - ref) / np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
        # (vertical velocity carries the sum of up to 255 downwash terms, each an fp32 exp: 5e-5 instead of the 2e-5
        # of the small-aviary one-step test)
        assert err.max() < 5e-5, (k, err.max(axis=1))
        # the reward is a sum over up to 256 drones of values <= 2: compare relative to its size
        np.testing.assert_allclose(core.reward.cpu().numpy(), rew, rtol=2e-5, atol=1e-4)
        np.testing.assert_array_equal(core.truncated.cpu().numpy().astype(bool), trunc)
    # and the same three steps again as one rollout from a fresh, identically initialised core: bitwise equal
    c1 = _core("cf2x", E, D, flags, S, "rpm", "multihover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    c2 = _core("cf2x", E, D, flags, S, "rpm", "multihover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    acts = torch.as_tensor((0.05 * rng.uniform(-1, 1, size=(4, E, D, 4))).astype(np.float32), device=gpu_device)
    for k in range(4):
        c1.step(acts[k])
    c2.rollout(acts)
    assert torch.equal(c1.kin, c2.kin) and torch.equal(c1.obs12, c2.obs12) and torch.equal(c1.reward, c2.reward)


def test_size_limits_are_reported(gpu_device):
    from gym_pybullet_drones_amd import engine
    with pytest.raises(ValueError):
        engine.SimCore(num_envs=1, drones_per_env=257, device=gpu_device)
    core = _core("cf2x", 4, 1, 0, 1, "rpm", "none", None, None, gpu_device)
    with pytest.raises(ValueError):
        core.rollout(torch.zeros((3, 5), device=gpu_device))                 # not K x N x A
    with pytest.raises(ValueError):
        core.rollout(torch.zeros((2, 4, 4), device=gpu_device), num_steps=3)  # neither one block nor K blocks


def test_rare_paths_gimbal_lock_and_tumbling(gpu_device):
    """The two rare per-lane cases of the step body, which the kernels enter through a wave-uniform unlikely test:
    Bullet's gimbal-lock branches of the Euler extraction (pitch = +-90 deg exactly) and the exact quaternion
    exponential for a drone that turns more than 1 rad per sub-step (> 480 rad/s at 240 Hz).  Mixed into a batch of
    ordinary drones so that some waves take the rare block for a few lanes only, some for none."""
    rng = np.random.default_rng(23)
    E = 1000
    xyz, rpy = _random_scene(rng, E, 1)
    gimbal_up, gimbal_dn, spin = np.arange(3, E, 97), np.arange(11, E, 89), np.arange(5, E, 61)
    rpy[gimbal_up, 0] = [0.3, np.pi / 2, 0.5]
    rpy[gimbal_dn, 0] = [-0.2, -np.pi / 2, 1.0]
    b = BatchedAviary(urdf("cf2x"), "cf2x", num_envs=E, num_drones=1, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240,
                      ctrl_freq=240, act="rpm", task="hover")
    core = _core("cf2x", E, 1, 0, 1, "rpm", "hover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    # gimbal lanes: no rotation during the step (hover RPMs, zero rates) so the pose stays on the branch
    b.rpy_rates[spin, 0] = rng.uniform(-1, 1, size=(len(spin), 3)) * np.array([700.0, 650.0, 900.0])
    _sync_from_oracle(core, b)
    kin0 = core.kin[:, :E].clone()
    sarg = -2 * (b.quat[:, 0, 0] * b.quat[:, 0, 2] - b.quat[:, 0, 3] * b.quat[:, 0, 1])
    assert (np.abs(sarg[gimbal_up]) >= 0.99999).all() and (np.abs(sarg[gimbal_dn]) >= 0.99999).all()
    a = np.zeros((E, 1, 4), dtype=np.float32)
    a[::2] = rng.uniform(-1, 1, size=a[::2].shape)
    a[gimbal_up] = a[gimbal_dn] = 0.0
    obs, *_ = b.step(a.astype(np.float64))
    core.step(torch.as_tensor(a, device=gpu_device))
    kin = core.kin[:, :E].cpu().numpy().astype(np.float64)
    ref = _oracle_kin(b)
    err = np.abs(kin - ref) / np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
    assert err.max() < 2e-5, err.max(axis=1)
    # the spinning drones really took the exact path: more than 1 rad in this sub-step
    assert (np.linalg.norm(b.rpy_rates[spin, 0], axis=1) / 480.0 > 1.0).all()
    o = core.obs12.cpu().numpy().astype(np.float64)
    lock = np.concatenate([gimbal_up, gimbal_dn])
    ang = np.abs((o[lock, 3:6] - obs[lock, 0, 3:6] + np.pi) % (2 * np.pi) - np.pi)
    assert ang.max() < 1e-5, ang.max(axis=0)
    assert np.allclose(np.abs(o[lock, 4]), np.pi / 2) and (o[lock, 3] == 0).all()     # Bullet's convention on the branch
    # ... and a rollout through the same states is bitwise the K single steps (the rare blocks are shared code)
    core.set_state(kin=kin0, step_counter=np.zeros(E, dtype=np.int32))
    core2 = _core("cf2x", E, 1, 0, 1, "rpm", "hover", xyz, rpy, gpu_device, target=b.TARGET_POS)
    core2.set_state(kin=kin0, step_counter=np.zeros(E, dtype=np.int32))
    acts = torch.as_tensor(np.stack([a, a, a]), device=gpu_device)
    singles = []
    for k in range(3):
        o1, *_ = core.step(acts[k])
        singles.append(o1.clone())
    o3, *_ = core2.rollout(acts)
    assert torch.equal(torch.stack(singles), o3) and torch.equal(core.kin, core2.kin)
