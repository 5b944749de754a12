#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE'S OWN PYTHON.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

The reference (utiasDSL/gym-pybullet-drones) cannot be imported as-is here: `pybullet`,
`pybullet_data`, `gymnasium` and `transforms3d` are absent.  This script installs stand-ins for
them (`oracle/pybullet_shim.py` = state store + restated Bullet quaternion utilities; the
package's minimal gymnasium stand-in; empty stubs for modules that are only imported, never
used on this path), imports the reference from /root/reference and drives its unmodified
`HoverAviary`, `MultiHoverAviary`, `CtrlAviary` (Physics.DYN) and `DSLPIDControl` on seeded
inputs.  Every array the scenarios produce is stored in `tests/golden/*.npz`; the oracle is
pinned against these files by tests/test_oracle_golden.py, and the fixtures travel to the GPU
box (the reference does not).

Scenarios (seeds and shapes are part of the fixture):
  hover_rpm         HoverAviary DYN, act RPM, 30 Hz ctrl / 240 Hz physics, 260 steps incl. the
                    time truncation at step 242
  hover_rpm_tilt    same with large actions: tilt truncation, tumbling
  hover_one_d_rpm   same with ONE_D_RPM (examples/learn.py's action type)
  hover_time_trunc  ONE_D_RPM, near-hover: stays in bounds until the time truncation on the 242nd step
  hover_pid         act PID (smooth waypoint path + noise), 150 steps
  hover_vel         act VEL, 120 steps
  hover_one_d_pid   act ONE_D_PID, 120 steps
  hover_240         ctrl_freq = pyb_freq = 240 (one sub-step per step), act RPM, 300 steps
  multihover_rpm    MultiHoverAviary DYN, 2 drones, act RPM, 260 steps
  multihover_pid    MultiHoverAviary DYN, 3 drones, act PID (smooth waypoints), 120 steps
  ctrl_pid_circle   examples/pid.py: CtrlAviary DYN, 3 drones, external DSLPIDControl tracking a
                    circle, 48 Hz ctrl / 240 Hz physics, 3 s
  ctrl_tumble       CtrlAviary DYN, cf2x/cf2p/racer, open-loop random RPMs, tumbling, 240 steps
  dslpid_calls      DSLPIDControl.computeControl on random states/targets (cf2x and cf2p)
  force_models      the forces `_groundEffect`, `_drag`, `_downwash` request from PyBullet
                    (captured by the shim) for random multi-drone configurations
  velocity_aviary   examples/pid_velocity.py: the reference's VelocityAviary DYN, 4 drones, piecewise-constant
                    velocity commands (incl. a zero-direction command), 48 Hz ctrl / 240 Hz physics, 200 steps
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = os.environ.get("GPD_REFERENCE", "/root/reference")


def load_reference():
    """Install the stand-in modules and import the reference package.  Returns (shim, modules)."""
    if REPO not in sys.path:
        sys.path.insert(0, REPO)
    import oracle.pybullet_shim as shim
    sys.modules["pybullet"] = shim
    pd = types.ModuleType("pybullet_data")
    pd.getDataPath = lambda: "/nonexistent"
    sys.modules["pybullet_data"] = pd
    spec = importlib.util.spec_from_file_location(
        "_gpd_gym_shim", os.path.join(REPO, "gym_pybullet_drones_amd", "_gym_shim.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    gym = types.ModuleType("gymnasium")
    gym.Env, gym.spaces = gs.Env, gs.spaces
    sp = types.ModuleType("gymnasium.spaces")
    sp.Box = gs.spaces.Box
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register = lambda **k: None
    genvs = types.ModuleType("gymnasium.envs")
    genvs.registration = reg
    t3 = types.ModuleType("transforms3d")
    t3q = types.ModuleType("transforms3d.quaternions")
    t3q.rotate_vector = t3q.qconjugate = None
    t3.quaternions = t3q
    sys.modules.update({"gymnasium": gym, "gymnasium.spaces": sp, "gymnasium.envs": genvs,
                        "gymnasium.envs.registration": reg, "transforms3d": t3,
                        "transforms3d.quaternions": t3q})
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    import gym_pybullet_drones  # noqa: F401
    from gym_pybullet_drones.envs.HoverAviary import HoverAviary
    from gym_pybullet_drones.envs.MultiHoverAviary import MultiHoverAviary
    from gym_pybullet_drones.envs.CtrlAviary import CtrlAviary
    from gym_pybullet_drones.envs.VelocityAviary import VelocityAviary
    from gym_pybullet_drones.control.DSLPIDControl import DSLPIDControl
    from gym_pybullet_drones.utils import enums
    return shim, dict(HoverAviary=HoverAviary, MultiHoverAviary=MultiHoverAviary,
                      CtrlAviary=CtrlAviary, VelocityAviary=VelocityAviary, DSLPIDControl=DSLPIDControl,
                      enums=enums)


def _silence():
    """The reference prints its URDF constants on every construction."""
    import contextlib
    import io
    return contextlib.redirect_stdout(io.StringIO())


def snapshot(env):
    """Full simulator state after a step, as the reference holds it."""
    n = env.NUM_DRONES
    sv = np.array([env._getDroneStateVector(i) for i in range(n)])
    d = dict(state20=sv, rpy_rates=np.array(env.rpy_rates, dtype=np.float64),
             step_counter=np.int64(env.step_counter))
    return d


def run_rl_env(env, actions, ctrl_state=False):
    """Drive an RL aviary with a fixed action sequence; record everything step() returns."""
    obs0, info0 = env.reset(seed=0)
    rec = dict(obs0=np.asarray(obs0, dtype=np.float64), state20_0=snapshot(env)["state20"])
    keys = ("obs", "reward", "terminated", "truncated", "state20", "rpy_rates", "step_counter")
    out = {k: [] for k in keys}
    if ctrl_state:
        out.update(integral_pos_e=[], last_rpy=[], integral_rpy_e=[])
    for a in actions:
        o, r, te, tr, info = env.step(a)
        s = snapshot(env)
        out["obs"].append(np.asarray(o, dtype=np.float64))
        out["reward"].append(float(r))
        out["terminated"].append(bool(te))
        out["truncated"].append(bool(tr))
        out["state20"].append(s["state20"])
        out["rpy_rates"].append(s["rpy_rates"])
        out["step_counter"].append(s["step_counter"])
        if ctrl_state:
            out["integral_pos_e"].append(np.array([c.integral_pos_e for c in env.ctrl]))
            out["last_rpy"].append(np.array([c.last_rpy for c in env.ctrl]))
            out["integral_rpy_e"].append(np.array([c.integral_rpy_e for c in env.ctrl]))
    rec.update({k: np.array(v) for k, v in out.items()})
    rec["actions"] = np.array(actions, dtype=np.float64)
    return rec


def main():
    shim, ref = load_reference()
    E = ref["enums"]
    rng = np.random.default_rng(20240915)
    os.makedirs(HERE, exist_ok=True)

    def save(name, **arrays):
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")

    # ---- HoverAviary, every action type ------------------------------------------------------
    def smooth_waypoints(steps, centre, n=1):
        """Slowly moving PID waypoints (+ small noise) around `centre` [n,3], inside [-1,1]^3."""
        i = np.arange(steps)[:, None, None]
        ph = np.arange(n)[None, :, None] * 0.7
        path = np.concatenate([0.3 * np.sin(0.05 * i + ph), 0.3 * (1 - np.cos(0.05 * i + ph)),
                               0.2 * np.sin(0.03 * i + ph)], axis=2)
        return np.clip(centre[None] + path + 0.02 * rng.uniform(-1, 1, size=(steps, n, 3)), -1, 1)

    def hover(act, steps, ctrl_freq=30, amp=1.0, init_xyz=None, init_rpy=None, acts=None):
        with _silence():
            env = ref["HoverAviary"](physics=E.Physics.DYN, act=act, ctrl_freq=ctrl_freq,
                                     initial_xyzs=init_xyz, initial_rpys=init_rpy)
        A = env.action_space.shape[1]
        if acts is None:
            acts = [amp * rng.uniform(-1, 1, size=(1, A)) for _ in range(steps)]
        rec = run_rl_env(env, acts, ctrl_state=act in (E.ActionType.PID, E.ActionType.VEL, E.ActionType.ONE_D_PID))
        rec["ctrl_freq"] = np.int64(ctrl_freq)
        if init_xyz is not None:
            rec["init_xyz"], rec["init_rpy"] = np.array(init_xyz), np.array(init_rpy)
        return rec

    save("hover_rpm", **hover(E.ActionType.RPM, 260, amp=0.02))
    save("hover_rpm_tilt", **hover(E.ActionType.RPM, 60, amp=0.3))
    save("hover_one_d_rpm", **hover(E.ActionType.ONE_D_RPM, 260))
    alt = [np.array([[0.02 * (-1) ** (i // 4) + 0.002 * rng.uniform(-1, 1)]]) for i in range(250)]
    save("hover_time_trunc", **hover(E.ActionType.ONE_D_RPM, 250, acts=alt))
    save("hover_pid", **hover(E.ActionType.PID, 150,
                              acts=list(smooth_waypoints(150, np.array([[0.0, 0.0, 0.6]])))))
    save("hover_vel", **hover(E.ActionType.VEL, 120,
                              init_xyz=np.array([[0.2, -0.1, 0.6]]), init_rpy=np.array([[0.05, -0.03, 0.4]])))
    save("hover_one_d_pid", **hover(E.ActionType.ONE_D_PID, 120))
    save("hover_240", **hover(E.ActionType.RPM, 300, ctrl_freq=240, amp=0.5,
                              init_xyz=np.array([[0.1, 0.2, 0.5]]), init_rpy=np.array([[0.1, -0.2, 0.3]])))

    # ---- MultiHoverAviary --------------------------------------------------------------------
    def multihover(act, n, steps, amp=1.0, smooth=False):
        with _silence():
            env = ref["MultiHoverAviary"](physics=E.Physics.DYN, act=act, num_drones=n)
        A = env.action_space.shape[1]
        if smooth:
            acts = list(smooth_waypoints(steps, np.array(env.TARGET_POS, dtype=np.float64) * np.array([1, 1, 0.6]), n))
        else:
            acts = [amp * rng.uniform(-1, 1, size=(n, A)) for _ in range(steps)]
        rec = run_rl_env(env, acts, ctrl_state=act == E.ActionType.PID)
        rec["target_pos"] = np.array(env.TARGET_POS, dtype=np.float64)
        rec["init_xyzs"] = np.array(env.INIT_XYZS, dtype=np.float64)
        return rec

    save("multihover_rpm", **multihover(E.ActionType.RPM, 2, 260, amp=0.02))
    save("multihover_pid", **multihover(E.ActionType.PID, 3, 120, smooth=True))

    # ---- examples/pid.py scenario on CtrlAviary(DYN) with external controllers ------------------
    def ctrl_pid_circle(model, num_drones=3, ctrl_hz=48, dur=3):
        H, H_STEP, R = .1, .05, .3
        init_xyzs = np.array([[R * np.cos((i / 6) * 2 * np.pi + np.pi / 2),
                               R * np.sin((i / 6) * 2 * np.pi + np.pi / 2) - R, H + i * H_STEP] for i in range(num_drones)])
        init_rpys = np.array([[0, 0, i * (np.pi / 2) / num_drones] for i in range(num_drones)])
        period = 10
        num_wp = ctrl_hz * period
        target = np.zeros((num_wp, 3))
        for i in range(num_wp):
            target[i, :] = (R * np.cos((i / num_wp) * (2 * np.pi) + np.pi / 2) + init_xyzs[0, 0],
                            R * np.sin((i / num_wp) * (2 * np.pi) + np.pi / 2) - R + init_xyzs[0, 1], 0)
        wp = np.array([int((i * num_wp / 6) % num_wp) for i in range(num_drones)])
        with _silence():
            env = ref["CtrlAviary"](drone_model=model, num_drones=num_drones, initial_xyzs=init_xyzs,
                                    initial_rpys=init_rpys, physics=E.Physics.DYN,
                                    pyb_freq=240, ctrl_freq=ctrl_hz)
            ctrl = [ref["DSLPIDControl"](drone_model=model) for _ in range(num_drones)]
        action = np.zeros((num_drones, 4))
        obs_l, act_l, tgt_l = [], [], []
        for i in range(int(dur * env.CTRL_FREQ)):
            obs, _, _, _, _ = env.step(action)
            tg = np.zeros((num_drones, 3))
            for j in range(num_drones):
                tg[j] = np.hstack([target[wp[j], 0:2], init_xyzs[j, 2]])
                action[j, :], _, _ = ctrl[j].computeControlFromState(
                    control_timestep=env.CTRL_TIMESTEP, state=obs[j], target_pos=tg[j], target_rpy=init_rpys[j, :])
            for j in range(num_drones):
                wp[j] = wp[j] + 1 if wp[j] < (num_wp - 1) else 0
            obs_l.append(np.array(obs))
            act_l.append(action.copy())
            tgt_l.append(tg)
        return dict(init_xyzs=init_xyzs, init_rpys=init_rpys, obs=np.array(obs_l), rpm=np.array(act_l),
                    target=np.array(tgt_l), ctrl_hz=np.int64(ctrl_hz), max_rpm=np.float64(env.MAX_RPM))

    save("ctrl_pid_circle_cf2x", **ctrl_pid_circle(E.DroneModel.CF2X))
    save("ctrl_pid_circle_cf2p", **ctrl_pid_circle(E.DroneModel.CF2P))

    # ---- open-loop tumbling on all three airframes --------------------------------------------
    def ctrl_tumble(model, steps=240, n=4):
        init_xyzs = rng.uniform(-1, 1, size=(n, 3)) + np.array([0, 0, 2.0])
        init_rpys = rng.uniform(-0.5, 0.5, size=(n, 3))
        with _silence():
            env = ref["CtrlAviary"](drone_model=model, num_drones=n, initial_xyzs=init_xyzs,
                                    initial_rpys=init_rpys, physics=E.Physics.DYN, pyb_freq=240, ctrl_freq=240)
        rpms = env.HOVER_RPM * (1 + 0.1 * rng.uniform(-1, 1, size=(steps, n, 4)))
        rpms[steps // 2] = env.MAX_RPM * 1.5        # exercises CtrlAviary's clip to MAX_RPM
        rpms[steps // 2 + 1] = -100.0               # ... and to 0
        obs_l, rr_l = [], []
        for k in range(steps):
            obs, _, _, _, _ = env.step(rpms[k])
            obs_l.append(np.array(obs))
            rr_l.append(np.array(env.rpy_rates))
        return dict(init_xyzs=init_xyzs, init_rpys=init_rpys, rpm=rpms, obs=np.array(obs_l),
                    rpy_rates=np.array(rr_l), max_rpm=np.float64(env.MAX_RPM), hover_rpm=np.float64(env.HOVER_RPM))

    for m in (E.DroneModel.CF2X, E.DroneModel.CF2P, E.DroneModel.RACE):
        save("ctrl_tumble_" + m.value, **ctrl_tumble(m))

    # ---- DSLPIDControl.computeControl, direct calls ---------------------------------------------
    def dslpid_calls(model, n=64, calls=6):
        with _silence():
            ctrls = [ref["DSLPIDControl"](drone_model=model) for _ in range(n)]
        import oracle.bullet_math as bm
        rec = {k: [] for k in ("pos", "quat", "vel", "tpos", "trpy", "tvel", "trates", "rpm", "pos_e", "yaw_e",
                               "integral_pos_e", "last_rpy", "integral_rpy_e")}
        dt = 1 / 48
        for c in range(calls):
            pos = rng.uniform(-1, 1, size=(n, 3))
            rpy = rng.uniform(-0.6, 0.6, size=(n, 3))
            quat = np.array([bm.quaternion_from_euler(r) for r in rpy])
            vel = rng.uniform(-1, 1, size=(n, 3))
            tpos = pos + rng.uniform(-0.5, 0.5, size=(n, 3))
            trpy = np.concatenate([np.zeros((n, 2)), rng.uniform(-1, 1, size=(n, 1))], axis=1)
            tvel = rng.uniform(-0.3, 0.3, size=(n, 3))
            trates = rng.uniform(-0.1, 0.1, size=(n, 3))
            rpm, pe, ye = [], [], []
            for i in range(n):
                r, p_e, y_e = ctrls[i].computeControl(dt, pos[i], quat[i], vel[i], np.zeros(3), tpos[i],
                                                      trpy[i], tvel[i], trates[i])
                rpm.append(r); pe.append(p_e); ye.append(y_e)
            for k, v in (("pos", pos), ("quat", quat), ("vel", vel), ("tpos", tpos), ("trpy", trpy), ("tvel", tvel),
                         ("trates", trates), ("rpm", np.array(rpm)), ("pos_e", np.array(pe)), ("yaw_e", np.array(ye)),
                         ("integral_pos_e", np.array([c_.integral_pos_e for c_ in ctrls])),
                         ("last_rpy", np.array([c_.last_rpy for c_ in ctrls])),
                         ("integral_rpy_e", np.array([c_.integral_rpy_e for c_ in ctrls]))):
                rec[k].append(v)
        out = {k: np.array(v) for k, v in rec.items()}
        out["dt"] = np.float64(dt)
        return out

    save("dslpid_calls_cf2x", **dslpid_calls(E.DroneModel.CF2X))
    save("dslpid_calls_cf2p", **dslpid_calls(E.DroneModel.CF2P))

    # ---- the add-on force models, as requested from PyBullet ------------------------------------
    def force_models(model, n=6, trials=12):
        recs = {k: [] for k in ("pos", "quat", "rpy", "vel", "rpm", "last_rpm", "gnd", "gnd_on", "drag_body", "dw")}
        for t in range(trials):
            init_xyzs = rng.uniform(-0.4, 0.4, size=(n, 3)) * np.array([1, 1, 0]) + \
                np.array([0, 0, 1]) * rng.uniform(0.02, 1.2, size=(n, 1))
            init_rpys = rng.uniform(-0.7, 0.7, size=(n, 3))
            if t == 0:
                init_rpys[0, 0] = 2.0           # |roll| > pi/2: ground effect switched off
            with _silence():
                env = ref["CtrlAviary"](drone_model=model, num_drones=n, initial_xyzs=init_xyzs,
                                        initial_rpys=init_rpys, physics=E.Physics.DYN)
            env.vel[:] = rng.uniform(-2, 2, size=(n, 3))
            rpm = env.HOVER_RPM * (1 + 0.2 * rng.uniform(-1, 1, size=(n, 4)))
            last = env.HOVER_RPM * (1 + 0.2 * rng.uniform(-1, 1, size=(n, 4)))
            gnd = np.zeros((n, 4)); gnd_on = np.zeros(n); drag = np.zeros((n, 3)); dw = np.zeros(n)
            for i in range(n):
                shim.clear_applied()
                env._groundEffect(rpm[i], i)
                for rec_ in shim.applied:
                    assert rec_[0] == "force" and rec_[5] == shim.LINK_FRAME and rec_[3][0] == 0 and rec_[3][1] == 0
                    gnd[i, rec_[2]] += rec_[3][2]
                gnd_on[i] = len(shim.applied) > 0
                shim.clear_applied()
                env._drag(last[i], i)
                (rec_,) = shim.applied
                assert rec_[2] == 4 and rec_[5] == shim.LINK_FRAME
                drag[i] = rec_[3]
                shim.clear_applied()
                env._downwash(i)
                for rec_ in shim.applied:
                    assert rec_[2] == 4 and rec_[5] == shim.LINK_FRAME and rec_[3][0] == 0 and rec_[3][1] == 0
                    dw[i] += rec_[3][2]
            for k, v in (("pos", np.array(env.pos)), ("quat", np.array(env.quat)), ("rpy", np.array(env.rpy)),
                         ("vel", np.array(env.vel)),
                         ("rpm", rpm), ("last_rpm", last), ("gnd", gnd), ("gnd_on", gnd_on), ("drag_body", drag), ("dw", dw)):
                recs[k].append(v)
        return {k: np.array(v) for k, v in recs.items()}

    save("force_models_cf2x", **force_models(E.DroneModel.CF2X))
    save("force_models_racer", **force_models(E.DroneModel.RACE))

    # ---- examples/pid_velocity.py scenario on the reference's VelocityAviary(DYN) ---------------------
    def velocity_aviary(model, n=4, ctrl_hz=48, steps=200):
        vrng = np.random.default_rng(777)                  # own generator: the fixtures above do not depend on it
        init_xyzs = np.array([[0, 0, .1], [.3, 0, .1], [.6, 0, .1], [0.9, 0, .1]])[:n]
        init_rpys = np.array([[0, 0, 0], [0, 0, np.pi / 3], [0, 0, np.pi / 4], [0, 0, np.pi / 2]])[:n]
        with _silence():
            env = ref["VelocityAviary"](drone_model=model, num_drones=n, initial_xyzs=init_xyzs, initial_rpys=init_rpys,
                                        physics=E.Physics.DYN, pyb_freq=240, ctrl_freq=ctrl_hz)
        # piecewise-constant commands like the example's, plus a zero-direction command (unit vector = 0 branch)
        acts = np.zeros((steps, n, 4))
        for k in range(steps):
            seg = k // 40
            r = np.random.default_rng(1000 + seg)
            acts[k] = np.hstack([r.uniform(-1, 1, size=(n, 3)), r.uniform(0.2, 1, size=(n, 1))])
        acts[80:90, 1, 0:3] = 0.0
        acts += 0.0 * vrng.uniform(size=acts.shape)
        obs_l, rr_l = [], []
        for k in range(steps):
            obs, _, _, _, _ = env.step(acts[k])
            obs_l.append(np.array(obs))
            rr_l.append(np.array(env.rpy_rates))
        return dict(init_xyzs=init_xyzs, init_rpys=init_rpys, actions=acts, obs=np.array(obs_l), rpy_rates=np.array(rr_l),
                    ctrl_hz=np.int64(ctrl_hz), speed_limit=np.float64(env.SPEED_LIMIT))

    save("velocity_aviary_cf2x", **velocity_aviary(E.DroneModel.CF2X))


if __name__ == "__main__":
    main()
