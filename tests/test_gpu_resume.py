"""Exact checkpoint / resume and the non-finite guard -- two auxiliary subsystems SURVEY.md section 5 lists as absent upstream
(the reference re-allocates its numpy state on every reset, envs/BaseAviary.py:468-477, keeps its action buffer across resets,
envs/BaseRLAviary.py:65-67, and never looks for NaNs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(dev, act, full_obs, **kw):
    from gym_pybullet_drones_amd.envs import VectorHoverAviary
    rng = np.random.default_rng(5)
    E = 512
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, 1, 3)) * np.array([1, 1, 0])
    rpy = rng.uniform(-0.1, 0.1, size=(E, 1, 3))
    return VectorHoverAviary(E, initial_xyzs=xyz, initial_rpys=rpy, ctrl_freq=30, act=act, full_obs=full_obs, auto_reset=True,
                             keep_terminal_obs=True, device=dev, **kw)


@pytest.mark.parametrize("act_name,full_obs", [("RPM", "lazy"), ("PID", "lazy"), ("RPM", True), ("VEL", False)])
def test_save_100_steps_restore_100_steps_is_bitwise(gpu_device, act_name, full_obs):
    """save -> 100 steps -> restore -> the same 100 steps: observations, history tails, rewards, flags and terminal observations
    equal bit for bit -- on the env that was saved AND on a fresh one built with the same arguments (the ring, its positions,
    the DSLPID members and the episode clocks travel with the snapshot; aviaries end and auto-reset inside the window)."""
    from gym_pybullet_drones_amd.utils.enums import ActionType
    act = ActionType[act_name]
    env = _mk(gpu_device, act, full_obs)
    A = env.ACT_DIM
    g = torch.Generator(device=gpu_device)
    g.manual_seed(11)
    acts = torch.rand((230, env.NUM_ENVS, 1, A), generator=g, device=gpu_device) * 2 - 1
    if act == ActionType.PID:
        acts = acts * 0.5
        acts[..., 2] += 1.0
    env.reset()
    for k in range(30):                     # some history: ring positions mid-way, integrators non-zero, a few resets behind
        env.step(acts[k])
    snap = env.get_state()
    assert {"kin", "step_counter", "obs12", "reward", "terminated", "truncated", "term_obs12"} <= set(snap)
    assert ("act_ring" in snap and "ring_pos" in snap) == bool(full_obs) and ("pid" in snap) == act.uses_pid

    def run(e):
        out = []
        for k in range(30, 130):
            obs, rew, term, trunc, info = e.step(acts[k])
            rows = e.full_rows() if full_obs == "lazy" else obs
            out.append((rows.clone(), rew.clone(), term.clone(), trunc.clone(), info["terminal_observation"].clone()))
        return out

    first = run(env)
    if act != ActionType.VEL:               # (the velocity controller keeps its drones in the box: no episode ends in 130 steps)
        assert sum(int((t | u).sum()) for _, _, t, u, _ in first) > 0      # episodes did end inside the window
    for k in range(130, 230):               # (drift further away before coming back)
        env.step(acts[k])
    fresh = _mk(gpu_device, act, full_obs)
    for e in (env, fresh):
        e.set_state(snap)
        again = run(e)
        for k, (a, b) in enumerate(zip(first, again)):
            for x, y in zip(a, b):
                assert torch.equal(x, y), (k, "fresh" if e is fresh else "same")
    # the snapshot is a copy: stepping did not touch it
    assert torch.equal(snap["kin"], env.get_state()["kin"]) is False
    with pytest.raises(TypeError):
        env.core.set_state(nonsense=1)
    with pytest.raises(ValueError):
        env.core.set_state(kin=torch.zeros((13, 3)))


def test_restore_continues_a_policy_rollout_bitwise(gpu_device):
    """the in-kernel policy loop starts from the latest observation rows and the ring: both are in the snapshot"""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    from gym_pybullet_drones_amd.utils.enums import ActionType
    env = _mk(gpu_device, ActionType.RPM, "lazy")
    pol = MlpPolicy.random(12 + env.ACTION_BUFFER_SIZE * 4, 4, seed=3, gain=1.0, device=gpu_device)
    env.reset()
    env.rollout_policy(pol, 20)
    snap = env.get_state()
    a = [t.clone() for t in env.rollout_policy(pol, 40)]
    env.rollout_policy(pol, 7)
    env.set_state(snap)
    b = env.rollout_policy(pol, 40)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("act_name", ["raw_rpm", "PID"])
def test_one_world_resumes_bitwise_whatever_the_sort_and_the_lists_held(gpu_device, act_name):
    """ONE aviary of 1 500 drones with the downwash over the whole swarm: save -> 23 control steps -> restore -> the same 23 steps.
    The sort, the wake lists and their margins are not in the snapshot (and differ: the restored run re-bins at once, the
    original was mid-interval); the forces are exact whatever they hold, so state vectors and forces repeat bit for bit -- on the
    world that was saved and on a fresh one."""
    from gym_pybullet_drones_amd.envs import SwarmAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
    rng = np.random.default_rng(77)
    sites = np.array([(x, y) for x in np.arange(-28, 29, 4.0) for y in np.arange(-18, 19, 4.0)])
    N = 1500
    idx = rng.permutation(len(sites) * 12)[:N]
    xyz = np.concatenate([sites[idx % len(sites)] + rng.uniform(-0.3, 0.3, size=(N, 2)), (1.0 + idx // len(sites))[:, None]], axis=1)
    act = "raw_rpm" if act_name == "raw_rpm" else ActionType.PID
    mk = lambda: SwarmAviary(N, initial_xyzs=xyz, physics=Physics.PYB_GND_DRAG_DW, pyb_freq=240, ctrl_freq=120, act=act, rebin_every=7,  # noqa: E731
                             device=gpu_device)
    env = mk()
    if act_name == "raw_rpm":
        acts = torch.as_tensor((env.HOVER_RPM * (1 + 0.04 * rng.uniform(-1, 1, size=(60, N, 4)))).astype(np.float32), device=gpu_device)
    else:
        acts = torch.as_tensor((xyz[None] + rng.uniform(-0.2, 0.2, size=(60, N, 3))).astype(np.float32), device=gpu_device)
    env.reset()
    for k in range(10):
        env.step(acts[k])
    snap = env.get_state()

    def run(e):
        out = []
        for k in range(10, 33):
            v, *_ = e.step(acts[k])
            out.append((v.clone(), e.dw_force[:N].clone()))
        return out

    first = run(env)
    for k in range(33, 45):
        env.step(acts[k])
    fresh = mk()
    fresh.reset()
    for e in (env, fresh):
        e.set_state(snap)
        for k, (a, b) in enumerate(zip(first, run(e))):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (k, e is fresh)
    assert env.step_counter == fresh.step_counter == 2 * 33 and float(first[-1][1].abs().max()) > 1e-3


@pytest.mark.parametrize("mode", ["step", "rollout", "rollout_pid", "multi"])
def test_nan_guard_flags_exactly_the_poisoned_aviaries(gpu_device, mode):
    """A NaN (or an infinity) that enters through an action poisons the drone's state for good, and no truncation bound ever trips
    on it (every comparison with a NaN is false): the reference runs on blind.  With `nan_guard=True` every call that stores the
    state leaves one byte per drone -- set for exactly the aviaries that were fed the bad action, clear for the rest, in the
    single-step kernel, the rollout kernels and multi-drone aviaries alike; a reset clears it with the next call."""
    from gym_pybullet_drones_amd.envs import VectorHoverAviary, VectorMultiHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    E = 1000
    rng = np.random.default_rng(2)
    poisoned = np.sort(rng.choice(E, 17, replace=False))
    if mode == "multi":
        env = VectorMultiHoverAviary(E, 3, ctrl_freq=240, act=ActionType.RPM, nan_guard=True, auto_reset=False, device=gpu_device,
                                     initial_xyzs=np.array([[0, 0, 0.3], [0.5, 0, 0.6], [1.0, 0, 0.9]]))
    else:
        env = VectorHoverAviary(E, ctrl_freq=240, act=ActionType.PID if mode == "rollout_pid" else ActionType.RPM, nan_guard=True,
                                auto_reset=False, device=gpu_device)
    D, A = env.NUM_DRONES, env.ACT_DIM
    env.reset()
    good = torch.zeros((4, E, D, A), device=gpu_device)
    if A == 3:
        good[..., 2] = 1.0
    bad = good.clone()
    bad[1, torch.as_tensor(poisoned, device=gpu_device), D - 1, 0] = float("nan")          # one drone of the aviary, one step
    bad[1, int(poisoned[0]), D - 1, 0] = float("inf")
    if mode == "step" or mode == "multi":
        for k in range(4):
            env.step(good[k])
        assert not env.bad_envs().any()
        for k in range(4):
            env.step(bad[k])
    else:
        env.rollout(good)
        assert not env.bad_envs().any()
        env.rollout(bad)
    flagged = np.flatnonzero(env.bad_envs().cpu().numpy())
    if mode == "rollout_pid":
        # DSLPID: a NaN SET-POINT never reaches the state -- every quantity on its way to the RPMs passes a clip, and the hardware
        # min / max return the bound for a NaN (numpy's clip, which the reference uses, propagates it: there the drone is gone for
        # good).  One step of saturated commands; everything stays finite, nothing is flagged.
        assert flagged.size == 0
        assert torch.isfinite(env.core.kin[:, :E]).all() and torch.isfinite(env.core.pid[:, :E]).all()
        sel = torch.as_tensor(poisoned, device=gpu_device)
        assert not torch.equal(env.core.kin[:, sel], env.core.kin[:, (sel + 1) % E])            # (they did see a different step)
        # ... the controller's own members heal the same way (each passes a clip or is overwritten every step) ...
        env.core.pid[4, int(poisoned[3])] = float("nan")
        env.core.pid[0, int(poisoned[4])] = float("nan")
        env.rollout(good)
        assert not env.bad_envs().any() and torch.isfinite(env.core.pid[:, :E]).all()
        # ... and a non-finite value that does sit in the kinematic state is found
        env.core.kin_V[int(poisoned[5]), 1] = float("inf")          # (vel y; the kinematic block lives in four planes, ABI 9)
        env.rollout(good)
        assert np.flatnonzero(env.bad_envs().cpu().numpy()).tolist() == [int(poisoned[5])]
    else:
        assert np.array_equal(flagged, poisoned), (flagged, poisoned)
        per_drone = env.core.bad.view(E, D).cpu().numpy()
        assert per_drone[poisoned, D - 1].all() and per_drone.sum() == len(poisoned)           # only the drone that got it
        # the reference's blind spot, demonstrated: none of the NaN aviaries was truncated (the one fed an infinity may be: +inf > z bound)
        assert not env.core.truncated.cpu().numpy()[poisoned[1:]].any()
    env.core.reset(reset_pid=True)
    assert not env.bad_envs().any()              # (the flags follow a reset at once, not at the next launch)
    env.step(good[0])
    assert not env.bad_envs().any()
    plain = VectorHoverAviary(8, device=gpu_device)
    with pytest.raises(ValueError):
        plain.bad_envs()


_DEBUG_PROBE = r'''
import ctypes, sys
import numpy as np, torch
from gym_pybullet_drones_amd import _native
from gym_pybullet_drones_amd.envs import SwarmAviary, VectorHoverAviary
from gym_pybullet_drones_amd.utils.enums import ActionType, Physics
L = _native.lib()
dev = torch.device("cuda:0")
def status(reset=0):
    out = (ctypes.c_uint32 * 4)()
    rc = L.gpd_debug_status(out, reset, None)
    assert rc == 0, (rc, L.gpd_last_error())
    return list(out)
env = VectorHoverAviary(300, ctrl_freq=30, act=ActionType.RPM, full_obs="lazy", device=dev)
a = torch.zeros((20, 300, 1, 4), device=dev)
env.reset()
for k in range(20):
    env.step(a[k])
env.rollout(a)
# (five layers 1 m apart on a 4 m lattice with jitter: no two drones at the same height within reach of each other -- the
# reference's model is singular there, alpha ~ 1 / dz^2)
rng = np.random.default_rng(0)
sites = np.array([(x, y) for x in np.arange(-24, 25, 4.0) for y in np.arange(-24, 25, 4.0)])
N = len(sites) * 5
xyz = np.concatenate([np.tile(sites, (5, 1)) + rng.uniform(-0.3, 0.3, size=(N, 2)), np.repeat(np.arange(1.0, 6.0), len(sites))[:, None]], axis=1)
sw = SwarmAviary(N, initial_xyzs=xyz, physics=Physics.PYB_DW, device=dev, rebin_every=4)
sw.reset()
rpm = torch.full((N, 4), float(sw.HOVER_RPM), device=dev)
for k in range(10):
    sw.step(rpm)
torch.cuda.synchronize()
print("CLEAN", status())
# a ring position beyond the ring: caught, clamped, the run goes on
env.core.ring_pos[7] = 1000
env.step(a[0])
torch.cuda.synchronize()
print("RING", status(reset=1))
env.core.ring_pos[7] = 0
env.rollout(a)
print("AFTER_RESET", status())
# a wake-list entry that points beyond the world, and a batch count beyond the capacity
while sw._since_bin >= sw.rebin_every - 1:          # (the next sub-step must REPLAY the lists, not rebuild them)
    sw.step(rpm)
assert int(sw._list_ok.sum()) > 0
sw._pair_list[:, 1, :64] = (5 << 26) | 0x3ffffff
sw._pair_nb[:, 2, 0] = 30000
sw.step(rpm)
torch.cuda.synchronize()
st = status(reset=1)
print("LIST", st)
'''


def test_debug_bounds_build_reports_a_corrupted_index_instead_of_following_it(gpu_device, tmp_path):
    """SURVEY.md section 5 lists no sanitizer or bounds checks upstream (numpy raises IndexError for the reference); a release build
    of this library follows whatever index memory holds.  The debug-bounds build (`_native.build(debug=True)`,
    `-DGPD_DEBUG_BOUNDS`, selected with GPD_LIB) checks every index its kernels read from memory: a clean run of the step, rollout
    and one-world kernels reports nothing; a ring position beyond the ring and wake-list entries / counts beyond the world are
    reported with their code, workgroup and value and CLAMPED -- the process survives and goes on; the release build answers
    `gpd_debug_status` with GPD_ENOTSUP."""
    import ctypes
    import os
    import subprocess
    import sys
    from conftest import REPO
    from gym_pybullet_drones_amd import _native
    out = (ctypes.c_uint32 * 4)()
    assert _native.lib().gpd_debug_status(out, 0, None) == _native.GPD_ENOTSUP          # this process runs the release build
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) and not os.path.exists(_native.DEBUG_LIB_PATH):
        pytest.skip("no hipcc and no prebuilt debug library")
    lib = _native.build(debug=True)                      # (travels with the snapshot when it was built before; rebuilt when stale)
    res = subprocess.run([sys.executable, "-c", _DEBUG_PROBE], cwd=REPO, env=dict(os.environ, GPD_LIB=lib), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = {l.split(" ", 1)[0]: eval(l.split(" ", 1)[1]) for l in res.stdout.splitlines() if l.split(" ", 1)[0] in ("CLEAN", "RING", "AFTER_RESET", "LIST")}
    assert lines["CLEAN"] == [0, 0, 0, 0], lines
    code, wg, val, count = lines["RING"]
    assert code == 1 and val == 1000 and count >= 1 and wg == 0, lines          # GPD_DBG_RING_POS, aviary 7 sits in workgroup 0
    assert lines["AFTER_RESET"] == [0, 0, 0, 0], lines
    code, wg, val, count = lines["LIST"]
    assert code in (5, 6) and count >= 2 and (val == 30000 if code == 5 else val == 0x3ffffff), lines
