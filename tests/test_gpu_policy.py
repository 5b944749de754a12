"""`gpd_rollout_policy`: K env steps per launch with an MLP policy evaluated inside the kernel (matrix cores, bf16 hi/lo
operands) -- against the float64 actor of oracle/policy_oracle.py, against stepping the same actions through `gpd_step`
(bitwise), and closed loop against the float64 oracle aviary."""
import numpy as np
import pytest
import torch

from conftest import urdf
from oracle.batched_oracle import BatchedAviary
from oracle.policy_oracle import mlp_actor, policy_loop

pytestmark = pytest.mark.gpu


def _weights(p):
    return [x.cpu().numpy().astype(np.float64) for x in (p.w1, p.b1, p.w2, p.b2, p.w3, p.b3)]


def _env(act, ctrl, mode, E, dev, **kw):
    from gym_pybullet_drones_amd.envs import VectorHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    rng = np.random.default_rng(E)
    xyz = np.array([0, 0, 0.5]) + rng.uniform(-0.4, 0.4, size=(E, 1, 3))
    rpy = rng.uniform(-0.2, 0.2, size=(E, 1, 3))
    return VectorHoverAviary(E, initial_xyzs=xyz, initial_rpys=rpy, act=ActionType(act), ctrl_freq=ctrl, full_obs=mode, device=dev, **kw)


CASES = [  # act, ctrl_freq (H = ctrl // 2), policy sees the history, activation
    ("rpm", 30, True, "tanh"), ("rpm", 30, False, "tanh"), ("one_d_rpm", 30, True, "tanh"), ("one_d_rpm", 240, False, "relu"),
    ("rpm", 24, True, "relu"), ("one_d_rpm", 40, True, "tanh"),
    # DSLPID action types: the policy's output is the controller's set-point (A = 3: the history shifts by one and a half registers)
    ("pid", 30, True, "tanh"), ("pid", 240, False, "tanh"), ("vel", 30, True, "tanh"), ("one_d_pid", 30, True, "relu"), ("pid", 24, True, "tanh"),
]


@pytest.mark.parametrize("act,ctrl,hist,activation", CASES)
def test_policy_actions_match_the_float64_actor_every_step(gpu_device, act, ctrl, hist, activation):
    """Every action the kernel chose equals the float64 actor evaluated on the kernel's OWN previous observation row (and
    action history): checks the feature staging, the lane-half swaps, the K permutation of the chained layers, the history
    shift and the reset handling at every step, independent of how the trajectories evolve.  Ragged batch, short episodes."""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K = 1000, 40
    env = _env(act, ctrl, "lazy" if hist else False, E, gpu_device, episode_len_sec=12.0 / ctrl)
    A, H = env.ACT_DIM, ctrl // 2
    pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=ctrl + A, gain=1.5, activation=activation, device=gpu_device)
    rng = np.random.default_rng(0)
    for _ in range(5):                           # some history and motion first
        env.step(torch.as_tensor(rng.uniform(-1, 1, size=(E, 1, A)).astype(np.float32), device=gpu_device))
    obs0 = env.core.obs12.clone()
    hist0 = env.history().clone() if hist else None
    obs, rew, term, trunc, acts = env.rollout_policy(pol, K)
    torch.cuda.synchronize()
    assert obs.shape == (K, E, 1, 12) and acts.shape == (K, E, 1, A) and float(acts.abs().max()) <= 1.0
    assert (term | trunc).any(), "the test must exercise resets inside the rollout"
    w = _weights(pol)
    o_prev = obs0.cpu().numpy().astype(np.float64).reshape(E, 12)
    h = hist0.cpu().numpy().astype(np.float64).reshape(E, H, A) if hist else None
    worst = 0.0
    for t in range(K):
        row = o_prev if h is None else np.concatenate([o_prev, h.reshape(E, -1)], axis=1)
        want = mlp_actor(row, *w, activation=activation)
        got = acts[t].cpu().numpy().astype(np.float64).reshape(E, A)
        worst = max(worst, float(np.abs(got - want).max()))
        if h is not None:
            h = np.concatenate([h[:, 1:], got[:, None, :]], axis=1)
        o_prev = obs[t].cpu().numpy().astype(np.float64).reshape(E, 12)
    print(f"{act} ctrl={ctrl} hist={hist} {activation}: max |kernel action - float64 actor| over {K} steps = {worst:.2e}")
    assert worst < 5e-5
    if hist:   # the ring holds the exact actions, oldest first
        np.testing.assert_array_equal(env.history().cpu().numpy().reshape(E, H, A)[:, -min(H, K):], acts[-min(H, K):].cpu().numpy().reshape(-1, E, A).transpose(1, 0, 2))


@pytest.mark.parametrize("act,ctrl,hist,phys", [("rpm", 30, True, "dyn"), ("one_d_rpm", 240, False, "dyn"), ("rpm", 48, False, "dyn"),
                                                ("rpm", 30, True, "pyb_gnd_drag_dw"), ("one_d_rpm", 30, True, "pyb_drag"),
                                                ("pid", 30, True, "dyn"), ("vel", 48, False, "pyb_gnd_drag_dw"), ("one_d_pid", 30, True, "dyn")])
@pytest.mark.parametrize("scramble", [False, True])
def test_policy_rollout_is_bitwise_stepping_its_actions(gpu_device, act, ctrl, hist, phys, scramble):
    """The physics inside the policy kernel is the shared `env_step`: feeding the actions it chose to `gpd_step` one at a time
    reproduces its observations, rewards, flags, state and action ring bit for bit.  `scramble`: with every float word of the
    by-value kernel arguments scaled by a factor of its own (test_gpu_rollout._scramble_arguments), so that both kernels must
    have read the same word for every field."""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K = 777, 25
    mode = "lazy" if hist else False
    from gym_pybullet_drones_amd.utils.enums import Physics
    a, b = (_env(act, ctrl, mode, E, gpu_device, episode_len_sec=10.0 / ctrl, physics=Physics(phys)) for _ in range(2))
    A, H = a.ACT_DIM, ctrl // 2
    if scramble:
        from test_gpu_rollout import _scramble_arguments
        _scramble_arguments(np.random.default_rng(78), a.core, b.core)
    pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=3, gain=1.2, device=gpu_device)
    obs, rew, term, trunc, acts = a.rollout_policy(pol, K)
    for t in range(K):
        o, r, te, tr, _ = b.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(te, term[t]) and torch.equal(tr, trunc[t]), t
    for name in ("kin", "last_rpm", "pid", "step_counter", "obs12", "reward", "terminated", "truncated"):
        if getattr(a.core, name) is not None:
            assert torch.equal(getattr(a.core, name), getattr(b.core, name)), name
    if hist:
        assert torch.equal(a.history(), b.history())


@pytest.mark.parametrize("act,ctrl,hist,sample", [("rpm", 30, True, False), ("rpm", 30, True, True), ("one_d_rpm", 240, False, False),
                                                  ("pid", 30, True, False), ("vel", 48, False, False)])
def test_policy_rollout_writes_the_terminal_observations_step_would(gpu_device, act, ctrl, hist, sample):
    """SB3's PPO bootstraps a time-out from `infos[i]["terminal_observation"]` (the loop `model.learn()` runs around
    examples/learn.py:61-95): the in-kernel collection loop hands them out too -- for every aviary that ends at step t of the
    launch, row (t, aviary) of the terminal block is, bit for bit, what `gpd_step` writes for the same action; rows of aviaries
    that did not end stay untouched; `term_obs12` afterwards holds what K single steps would have left.  The VecEnv-style
    infos of a policy rollout are thereby reconstructible from its outputs alone."""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K = 777, 30
    mode = "lazy" if hist else False
    a, b = (_env(act, ctrl, mode, E, gpu_device, episode_len_sec=10.0 / ctrl, keep_terminal_obs=True) for _ in range(2))
    A, H = a.ACT_DIM, ctrl // 2
    pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=5, gain=1.2, device=gpu_device)
    kw = {}
    if sample:
        g = torch.Generator(device=gpu_device)
        g.manual_seed(1)
        kw = dict(noise=torch.randn((K, E, 1, A), generator=g, device=gpu_device), action_std=[0.5] * A)
    sentinel = -123.0
    a.core._rollout_buffers(K)[4].fill_(sentinel)
    obs, rew, term, trunc, acts = a.rollout_policy(pol, K, **kw)
    tob = a.core.terminal_observations(K).view(K, E, 1, 12)
    done_any = torch.zeros(E, dtype=torch.bool, device=gpu_device)
    for t in range(K):
        o, r, te, tr, info = b.step(acts[t])
        done = te | tr
        assert torch.equal(te, term[t]) and torch.equal(tr, trunc[t]), t
        assert torch.equal(tob[t][done], info["terminal_observation"][done]), t
        assert bool((tob[t][~done] == sentinel).all()), t                     # nothing written where no episode ended
        done_any |= done
    assert int(done_any.sum()) > E // 2                                       # (short episodes: most aviaries ended at least once)
    assert torch.equal(a.core.term_obs12, b.core.term_obs12)
    # the VecEnv infos, rebuilt from the rollout's outputs alone
    t_last = K - 1
    idx = torch.nonzero(term[t_last] | trunc[t_last]).flatten()
    for i in idx[:5].tolist():
        info_i = {"terminal_observation": tob[t_last, i], "TimeLimit.truncated": bool(trunc[t_last, i] and not term[t_last, i])}
        assert torch.equal(info_i["terminal_observation"], b.core.term_obs12.view(E, 1, 12)[i])


def test_policy_closed_loop_against_the_float64_oracle(gpu_device):
    """`examples/learn.py:157-192` end to end: HoverAviary at the reference's 30 Hz, RPM actions, the policy sees the full
    row; fp32 HIP kernel vs float64 oracle aviary + float64 actor, 45 control steps (1.5 s)."""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K, ctrl, A, H = 512, 45, 30, 4, 15
    env = _env("rpm", ctrl, "lazy", E, gpu_device, auto_reset=False)
    orc = BatchedAviary(urdf("cf2x"), "cf2x", E, 1, initial_xyzs=env.INIT_XYZS, initial_rpys=env.INIT_RPYS, pyb_freq=240,
                        ctrl_freq=ctrl, act="rpm", task="hover")
    pol = MlpPolicy.random(12 + H * A, A, seed=11, gain=1.0, device=gpu_device)
    obs0, _ = env.reset()
    ref = policy_loop(orc, _weights(pol), K, obs0.cpu().numpy().astype(np.float64), history=np.zeros((E, 1, H, A)))
    obs, rew, term, trunc, acts = env.rollout_policy(pol, K)
    o32 = obs.cpu().numpy().astype(np.float64)
    for t in (0, 4, 14, 29, 44):
        eo = np.abs(o32[t] - ref["obs"][t]).max(axis=(0, 1))
        ea = np.abs(acts[t].cpu().numpy() - ref["actions"][t]).max()
        print(f"t={t + 1:3d} obs err pos {eo[:3].max():.2e} rpy {eo[3:6].max():.2e} vel {eo[6:9].max():.2e} ang_v {eo[9:].max():.2e} | action err {ea:.2e}")
    scale = np.maximum(np.abs(ref["obs"]).max(axis=(0, 1, 2)), 1.0)
    assert (np.abs(o32 - ref["obs"]) / scale).max() < 1e-4
    assert np.abs(acts.cpu().numpy() - ref["actions"]).max() < 2e-4
    np.testing.assert_allclose(rew.cpu().numpy(), ref["reward"], rtol=1e-3, atol=1e-3)


def test_policy_argument_errors(gpu_device):
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.envs import VectorMultiHoverAviary
    from gym_pybullet_drones_amd.policy import MlpPolicy
    env = _env("rpm", 30, False, 64, gpu_device)
    with pytest.raises(_native.GpdError, match="in_dim"):
        env.rollout_policy(MlpPolicy.random(72, 4, device=gpu_device), 4)          # wants the history, the env keeps none
    with pytest.raises(_native.GpdError, match="single-drone"):
        VectorMultiHoverAviary(8, 2, device=gpu_device).rollout_policy(MlpPolicy.random(12, 4, device=gpu_device), 4)
    with pytest.raises(_native.GpdError, match="history too long"):
        _env("rpm", 240, "lazy", 64, gpu_device).rollout_policy(MlpPolicy.random(12 + 120 * 4, 4, device=gpu_device), 4)
    with pytest.raises(ValueError):
        MlpPolicy(np.zeros((32, 12)), np.zeros(32), np.zeros((32, 32)), np.zeros(32), np.zeros((4, 32)), np.zeros(4), device=gpu_device)


@pytest.mark.parametrize("act,ctrl,hist", [("rpm", 30, True), ("rpm", 240, False), ("one_d_rpm", 30, True), ("one_d_rpm", 48, False),
                                           ("vel", 240, False), ("pid", 30, True)])
@pytest.mark.parametrize("tweak", [dict(task=0), dict(auto_reset=0), dict(xy_bound=0.2, tilt_bound=0.05), dict(trunc_counter=3),
                                   dict(z_bound=1e9, xy_bound=1e9, tilt_bound=1e9, trunc_counter=2 ** 30)])
@pytest.mark.parametrize("sampled", [False, True])
def test_policy_rollout_reads_the_step_configuration_like_gpd_step(gpu_device, act, ctrl, hist, tweak, sampled):
    """Every field of GpdStepCfg the task evaluation reads (task switch, auto-reset, the truncation box and tilt bound, the time
    limit), changed one at a time: the policy kernel must keep following gpd_step bit for bit -- rewards and flags included.
    (A build of the VEL variant under another instruction scheduler once returned truncated = 1 and a task reward for every
    aviary regardless of these fields, DESIGN.md section 3.7: this is the test that would have caught it in any variant.)"""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K = 300, 12
    mode = "lazy" if hist else False
    a, b = (_env(act, ctrl, mode, E, gpu_device, episode_len_sec=1.0) for _ in range(2))
    for env in (a, b):
        for k, v in tweak.items():
            setattr(env.core._cfg, k, v)
    A, H = a.ACT_DIM, ctrl // 2
    pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=5, gain=1.2, device=gpu_device)
    if sampled:        # the sampling kernels are separate instantiations (own register allocation): the same matrix for them
        if act not in ("rpm", "one_d_rpm"):
            pytest.skip("sampling: RPM action types")
        noise = torch.randn((K, E, 1, A), generator=torch.Generator(device=gpu_device).manual_seed(1), device=gpu_device)
        obs, rew, term, trunc, acts = a.rollout_policy(pol, K, noise=noise, action_std=[0.5] * A)
    else:
        obs, rew, term, trunc, acts = a.rollout_policy(pol, K)
    for t in range(K):
        o, r, te, tr, _ = b.step(acts[t])
        assert torch.equal(r, rew[t]) and torch.equal(te, term[t]) and torch.equal(tr, trunc[t]) and torch.equal(o, obs[t]), (t, tweak)
    if tweak.get("task", 1) == 0:
        assert bool((rew == -1).all()) and not bool(trunc.any())
    if "trunc_counter" in tweak and tweak["trunc_counter"] == 3:
        assert bool(trunc.any())


@pytest.mark.parametrize("act,ctrl,hist,activation", [("rpm", 30, True, "tanh"), ("rpm", 240, False, "relu"), ("one_d_rpm", 30, True, "tanh"),
                                                      ("one_d_rpm", 48, False, "tanh")])
def test_sampled_actions_for_training_rollouts(gpu_device, act, ctrl, hist, activation):
    """`noise` + `action_std` (SB3's PPO collection: DiagGaussianDistribution, state-independent log_std, actions clipped before
    env.step): (1) zero noise reproduces the deterministic kernel bit for bit and `mean_out` holds the unclipped means;
    (2) with standard-normal noise every action is clip(mean + std * noise) of the kernel's own mean, the means are the float64
    actor's on the kernel's own previous row, and stepping the sampled actions through gpd_step reproduces the rollout bit for
    bit (ring included)."""
    from gym_pybullet_drones_amd.policy import MlpPolicy
    E, K = 900, 30
    mode = "lazy" if hist else False
    a, b, c = (_env(act, ctrl, mode, E, gpu_device, episode_len_sec=10.0 / ctrl) for _ in range(3))
    A, H = a.ACT_DIM, ctrl // 2
    pol = MlpPolicy.random(12 + (H * A if hist else 0), A, seed=9, gain=1.3, activation=activation, device=gpu_device)
    std = [0.6, 0.45, 0.7, 0.5][:A]
    g = torch.Generator(device=gpu_device).manual_seed(3)
    noise = torch.randn((K, E, 1, A), generator=g, device=gpu_device)
    # (1) zero noise
    det = [x.clone() for x in a.rollout_policy(pol, K)]
    mean0 = torch.empty((K, E, 1, A), device=gpu_device)
    zer = [x.clone() for x in b.rollout_policy(pol, K, noise=torch.zeros_like(noise), action_std=std, mean_out=mean0)]
    for x, y in zip(det, zer):
        assert torch.equal(x, y)
    assert torch.equal(mean0.clamp(-1, 1), det[4]) and float(mean0.abs().max()) > 0.2
    # (2) sampled
    obs0 = c.core.obs12.clone()
    hist0 = c.history().clone() if hist else None
    mean = torch.empty((K, E, 1, A), device=gpu_device)
    obs, rew, term, trunc, acts = c.rollout_policy(pol, K, noise=noise, action_std=std, mean_out=mean)
    want = (mean.double() + torch.tensor(std, dtype=torch.float64, device=gpu_device) * noise.double()).clamp(-1, 1)
    assert float((acts.double() - want).abs().max()) < 2e-7
    assert float((acts.abs() == 1).float().mean()) > 0.01, "the test must exercise the clipping"
    w = _weights(pol)
    o_prev = obs0.cpu().numpy().astype(np.float64).reshape(E, 12)
    h = hist0.cpu().numpy().astype(np.float64).reshape(E, H, A) if hist else None
    worst = 0.0
    for t in range(K):
        row = o_prev if h is None else np.concatenate([o_prev, h.reshape(E, -1)], axis=1)
        mu = mlp_actor(row, *w, activation=activation, clip=False)
        worst = max(worst, float(np.abs(mean[t].cpu().numpy().astype(np.float64).reshape(E, A) - mu).max()))
        got = acts[t].cpu().numpy().astype(np.float64).reshape(E, A)
        if h is not None:
            h = np.concatenate([h[:, 1:], got[:, None, :]], axis=1)
        o_prev = obs[t].cpu().numpy().astype(np.float64).reshape(E, 12)
    print(f"{act} ctrl={ctrl} hist={hist}: max |kernel mean - float64 actor| over {K} steps = {worst:.2e}")
    assert worst < 1e-4
    d = _env(act, ctrl, mode, E, gpu_device, episode_len_sec=10.0 / ctrl)
    for t in range(K):
        o, r, te, tr, _ = d.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(te, term[t]) and torch.equal(tr, trunc[t]), t
    if hist:
        assert torch.equal(c.history(), d.history())
    assert (term | trunc).any()


def test_sampling_argument_errors(gpu_device):
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.policy import MlpPolicy
    env = _env("rpm", 30, False, 64, gpu_device)
    pol = MlpPolicy.random(12, 4, device=gpu_device)
    noise = torch.zeros((4, 64, 1, 4), device=gpu_device)
    with pytest.raises(ValueError):
        env.rollout_policy(pol, 4, noise=noise[:2], action_std=[1, 1, 1, 1])
    with pytest.raises(ValueError):
        env.rollout_policy(pol, 4, noise=noise, action_std=[1, 1])
    with pytest.raises(_native.GpdError, match="noise"):
        env.rollout_policy(pol, 4, mean_out=torch.zeros_like(noise))
    penv = _env("pid", 30, False, 64, gpu_device)
    with pytest.raises(_native.GpdError, match="RPM"):
        penv.rollout_policy(MlpPolicy.random(12, 3, device=gpu_device), 4, noise=torch.zeros((4, 64, 1, 3), device=gpu_device), action_std=[1, 1, 1])


def test_vel_policy_kernel_is_right_under_both_schedulers(gpu_device, tmp_path):
    """VERDICT r03 #6.  The VEL policy kernel once came out of hipcc with four configuration words replaced by drone parameters
    (a register-allocation defect) when it was compiled in the policy unit under the default scheduler.
    This test BUILDS that configuration on the GPU box -- both units with -DGPD_PID_POLICY_IN_POLICY_TU, the policy unit under
    the default scheduler -- and runs the DSLPID cases of the bitwise cross-kernel test (scrambled arguments included) against it
    in a fresh process (`GPD_LIB`); the shipped library (those kernels in the max-ilp unit) runs them in this one."""
    import os
    import subprocess
    import sys
    from conftest import REPO
    from gym_pybullet_drones_amd import _native
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    objs, procs = [], []
    for unit, extra in _native.UNITS:
        obj = str(tmp_path / unit.replace(".hip", ".o"))
        cmd = [hipcc] + _native.COMMON_FLAGS + extra + ["-DGPD_PID_POLICY_IN_POLICY_TU", "-I", _native.INCLUDE, "-c", os.path.join(_native.CSRC, unit), "-o", obj]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        objs.append(obj)
    for pr in procs:
        out, _ = pr.communicate()
        assert pr.returncode == 0, out[-3000:]
    lib = str(tmp_path / "libgpd_pid_in_policy_unit.so")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], check=True, capture_output=True)
    # the DSLPID policy kernels are now defined by the policy unit's object (default scheduler), not the main one
    assert "-amdgpu-sched-strategy=max-ilp" in " ".join(_native.UNITS[0][1]) and "max-ilp" not in " ".join(_native.UNITS[1][1])
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_gpu_policy.py"), "-q", "-x", "-k",
                          "bitwise_stepping and (vel or pid)"], cwd=REPO, env=dict(os.environ, GPD_LIB=lib), capture_output=True, text=True, timeout=900)
    assert res.returncode == 0 and " passed" in res.stdout, res.stdout[-3000:] + res.stderr[-2000:]
    print(res.stdout.strip().splitlines()[-1])
