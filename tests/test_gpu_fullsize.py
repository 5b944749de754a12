"""Parity at BASELINE.json's full sizes: against the C oracle where it finishes in seconds, and through
size-independent properties (analytic known answers, batch-size independence, permutation equivariance,
norm preservation) at 524 288 drones."""
import numpy as np
import pytest
import torch

from conftest import urdf
from oracle.c_oracle import CAviary
from test_gpu_parity import GROUPS, _core, _oracle_kin, _sync_from_oracle

pytestmark = pytest.mark.gpu


def _run(core, orc, acts, dev, checkpoints):
    N = orc.E * orc.D
    maxima = {g: fl for g, (_, fl) in GROUPS.items()}
    snaps = {}
    for k in range(acts.shape[0]):
        orc.step(acts[k].astype(np.float64))
        core.step(torch.as_tensor(acts[k], device=dev))
        ref = _oracle_kin(orc)
        for g, (sl, _) in GROUPS.items():
            maxima[g] = max(maxima[g], float(np.abs(ref[sl]).max()))
        t = (k + 1) * orc.S
        if t in checkpoints:
            snaps[t] = (core.kin[:, :N].cpu().numpy().astype(np.float64), ref.copy())
    return {t: {g: float(np.abs(a[sl] - b[sl]).max() / maxima[g]) for g, (sl, _) in GROUPS.items()}
            for t, (a, b) in snaps.items()}


def test_config2_65536_hover_1920_physics_steps(gpu_device):
    """BASELINE metric config: 65 536 HoverAviaries, DYN, 30 Hz control / 240 Hz physics, 1920 physics steps."""
    rng = np.random.default_rng(65536)
    E, D, S = 65536, 1, 8
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240, ctrl_freq=30, act="rpm",
                  task="hover")
    core = _core("cf2x", E, D, 0, S, "rpm", "hover", xyz, rpy, gpu_device, target=orc.TARGET_POS)
    _sync_c(core, orc)
    acts = (0.01 * rng.uniform(-1, 1, size=(1, E, D, 4)) + 0.01 * rng.uniform(-1, 1, size=(240, E, D, 4))).astype(np.float32)
    errs = _run(core, orc, acts, gpu_device, {8, 240, 1920})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        assert max(e.values()) < 1e-4, (t, e)
    # task outputs of the last step
    np.testing.assert_allclose(core.reward.cpu().numpy(), orc.reward, rtol=1e-3, atol=1e-3)
    assert (core.truncated.cpu().numpy() != orc.truncated.astype(bool)).mean() < 0.002


def test_config3_stacks_of_8_all_force_terms(gpu_device):
    """BASELINE config 3 (ii): 8192 aviaries x 8 stacked drones, GND|DRAG|DW, 0.25 s (60 physics steps), on THE scene bench.py
    times (`bench.stack_scene`: workload stack8x8192_ext_240hz), against the tolerance.

    Short on purpose: among 65 536 drones with random tilts a few drift under their upper neighbour within
    half a second, where the downwash Gaussian exp(-(dxy/beta)^2/2) with |beta| ~ 0.07 m has a relative
    condition number of (dxy/beta)^2 ~ 10-100 and its 1/dz^2 prefactor is ~3x the drone's weight; fp32 and
    fp64 then separate by 1e-3..1e-2 m although every single step agrees to 1e-6 (one-step tests).  The 256-step horizon of
    the bench is held against the float64 envelope instead (next test)."""
    import bench
    E, D, S = 8192, 8, 1
    xyz, rpy = bench.stack_scene(np.random.default_rng(1000), E, D)            # (seed 1000: rank 0's aviaries in bench.make_env)
    rng = np.random.default_rng(8192)
    orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=7, pyb_freq=240,
                  ctrl_freq=240, act="rpm", task="multihover")
    core = _core("cf2x", E, D, 7, S, "rpm", "multihover", xyz, rpy, gpu_device, target=orc.TARGET_POS)
    _sync_c(core, orc)
    acts = (0.2 + 0.02 * rng.uniform(-1, 1, size=(60, E, D, 4))).astype(np.float32)
    errs = _run(core, orc, acts, gpu_device, {1, 30, 60})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        assert max(e.values()) < 1e-4, (t, e)
    np.testing.assert_allclose(core.reward.cpu().numpy(), orc.reward, rtol=1e-3, atol=1e-3)


def test_config3_conditioned_stacks_meet_the_plain_tolerance_over_256_steps(gpu_device):
    """BASELINE config 3 (ii) at full size -- 8192 aviaries x 8 stacked drones, GND|DRAG|DW -- over the bench's 256-step horizon against
    the PLAIN 1e-4 tolerance (VERDICT r04 "next" #5), on a scene the downwash model is well-conditioned on for that long: the drones
    0.3 m apart in height on a staircase of 0.2 m per drone (|dxy| = 3.2 |beta|, beta = DW2 * 0.3 + DW3 = -0.062 m: every drone sits
    on the far shoulder of its upper neighbour's wake, exp(-5.2) of its peak), tilts of +-0.01 rad, per-rotor RPM noise of +-0.1 %
    (the bench's +-5 % tumbles a drone past 0.4 rad within the second and sends it through its neighbours' wakes: that horizon is
    held against the float64 envelope instead, next test).  No task: a stack of eight 0.3 m apart is 2.2 m tall, above the
    MultiHover ceiling.  Two float64 runs of this scene, one nudged by half an fp32 ulp per step, stay within 2.4e-5 of each other
    in their worst aviary (median 3.8e-6): what is left for the fp32 path is its own rounding."""
    E, D, S = 8192, 8, 1
    rng = np.random.default_rng(8)
    xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.2, 0.0, 0.3]) + np.array([0, 0, 0.1])
    rpy = rng.uniform(-0.01, 0.01, size=(E, D, 3))
    orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=7, pyb_freq=240,
                  ctrl_freq=240, act="rpm", task="none")
    core = _core("cf2x", E, D, 7, S, "rpm", "none", xyz, rpy, gpu_device)
    _sync_c(core, orc)
    _all_threads()
    acts = (0.02 * rng.uniform(-1, 1, size=(256, E, D, 4))).astype(np.float32)
    errs = _run(core, orc, acts, gpu_device, {1, 64, 128, 256})
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        assert max(e.values()) < 1e-4, (t, e)
    # the downwash term is active on this scene: the second drone from the top feels its neighbour (a few percent of its weight)
    N = E * D
    z = core.kin[2, :N].view(E, D).cpu().numpy()
    free = 0.1 + 0.3 * np.arange(D)
    assert np.abs(z - free).max() > 1e-3


@pytest.mark.parametrize("workload", ["stack8x8192_ext_240hz", "stack8x8192_ext_pid_240hz"])
def test_stack8_downwash_stays_inside_the_float64_envelope(gpu_device, workload):
    """The bench's OWN stack8 workload -- its scene, its random +-5 % RPM (or DSLPID waypoint) blocks, same-step auto-reset, its
    64-step rollout launches -- for the bench's 256-step horizon, through the bench's own checker (`oracle/bench_checks.py: parity_check`):
    the fp32 HIP run against the float64 C oracle, AND against the envelope of a second float64 run nudged by half an fp32
    ulp after every step.  Over 256 steps of random actions some drones cross a neighbour's wake, where the reference's model
    amplifies any rounding; the claim that this, not a kernel defect, is what the plain tolerance sees is measured here:
    median and 95th percentile of the fp32 error over the aviaries stay within 4x the float64-vs-float64 envelope at every
    checkpoint."""
    import bench
    w = bench.WORKLOADS[workload]
    env = bench.make_env(w, gpu_device, seed=1000)
    acts = bench.make_actions(w, env, gpu_device, seed=2000, pool=64)
    for n in (64, 64):                                   # (some history first, like the bench's timed region before its check)
        bench.launch_rollout(env, acts, n)
    _all_threads()
    from oracle.bench_checks import parity_check
    res = parity_check(w, env, acts, 256, 64, bench.launch_rollout, max_steps=256)
    e = res["envelope"]
    for r in e["rows"]:
        print("t=%3d %-5s median fp32 %.2e envelope %.2e | p95 fp32 %.2e envelope %.2e | max fp32 %.2e envelope %.2e | %d aviaries" % tuple(r))
    print({k: res[k] for k in ("max", "flag_mismatch_frac", "ok", "ok_by", "episodes_ended_in_window")}, "ratio", e["ratio"], e["worst"])
    assert res["checked_steps"] == 256 and len(e["rows"]) >= 4 * 16
    assert e["ratio"] <= 4.0 and e["ok"] and res["ok_envelope"], e["worst"]
    # `ok` is the plain tolerance's verdict and nothing else; the line also says how many aviaries are inside it on their
    # own, and how close the worst one's drones came to each other in height
    assert res["ok_plain_all_aviaries"] == (res["max"] < 1e-4) and res["frac_aviaries_within_tolerance"] >= 0.97, res["frac_aviaries_within_tolerance"]
    print("aviaries inside 1e-4:", res["frac_aviaries_within_tolerance"], res.get("worst_aviary"))
    # ... and `ok` itself follows the WRITTEN rule for drones in each other's wake (VERDICT r05 #1d): the aviaries whose drones never came
    # within 2 cm of each other in height -- at least 95 % of them -- are ALL inside the plain tolerance; the rest is counted, not judged
    wr = res["wake_rule"]
    print("WAKE RULE", {k: wr[k] for k in ("min_abs_dz_m", "aviaries_kept", "excluded_frac", "excluded_for_wake_frac", "max_over_kept", "max_over_excluded", "ok")},
          wr["threshold_scan"])
    assert res["ok"] == wr["ok"] and wr["min_abs_dz_m"] == 0.02 and wr["needs_kept_frac"] == 0.95
    # the first steps are plain rounding on both sides ...
    first = [r for r in e["rows"] if r[0] == 1]
    assert all(r[4] < 2e-6 for r in first), first
    # ... and the divergence the tolerance sees is there in float64 as well: the two float64 runs separate by more than the tolerance
    # in their worst aviaries, or the fp32 run passed the tolerance outright
    assert res["ok_by"] in ("tolerance", "wake_rule") or max(r[7] for r in e["rows"]) > 1e-4
    assert e["flag_mismatch_frac_between_the_two_float64_runs"] < 0.05 and res["flag_mismatch_frac"] < 0.05


def _all_threads():
    """the C oracle on every usable host thread (the aviaries are independent: static OpenMP chunks)"""
    import os
    from oracle import c_oracle
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return c_oracle.lib().orc_set_threads(max(1, min(n, 64)))


def test_config5_16384x2_pairwise_downwash_240_steps(gpu_device):
    """BASELINE config 5 at its per-GPU size: 16 384 two-drone MultiHover aviaries, pairwise downwash ON (the D == 2 pair
    path of the wave-local exchange), RPM actions, 240 steps -- HIP vs the float64 C oracle.

    Scene: the two drones 0.3 m above each other, 5 cm apart laterally.  (MultiHoverAviary's DEFAULT initial poses put
    both drones at the same height, z = 0.1125: the reference's downwash model is singular there -- alpha = DW1
    (r / 4 dz)^2 with dz -> 0+ gives ~1e10 N as soon as rounding separates the heights -- so no finite-precision
    trajectory, float64 included, is meaningful from that start)  The upper drone pushes the lower
    one down with ~2x its weight; it falls away, dz only grows, the comparison stays well-conditioned."""
    rng = np.random.default_rng(16384)
    E, D, S = 16384, 2, 1
    xyz = rng.uniform(-0.02, 0.02, size=(E, D, 3)) + np.arange(D)[None, :, None] * np.array([0.05, 0.0, 0.3]) + \
        np.array([0, 0, 0.6])
    rpy = rng.uniform(-0.05, 0.05, size=(E, D, 3))
    _all_threads()
    try:
        orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=4, pyb_freq=240,
                      ctrl_freq=240, act="rpm", task="multihover")
        core = _core("cf2x", E, D, 4, S, "rpm", "multihover", xyz, rpy, gpu_device, target=orc.TARGET_POS)
        _sync_c(core, orc)
        acts = (0.02 * rng.uniform(-1, 1, size=(240, E, D, 4))).astype(np.float32)
        errs = _run(core, orc, acts, gpu_device, {1, 24, 120, 240})
    finally:
        from oracle import c_oracle
        c_oracle.lib().orc_set_threads(1)
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        assert max(e.values()) < 1e-4, (t, e)
    # the downwash really acted: the lower drones fell well below the upper ones' free trajectory
    z = core.kin[2, :E * D].view(E, D)
    assert float((z[:, 1] - z[:, 0]).min()) > 0.5
    np.testing.assert_allclose(core.reward.cpu().numpy(), orc.reward, rtol=1e-3, atol=1e-3)
    assert (core.truncated.cpu().numpy() != orc.truncated.astype(bool)).mean() < 0.002


def test_config3i_65536_all_force_terms_single_drone_aviaries(gpu_device):
    """BASELINE config 3 (i): 65 536 single-drone aviaries with GND|DRAG|DW compiled in and enabled (no neighbours, so the
    downwash term is exercised as "present and zero"), start heights in [0.05, 1] m so that the ground effect is active,
    240 steps, against the float64 C oracle."""
    rng = np.random.default_rng(3)
    E, D, S = 65536, 1, 1
    xyz = rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0]) + np.array([0, 0, 1]) * rng.uniform(0.05, 1.0, size=(E, D, 1))
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    _all_threads()
    try:
        orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, physics_flags=7, pyb_freq=240,
                      ctrl_freq=240, act="rpm", task="hover")
        core = _core("cf2x", E, D, 7, S, "rpm", "hover", xyz, rpy, gpu_device, target=orc.TARGET_POS)
        _sync_c(core, orc)
        acts = (0.01 * rng.uniform(-1, 1, size=(1, E, D, 4)) + 0.01 * rng.uniform(-1, 1, size=(240, E, D, 4))).astype(np.float32)
        errs = _run(core, orc, acts, gpu_device, {1, 24, 120, 240})
    finally:
        from oracle import c_oracle
        c_oracle.lib().orc_set_threads(1)
    for t, e in sorted(errs.items()):
        print(f"t={t:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
        assert max(e.values()) < 1e-4, (t, e)
    np.testing.assert_allclose(core.reward.cpu().numpy(), orc.reward, rtol=1e-3, atol=1e-3)


def test_config4_524288_drones_240_steps_against_the_c_oracle(gpu_device):
    """BASELINE config 4's whole-node size (8 x 65 536 = 524 288 HoverAviaries) on ONE GPU: an oracle trajectory, not
    only size-independent properties -- 240 steps at 240 Hz, RPM actions, through gpd_rollout (4 launches of 60 steps:
    also the launch mode the scaling bench uses)."""
    rng = np.random.default_rng(524288)
    E, D, S, K = 524288, 1, 1, 60
    xyz = np.array([0, 0, 0.1125]) + rng.uniform(-0.5, 0.5, size=(E, D, 3)) * np.array([1, 1, 0])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    _all_threads()
    try:
        orc = CAviary(urdf("cf2x"), "cf2x", E, D, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240, ctrl_freq=240, act="rpm",
                      task="hover")
        core = _core("cf2x", E, D, 0, S, "rpm", "hover", xyz, rpy, gpu_device, target=orc.TARGET_POS)
        _sync_c(core, orc)
        bias = 0.01 * rng.uniform(-1, 1, size=(1, E, D, 4))
        maxima = {g: fl for g, (_, fl) in GROUPS.items()}
        for r in range(4):
            acts = (bias + 0.01 * rng.uniform(-1, 1, size=(K, E, D, 4))).astype(np.float32)
            obs, rew, te, tr = core.rollout(torch.as_tensor(acts, device=gpu_device))
            for k in range(K):
                orc.step_in_place(acts[k].astype(np.float64))
            ref = _oracle_kin(orc)
            for g, (sl, _) in GROUPS.items():
                maxima[g] = max(maxima[g], float(np.abs(ref[sl]).max()))
            kin = core.kin[:, :E].cpu().numpy().astype(np.float64)
            e = {g: float(np.abs(kin[sl] - ref[sl]).max() / maxima[g]) for g, (sl, _) in GROUPS.items()}
            print(f"t={(r + 1) * K:5d} " + " ".join(f"{g}={v:.2e}" for g, v in e.items()))
            assert max(e.values()) < 1e-4, (r, e)
            # the last step's observation rows and rewards of the rollout block
            o64 = orc.obs.reshape(E, 12)
            oerr = np.abs(obs[K - 1].cpu().numpy().astype(np.float64) - o64)
            oerr[:, 3:6] = np.abs((oerr[:, 3:6] + np.pi) % (2 * np.pi) - np.pi)
            assert oerr.max() < 1e-4
            np.testing.assert_allclose(rew[K - 1].cpu().numpy(), orc.reward, rtol=1e-3, atol=1e-3)
    finally:
        from oracle import c_oracle
        c_oracle.lib().orc_set_threads(1)


def test_population_of_65536_closed_loop_hover_episodes_matches_the_float64_population(gpu_device):
    """VERDICT r05 #1(b).  Past half a second of DSLPID at the reference's 30 Hz (HoverAviary's default, envs/HoverAviary.py:16-17;
    control/DSLPIDControl.py:212-259 rides its torque clip) no two runs agree drone by drone -- two float64 runs do not -- so
    what must hold is that the fp32 POPULATION behaves like the float64 one: 65 536 HoverAviaries spread over the 1.5 m box,
    ActionType.PID towards one random waypoint each up to 0.5 m away (about a third of the episodes end early -- the first
    lunge tilts the drone past 0.4 rad, or carries it over the edge of the box -- the rest at the 8 s limit), same-step auto-reset,
    260 control steps = one full 8 s episode and the start of the next, fp32 HIP against oracle/gpd_oracle.c.  Compared:
      * the distributions of the first episode's return and of the step it ended at: relative difference of the means < 1e-3,
        two-sample Kolmogorov-Smirnov p > 0.01;
      * at EVERY step the population mean and standard deviation of position and attitude against what two float64 populations
        differ by (the second one nudged by the fp32 state array's rounding, relative 2^-24 sqrt(8) per control step): within 3x of
        that envelope -- over ALL aviaries, and over the aviaries that are in the same episode on both sides (1.2 % are not: a tilt
        within rounding of 0.4 rad ends the episode on one side a step earlier, and from then on one side shows the reset pose where
        the other still flies).  VERDICT r05 asked for "within 1e-4 of the scale" here; two float64 populations do not meet that
        (measured, round 6, nudged float64 against float64 over all aviaries: rpy mean 6.6e-4, rpy std 4.3e-4, pos mean 1.8e-4, pos
        std 1.9e-4 of the scale -- 65 536 chattering attitude loops average to ~1e-4 rad, not to zero), so the float64 pair is the
        yardstick: the fp32 population reads 6.3e-4 / 4.3e-4 / 1.5e-4 / 1.7e-4, 0.84 - 0.99 of it; over the aviaries in the same episode
        fp32 | float64 pair: rpy mean 4.6e-4 | 5.7e-4, rpy std 2.7e-4 | 2.3e-4, pos mean 9.4e-5 | 6.8e-5, pos std 1.2e-4 | 1.3e-4
        (profiles/r06_parity_measured.log)."""
    import os
    from scipy import stats
    from oracle import bullet_math as bm
    from oracle import c_oracle
    rng = np.random.default_rng(20260)
    E, S, T = 65536, 8, 260
    xyz = (np.array([0, 0, 0.8]) + rng.uniform(-1, 1, size=(E, 1, 3)) * np.array([1.3, 1.3, 0.5])).astype(np.float32).astype(np.float64)
    rpy = rng.uniform(-0.1, 0.1, size=(E, 1, 3)).astype(np.float32).astype(np.float64)
    wp = (xyz + rng.uniform(-1, 1, size=(E, 1, 3)) * np.array([0.5, 0.5, 0.3])).astype(np.float32)
    mk = lambda: CAviary(urdf("cf2x"), "cf2x", E, 1, initial_xyzs=xyz, initial_rpys=rpy, pyb_freq=240, ctrl_freq=30, act="pid",  # noqa: E731
                         task="hover", auto_reset=True)
    orc, orn = mk(), mk()
    core = _core("cf2x", E, 1, 0, S, "pid", "hover", xyz, rpy, gpu_device, auto_reset=True, target=orc.TARGET_POS)
    wp_dev, wp64 = torch.as_tensor(wp, device=gpu_device), wp.astype(np.float64)
    sides = ("hip", "f64", "f64n")
    ret = {s: np.zeros(E) for s in sides}
    end = {s: np.zeros(E, dtype=np.int64) for s in sides}
    hist = {s: np.zeros(E, dtype=np.int64) for s in sides}        # episodes ended so far: equal on two sides = the same episode
    groups = (("pos", slice(0, 3)), ("rpy", slice(3, 6)))
    keys = [f"{g}_{m}" for g, _ in groups for m in ("mean", "std")]
    matched = {k: [0.0, 0.0] for k in keys}                        # [fp32 vs f64, f64 nudged vs f64], worst step, aviaries in the same episode
    everyone = {k: [0.0, 0.0] for k in keys}                       # the same over all aviaries
    eps = 2.0 ** -24 * np.sqrt(S)
    c_oracle.lib().orc_set_threads(min(len(os.sched_getaffinity(0)), c_oracle.lib().orc_max_threads()))
    try:
        for k in range(T):
            orc.step_in_place(wp64)
            orn.step_in_place(wp64)
            for name in ("pos", "quat", "vel", "rpy_rates"):
                arr = getattr(orn, name)
                arr *= 1.0 + eps * rng.choice([-1.0, 1.0], size=arr.shape)
            orn.rpy = np.ascontiguousarray(bm.euler_from_quaternion_b(orn.quat))
            obs, rew, term, trunc = core.step(wp_dev)
            o = {"hip": obs.cpu().numpy().astype(np.float64).reshape(E, 12), "f64": orc.obs.reshape(E, 12), "f64n": orn.obs.reshape(E, 12)}
            for side, r, done in (("hip", rew.cpu().numpy().astype(np.float64), (term | trunc).cpu().numpy()),
                                  ("f64", orc.reward, (orc.terminated | orc.truncated).astype(bool)),
                                  ("f64n", orn.reward, (orn.terminated | orn.truncated).astype(bool))):
                first = end[side] == 0
                ret[side] += np.where(first, r, 0.0)
                end[side] = np.where(first & done, k + 1, end[side])
                hist[side] += done
            # (the observation row of an aviary that ended in this step is its reset pose: same-step auto-reset)
            same = {"hip": hist["hip"] == hist["f64"], "f64n": hist["f64n"] == hist["f64"]}
            for g, sl in groups:
                scale = max(float(np.abs(o["f64"][:, sl]).max()), 1.0)
                for m, fn in (("mean", np.mean), ("std", np.std)):
                    key = f"{g}_{m}"
                    ref = fn(o["f64"][:, sl], axis=0)
                    for j, side in enumerate(("hip", "f64n")):
                        sel = same[side]
                        matched[key][j] = max(matched[key][j], float(np.abs(fn(o[side][sel][:, sl], axis=0) - fn(o["f64"][sel][:, sl], axis=0)).max() / scale))
                        everyone[key][j] = max(everyone[key][j], float(np.abs(fn(o[side][:, sl], axis=0) - ref).max() / scale))
    finally:
        c_oracle.lib().orc_set_threads(1)
    assert all((end[s] > 0).all() for s in sides)                       # every aviary finished its first episode (time limit: step 241)
    early = float((end["f64"] < 241).mean())
    d_ret = abs(ret["hip"].mean() - ret["f64"].mean()) / abs(ret["f64"].mean())
    d_end = abs(end["hip"].mean() - end["f64"].mean()) / end["f64"].mean()
    ks_ret, ks_end = stats.ks_2samp(ret["hip"], ret["f64"]), stats.ks_2samp(end["hip"], end["f64"])
    same_end, same_end_n = float((end["hip"] == end["f64"]).mean()), float((end["f64n"] == end["f64"]).mean())
    print(f"POPULATION 65536 x PID @ 30 Hz: {early:.3f} of the episodes end before the time limit; return mean {ret['f64'].mean():.3f} (f64) "
          f"rel diff {d_ret:.2e}, KS D {ks_ret.statistic:.2e} p {ks_ret.pvalue:.3f}; end step mean {end['f64'].mean():.2f} rel diff {d_end:.2e}, "
          f"KS D {ks_end.statistic:.2e} p {ks_end.pvalue:.3f}; same end step drone by drone: fp32 {same_end:.4f}, nudged float64 {same_end_n:.4f}; "
          f"still in the same episode at the end {float((hist['hip'] == hist['f64']).mean()):.4f}")
    for name, table in (("aviaries in the same episode on both sides", matched), ("ALL aviaries", everyone)):
        print(f"POPULATION moments, worst step, {name} [fp32 vs f64 | nudged f64 vs f64]: " + " ".join(f"{n}={a:.2e}|{b:.2e}" for n, (a, b) in table.items()))
    assert 0.15 < early < 0.8                                           # the scene does produce a distribution of episode lengths
    assert d_ret < 1e-3 and d_end < 1e-3
    assert ks_ret.pvalue > 0.01 and ks_end.pvalue > 0.01
    assert abs(same_end - same_end_n) < 0.01                            # the fp32 run leaves the float64 one as often as a float64 run does
    for table in (matched, everyone):
        for key, (a, b_) in table.items():
            assert a <= 3.0 * b_ + 2e-5, (key, a, b_)
            assert a < 1e-3, (key, a)                                   # ... and small in absolute terms whatever the yardstick reads


def _sync_c(core, orc):
    """CAviary has the same state attributes as BatchedAviary (the PID block aside, unused here)."""
    assert core.pid is None
    _sync_from_oracle(core, orc)


def test_524288_drones_free_fall_and_hover_known_answers(gpu_device):
    """SURVEY App. A.6 KATs at config-4 size on one GPU: rpm = 0 -> exact semi-implicit free fall;
    rpm = HOVER_RPM -> equilibrium."""
    E, k = 524288, 240
    core = _core("cf2x", E, 1, 0, 1, "raw_rpm", "none", None, None, gpu_device)
    z0 = 0.1125
    zero = torch.zeros((E, 4), device=gpu_device)
    for _ in range(k):
        core.step(zero)
    kin = core.kin[:, :E]
    g, dt = 9.8, 1.0 / 240
    vz = kin[9].cpu().numpy().astype(np.float64)
    z = kin[2].cpu().numpy().astype(np.float64)
    assert np.ptp(vz) == 0 and np.ptp(z) == 0                       # every lane computes the same thing
    assert vz[0] == pytest.approx(-g * dt * k, rel=2e-6)
    assert z[0] == pytest.approx(z0 - g * dt * dt * k * (k + 1) / 2, rel=2e-6)
    assert float(kin[0:2].abs().max()) == 0 and float(kin[10:13].abs().max()) == 0
    np.testing.assert_array_equal(kin[3:7, 0].cpu().numpy(), [0, 0, 0, 1])
    # hover: 4*KF*HOVER_RPM^2 = GRAVITY -> |a| at fp32 rounding level
    core.reset()
    hov = torch.full((E, 4), float(core.P.HOVER_RPM), device=gpu_device)
    for _ in range(k):
        core.step(hov)
    assert abs(float(core.kin[9, 0])) < 2e-5 and abs(float(core.kin[2, 0]) - z0) < 2e-5


def test_batch_size_independence_and_permutation_equivariance(gpu_device):
    """A drone's trajectory does not depend on how many others share the launch nor on its lane:
    bitwise identical results for the same drone in a 4096- and a 524 288-drone batch, and under a
    permutation of the batch."""
    rng = np.random.default_rng(99)
    big, small, steps = 524288, 4096, 24
    xyz = (np.array([0, 0, 0.5]) + rng.uniform(-0.5, 0.5, size=(big, 1, 3))).astype(np.float32).astype(np.float64)
    rpy = rng.uniform(-0.3, 0.3, size=(big, 1, 3))
    acts = torch.as_tensor(rng.uniform(-1, 1, size=(steps, big, 1, 3)).astype(np.float32) * 0.5, device=gpu_device)
    perm = torch.as_tensor(rng.permutation(big), device=gpu_device)
    a = _core("cf2x", big, 1, 3, 2, "pid", "hover", xyz, rpy, gpu_device)
    b = _core("cf2x", small, 1, 3, 2, "pid", "hover", xyz[:small], rpy[:small], gpu_device)
    pc = perm.cpu().numpy()
    c = _core("cf2x", big, 1, 3, 2, "pid", "hover", xyz[pc], rpy[pc], gpu_device)
    for k in range(steps):
        a.step(acts[k])
        b.step(acts[k, :small].contiguous())
        c.step(acts[k][perm].contiguous())
    assert torch.equal(a.kin[:, :small], b.kin[:, :small])
    assert torch.equal(a.obs12[:small], b.obs12[:small])
    assert torch.equal(a.pid[:, :small], b.pid[:, :small])
    assert torch.equal(a.reward[:small], b.reward[:small])
    assert torch.equal(a.kin[:, :big][:, perm], c.kin[:, :big])
    assert torch.equal(a.obs12[perm], c.obs12)
    # quaternion norm is preserved by the exponential update (never renormalised, App. B.6)
    qn = a.kin[3:7, :big].pow(2).sum(0).sqrt()
    assert float((qn - 1).abs().max()) < 1e-5


def test_multihover_131072x2_reward_is_sum_of_hover_rewards(gpu_device):
    """BASELINE config 5 size on one GPU: with downwash off, a 2-drone MultiHover aviary is two independent
    drones; its kinematics are bitwise those of single-drone runs and its reward is the fp32 sum of theirs."""
    rng = np.random.default_rng(5)
    E, D, steps = 131072, 2, 16
    xyz = rng.uniform(-0.3, 0.3, size=(E, D, 3)) + np.array([0, 0, 0.6])
    rpy = rng.uniform(-0.1, 0.1, size=(E, D, 3))
    tgt = xyz + np.array([[0, 0, 1.0], [0, 0, 0.5]])
    acts = torch.as_tensor(rng.uniform(-1, 1, size=(steps, E, D, 4)).astype(np.float32) * 0.3, device=gpu_device)
    m = _core("cf2x", E, D, 1, 8, "rpm", "multihover", xyz, rpy, gpu_device, target=tgt)
    s = _core("cf2x", E * D, 1, 1, 8, "rpm", "hover", xyz.reshape(E * D, 1, 3), rpy.reshape(E * D, 1, 3), gpu_device,
              target=tgt.reshape(E * D, 1, 3))
    for k in range(steps):
        m.step(acts[k])
        s.step(acts[k].reshape(E * D, 1, 4))
        # two template instantiations of the kernel (MULTI / not): same source expressions, so the results
        # are expected to be bitwise equal; 1e-6 allows for a different FMA contraction choice
        assert torch.allclose(m.kin[:, :E * D], s.kin[:, :E * D], rtol=0, atol=1e-6)
        assert torch.allclose(m.obs12, s.obs12, rtol=0, atol=1e-6)
        r = s.reward.view(E, D)
        assert torch.allclose(m.reward, (r[:, 0] + r[:, 1]), rtol=0, atol=2e-6)
    assert int(m.step_counter[0]) == steps * 8


@pytest.mark.parametrize("workload,K,steps", [("hover65536_240hz", 20, 240), ("hover65536_240hz", 64, 128),
                                              ("hover65536_ext_240hz", 20, 120), ("multihover2x16384_240hz", 20, 120),
                                              # config 2 at 30 Hz, and the closed-loop (ActionType.PID) runs
                                              ("hover4096_30hz", 64, 128), ("hover65536_pid_240hz", 20, 120),
                                              ("hover65536_ext_pid_240hz", 20, 120), ("multihover2x16384_pid_240hz", 20, 120)])
def test_the_timed_workload_of_bench_py_against_the_c_oracle(gpu_device, workload, K, steps):
    """What bench.py TIMES, not a gentler stand-in: 65 536 HoverAviaries at 240 Hz, U(-1, 1) RPM actions that change every step,
    same-step auto-reset on, through `gpd_rollout` with K steps per launch (K = 20: the driver's `--steps 20`) -- replayed
    through the float64 C oracle from the device's own state by bench.py's checker, `oracle.bench_checks.parity_check` (the block the bench line carries).
    SURVEY.md section 8(d)'s metric, every field group below 1e-4; episodes do end inside the window (tilt / box truncation:
    the reset path is exercised), and the aviaries whose flags flip within rounding of a threshold stay a handful.  The other
    cases: the same for the workloads of the other BASELINE configs, open loop and with DSLPID closing the loop in the kernel
    (waypoints around TARGET_POS that change every step)."""
    import bench
    w = bench.WORKLOADS[workload]
    env = bench.make_env(w, gpu_device, seed=1000)
    acts = bench.make_actions(w, env, gpu_device, seed=2000, pool=64)
    for _ in range(3):                                   # not from the reset poses: a few launches in, like after a timed region
        bench.launch_rollout(env, acts, K)
    from oracle.bench_checks import parity_check
    res = parity_check(w, env, acts, steps, K, bench.launch_rollout, max_steps=steps)
    print(res)
    assert res["checked_steps"] == steps and res["launches"] == [f"rollout{K}"] * (steps // K)
    assert res["max"] < 1e-4 and res["ok"], res
    assert max(res["obs_every_step"].values()) < 1e-4, res
    assert res["flag_mismatch_frac"] < 2e-3, res
    assert res["reward_max_abs"] < 1e-3, res
    if w["D"] == 1 and not w["phys"]:
        assert res["episodes_ended_in_window"] > w["E"] // 4, res


def test_swarm_of_65536_drones_forces_and_a_second_of_flight(gpu_device):
    """ONE world at bench size (`swarm65536_ext_240hz`): after 48 steps through the persistent path (a binning every 16th
    sub-step, stale cell order, wake lists) the forces of all 65 536 drones against the float64 all-pairs loop of the reference
    (`BaseAviary._downwash`, 4.3e9 pairs in C) on the same positions -- bench_extra.py's checker, `oracle.bench_checks.swarm_parity_check`, the block its swarm
    lines carry -- and the trajectory bit for bit that of a twin that bins before every force evaluation."""
    import bench
    import bench_extra  # noqa: F401 -- registers the one-world workloads
    w = bench.WORKLOADS["swarm65536_ext_240hz"]
    env = bench.make_env(w, gpu_device, seed=1000)
    from gym_pybullet_drones_amd.envs import SwarmAviary
    from gym_pybullet_drones_amd.utils.enums import Physics
    twin = SwarmAviary(w["D"], initial_xyzs=env.INIT_XYZS, initial_rpys=env.INIT_RPYS, physics=Physics.PYB_GND_DRAG_DW, pyb_like="damped", pyb_freq=240,
                       ctrl_freq=240, act="raw_rpm", device=gpu_device, cell=env.cell, rebin_every=1)       # (the bench's world: every term on)
    acts = bench.make_actions(w, env, gpu_device, seed=2000, pool=8)
    va, _ = env.reset()
    vb, _ = twin.reset()
    for k in range(48):
        va, *_ = env.step(acts[k % 8])
        vb, *_ = twin.step(acts[k % 8])
    assert torch.equal(va, vb) and torch.equal(env.dw_force, twin.dw_force)
    assert env.wake_lists and float(env._list_ok.float().mean()) > 0.99
    from oracle.bench_checks import swarm_parity_check
    res = swarm_parity_check(env)
    print(res)
    assert res["ok"] and res["drones_with_a_force"] > 30000, res
