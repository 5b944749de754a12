"""The example scripts run end to end and do what they say (the reference's own test strategy: `tests/test_examples.py`
calls each example's `run(...)`; here the results are asserted as well)."""
import importlib.util
import os

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location("example_" + name, os.path.join(REPO, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_pid(gpu_device, tmp_path):
    m = _load("pid")
    err, final = m.run(duration_sec=2, output_folder=str(tmp_path), device=gpu_device)
    assert err < 0.15 and final.shape == (3, 3) and np.all(np.abs(final[:, 2] - [0.1, 0.15, 0.2]) < 0.05)
    err_b, final_b = m.run(duration_sec=2, mode="batched", num_envs=256, log=False, device=gpu_device)
    assert err_b < 0.15
    np.testing.assert_allclose(final_b, final, atol=0.05)     # same scene; the 48 Hz loop chatters at the cm level


def test_pid_velocity(gpu_device):
    m = _load("pid_velocity")
    vel, limit = m.run(duration_sec=3, device=gpu_device)
    speed = np.linalg.norm(vel, axis=1)
    assert np.all(speed > 0.5 * limit) and np.all(speed < 1.5 * limit)
    assert vel[0, 0] > 0 and vel[1, 1] > 0 and vel[2, 0] < 0 and vel[3, 2] > 0


def test_downwash(gpu_device):
    m = _load("downwash")
    z_dw, z_no, sag = m.run(duration_sec=2, swarm=4000, device=gpu_device)
    assert z_dw < z_no - 1e-3              # the wake pushes the lower drone down
    assert sag is not None and sag > 1e-4  # ... in the swarm too


def test_rollout(gpu_device):
    m = _load("rollout")
    shape = m.run(num_envs=2048, steps=32, device=gpu_device)
    assert tuple(shape) == (32, 2048, 1, 12 + 15)


def test_learn(gpu_device):
    """examples/learn.py: PPO on 2048 GPU-resident HoverAviaries (the reference's learn.py trains SB3-PPO on one
    aviary and stops at a mean episode reward of 474 -- essentially the bang-bang optimum of this task: 484 minus the
    ~10 lost while climbing 0.89 m at +-10 % thrust).  A few seconds of training must get most of the way there."""
    m = _load("learn")
    history, eval_ret = m.run(num_envs=2048, iters=25, verbose=False, device=gpu_device)      # rollouts collected INSIDE the kernel
    assert history[0] < 250                      # an untrained policy drifts away / times out low
    assert eval_ret > 440 and max(history) > 430
    # ... and with the policy as torch operations between the steps (how the loop looked before the kernel could sample)
    history_t, eval_t = m.run(num_envs=2048, iters=25, verbose=False, device=gpu_device, collect="torch")
    assert history_t[0] < 250 and eval_t > 440 and max(history_t) > 430
    # the evaluation episode ran twice: 242 launches with the actor as torch operations in between, and ONE launch with the
    # trained actor inside the kernel -- the same policy from the same start, the same return (fp32 torch vs bf16 hi/lo MFMA)
    assert abs(m.run.last_eval_torch - eval_t) < 0.1
