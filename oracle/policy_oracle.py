"""Float64 restatement of the deterministic MLP actor the in-kernel policy rollout evaluates (TEST INFRASTRUCTURE, see
oracle/__init__.py).

What it follows.  The reference's evaluation loop, `examples/learn.py:157-192`:
    action, _states = model.predict(obs, deterministic=True); obs, reward, terminated, truncated, info = test_env.step(action)
with `model = PPO('MlpPolicy', train_env)` (`learn.py:61-66`).  The policy is third-party: `stable-baselines3 ^2.0`
(`pyproject.toml` of the reference; not vendored, not installed here).  Its published algorithm for the default
`MlpPolicy` actor (`common/policies.py: ActorCriticPolicy`, `common/torch_layers.py: MlpExtractor`, `net_arch =
dict(pi=[64, 64], vf=[64, 64])`, `activation_fn = nn.Tanh`):
    features = flatten(obs)                                   (FlattenExtractor)
    latent   = tanh(W2 tanh(W1 features + b1) + b2)           (mlp_extractor.policy_net)
    mean     = W3 latent + b3                                 (action_net); deterministic action = mean
    predict(): clip(mean, action_space.low, action_space.high) = clip(mean, -1, 1)   (`base_class.py: predict`, Box spaces)
PARITY UNPINNED against SB3 itself (no SB3 here); the arithmetic is three affine maps and tanh.
"""
import numpy as np


def mlp_actor(obs, w1, b1, w2, b2, w3, b3, activation="tanh", clip=True):
    """obs [..., in_dim] -> clipped deterministic action [..., act_dim], all float64 (`clip=False`: the mean of the action
    distribution before `predict()` clips it to the Box)."""
    f = np.tanh if activation == "tanh" else (lambda x: np.maximum(x, 0.0))
    x = np.asarray(obs, dtype=np.float64)
    h = f(x @ np.asarray(w1, dtype=np.float64).T + np.asarray(b1, dtype=np.float64))
    h = f(h @ np.asarray(w2, dtype=np.float64).T + np.asarray(b2, dtype=np.float64))
    y = h @ np.asarray(w3, dtype=np.float64).T + np.asarray(b3, dtype=np.float64)
    return np.clip(y, -1.0, 1.0) if clip else y


def policy_loop(aviary, weights, num_steps, obs0, history=None, activation="tanh"):
    """`examples/learn.py:157-192` on a batched oracle aviary (`BatchedAviary` / `CAviary`): from the observation rows `obs0`
    [E, D, 12] (and the action history [E, D, H, A], oldest first, or None for a policy that sees the 12 kinematic floats only)
    -> lists of observations, rewards, terminated, truncated, actions per step."""
    obs = np.asarray(obs0, dtype=np.float64)
    hist = None if history is None else np.array(history, dtype=np.float64)
    out = {"obs": [], "reward": [], "terminated": [], "truncated": [], "actions": []}
    for _ in range(num_steps):
        row = obs if hist is None else np.concatenate([obs, hist.reshape(hist.shape[:2] + (-1,))], axis=-1)
        a = mlp_actor(row, *weights, activation=activation)
        o, r, te, tr, _ = aviary.step(a)
        if hist is not None:
            hist = np.concatenate([hist[:, :, 1:], a[:, :, None, :]], axis=2)
        obs = o
        for k, v in zip(("obs", "reward", "terminated", "truncated", "actions"), (o, r, te, tr, a)):
            out[k].append(np.array(v))
    return {k: np.stack(v) for k, v in out.items()}
