"""`MlpPolicy`: the weights of a deterministic MLP actor, resident on the device, for `gpd_rollout_policy` -- K env steps
per launch with the policy evaluated INSIDE the kernel (the loop of the reference's `examples/learn.py:157-192`:
`action, _ = model.predict(obs, deterministic=True); obs, ... = env.step(action)`).

The architecture is the actor of Stable-Baselines3's default `MlpPolicy` (`stable-baselines3 ^2.0`, a dependency of the
reference, `pyproject.toml`; `policies.py: ActorCriticPolicy` with `net_arch = dict(pi=[64, 64], vf=[64, 64])`,
`activation_fn = nn.Tanh`): flatten -> Linear(in, 64) -> tanh -> Linear(64, 64) -> tanh -> Linear(64, act_dim) = the mean
of the action distribution = the deterministic action, clipped to the Box bounds [-1, 1] by `predict()`.
"""
import ctypes

import torch

from . import _native


class GpdPolicy(ctypes.Structure):
    """mirror of `struct GpdPolicy` (include/gpd.h)"""
    _fields_ = [("w1", ctypes.c_void_p), ("b1", ctypes.c_void_p), ("w2", ctypes.c_void_p), ("b2", ctypes.c_void_p),
                ("w3", ctypes.c_void_p), ("b3", ctypes.c_void_p), ("in_dim", ctypes.c_int32), ("hidden", ctypes.c_int32),
                ("activation", ctypes.c_int32), ("pad_", ctypes.c_int32)]


class MlpPolicy:
    """in_dim -> 64 -> 64 -> act_dim, tanh (default) or ReLU; torch `Linear` layout (`weight [out, in]`, `bias [out]`)."""

    HIDDEN = 64

    def __init__(self, w1, b1, w2, b2, w3, b3, activation: str = "tanh", device=None):
        dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        t = lambda x: torch.as_tensor(x, dtype=torch.float32).to(dev).contiguous()  # noqa: E731
        self.w1, self.b1, self.w2, self.b2, self.w3, self.b3 = t(w1), t(b1), t(w2), t(b2), t(w3), t(b3)
        self.in_dim, self.act_dim = int(self.w1.shape[1]), int(self.w3.shape[0])
        if self.w1.shape[0] != self.HIDDEN or tuple(self.w2.shape) != (self.HIDDEN, self.HIDDEN) or self.w3.shape[1] != self.HIDDEN \
                or self.b1.numel() != self.HIDDEN or self.b2.numel() != self.HIDDEN or self.b3.numel() != self.act_dim:
            raise ValueError("MlpPolicy expects Linear(in, 64) -> Linear(64, 64) -> Linear(64, act_dim) weights")
        if activation not in ("tanh", "relu"):
            raise ValueError("activation must be 'tanh' or 'relu'")
        self.activation = activation
        self.device = dev

    def struct(self) -> GpdPolicy:
        return GpdPolicy(w1=self.w1.data_ptr(), b1=self.b1.data_ptr(), w2=self.w2.data_ptr(), b2=self.b2.data_ptr(),
                         w3=self.w3.data_ptr(), b3=self.b3.data_ptr(), in_dim=self.in_dim, hidden=self.HIDDEN,
                         activation=1 if self.activation == "relu" else 0)

    @classmethod
    def random(cls, in_dim: int, act_dim: int, seed: int = 0, gain: float = 1.0, activation: str = "tanh", device=None):
        """Random weights (normal, std = gain / sqrt(fan_in); small output layer like SB3's 0.01-gain action head)."""
        g = torch.Generator().manual_seed(seed)
        n = lambda o, i, s: torch.randn((o, i), generator=g) * (s / i ** 0.5)  # noqa: E731
        return cls(n(64, in_dim, gain), torch.randn(64, generator=g) * 0.1, n(64, 64, gain), torch.randn(64, generator=g) * 0.1,
                   n(act_dim, 64, gain * 0.5), torch.randn(act_dim, generator=g) * 0.05, activation=activation, device=device)

    @classmethod
    def from_sb3(cls, model, device=None):
        """The actor of a Stable-Baselines3 on-policy model (`PPO("MlpPolicy", env)`, `examples/learn.py:61-66`) with the
        default `net_arch` (two 64-unit layers)."""
        pol = model.policy
        layers = [m for m in pol.mlp_extractor.policy_net if hasattr(m, "weight")]
        if len(layers) != 2:
            raise ValueError("expected the default two-layer policy network")
        act = type([m for m in pol.mlp_extractor.policy_net if not hasattr(m, "weight")][0]).__name__.lower()
        return cls(layers[0].weight.detach(), layers[0].bias.detach(), layers[1].weight.detach(), layers[1].bias.detach(),
                   pol.action_net.weight.detach(), pol.action_net.bias.detach(), activation="relu" if "relu" in act else "tanh",
                   device=device)

    def __call__(self, obs: torch.Tensor) -> torch.Tensor:
        """The same policy as plain torch operations (one launch per layer): what an RL loop would run BETWEEN two
        `env.step()` calls; `(..., in_dim) -> (..., act_dim)` in [-1, 1]."""
        f = torch.relu if self.activation == "relu" else torch.tanh
        h = f(torch.addmm(self.b1, obs.reshape(-1, self.in_dim), self.w1.t()))
        h = f(torch.addmm(self.b2, h, self.w2.t()))
        return torch.addmm(self.b3, h, self.w3.t()).clamp_(-1.0, 1.0).reshape(obs.shape[:-1] + (self.act_dim,))
