"""Batched aviaries: E independent Hover / MultiHover / control aviaries advanced by ONE kernel launch.

New relative to the reference, which vectorises only through SB3's `make_vec_env(..., n_envs=1)`
(`examples/learn.py:54-58`).  The per-aviary semantics are exactly those of `HoverAviary` /
`MultiHoverAviary` / `CtrlAviary`; on top of that the batch follows the SB3 `DummyVecEnv`
convention: an aviary that terminates or is truncated is reset inside the same `step()` and the
observation returned for it is the first one of the new episode (the last one of the finished
episode is available as `info["terminal_observation"]` when `keep_terminal_obs=True`).  As in the
reference, a reset neither clears the action history nor the embedded PID state
(SURVEY.md App. B.2/B.3).

Everything stays on the GPU: actions come in and observations / rewards / flags go out as torch
tensors on the aviary's device, nothing synchronises with the host.
"""
import numpy as np
import torch

from .. import engine
from ..params import DroneParams
from ..utils.enums import ACT_RAW_RPM, ActionType, DroneModel, ObservationType, Physics

_TASKS = {"none": engine.TASK_NONE, "hover": engine.TASK_HOVER, "multihover": engine.TASK_MULTIHOVER}


class VectorAviary:
    """E aviaries x D drones.  Observations `(E, D, 12)` (or `(E, D, 12 + H*A)` with `full_obs`)."""

    def __init__(self,
                 num_envs: int,
                 num_drones: int = 1,
                 drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None,
                 initial_rpys=None,
                 physics: Physics = Physics.DYN,
                 pyb_freq: int = 240,
                 ctrl_freq: int = 30,
                 obs: ObservationType = ObservationType.KIN,
                 act=ActionType.RPM,
                 task: str = "hover",
                 target_pos=None,
                 episode_len_sec: float = 8,
                 auto_reset: bool = True,
                 full_obs: bool = False,
                 keep_terminal_obs: bool = False,
                 track_rpm: bool = False,
                 pyb_like: bool = None,
                 nan_guard: bool = False,
                 device=None):
        if obs != ObservationType.KIN:
            raise NotImplementedError("only ObservationType.KIN is on the MI355X hot path")
        self.NUM_ENVS, self.NUM_DRONES = int(num_envs), int(num_drones)
        self.DRONE_MODEL, self.PHYSICS = drone_model, physics
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.PYB_STEPS_PER_CTRL = pyb_freq // ctrl_freq
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1. / ctrl_freq, 1. / pyb_freq
        self.EPISODE_LEN_SEC = episode_len_sec
        self.ACT_TYPE = act
        act_code = ACT_RAW_RPM if act == "raw_rpm" else act.code
        P = DroneParams(drone_model)
        self.HOVER_RPM, self.MAX_RPM = P.HOVER_RPM, P.MAX_RPM
        if initial_xyzs is None:
            initial_xyzs = P.default_init_xyzs(self.NUM_DRONES)
        init = np.asarray(initial_xyzs, dtype=np.float64)
        if target_pos is None:
            if task == "hover":
                target_pos = np.broadcast_to(np.array([0., 0., 1.]), (self.NUM_DRONES, 3)).copy()
            elif task == "multihover":
                target_pos = init + np.array([[0, 0, 1 / (i + 1)] for i in range(self.NUM_DRONES)])
        xy = 1.5 if task == "hover" else 2.0
        self.core = engine.SimCore(drone_model=drone_model, num_envs=num_envs, drones_per_env=num_drones,
                                   physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, act_code=act_code,
                                   task=_TASKS[task], initial_xyzs=init, initial_rpys=initial_rpys,
                                   target_pos=target_pos, episode_len_sec=episode_len_sec, xy_bound=xy,
                                   auto_reset=auto_reset, track_rpm=track_rpm, keep_terminal_obs=keep_terminal_obs,
                                   device=device, pyb_like=pyb_like, nan_guard=nan_guard)
        self.device = self.core.device
        self.ACT_DIM = self.core.A
        self.INIT_XYZS, self.INIT_RPYS, self.TARGET_POS = self.core.INIT_XYZS, self.core.INIT_RPYS, self.core.TARGET_POS
        # full_obs: False -> (E, D, 12) rows only, no action history kept;
        #           True  -> the reference's (E, D, 12 + H*A) rows, materialised after every step (one gather kernel);
        #           "lazy" -> the action ring is kept (pushed inside the step kernel, +2 x 16 B per drone and step) and
        #                     step() returns the (E, D, 12) rows; `history()` is a zero-copy strided view of the ring,
        #                     `full_rows()` materialises the reference's rows when a consumer wants them
        if full_obs not in (False, True, "lazy"):
            raise ValueError("full_obs must be False, True or 'lazy'")
        self.lazy_history = full_obs == "lazy"
        self.full_obs = full_obs is True
        self.ACTION_BUFFER_SIZE = int(ctrl_freq // 2)
        self.OBS_DIM = 12 + (self.ACTION_BUFFER_SIZE * self.ACT_DIM if self.full_obs else 0)
        if full_obs:
            self.core.enable_history(self.ACTION_BUFFER_SIZE)

    # ---- gymnasium-VectorEnv-like surface ----------------------------------------------------
    @property
    def num_envs(self):
        return self.NUM_ENVS

    def _obs(self):
        E, D = self.NUM_ENVS, self.NUM_DRONES
        if not self.full_obs:
            return self.core.obs12.view(E, D, 12)
        return self.core.obs_full.view(E, D, self.OBS_DIM)

    def history(self) -> torch.Tensor:
        """(E, D, H, A) zero-copy strided VIEW of the action ring: the last H actions of every drone, oldest first
        (the tail the reference appends to the observation row, BaseRLAviary.py:317-318).  Valid until the next step."""
        H = self.ACTION_BUFFER_SIZE
        return self.core.history_view().unflatten(0, (self.NUM_ENVS, self.NUM_DRONES))

    def action_history(self) -> torch.Tensor:
        """(E*D, H, A) copy of the last H actions, oldest first."""
        return self.core.history_view().contiguous()

    def full_rows(self) -> torch.Tensor:
        """(E, D, 12 + H*A): the reference's observation rows for the current state, materialised now."""
        return self.core.history_rows().view(self.NUM_ENVS, self.NUM_DRONES, -1)

    def reset(self, seed=None, options=None, mask=None):
        """Reset all aviaries (or those selected by the boolean/uint8 tensor `mask` [E]).  As in the reference the
        action history (and the embedded PID state) survives a reset."""
        self.core.reset(mask=mask)
        if self.full_obs:      # rows of the reset poses with the unchanged history tail (ring not advanced)
            self.core.history_rows()
        return self._obs(), {}

    def step(self, action: torch.Tensor):
        """action: float32 tensor (E, D, A) on `self.device` -> (obs, reward[E], terminated[E], truncated[E], info)."""
        _, reward, terminated, truncated = self.core.step(action)      # (pushes the action into the ring, if there is one)
        if self.full_obs:
            self.core.history_rows()                         # row assembly: one more launch
        info = {}
        if self.core.term_obs12 is not None:
            info["terminal_observation"] = self.core.term_obs12.view(self.NUM_ENVS, self.NUM_DRONES, 12)
        return self._obs(), reward, terminated, truncated, info

    def rollout(self, actions: torch.Tensor):
        """K env steps in ONE launch (`gpd_rollout`): actions (K, E, D, A) known up front (open-loop sequences,
        DSLPID waypoint lists).  Returns (obs (K,E,D,OBS_DIM), reward (K,E), terminated (K,E), truncated (K,E))."""
        K = actions.shape[0]
        # (lazy history: the rollout kernel pushes the actions into the ring itself when it has a variant for the shape)
        obs, reward, terminated, truncated = self.core.rollout(actions, push_history=bool(self.lazy_history) and not self.full_obs)
        if self.full_obs:
            obs = self.core.full_obs(actions, obs12=obs, num_steps=K)
            if self.core.obs_full is None:
                self.core.history_rows()
            self.core.obs_full.copy_(obs[K - 1])
        elif self.lazy_history and not self.core.pushed_history:
            self.core.full_obs(actions, num_steps=K, want_rows=False)      # ring update only
        return obs.view(K, self.NUM_ENVS, self.NUM_DRONES, -1), reward, terminated, truncated

    def rollout_policy(self, policy, num_steps: int, noise: torch.Tensor = None, action_std=None, mean_out: torch.Tensor = None):
        """K env steps in ONE launch with the policy in the loop (`policy.MlpPolicy`; the loop of
        `examples/learn.py:157-192`).  The policy's input is the (12,) kinematic row, or -- `full_obs` True / "lazy" and
        `policy.in_dim == 12 + H*A` -- the reference's full row with the action history.  Returns
        `(obs (K,E,1,12), reward (K,E), terminated (K,E), truncated (K,E), actions (K,E,1,A))`.
        `noise` (K,E,1,A) + `action_std` (A): sampled actions for training, `clip(mean + std * noise, -1, 1)`; `mean_out`
        (K,E,1,A) receives the unclipped means (see `SimCore.rollout_policy`)."""
        obs, reward, terminated, truncated, acts = self.core.rollout_policy(policy, num_steps, noise=noise, action_std=action_std,
                                                                            mean_out=mean_out)
        if self.full_obs:
            self.core.history_rows()
        E, D = self.NUM_ENVS, self.NUM_DRONES
        return obs.view(-1, E, D, 12), reward, terminated, truncated, acts.view(-1, E, D, self.ACT_DIM)

    def state_vectors(self) -> torch.Tensor:
        """(E, D, 20) `_getDroneStateVector`-ordered states (needs `track_rpm=True` for the RPM columns)."""
        return self.core.state_vectors().view(self.NUM_ENVS, self.NUM_DRONES, 20)

    # ---- checkpoint / resume; non-finite guard (both absent upstream) ----------------------
    def get_state(self) -> dict:
        """Complete snapshot (device clones): integrator state, controller members, episode clocks, latest observations and -- with
        `full_obs` -- the action ring and its positions.  `set_state(get_state())` resumes bit for bit, history tails included."""
        return self.core.get_state()

    def set_state(self, state: dict):
        self.core.set_state(**state)
        if self.full_obs:                    # the materialised rows follow from obs12 + ring
            self.core.history_rows()
        return self._obs()

    def bad_envs(self) -> torch.Tensor:
        """[E] bool device tensor: aviaries whose state holds a NaN / infinity after the latest step (`nan_guard=True`)."""
        return self.core.bad_envs()

    def close(self):
        pass


class VectorHoverAviary(VectorAviary):
    """E x `HoverAviary` (one drone each)."""

    def __init__(self, num_envs: int, drone_model: DroneModel = DroneModel.CF2X, initial_xyzs=None, initial_rpys=None,
                 physics: Physics = Physics.DYN, pyb_freq: int = 240, ctrl_freq: int = 30,
                 obs: ObservationType = ObservationType.KIN, act: ActionType = ActionType.RPM, **kw):
        super().__init__(num_envs=num_envs, num_drones=1, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, obs=obs,
                         act=act, task="hover", **kw)


class VectorMultiHoverAviary(VectorAviary):
    """E x `MultiHoverAviary` (`num_drones` drones each, default 2)."""

    def __init__(self, num_envs: int, num_drones: int = 2, drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.DYN, pyb_freq: int = 240,
                 ctrl_freq: int = 30, obs: ObservationType = ObservationType.KIN, act: ActionType = ActionType.RPM, **kw):
        super().__init__(num_envs=num_envs, num_drones=num_drones, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq, obs=obs,
                         act=act, task="multihover", **kw)


class VectorCtrlAviary(VectorAviary):
    """E x `CtrlAviary`: raw RPM actions clipped to [0, MAX_RPM], no task."""

    def __init__(self, num_envs: int, num_drones: int = 1, drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.DYN, pyb_freq: int = 240,
                 ctrl_freq: int = 240, **kw):
        kw.setdefault("auto_reset", False)
        kw.setdefault("track_rpm", True)
        super().__init__(num_envs=num_envs, num_drones=num_drones, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq,
                         act="raw_rpm", task="none", **kw)


class VectorVelocityAviary(VectorAviary):
    """E x `VelocityAviary`: velocity commands `[vx, vy, vz, fraction of SPEED_LIMIT]` tracked by the embedded DSLPID
    controllers (`GPD_ACT_VEL`), no task; `state_vectors()` gives the (E, D, 20) observation of the reference class."""

    def __init__(self, num_envs: int, num_drones: int = 1, drone_model: DroneModel = DroneModel.CF2X,
                 initial_xyzs=None, initial_rpys=None, physics: Physics = Physics.DYN, pyb_freq: int = 240,
                 ctrl_freq: int = 240, **kw):
        kw.setdefault("auto_reset", False)
        kw.setdefault("track_rpm", True)
        super().__init__(num_envs=num_envs, num_drones=num_drones, drone_model=drone_model, initial_xyzs=initial_xyzs,
                         initial_rpys=initial_rpys, physics=physics, pyb_freq=pyb_freq, ctrl_freq=ctrl_freq,
                         act=ActionType.VEL, task="none", **kw)


class VecEnvAdapter:
    """Stable-Baselines3 `VecEnv`-shaped front end of a `VectorAviary` (duck-typed: SB3 is not installed here).

    Follows `DummyVecEnv` (what `examples/learn.py:54-58` builds through `make_vec_env`): numpy in / numpy out,
    `dones = terminated | truncated`, aviaries that end are reset inside the same `step_wait()` and report the
    last observation of the finished episode as `infos[i]["terminal_observation"]` together with
    `infos[i]["TimeLimit.truncated"]`.  Observations are `(num_envs, D, 12 [+ H*A])` (squeezed to
    `(num_envs, 12 [+ H*A])` for one drone per aviary when `squeeze=True`).  The device tensors of the last step
    stay available as `last` for a GPU-resident learner that does not want the host copies.
    """

    def __init__(self, env: "VectorAviary", squeeze: bool = False):
        if not env.core.auto_reset:
            raise ValueError("VecEnvAdapter needs a VectorAviary built with auto_reset=True")
        self.env = env
        self.num_envs = env.NUM_ENVS
        self.squeeze = bool(squeeze and env.NUM_DRONES == 1)
        from .._gym_shim import spaces
        D, W, A = env.NUM_DRONES, env.OBS_DIM, env.ACT_DIM
        oshape, ashape = ((W,), (A,)) if self.squeeze else ((D, W), (D, A))
        self.observation_space = spaces.Box(low=np.full(oshape, -np.inf), high=np.full(oshape, np.inf), dtype=np.float32)
        self.action_space = spaces.Box(low=-np.ones(ashape), high=np.ones(ashape), dtype=np.float32)
        self._actions = None
        self.last = None
        if env.core.term_obs12 is None:     # terminal observations are part of the VecEnv contract
            env.core.term_obs12 = torch.zeros((env.core.N, 12), dtype=torch.float32, device=env.device)

    def _np_obs(self, obs):
        o = obs.cpu().numpy()
        return o[:, 0, :] if self.squeeze else o

    def reset(self):
        obs, _ = self.env.reset()
        return self._np_obs(obs)

    def step_async(self, actions):
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32), device=self.env.device)
        self._actions = a.reshape(self.num_envs, self.env.NUM_DRONES, self.env.ACT_DIM)

    def step_wait(self):
        env = self.env
        obs, reward, terminated, truncated, _ = env.step(self._actions)
        self.last = (obs, reward, terminated, truncated)
        packed = torch.stack([reward, terminated.to(torch.float32), truncated.to(torch.float32)]).cpu().numpy()
        rew, term, trunc = packed[0], packed[1] != 0, packed[2] != 0
        dones = term | trunc
        infos = [{} for _ in range(self.num_envs)]
        idx = np.flatnonzero(dones)
        if idx.size:
            tobs = env.core.term_obs12.view(self.num_envs, env.NUM_DRONES, 12)[torch.as_tensor(idx, device=env.device)].cpu().numpy()
            if env.full_obs:    # the history tail of the terminal observation is the one of the returned row
                tail = obs[torch.as_tensor(idx, device=env.device)][..., 12:].cpu().numpy()
                tobs = np.concatenate([tobs, tail], axis=-1)
            for j, i in enumerate(idx):
                infos[i]["terminal_observation"] = tobs[j, 0] if self.squeeze else tobs[j]
                infos[i]["TimeLimit.truncated"] = bool(trunc[i] and not term[i])
        return self._np_obs(obs), rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.env.close()

    def get_attr(self, name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self.env, name)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [False] * n

    def seed(self, seed=None):
        return [seed] * self.num_envs          # the simulator is deterministic: nothing draws random numbers


class GymVectorEnvAdapter:
    """`gymnasium.vector.VectorEnv`-shaped front end of a `VectorAviary` (duck-typed: gymnasium is optional here).

    The gymnasium >= 1.0 vector API (what `gymnasium.make_vec` / `SyncVectorEnv` give a learner that does not go through
    SB3's `make_vec_env`, `examples/learn.py:54-58`): `reset(seed=, options=) -> (obs, infos)`, `step(actions) ->
    (obs, rewards, terminations, truncations, infos)` with batched numpy arrays, `num_envs`, `single_observation_space`
    / `single_action_space` and their batched versions, and an explicit autoreset mode.  The kernel resets an aviary in
    the SAME step it ends in (`AutoresetMode.SAME_STEP`): the returned observation is the first one of the new episode
    and the last one of the finished episode is `infos["final_obs"][i]` (with the mask `infos["_final_obs"]`), as
    gymnasium's own same-step vector envs report it.  `last` keeps the device tensors of the latest step for a
    GPU-resident learner."""

    metadata = {"autoreset_mode": "SameStep", "render_modes": []}

    def __init__(self, env: "VectorAviary"):
        if not env.core.auto_reset:
            raise ValueError("GymVectorEnvAdapter needs a VectorAviary built with auto_reset=True")
        self.env = env
        self.num_envs = env.NUM_ENVS
        from .._gym_shim import spaces
        D, W, A = env.NUM_DRONES, env.OBS_DIM, env.ACT_DIM
        box = lambda shape, lo, hi: spaces.Box(low=np.full(shape, lo), high=np.full(shape, hi), dtype=np.float32)  # noqa: E731
        self.single_observation_space = box((D, W), -np.inf, np.inf)
        self.single_action_space = box((D, A), -1.0, 1.0)
        self.observation_space = box((self.num_envs, D, W), -np.inf, np.inf)
        self.action_space = box((self.num_envs, D, A), -1.0, 1.0)
        self.last = None
        self.closed = False
        if env.core.term_obs12 is None:
            env.core.term_obs12 = torch.zeros((env.core.N, 12), dtype=torch.float32, device=env.device)

    def reset(self, *, seed=None, options=None):
        mask = None if not options or "reset_mask" not in options else torch.as_tensor(options["reset_mask"], device=self.env.device)
        obs, _ = self.env.reset(seed=seed, mask=mask)
        return obs.cpu().numpy(), {}

    def step(self, actions):
        env = self.env
        a = torch.as_tensor(np.asarray(actions, dtype=np.float32), device=env.device).reshape(self.num_envs, env.NUM_DRONES, env.ACT_DIM)
        obs, reward, terminated, truncated, _ = env.step(a)
        self.last = (obs, reward, terminated, truncated)
        packed = torch.stack([reward, terminated.to(torch.float32), truncated.to(torch.float32)]).cpu().numpy()
        rew, term, trunc = packed[0], packed[1] != 0, packed[2] != 0
        obs_np = obs.cpu().numpy()
        infos = {}
        done = term | trunc
        if done.any():
            final = np.zeros_like(obs_np)
            tob = env.core.term_obs12.view(self.num_envs, env.NUM_DRONES, 12).cpu().numpy()
            final[done, :, :12] = tob[done]
            final[done, :, 12:] = obs_np[done, :, 12:]       # the history tail is not cleared by a reset (App. B.2)
            infos["final_obs"], infos["_final_obs"] = final, done
        return obs_np, rew, term, trunc, infos

    def close(self, **kwargs):
        self.closed = True
        self.env.close()

    def render(self):
        return None

    @property
    def unwrapped(self):
        return self
