"""Host side of the hot path: owns the structure-of-arrays drone state in HBM (PyTorch-ROCm
tensors) and launches the HIP kernels of `csrc/libgpd.so` through the C-ABI (`include/gpd.h`).

PyTorch is used for device memory and streams only; every arithmetic step of the simulator runs
in the hand-written kernels.  Layout (N = num_envs * drones_per_env, drone n = env*D + d):

    kin_store float32 [13*ld]    the kinematic block in four planes (include/gpd.h, GpdState.kin; ld = N rounded up to 64):
                                 kin_P [ld][4] pos xyz + body rate x | kin_Q [ld][4] quat xyzw | kin_V [ld][4] vel xyz + body rate y |
                                 kin_W [ld] body rate z -- views of the one buffer; `kin` is the logical [13][ld] matrix
                                 (pos xyz | quat xyzw | vel xyz | body rates xyz), assembled on request (a copy)
    last_rpm  float32 [4][ld]    last applied RPMs (optional)
    pid       float32 [9][ld]    DSLPID integrators / last rpy (PID action types only)
    counter   int32   [E]        physics steps since the env's last reset
    obs12     float32 [N][12]    pos | rpy | vel | ang_v — row-major, ready for a policy / all-gather
    reward    float32 [E], terminated/truncated bool (1 byte) [E]
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import _native
from .params import DroneParams, PIDGains, euler_to_quat, trunc_counter
from .utils.enums import DroneModel, PHYS_DRAG, Physics

TASK_NONE, TASK_HOVER, TASK_MULTIHOVER = 0, 1, 2
_ACT_DIM = {0: 4, 1: 3, 2: 4, 3: 1, 4: 1, 5: 4, 6: 4}
_PID_ACTS = (1, 2, 4)


def action_needs_fix(a, device):
    return a.device != device or a.dtype != torch.float32 or not a.is_contiguous()


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _raw_stream(device) -> int:
    """hipStream_t of torch's current stream on `device` (the accessor torch's own compiled kernels use: no Stream object per call)"""
    if _get_raw_stream is not None and device.index is not None:
        return _get_raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


#: MI355X: 256 CUs x 4 SIMDs
_NUM_SIMDS = 1024


def lanes_per_wave(num_drones: int, drones_per_env: int) -> int:
    """Active lanes per 64-wide wavefront for single-drone aviaries (`GpdStepCfg.lanes_per_wave`).

    Measured on MI355X (profiles/r01_lanes_per_wave.txt): at N = 65 536 the full width wins (5.76 us vs
    6.22 us at 32 and 7.02 us at 16 lanes) -- the launch is bound by the serialised load -> compute ->
    store phases of waves that all start together, not by per-wave instruction latency, so spreading the
    batch over more, narrower waves only adds wave launches.  The knob stays in the ABI for tuning;
    `GPD_LANES_PER_WAVE` overrides."""
    env = os.environ.get("GPD_LANES_PER_WAVE")
    if env:
        return int(env)
    return 64


def kin_rows_from_planes(store, ld: int):
    """The four planes of `GpdState.kin` (a flat array of 13 * ld floats: torch tensor or numpy array) -> the logical [13, ld] matrix
    pos xyz | quat xyzw | vel xyz | body rates xyz (a copy)."""
    P, Q, V, W = store[:4 * ld].reshape(ld, 4), store[4 * ld:8 * ld].reshape(ld, 4), store[8 * ld:12 * ld].reshape(ld, 4), store[12 * ld:13 * ld]
    rows = [P[:, 0], P[:, 1], P[:, 2], Q[:, 0], Q[:, 1], Q[:, 2], Q[:, 3], V[:, 0], V[:, 1], V[:, 2], P[:, 3], V[:, 3], W]
    return torch.stack(rows) if torch.is_tensor(store) else np.stack(rows)


_KIN_COPY = ("SimCore.kin is a copy of the state (the device keeps it in four planes since ABI 9): write through the views "
             "kin_P / kin_Q / kin_V / kin_W (positions(), quaternions(), velocities()), or pass an edited clone to set_state(kin=...)")


class KinRows(torch.Tensor):
    """What `SimCore.kin` returns: a COPY of the state as the logical [13, ld] matrix.  Writing into it (or into a slice or row of it)
    would be lost without a trace -- up to ABI 8 `kin` WAS the device's state -- so item assignment AND every in-place torch operation
    on it or on a view of it (`kin[0:3].copy_(x)`, `kin[2].fill_(1.)`, `kin[:, i].zero_()` ...) raise (`kin += x` rebinds the name to a new tensor: torch turns the
    TypeError of an augmented assignment into Python's out-of-place fallback); `clone()` gives a plain
    tensor to edit.  Every access builds the full 13 x ld copy: loops read `positions()` / `velocities()` / `kin_P` ... instead."""

    def __setitem__(self, key, value):
        raise TypeError(_KIN_COPY)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        name = getattr(func, "__name__", "")
        if name.endswith("_") and not name.endswith("__") and args and isinstance(args[0], KinRows):      # torch's in-place naming rule
            raise TypeError(_KIN_COPY)
        return super().__torch_function__(func, types, args, kwargs or {})

    def clone(self, *args, **kwargs):
        return super().clone(*args, **kwargs).as_subclass(torch.Tensor)


class SimCore:
    """E aviaries x D drones advanced by one fused kernel launch per `step()`."""

    def __init__(self, drone_model: DroneModel = DroneModel.CF2X, num_envs: int = 1, drones_per_env: int = 1,
                 physics=Physics.DYN, pyb_freq: int = 240, ctrl_freq: int = 240, act_code: int = 0,
                 task: int = TASK_NONE, initial_xyzs=None, initial_rpys=None, target_pos=None,
                 episode_len_sec: float = 8.0, xy_bound: float = 1.5, z_bound: float = 2.0, tilt_bound: float = 0.4,
                 term_dist: float = 1e-4, auto_reset: bool = False, track_rpm: bool = True,
                 keep_terminal_obs: bool = False, device=None, gains: PIDGains = None, force_pid: bool = False,
                 pyb_like: bool = None, nan_guard: bool = False, host_visible: bool = False):
        """`host_visible`: keep the state and the per-step outputs in page-locked HOST memory that the device addresses directly
        (one aviary of a few drones: the reference-shaped `HoverAviary()` & co).  A step is then ONE launch + one stream
        synchronisation -- the kernel reads the action and the state over the link and writes state, observation row, reward and
        flags where the host reads them, no copy engine, no second launch; `step()` / `reset()` return after the stream has
        drained.  Same kernels, same bits (`test_host_visible_core_is_bitwise_the_device_core`).  Not for batches: every byte
        crosses PCIe."""
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("[ERROR] pyb_freq is not divisible by ctrl_freq.")
        self.lib = _native.lib()                      # raises if the HIP extension is missing
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is None or torch.device(device).type != "cuda":
            raise _native.GpdError("the simulator's hot path runs on an MI355X only: no CUDA/HIP device available "
                                   "(there is no CPU fallback)")
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._dev_index = self.device.index
        self.E, self.D = int(num_envs), int(drones_per_env)
        self.N = self.E * self.D
        if not (1 <= self.D <= 256):
            raise ValueError("drones_per_env must be in 1..256")
        self.ld = (self.N + 63) // 64 * 64
        self.P = DroneParams(drone_model)
        # a `Physics` member selects the add-on force models and, for PYB* (unless pyb_like is off), the ground plane and Bullet's
        # default damping; an int is the raw GPD_PHYS_* mask
        self.physics_flags = physics.mask(pyb_like) if isinstance(physics, Physics) else int(physics)
        self.act_code = int(act_code)
        self.A = _ACT_DIM[self.act_code]
        self.uses_pid = self.act_code in _PID_ACTS
        if self.uses_pid and drone_model == DroneModel.RACE:
            raise ValueError("[ERROR] no controller is available for the specified drone_model")
        self.S = pyb_freq // ctrl_freq
        self.pyb_freq, self.ctrl_freq = pyb_freq, ctrl_freq
        self.task = task
        self.auto_reset = bool(auto_reset)
        self._params = self.P.to_struct(pid_model=DroneModel.CF2X, gains=gains)   # BaseRLAviary.py:75-76

        dev, f32 = self.device, torch.float32
        self.host_visible = bool(host_visible)

        def buf(shape, dtype=f32):
            """a zeroed buffer the kernels read and write: HBM, or (host_visible) page-locked host memory mapped into the device's
            address space (hipHostMalloc through torch's pinned allocator: the host pointer IS the device pointer)"""
            if self.host_visible:
                with torch.cuda.device(dev):
                    return torch.zeros(shape, dtype=dtype).pin_memory()
            return torch.zeros(shape, dtype=dtype, device=dev)

        self.kin_store = buf((13 * self.ld,))
        ld = self.ld
        self.kin_P, self.kin_Q = self.kin_store[:4 * ld].view(ld, 4), self.kin_store[4 * ld:8 * ld].view(ld, 4)
        self.kin_V, self.kin_W = self.kin_store[8 * ld:12 * ld].view(ld, 4), self.kin_store[12 * ld:]
        need_rpm = track_rpm or bool(self.physics_flags & PHYS_DRAG)
        self.last_rpm = buf((4, self.ld)) if need_rpm else None
        # (force_pid: the controller state exists although the kernel is fed RPMs -- a host-side caller runs gpd_pid on it)
        self.pid = buf((9, self.ld)) if (self.uses_pid or force_pid) else None
        self.step_counter = buf((self.E,), torch.int32)
        self.obs12 = buf((self.N, 12))
        self.reward = buf((self.E,))
        # torch.bool is one byte holding 0/1, exactly what the kernel stores: no conversion kernel needed
        self.terminated = buf((self.E,), torch.bool)
        self.truncated = buf((self.E,), torch.bool)
        self.term_obs12 = buf((self.N, 12)) if keep_terminal_obs else None
        # host_visible: the action row is written by the host where the kernel reads it (`step(core.action_host)`)
        self.action_host = buf((self.N, self.A)) if self.host_visible else None

        # initial poses: (D,3) shared by all envs, or (E,D,3) per env
        if initial_xyzs is None:
            initial_xyzs = self.P.default_init_xyzs(self.D)
        xyz = np.asarray(initial_xyzs, dtype=np.float64)
        rpy = np.zeros((self.D, 3)) if initial_rpys is None else np.asarray(initial_rpys, dtype=np.float64)
        if xyz.shape not in ((self.D, 3), (self.E, self.D, 3)):
            raise ValueError(f"initial_xyzs must have shape ({self.D},3) or ({self.E},{self.D},3), got {xyz.shape}")
        if rpy.shape not in ((self.D, 3), (self.E, self.D, 3)):
            raise ValueError(f"initial_rpys must have shape ({self.D},3) or ({self.E},{self.D},3), got {rpy.shape}")
        self.init_per_env = int(xyz.ndim == 3 or rpy.ndim == 3)
        if self.init_per_env:
            xyz = np.broadcast_to(xyz, (self.E, self.D, 3))
            rpy = np.broadcast_to(rpy, (self.E, self.D, 3))
        self.INIT_XYZS, self.INIT_RPYS = np.array(xyz), np.array(rpy)
        pose = np.concatenate([xyz, euler_to_quat(rpy)], axis=-1).reshape(-1, 7)
        self.init_pose = torch.tensor(pose, dtype=f32, device=dev).contiguous()
        self.set_target(target_pos)

        # nan_guard: one byte per drone, rewritten by every call that stores the state: 1 = a NaN / infinity sits in the drone's
        # kinematic state (GpdState.bad; the reference has no such check)
        self.bad = buf((self.N,), torch.bool) if nan_guard else None
        self._state = _native.GpdState(kin=self.kin_store.data_ptr(),
                                       last_rpm=self.last_rpm.data_ptr() if self.last_rpm is not None else None,
                                       pid=self.pid.data_ptr() if self.pid is not None else None,
                                       step_counter=self.step_counter.data_ptr(), ld=self.ld,
                                       bad=self.bad.data_ptr() if nan_guard else None)
        self._cfg = _native.GpdStepCfg(
            num_envs=self.E, drones_per_env=self.D, act_type=self.act_code, substeps=self.S,
            physics_flags=self.physics_flags, pyb_dt=1.0 / pyb_freq, ctrl_dt=1.0 / ctrl_freq,
            inv_ctrl_dt=float(ctrl_freq), lanes_per_wave=lanes_per_wave(self.N, self.D), task=task,
            xy_bound=xy_bound, z_bound=z_bound, tilt_bound=tilt_bound, term_dist=term_dist,
            trunc_counter=trunc_counter(episode_len_sec, pyb_freq), target_per_env=self.target_per_env,
            init_per_env=self.init_per_env, auto_reset=int(self.auto_reset))
        self.reset()

    # ------------------------------------------------------------------------------------------
    def set_target(self, target_pos):
        """TARGET_POS of the task: (D,3) shared, or (E,D,3) per env; None = zeros."""
        if target_pos is None:
            target_pos = np.zeros((self.D, 3))
        tp = np.asarray(target_pos, dtype=np.float64)
        if tp.shape not in ((self.D, 3), (self.E, self.D, 3)):
            raise ValueError(f"target_pos must have shape ({self.D},3) or ({self.E},{self.D},3), got {tp.shape}")
        self.target_per_env = int(tp.ndim == 3)
        self.TARGET_POS = tp
        self.target = torch.tensor(tp.reshape(-1, 3), dtype=torch.float32, device=self.device).contiguous()
        self._step_args = self._host_args = None     # (hold the old buffer's address)
        if hasattr(self, "_cfg"):
            self._cfg.target_per_env = self.target_per_env

    def _stream(self):
        if self._own_stream is not None:
            return self._own_stream
        return ctypes.c_void_p(_raw_stream(self.device))

    _own_stream = None
    _pinned = None
    _host_args = None
    _step_args = None       # the arguments of gpd_step that never change between two calls, as ctypes objects (built on first use)

    def use_stream(self, stream: "torch.cuda.Stream" = None):
        """Pin every launch of this core to `stream` (None: back to torch's current stream).  For independent batches that
        advance as independent chains of launches on their own streams (bench.py --split): the caller orders them against the
        rest of the program with events / `wait_stream`; the torch-side copies of `rollout(update_latest=True)` still go to
        torch's current stream."""
        self._pinned = stream           # (keeps the torch stream object alive)
        self._own_stream = None if stream is None else ctypes.c_void_p(stream.cuda_stream)

    #: bumped by every method that changes the kinematic state (callers that cache something derived from the positions --
    #: SwarmAviary's downwash forces -- compare it)
    state_version = 0

    def reset(self, mask=None, reset_pid: bool = False):
        """Masked reset (mask: uint8/bool tensor [E] on the device, None = all envs)."""
        self.state_version += 1
        if mask is not None:
            mask = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_reset(ctypes.byref(self._state), _ptr(self.init_pose), self.init_per_env, _ptr(mask),
                                    self.E, self.D, int(reset_pid), _ptr(self.obs12), self._stream())
        _native.check(rc, "gpd_reset")
        if self.host_visible:
            self.drain()
        if self.bad is not None:          # (the non-finite flags describe the state the last launch left: a reset pose is finite)
            # on the stream the kernels of this core run on: behind an in-flight step's flag store, before the next launch
            with torch.cuda.stream(self._pinned) if self._pinned is not None else contextlib.nullcontext():
                if mask is None:
                    self.bad.zero_()
                else:
                    self.bad.view(self.E, self.D).masked_fill_(mask.to(device=self.bad.device, dtype=torch.bool).unsqueeze(1), False)   # (no host sync)
        return self.obs12

    def drain(self):
        """Wait until the stream this core launches on is empty (what makes host-visible buffers readable)."""
        if self._pinned is not None:
            self._pinned.synchronize()
        else:
            torch.cuda.current_stream(self.device).synchronize()

    def step_host(self):
        """`step(self.action_host)` of a host-visible core with nothing on the way: ONE library call (`gpd_step_sync`: the launch
        and the wait for its stream), the argument list built once, the stream from torch's raw-handle accessor (no Stream object,
        no device guard when the device is current).  What `BaseAviary.step()` calls."""
        a = self._host_args
        if a is None:
            if not self.host_visible:
                raise ValueError("step_host() needs a core built with host_visible=True")
            a = self._host_args = (ctypes.byref(self._params), ctypes.byref(self._state), ctypes.byref(self._cfg), _ptr(self.action_host),
                                   _ptr(self.target), _ptr(self.init_pose), _ptr(self.obs12), _ptr(self.reward), _ptr(self.terminated),
                                   _ptr(self.truncated), _ptr(self.term_obs12), self.lib.gpd_step_sync)
        if self._own_stream is not None or _get_raw_stream is None or torch.cuda.current_device() != self._dev_index:
            self.step(self.action_host)
            return
        self.state_version += 1
        rc = a[11](a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], _get_raw_stream(self._dev_index))
        if rc:
            _native.check(rc, "gpd_step_sync")

    # ---- the kinematic block as the logical [13][ld] matrix (rows: pos xyz | quat xyzw | vel xyz | body rates xyz) -------------
    @property
    def kin(self) -> torch.Tensor:
        """A COPY of the kinematic state as the [13, ld] matrix of rounds 1-4 (the device keeps it in four planes, `kin_P` /
        `kin_Q` / `kin_V` / `kin_W`: write through those views, or through `set_state(kin=...)`)."""
        return kin_rows_from_planes(self.kin_store, self.ld).as_subclass(KinRows)

    def positions(self, n: int = None) -> torch.Tensor:
        """[n, 3] VIEW of the positions"""
        return self.kin_P[:self.N if n is None else n, :3]

    def quaternions(self, n: int = None) -> torch.Tensor:
        """[n, 4] VIEW of the quaternions (x, y, z, w)"""
        return self.kin_Q[:self.N if n is None else n]

    def velocities(self, n: int = None) -> torch.Tensor:
        """[n, 3] VIEW of the linear velocities"""
        return self.kin_V[:self.N if n is None else n, :3]

    def body_rates(self, n: int = None) -> torch.Tensor:
        """[n, 3] COPY of the body rates (the reference's rpy_rates)"""
        n = self.N if n is None else n
        return torch.stack([self.kin_P[:n, 3], self.kin_V[:n, 3], self.kin_W[:n]], dim=1)

    def step(self, action: torch.Tensor):
        """One env step for every aviary.  `action`: float32 device tensor with E*D*A elements.

        Returns views of the persistent output tensors (obs12 [N,12], reward [E], terminated [E],
        truncated [E]); they are overwritten by the next call.  Asynchronous on the current stream.
        """
        if action is not self.action_host and (action.device != self.device or action.dtype != torch.float32 or not action.is_contiguous()):
            action = action.to(device=self.device, dtype=torch.float32).contiguous()
        if action.numel() != self.N * self.A:
            raise ValueError(f"action has {action.numel()} elements, expected {self.N}x{self.A}")
        self.state_version += 1
        # (an eager loop is bound by this function, not by the kernel -- 9.3 us of host time per call against 3-4 us of GPU
        # time, scratch/host_overhead.py: the structs' references and the persistent buffers' addresses are built once, the
        # stream comes from torch's raw-handle accessor, the device guard is skipped when the device is current already)
        a = self._step_args
        if a is None:
            a = self._step_args = (ctypes.byref(self._params), ctypes.byref(self._state), ctypes.byref(self._cfg),
                                   _ptr(self.target), _ptr(self.init_pose), _ptr(self.obs12), _ptr(self.reward),
                                   _ptr(self.terminated), _ptr(self.truncated), _ptr(self.term_obs12))
        if torch.cuda.current_device() == self._dev_index:
            rc = self.lib.gpd_step(a[0], a[1], a[2], ctypes.c_void_p(action.data_ptr()), a[3], a[4], a[5], a[6], a[7], a[8], a[9],
                                   self._stream())
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.gpd_step(a[0], a[1], a[2], ctypes.c_void_p(action.data_ptr()), a[3], a[4], a[5], a[6], a[7], a[8], a[9],
                                       self._stream())
        if rc:
            _native.check(rc, "gpd_step")
        if self.host_visible:
            self.drain()
        return self.obs12, self.reward, self.terminated, self.truncated

    def rollout(self, actions: torch.Tensor, num_steps: int = None, last_only: bool = False,
                update_latest: bool = True, push_history: bool = False):
        """K consecutive env steps of every aviary in ONE kernel launch (`gpd_rollout`).

        `actions`: float32 device tensor `[K, E*D*A]` (any shape with K leading and E*D*A trailing
        elements): step t uses block t.  With `num_steps=K` and a single action block (E*D*A elements) the
        same action is applied K times (e.g. ActionType.PID holding one waypoint).

        Returns `(obs12 [K,N,12], reward [K,E], terminated [K,E], truncated [K,E])` -- persistent buffers that
        the next rollout of the same length overwrites -- or, with `last_only=True`, only the last step's
        values in the usual `step()` output tensors.  Bitwise identical to K calls of `step()`.
        """
        per = self.N * self.A
        if action_needs_fix(actions, self.device):
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        if num_steps is None:
            if actions.numel() % per != 0 or actions.numel() == 0:
                raise ValueError(f"actions has {actions.numel()} elements, expected K x {self.N}x{self.A}")
            K, a_stride = actions.numel() // per, per
        else:
            K = int(num_steps)
            if actions.numel() == per:
                a_stride = 0
            elif actions.numel() == K * per:
                a_stride = per
            else:
                raise ValueError(f"actions has {actions.numel()} elements, expected {per} or {K}x{per}")
        if K < 1:
            raise ValueError("num_steps must be >= 1")
        if last_only:
            obs, rew, term, trunc, tobs = self.obs12, self.reward, self.terminated, self.truncated, self.term_obs12
            o_stride = e_stride = 0
        else:
            buf = self._rollout_buffers(K)
            obs, rew, term, trunc, tobs = buf
            o_stride, e_stride = self.N * 12, self.E
        self.pushed_history = False
        self.state_version += 1
        with torch.cuda.device(self.device):
            if push_history and getattr(self, "act_ring", None) is not None and tobs is None:
                # the kernel pushes every step's action into the ring itself (`gpd_rollout_history`)
                rc = self.lib.gpd_rollout_history(ctypes.byref(self._params), ctypes.byref(self._state), ctypes.byref(self._cfg),
                                                  K, _ptr(actions), a_stride, _ptr(self.target), _ptr(self.init_pose),
                                                  _ptr(obs), o_stride, _ptr(rew), _ptr(term), _ptr(trunc), e_stride, self._stream())
                if rc != _native.GPD_ENOTSUP:     # only "no fused variant for this shape" falls through (nothing was launched);
                    _native.check(rc, "gpd_rollout_history")     # a real failure must not be retried on an already advanced state
                    self.pushed_history = True
            if not self.pushed_history:           # (no fused variant for this shape: the caller updates the ring with full_obs())
                rc = self.lib.gpd_rollout(ctypes.byref(self._params), ctypes.byref(self._state), ctypes.byref(self._cfg),
                                          K, _ptr(actions), a_stride, _ptr(self.target), _ptr(self.init_pose),
                                          _ptr(obs), o_stride, _ptr(rew), _ptr(term), _ptr(trunc), e_stride,
                                          _ptr(tobs), self._stream())
                _native.check(rc, "gpd_rollout")
        if not last_only and update_latest:
            self.obs12.copy_(obs[K - 1])
            self.reward.copy_(rew[K - 1])
            self.terminated.copy_(term[K - 1])
            self.truncated.copy_(trunc[K - 1])
            if tobs is not None and self.auto_reset:      # term_obs12 = what K single steps would have left
                self._latest_terminal(tobs, term, trunc, K)
        return obs, rew, term, trunc

    def rollout_policy(self, policy, num_steps: int, want_actions: bool = True, noise: torch.Tensor = None, action_std=None,
                       mean_out: torch.Tensor = None):
        """K env steps in ONE launch with `policy` (a `policy.MlpPolicy`) evaluated inside the kernel
        (`gpd_rollout_policy`): a_t = policy(o_t), o_{t+1}, r_t, ... = step(a_t), starting from the latest observation.
        Returns `(obs12 [K,N,12], reward [K,E], terminated [K,E], truncated [K,E], actions [K,N,A] or None)` -- the same
        persistent buffers `rollout()` uses; the latest-step tensors (`obs12`, `reward`, ...) are updated as well.

        Training rollouts: `noise` `[K,N,A]` (standard-normal draws, e.g. `torch.randn`) and `action_std` (A floats =
        exp(log_std)) make it `a_t = clip(mean_t + action_std * noise_t, -1, 1)`, SB3's collection loop; `mean_out`
        `[K,N,A]` receives the unclipped means.  With `keep_terminal_obs` the kernel also writes the last observation of every
        episode that ends inside the launch: `terminal_observations(K)` returns the `[K,N,12]` block (rows of step t of the
        aviaries that ended at step t; the flags say which), and `term_obs12` holds those of the last step -- what SB3's
        `VecEnv` hands its PPO as `infos[i]["terminal_observation"]` (examples/learn.py:61-95)."""
        K = int(num_steps)
        if K < 1:
            raise ValueError("num_steps must be >= 1")
        std = None
        if noise is not None and action_std is None:
            raise ValueError("noise comes with action_std (A floats = exp(log_std))")
        if noise is not None:
            if noise.device != self.device or noise.dtype != torch.float32 or not noise.is_contiguous() or noise.numel() != K * self.N * self.A:
                raise ValueError(f"noise must be a contiguous float32 tensor of {K}x{self.N}x{self.A} elements on {self.device}")
            vals = [float(v) for v in (action_std.detach().cpu().reshape(-1).tolist() if torch.is_tensor(action_std) else list(action_std))]
            if len(vals) != self.A:
                raise ValueError(f"action_std must hold {self.A} values")
            std = (ctypes.c_float * self.A)(*vals)
            if mean_out is not None and (mean_out.device != self.device or mean_out.dtype != torch.float32 or not mean_out.is_contiguous()
                                         or mean_out.numel() != K * self.N * self.A):
                raise ValueError("mean_out must be a contiguous float32 tensor like noise")
        obs, rew, term, trunc, tobs = self._rollout_buffers(K)
        cache = self.__dict__.setdefault("_policy_actions", {})
        acts = None
        if want_actions:
            acts = cache.get(K)
            if acts is None:
                cache.clear()
                acts = cache[K] = torch.zeros((K, self.N, self.A), dtype=torch.float32, device=self.device)
        ps = policy.struct()
        self.state_version += 1
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_rollout_policy(ctypes.byref(self._params), ctypes.byref(self._state), ctypes.byref(self._cfg),
                                             ctypes.byref(ps), K, _ptr(self.obs12), _ptr(self.target), _ptr(self.init_pose),
                                             _ptr(acts), _ptr(obs), self.N * 12, _ptr(rew), _ptr(term), _ptr(trunc), self.E,
                                             _ptr(noise), std, _ptr(mean_out), _ptr(tobs), self._stream())
        _native.check(rc, "gpd_rollout_policy")
        if tobs is not None and self.auto_reset:       # (the kernel writes terminal rows only when it resets: K single steps of an
            self._latest_terminal(tobs, term, trunc, K)   # aviary without auto-reset leave term_obs12 untouched, and so does this)
        self.obs12.copy_(obs[K - 1])
        self.reward.copy_(rew[K - 1])
        self.terminated.copy_(term[K - 1])
        self.truncated.copy_(trunc[K - 1])
        return obs, rew, term, trunc, acts

    def _latest_terminal(self, tobs, term, trunc, K):
        """`term_obs12` after a K-step launch = what K single steps would have left: for every aviary the terminal observation
        of the LAST step it ended in (aviaries that did not end keep what they held).  A few small torch kernels, off the
        hot path (only with keep_terminal_obs)."""
        done = term | trunc                                                             # [K, E]
        last = (done.to(torch.int32) * torch.arange(1, K + 1, dtype=torch.int32, device=self.device).unsqueeze(1)).amax(dim=0)   # 0: never
        W = self.D * 12
        rows = tobs.view(K, self.E, W).gather(0, (last - 1).clamp(min=0).to(torch.int64).view(1, self.E, 1).expand(1, self.E, W))[0]
        cur = self.term_obs12.view(self.E, W)
        torch.where((last > 0).unsqueeze(1), rows, cur, out=cur)

    def terminal_observations(self, K: int) -> torch.Tensor:
        """`[K, N, 12]` terminal-observation block of the latest K-step `rollout()` / `rollout_policy()` (needs
        `keep_terminal_obs`): row (t, n) is meaningful where aviary n ended at step t."""
        buf = self.__dict__.get("_rollout_cache", {}).get(int(K))
        if buf is None or buf[4] is None:
            raise ValueError("no terminal observations: build with keep_terminal_obs=True and run a K-step rollout first")
        return buf[4]

    # ---- action history / full KIN observation rows (envs/BaseRLAviary.py:65-67, 153-154, 187, 307-320) ----
    def enable_history(self, hist_len: int):
        """Allocate the action ring: a DOUBLE ring `[2H][N][A]` (zeros, like the reference's pre-filled deque; never
        cleared by a reset, App. B.2) + the per-aviary ring positions, and hand them to the kernels through `GpdState`:
        from now on `step()` pushes its action into the ring inside the step kernel."""
        self.H = int(hist_len)
        self.act_ring = torch.zeros((2 * self.H, self.N, self.A), dtype=torch.float32, device=self.device)
        self.ring_pos = torch.zeros((self.E,), dtype=torch.int32, device=self.device)
        self.obs_full = None            # materialised rows, allocated on first request
        self._full_buf = None
        self._state.act_ring = self.act_ring.data_ptr()
        self._state.ring_pos = self.ring_pos.data_ptr()
        self._state.hist_len = self.H

    def history_view(self) -> torch.Tensor:
        """`[N, H, A]` STRIDED VIEW of the ring: the H most recent actions of every drone, oldest first (what the
        reference appends to the observation row, BaseRLAviary.py:317-318).  No copy; valid until the next step.  Reads
        the ring position from the device (one 4-byte copy): the aviaries of a core advance in lock-step."""
        p = int(self.ring_pos[0].item())
        return self.act_ring[p:p + self.H].permute(1, 0, 2)

    def history_rows(self, obs12: torch.Tensor = None) -> torch.Tensor:
        """Materialise the current full rows `[N, 12 + H*A]` (`gpd_hist_rows`: one gather kernel, no host sync)."""
        if self.obs_full is None:
            self.obs_full = torch.zeros((self.N, 12 + self.H * self.A), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_hist_rows(ctypes.byref(self._state), self.N, self.D, self.A,
                                        _ptr(self.obs12 if obs12 is None else obs12), _ptr(self.obs_full), self._stream())
        _native.check(rc, "gpd_hist_rows")
        return self.obs_full

    def full_obs(self, actions: torch.Tensor, obs12: torch.Tensor = None, num_steps: int = 1, want_rows: bool = True):
        """After `rollout(actions)`: (want_rows) assemble the full observation rows `[K, N, 12 + H*A]` of the K steps,
        then push the K actions into the ring (`gpd_full_obs`).  `actions`: what was passed to the rollout; `obs12`: its
        observation output (default: the latest rollout buffer)."""
        K = int(num_steps)
        per = self.N * self.A
        if action_needs_fix(actions, self.device):
            actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        a_stride = 0 if actions.numel() == per else per
        if actions.numel() not in (per, K * per):
            raise ValueError(f"actions has {actions.numel()} elements, expected {per} or {K}x{per}")
        W = 12 + self.H * self.A
        out = None
        if want_rows:
            if self._full_buf is None or self._full_buf.shape[0] != K:
                self._full_buf = torch.zeros((K, self.N, W), dtype=torch.float32, device=self.device)
            out = self._full_buf
            if obs12 is None:
                obs12 = self._rollout_buf[0]
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_full_obs(ctypes.byref(self._state), K, self.N, self.D, self.A,
                                       _ptr(obs12) if want_rows else _ptr(None), self.N * 12, _ptr(actions), a_stride,
                                       _ptr(out), self.N * W, self._stream())
        _native.check(rc, "gpd_full_obs")
        return out

    def _rollout_buffers(self, K: int):
        """Persistent output buffers of a K-step rollout (kept per K: the four most recent lengths stay allocated)."""
        cache = self.__dict__.setdefault("_rollout_cache", {})
        buf = cache.get(K)
        if buf is None:
            dev = self.device
            buf = (torch.zeros((K, self.N, 12), dtype=torch.float32, device=dev),
                   torch.zeros((K, self.E), dtype=torch.float32, device=dev),
                   torch.zeros((K, self.E), dtype=torch.bool, device=dev),
                   torch.zeros((K, self.E), dtype=torch.bool, device=dev),
                   torch.zeros((K, self.N, 12), dtype=torch.float32, device=dev) if self.term_obs12 is not None else None)
            while len(cache) >= 4:
                cache.pop(next(iter(cache)))
            cache[K] = buf
        self._rollout_buf = buf
        return buf

    def bytes_per_rollout(self, K: int, action_stride_zero: bool = False, last_only: bool = False) -> int:
        """Algorithmic HBM bytes of one `rollout()` of K steps: the state once, and per step the action
        row in, the observation row and the aviary's reward + flags out."""
        state = 2 * 13 * 4
        if self.uses_pid:
            state += 2 * 9 * 4
        if self.physics_flags & PHYS_DRAG:
            state += 4 * 4
        if self.last_rpm is not None:
            state += 4 * 4
        ka = 1 if action_stride_zero else K
        ko = 1 if last_only else K
        per_drone = state + ka * self.A * 4 + ko * 12 * 4
        per_env = 2 * 4 + ko * (4 + 2)
        return per_drone * self.N + per_env * self.E

    def bytes_full_rows(self, K: int = 1, push: bool = False) -> int:
        """Algorithmic HBM bytes of materialising the full observation rows of K steps (`history_rows` / `full_obs`): every
        row float is read once (obs12 row, ring slot or action block) and written once; `push`: plus the ring update of a
        rollout's post-pass (min(K, H) action blocks read and written twice)."""
        W = 12 + self.H * self.A
        b = 2 * K * self.N * W * 4
        if push:
            b += 3 * min(K, self.H) * self.N * self.A * 4 + 2 * 4 * self.E
        return b

    def state_vectors(self) -> torch.Tensor:
        """[N,20] state vectors in `_getDroneStateVector` order (BaseAviary.py:559-561)."""
        out = torch.empty((self.N, 20), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_state_vectors(ctypes.byref(self._state), _ptr(self.obs12), _ptr(out), self.N,
                                            self._stream())
        _native.check(rc, "gpd_state_vectors")
        return out

    def bad_envs(self) -> torch.Tensor:
        """[E] bool: aviaries with a non-finite kinematic state after the latest step / rollout (needs `nan_guard=True`).  A
        device tensor -- `.any().item()` is the one host sync a training loop pays when it wants to stop on the first NaN."""
        if self.bad is None:
            raise ValueError("built without nan_guard=True")
        return self.bad.view(self.E, self.D).any(dim=1)

    # ---- checkpoint / resume (host <-> device copies, off the hot path) ------------------------------
    #: everything a later step can depend on: the integrator state, the controllers' members, the last applied RPMs (drag), the
    #: episode clocks, the latest observation rows and task outputs (a policy rollout starts from `obs12`), and -- with an action
    #: history -- the ring with its positions (the reference's never-reset `action_buffer`, envs/BaseRLAviary.py:65-67)
    _STATE_FIELDS = ("kin", "last_rpm", "pid", "step_counter", "obs12", "reward", "terminated", "truncated", "term_obs12",
                     "act_ring", "ring_pos", "bad")

    def get_state(self) -> dict:
        """Snapshot (device clones) of the complete simulator state: `set_state(**get_state())` later -- on this core or on
        another one built with the same arguments -- continues bit for bit (same observations, history tails, rewards)."""
        n = self.N
        out = {}
        for name in self._STATE_FIELDS:
            t = getattr(self, name, None)
            if t is not None:
                out[name] = (t[:, :n] if name in ("kin", "last_rpm", "pid") else t).clone()      # (`kin`: the logical rows)
        return out

    def set_state(self, kin=None, last_rpm=None, pid=None, step_counter=None, **rest):
        """Overwrite the given parts of the state (tensors or arrays shaped like `get_state()`'s; absent or None: untouched)."""
        n = self.N
        self.state_version += 1
        given = dict(rest, kin=kin, last_rpm=last_rpm, pid=pid, step_counter=step_counter)
        unknown = set(given) - set(self._STATE_FIELDS)
        if unknown:
            raise TypeError(f"set_state() got unknown state fields {sorted(unknown)}")
        for name, v in given.items():
            t = getattr(self, name, None)
            if v is None or t is None:       # (a snapshot of a core with more optional parts than this one: the rest applies)
                continue
            dst = t[:, :n] if name in ("kin", "last_rpm", "pid") else t
            src = torch.as_tensor(v, device=self.device)
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"set_state: {name} has shape {tuple(src.shape)}, expected {tuple(dst.shape)}")
            if name == "kin" and self.bad is not None and given.get("bad") is None:
                self.bad[:n] = ~torch.isfinite(torch.as_tensor(v, device=self.device, dtype=torch.float32)).all(dim=0)       # (flags follow the state written)
            if name == "kin":                # logical rows -> the four planes
                src = src.to(torch.float32)
                self.kin_P[:n, :3] = src[0:3].t(); self.kin_P[:n, 3] = src[10]
                self.kin_Q[:n] = src[3:7].t()
                self.kin_V[:n, :3] = src[7:10].t(); self.kin_V[:n, 3] = src[11]
                self.kin_W[:n] = src[12]
                continue
            dst.copy_(src.to(dst.dtype))

    def bytes_per_step(self) -> int:
        """Algorithmic HBM bytes one `step()` moves (SURVEY.md §8d accounting)."""
        per_drone = (13 + self.A) * 4 + (13 + 12) * 4            # state+action in, state+obs out
        if self.uses_pid:
            per_drone += 2 * 9 * 4
        if self.physics_flags & PHYS_DRAG:
            per_drone += 4 * 4
        if self.last_rpm is not None:
            per_drone += 4 * 4
        per_env = 4 + 2 + 2 * 4                                   # reward + 2 flags + counter r/w
        if getattr(self, "act_ring", None) is not None:           # action pushed into both halves of the double ring
            per_drone += 2 * self.A * 4
            per_env += 2 * 4                                      # ring position r/w
        return per_drone * self.N + per_env * self.E
