"""`SwarmAviary`: ONE aviary of N drones, any N — with the pairwise downwash of the whole swarm.

The reference simulates one world per aviary and couples its drones only through `_downwash`
(`envs/BaseAviary.py:785-811`): an O(N²) Python loop over every pair with `dz > 0` and `dxy < 10 m`, once per
physics sub-step, on the positions all drones had at the start of the sub-step (`:346-347`).  The fused step
kernel covers aviaries of up to 256 drones (one workgroup, positions exchanged through LDS).  This class is the
large-world counterpart: the swarm is stepped as N single-drone lanes of the same kernel, and
the downwash force of each drone is computed per sub-step by `gpd_downwash_global` — uniform, periodic 10 m grid, counting
sort by cell, 3×3-cell neighbourhood search, order-independent fixed-point accumulation — and handed to the step
kernel as `state.dw_force`.

Interface: `CtrlAviary`-like.  `step(action)` takes raw RPMs `(N, 4)` clipped to `[0, MAX_RPM]`
(`envs/CtrlAviary.py:140`) — or, with `act=ActionType.PID`, waypoints `(N, 3)` tracked by N `DSLPIDControl`s
(`examples/downwash.py:93-113`) — and returns the `(N, 20)` state vectors.  Everything stays on the GPU.
One process / one GPU: a single world does not shard by aviary.
"""
import ctypes

import numpy as np
import torch

from .. import _native, engine
from ..control.DSLPIDControl import DSLPIDControlBatch
from ..params import DroneParams
from ..utils.enums import ACT_DIRECT_RPM, ACT_RAW_RPM, ActionType, DroneModel, PHYS_DW, Physics, warn_if_pyb


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class SwarmAviary:
    """One world, `num_drones` drones, explicit integrator + the selected force models over the whole swarm."""

    def __init__(self, num_drones: int, drone_model: DroneModel = DroneModel.CF2X, initial_xyzs=None, initial_rpys=None,
                 physics: Physics = Physics.PYB_DW, pyb_freq: int = 240, ctrl_freq: int = 240, act="raw_rpm",
                 world_min=None, world_max=None, cell: float = 10.0, zbin: float = 1.0, nz: int = 1, device=None,
                 pyb_like: bool = None):
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("[ERROR] in SwarmAviary.__init__(), pyb_freq is not divisible by ctrl_freq.")
        if act not in ("raw_rpm", ActionType.RPM, ActionType.PID):
            raise ValueError("SwarmAviary supports act = 'raw_rpm', ActionType.RPM or ActionType.PID")
        self.NUM_DRONES = N = int(num_drones)
        self.DRONE_MODEL, self.PHYSICS, self.ACT_TYPE = drone_model, physics, act
        warn_if_pyb(physics)
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.PYB_STEPS_PER_CTRL = pyb_freq // ctrl_freq
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1. / ctrl_freq, 1. / pyb_freq
        P = DroneParams(drone_model)
        self.HOVER_RPM, self.MAX_RPM = P.HOVER_RPM, P.MAX_RPM
        if initial_xyzs is None:
            initial_xyzs = P.default_init_xyzs(N)
        xyz = np.asarray(initial_xyzs, dtype=np.float64).reshape(N, 1, 3)
        rpy = np.zeros((N, 1, 3)) if initial_rpys is None else np.asarray(initial_rpys, dtype=np.float64).reshape(N, 1, 3)
        self.INIT_XYZS, self.INIT_RPYS = xyz[:, 0], rpy[:, 0]
        # the kernel runs N single-drone lanes, one physics sub-step per launch.  The action -> RPM mapping is the kernel's own
        # (GPD_ACT_RAW_RPM: clip to [0, MAX_RPM], envs/CtrlAviary.py:140; GPD_ACT_RPM: HOVER_RPM (1 + 0.05 a),
        # envs/BaseRLAviary.py:191-192); waypoint actions go through the batched DSLPID kernel and arrive as RPMs.
        act_code = {"raw_rpm": ACT_RAW_RPM, ActionType.RPM: ActionType.RPM.code, ActionType.PID: ACT_DIRECT_RPM}[act]
        self.core = engine.SimCore(drone_model=drone_model, num_envs=N, drones_per_env=1, physics=physics, pyb_freq=pyb_freq,
                                   ctrl_freq=pyb_freq, act_code=act_code, task=engine.TASK_NONE, initial_xyzs=xyz,
                                   initial_rpys=rpy, auto_reset=False, track_rpm=True, device=device, pyb_like=pyb_like)
        self.device = dev = self.core.device
        self.flags = self.core.physics_flags
        self.ctrl = DSLPIDControlBatch(N, drone_model, device=dev) if act == ActionType.PID else None
        # ---- downwash grid ------------------------------------------------------------------------
        self.cell = float(cell)
        lo = xyz[:, 0, :2].min(axis=0) - 2 * self.cell if world_min is None else np.asarray(world_min, dtype=np.float64)
        hi = xyz[:, 0, :2].max(axis=0) + 2 * self.cell if world_max is None else np.asarray(world_max, dtype=np.float64)
        self.x0, self.y0 = float(lo[0]), float(lo[1])
        self.nx = max(3, int(np.ceil((hi[0] - lo[0]) / self.cell)))
        self.ny = max(3, int(np.ceil((hi[1] - lo[1]) / self.cell)))
        while self.nx * self.ny > 65536:          # coarser cells keep the search exact (cell >= 10 m), only less selective
            self.cell *= 2
            self.nx, self.ny = max(3, int(np.ceil((hi[0] - lo[0]) / self.cell))), max(3, int(np.ceil((hi[1] - lo[1]) / self.cell)))
        cells = self.nx * self.ny
        # optional height bins inside every cell (sort key = cell * nz + bin): a group of 64 drones sweeps the candidates from
        # its lowest bin upwards.  Off by default (nz = 1): it pays only when the 64 drones of a group share a height band --
        # with twelve layers mixed in every cell it prunes nothing and the 16x larger key space costs 20 us per step (measured)
        self.zbin = float(zbin)
        self.z0 = float(xyz[:, 0, 2].min() - self.zbin)
        self.nz = int(max(1, min(nz, 65536 // cells)))
        i32 = dict(dtype=torch.int32, device=dev)
        self._count, self._start = torch.zeros(2 * (cells * self.nz + 1), **i32), torch.zeros(cells * self.nz + 1, **i32)
        self._order = torch.zeros(N, **i32)          # sorted slot -> drone, filled by every call ...
        self._visit = torch.zeros(N, **i32)          # ... and the previous call's, which the next sort visits the drones in
        self._have_visit = False
        self._sorted = torch.zeros((N, 4), dtype=torch.float32, device=dev)
        self.dw_force = torch.zeros(self.core.ld, dtype=torch.float32, device=dev)
        if self.flags & PHYS_DW:
            self.core._state.dw_force = self.dw_force.data_ptr()
        self.step_counter = 0
        self._dw_version = -1                        # core.state_version the forces in dw_force were computed for

    # ------------------------------------------------------------------------------------------
    def downwash(self, vectors: torch.Tensor = None) -> torch.Tensor:
        """Body-z downwash force of every drone for the current positions (`gpd_downwash_global`) -> [N] view.
        `vectors`: an (N, 20) tensor that the sort's first pass fills with the state vectors on the way (one launch less
        than `state_vectors()` after it)."""
        c = self.core
        if vectors is not None and (vectors.device != self.device or vectors.dtype != torch.float32 or not vectors.is_contiguous()
                                    or tuple(vectors.shape) != (self.NUM_DRONES, 20)):
            raise ValueError(f"vectors must be a contiguous float32 ({self.NUM_DRONES}, 20) tensor on {self.device}")
        with torch.cuda.device(self.device):
            self._order, self._visit = self._visit, self._order      # ping-pong: last call's order is this call's visit order
            rc = c.lib.gpd_downwash_global(ctypes.byref(c._params), _ptr(c.kin), c.ld, self.NUM_DRONES, self.cell, self.x0,
                                           self.y0, self.nx, self.ny, self.z0, self.zbin, self.nz,
                                           _ptr(self._visit) if self._have_visit else None,
                                           _ptr(self._count), _ptr(self._start), _ptr(self._order), _ptr(self._sorted),
                                           _ptr(self.dw_force), ctypes.byref(c._state) if vectors is not None else None,
                                           _ptr(c.obs12) if vectors is not None else None, _ptr(vectors), c._stream())
        _native.check(rc, "gpd_downwash_global")
        self._have_visit = True
        self._dw_version = c.state_version                           # (the forces belong to this state)
        return self.dw_force[:self.NUM_DRONES]

    def reset(self, seed=None, options=None):
        self.core.reset()
        if self.ctrl is not None:
            self.ctrl.reset()
        self.step_counter = 0
        if not (self.flags & PHYS_DW):
            return self.state_vectors(), {"answer": 42}
        vectors = torch.empty((self.NUM_DRONES, 20), dtype=torch.float32, device=self.device)
        self.downwash(vectors)                        # the forces of the first sub-step (see step())
        return vectors, {"answer": 42}

    def _kernel_action(self, action) -> torch.Tensor:
        """What the step kernel is fed: the raw action itself (the kernel maps it to RPMs), or -- waypoint actions -- the
        RPMs of the N embedded DSLPID controllers (`gpd_pid`, one launch)."""
        N = self.NUM_DRONES
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        if self.ctrl is None:
            return a.reshape(N, 4)
        k = self.core.kin[:, :N]
        rpm, _, _ = self.ctrl.computeControl(self.CTRL_TIMESTEP, k[0:3].t(), k[3:7].t(), k[7:10].t(), None, a.reshape(N, 3))
        return rpm

    def step(self, action):
        """One control step = PYB_STEPS_PER_CTRL × { downwash of the snapshot, one physics sub-step }.

        The forces a sub-step uses are computed right AFTER the sub-step before it, on the snapshot it left (the same
        positions: `envs/BaseAviary.py:346-347, 785-811`), so that the pass that bins the drones also writes the state vectors
        this method returns -- five dependent launches per step instead of six.  After a reset, a `set_state` or any other
        change of the state behind this class's back (call `invalidate()` then), the first sub-step computes its own."""
        rpm = self._kernel_action(action).contiguous()
        if not (self.flags & PHYS_DW):
            for _ in range(self.PYB_STEPS_PER_CTRL):
                self.core.step(rpm)
            self.step_counter += self.PYB_STEPS_PER_CTRL
            return self.state_vectors(), -1, False, False, {"answer": 42}
        vectors = torch.empty((self.NUM_DRONES, 20), dtype=torch.float32, device=self.device)
        for s in range(self.PYB_STEPS_PER_CTRL):
            if self._dw_version != self.core.state_version:
                self.downwash()
            self.core.step(rpm)
            self.downwash(vectors if s == self.PYB_STEPS_PER_CTRL - 1 else None)
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return vectors, -1, False, False, {"answer": 42}

    def invalidate(self):
        """Tell the aviary that the state was changed without going through `reset()` / `core.set_state()` (e.g. by writing
        into `core.kin`): the next step recomputes the downwash forces first."""
        self._dw_version = -1

    def state_vectors(self) -> torch.Tensor:
        """(N, 20) `_getDroneStateVector` rows (envs/BaseAviary.py:559-561)."""
        return self.core.state_vectors()

    def close(self):
        pass
