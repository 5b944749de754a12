"""`SwarmAviary`: ONE aviary of N drones, any N — with the pairwise downwash of the whole swarm — on one GPU or sharded
across several.

The reference simulates one world per aviary and couples its drones only through `_downwash`
(`envs/BaseAviary.py:785-811`): an O(N²) Python loop over every pair with `dz > 0` and `dxy < 10 m`, once per
physics sub-step, on the positions all drones had at the start of the sub-step (`:346-347`).  The fused step
kernel covers aviaries of up to 256 drones (one workgroup, positions exchanged through LDS).  This class is the
large-world counterpart, built on the `GpdSwarm` entries of the C-ABI (`include/gpd.h`):

* a physics sub-step is TWO launches: `gpd_swarm_step` (the step kernel's arithmetic for this rank's drones; it also writes
  each drone's new position into the packed `pos4` array, tracks the largest lateral displacement since the drones were last
  binned, and — on the last sub-step — the `(n, 20)` state vectors) and `gpd_swarm_forces` (downwash of the NEXT sub-step on
  the snapshot this one left);
* the counting sort by grid cell (`gpd_swarm_bin`) runs every `rebin_every` sub-steps only; in between the force kernel
  searches the stale cell order with current positions and a radius that grows with the tracked displacement (measured against
  the swarm's common drift: a swarm in transit counts as standing still) — every pair the reference would sum is evaluated
  whatever the drones do, in order-independent 64-bit fixed point;
* `world_size` ranks share one world: rank r owns a block of drones (by default a stripe of the world: the drones are dealt in
  the cell order of their initial positions; `GLOBAL_IDS`), and after every sub-step the ranks all-gather
  their 16 bytes per drone (`exchange`: RCCL through the C-ABI's `gpd_allgather_obs`, or `torch.distributed`) — the one
  collective of a sub-step.  Every rank bins all positions and evaluates the forces of its own drones; the sums are integers,
  so a world stepped by 1, 2 or 8 ranks follows the same trajectory bit for bit.

Interface: `CtrlAviary`-like.  `step(action)` takes raw RPMs `(n_own, 4)` clipped to `[0, MAX_RPM]`
(`envs/CtrlAviary.py:140`) — or, with `act=ActionType.PID`, waypoints `(n_own, 3)` tracked by `DSLPIDControl`s
(`examples/downwash.py:93-113`) — and returns the `(n_own, 20)` state vectors of this rank's drones.  Everything stays on the
GPU.
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


def swarm_partition(num_drones: int, world_size: int):
    """How one world of `num_drones` is dealt to `world_size` ranks: `(per, slab, counts)` -- balanced contiguous blocks (sizes
    differ by at most one: the first `N mod W` ranks own `per = ceil(N / W)` drones, the others one fewer; rank r's first drone is
    `swarm_first_drone(N, W, r)`), which sit in the rows `r*slab ..` of the packed position array; `slab = per + meta` rows per
    rank, the same on every rank, the last `meta = ceil(per / 256)` being the rank's meta rows (one per workgroup of the step
    kernel; `include/gpd.h`, `GpdSwarm`).  Any N >= W works."""
    if num_drones < world_size:
        raise ValueError(f"{num_drones} drones cannot be shared by {world_size} ranks: every rank needs at least one drone")
    base, rem = divmod(num_drones, world_size)
    per = base + (1 if rem else 0)
    counts = [base + (1 if r < rem else 0) for r in range(world_size)]
    return per, per + -(-per // 256), counts


def swarm_first_drone(num_drones: int, world_size: int, rank: int) -> int:
    """Index (in the order the drones are dealt in) of the first drone of `rank` under `swarm_partition`."""
    base, rem = divmod(num_drones, world_size)
    return rank * base + min(rank, rem)


def swarm_spatial_order(xyz, cell: float) -> np.ndarray:
    """The order `partition="spatial"` deals the drones to the ranks in: row-major over the cells (of size `cell`, anchored at the
    lowest x / y) of their INITIAL positions, the caller's index breaking ties -- a rank's contiguous block of it is a stripe of
    the world.  Returns a permutation of `range(len(xyz))`."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    cxy = np.floor((xyz[:, :2] - xyz[:, :2].min(axis=0)) / cell).astype(np.int64)
    return np.lexsort((np.arange(len(xyz)), cxy[:, 0], cxy[:, 1]))


import contextlib
_NO_GUARD = contextlib.nullcontext()


class TorchSlabExchange:
    """The all-gather of the ranks' position slabs through `torch.distributed` (RCCL with the "nccl" backend; with gloo the
    slabs are staged through host memory: CPU tests and the single-device test hook)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.stage = dist.get_backend(group) == "gloo"

    def __call__(self, pos4: torch.Tensor, rank: int, slab: int):
        mine = pos4[rank * slab:(rank + 1) * slab].clone()       # (the send buffer must not alias the output for torch)
        if self.stage:
            host = torch.empty(pos4.shape, dtype=pos4.dtype)
            self.dist.all_gather_into_tensor(host.view(-1), mine.cpu().view(-1).contiguous(), group=self.group)
            pos4.copy_(host)
        else:
            self.dist.all_gather_into_tensor(pos4.view(-1), mine.view(-1), group=self.group)


class NativeSlabExchange:
    """The same all-gather through the C-ABI (`gpd_allgather_obs`: `ncclAllGather`, IN PLACE -- the send buffer is the rank's
    own slab inside the receive buffer) on the process's one communicator (`dist.NativeComm`)."""

    def __init__(self, comm=None, device=None):
        from ..dist import NativeComm
        self.nc = comm if comm is not None else NativeComm.shared(device=device)

    def __call__(self, pos4: torch.Tensor, rank: int, slab: int):
        with torch.cuda.device(pos4.device):
            rc = self.nc.lib.gpd_allgather_obs(self.nc.comm, ctypes.c_void_p(pos4.data_ptr() + rank * slab * 16), _ptr(pos4), slab * 4,
                                               ctypes.c_void_p(torch.cuda.current_stream(pos4.device).cuda_stream))
        _native.check(rc, "gpd_allgather_obs")


# ---- halo exchange: a rank's neighbours' border drones instead of every position -------------------------------------------
#: lateral cut-off of the reference's downwash model (envs/BaseAviary.py:801: `delta_xy < 10`)
DW_LATERAL_CUTOFF = 10.0


class HaloPlan:
    """Which of a rank's drones every other rank needs, and where the blocks it receives go.

    The ranks hold stripes of the world (`partition="spatial"`: blocks of the row-major cell order, i.e. bands in y), and a
    drone feels only drones within 10 m laterally (envs/BaseAviary.py:798-811).  Rank d therefore needs, of rank s's drones,
    those whose y lies within `reach = 10 m + margin` of the y-interval its own drones occupy -- for stripes: blocks from the two
    neighbouring ranks, nothing from the others.  The plan is made whenever the drones are re-binned (and after a reset) from
    CURRENT positions, in three phases separated by two tiny all-gathers (3 floats, then W counts per rank):
        phase1  this rank's y-interval (+ whether the PREVIOUS plan's margin held)            -> [lo, hi, violated]
        phase2  from everybody's intervals: the rows of this rank every other rank needs      -> counts [W]
        phase3  from everybody's counts: send offsets, receive places; the rest of every other rank's rows is marked empty
    Between two plans the SAME rows travel every sub-step (`gather()` packs them; `sends` / `recvs` say where the blocks go), so
    row j of rank s's region is the same drone until the next plan -- what the stale-cell-order search of the force kernel
    needs.  A halo row has no identity beyond that: received blocks are stored compactly at the start of the sender's region
    of `pos4`, and the force sums are integers, so the forces are the all-gather path's bit for bit.

    Exactness: a pair within 10 m NOW was within 10 m + d_i + d_j in y at plan time (d = |y now - y at plan time|), so the plan
    covers every pair while every drone has moved less than margin / 2 in y since the plan.  `phase1` of the next plan checks
    that for the interval that ends (every rank learns of a violation anywhere and raises): results are exact, or the run
    stops and says which knob to turn (`halo_margin`, `rebin_every`).  The plan also sends every rank's meta rows (the per-
    workgroup displacement maxima and sums the force kernel derives its search radius and the common drift from) to every
    other rank: meta_rows x 16 bytes per pair."""

    def __init__(self, margin: float):
        if not margin > 0:
            raise ValueError("halo margin must be positive")
        self.margin = float(margin)
        self.reach = DW_LATERAL_CUTOFF + self.margin
        self.ready = False
        self.y_plan = None

    def forget(self):
        """Drop the plan AND the positions it was made from: the next exchange plans from scratch and checks nothing.  Called
        whenever positions change discontinuously (reset(), set_state(), invalidate() -- everything that re-packs): a teleport
        is not a margin violation, and the interval that ends with it has no pairs left to miss."""
        self.ready = False
        self.y_plan = None

    def _own(self, env):
        r0 = env.RANK * env.slab
        return env.pos4[r0:r0 + env.NUM_DRONES]

    def phase1(self, env) -> torch.Tensor:
        own = self._own(env)
        y = own[:, 1]
        fin = torch.isfinite(own[:, :3]).all(dim=1)
        lo = torch.where(fin, y, torch.full_like(y, 3.0e38)).min()
        hi = torch.where(fin, y, torch.full_like(y, -3.0e38)).max()
        if self.y_plan is not None:
            moved = ((y - self.y_plan).abs() > 0.5 * self.margin) & fin & self.fin_plan
            viol = moved.any().to(torch.float32)
        else:
            viol = torch.zeros((), dtype=torch.float32, device=y.device)
        self._y, self._fin = y, fin
        return torch.stack([lo, hi, viol])

    def phase2(self, env, bounds: torch.Tensor) -> torch.Tensor:
        """`bounds`: [W, 3] of phase1, every rank's row -> float counts [W] (rows of mine rank d needs; 0 for myself)"""
        if bool((bounds[:, 2] > 0).any().item()):
            who = torch.nonzero(bounds[:, 2] > 0).flatten().tolist()
            self.forget()           # (reported once: the next call plans afresh from the current positions)
            raise RuntimeError(f"halo exchange: drones of rank(s) {who} moved more than margin / 2 = {0.5 * self.margin:g} m in y since the last "
                               f"plan -- the halo may have missed pairs; results since the last binning are not guaranteed.  Use a larger "
                               f"halo_margin, a smaller rebin_every, or exchange='allgather'.")
        y, fin = self._y, self._fin
        lo, hi = bounds[:, 0:1] - self.reach, bounds[:, 1:2] + self.reach
        self._masks = (y.unsqueeze(0) >= lo) & (y.unsqueeze(0) <= hi) & fin.unsqueeze(0)
        self._masks[env.RANK] = False
        return self._masks.sum(dim=1).to(torch.float32)

    def phase3(self, env, counts: torch.Tensor):
        """`counts`: [W, W] of phase2 (row s = what rank s sends to each rank).  Builds the plan; marks the rows of the other
        ranks that will not be received as empty."""
        W, r, slab, meta = env.WORLD_SIZE, env.RANK, env.slab, env.slab - env.per
        c = counts.to(torch.int64).cpu()
        self.send_cnt, self.recv_cnt = c[r].tolist(), c[:, r].tolist()
        own = self._own(env)
        idx = [torch.nonzero(self._masks[p]).flatten() for p in range(W) if self.send_cnt[p] > 0]
        self.idx = torch.cat(idx) if idx else torch.zeros(0, dtype=torch.int64, device=own.device)
        assert int(self.idx.numel()) == sum(self.send_cnt)
        self.sendbuf = torch.empty((max(1, int(self.idx.numel())), 4), dtype=torch.float32, device=own.device)
        v = env.pos4.view(W, slab, 4)
        self.sends, self.recvs, off = [], [], 0
        meta_own = v[r, slab - meta:]
        for p in range(W):
            if p == r:
                continue
            if self.send_cnt[p]:
                self.sends.append((p, self.sendbuf[off:off + self.send_cnt[p]]))
                off += self.send_cnt[p]
            self.sends.append((p, meta_own))
            if self.recv_cnt[p]:
                self.recvs.append((p, v[p, :self.recv_cnt[p]]))
            self.recvs.append((p, v[p, slab - meta:]))
            v[p, self.recv_cnt[p]:slab - meta] = float("nan")          # rows without a drone (a non-finite x: include/gpd.h)
        self.y_plan, self.fin_plan = self._y.clone(), self._fin.clone()
        self._masks = None
        self.bytes_sent = (int(self.idx.numel()) + (W - 1) * meta) * 16
        self.bytes_allgather = (W - 1) * slab * 16                     # what the in-place all-gather makes this rank receive
        self.ready = True

    def gather(self, env):
        """pack the rows of the plan into the send buffer (one gather kernel per sub-step)"""
        if self.idx.numel():
            torch.index_select(self._own(env), 0, self.idx, out=self.sendbuf[:self.idx.numel()])


class _HaloExchange:
    """Plan (when a binning is due and the stream is not being captured) + move the blocks; subclasses are the transports."""
    halo = True

    def __init__(self, margin: float = 2.0):
        self.plan = HaloPlan(margin)
        self.plans_made = 0

    # transport hooks ------------------------------------------------------------------------------
    def _allgather_small(self, t: torch.Tensor) -> torch.Tensor:        # [k] on the device -> [W, k]
        raise NotImplementedError

    def _move(self, env):
        raise NotImplementedError

    def _plan_changed(self, env):
        pass

    def exchange(self, env, replan: bool):
        P = self.plan
        capturing = env.pos4.is_cuda and torch.cuda.is_current_stream_capturing()
        if (replan or not P.ready) and not capturing:
            bounds = self._allgather_small(P.phase1(env))
            counts = self._allgather_small(P.phase2(env, bounds))
            P.phase3(env, counts)
            self.plans_made += 1
            self._plan_changed(env)
        elif not P.ready:
            raise RuntimeError("halo exchange: no plan yet -- run one eager step (or reset()) before capturing a hipGraph")
        P.gather(env)
        self._move(env)

    def check(self, env):
        """Collective: raises on every rank if any drone has outrun the margin since the last plan (what the next plan would
        find): call it after replaying a captured graph, whose plan is the one made before the capture."""
        bounds = self._allgather_small(self.plan.phase1(env))
        if bool((bounds[:, 2] > 0).any().item()):
            self.plan.phase2(env, bounds)

    @property
    def bytes_per_substep(self):
        return self.plan.bytes_sent if self.plan.ready else None


class TorchHaloExchange(_HaloExchange):
    """The halo blocks as `torch.distributed` point-to-point operations in one batch (`batch_isend_irecv`: RCCL's grouped
    ncclSend / ncclRecv with the "nccl" backend; with gloo -- CPU tests, the single-device test hook -- device blocks are staged
    through host memory)."""

    def __init__(self, margin: float = 2.0, group=None):
        super().__init__(margin)
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.stage = dist.get_backend(group) == "gloo"

    def _allgather_small(self, t):
        W = self.dist.get_world_size(self.group)
        if self.stage:
            out = torch.empty((W, t.numel()), dtype=t.dtype)
            self.dist.all_gather_into_tensor(out, t.detach().cpu().reshape(1, -1).contiguous(), group=self.group)
            return out.to(t.device)
        out = torch.empty((W, t.numel()), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.reshape(1, -1).contiguous(), group=self.group)
        return out

    def _move(self, env):
        P, d = self.plan, self.dist
        ops, landing = [], []
        for p, blk in P.sends:
            ops.append(d.P2POp(d.isend, blk.cpu().contiguous() if self.stage else blk, p, group=self.group))
        for p, blk in P.recvs:
            buf = torch.empty(blk.shape, dtype=blk.dtype) if self.stage else blk
            landing.append((blk, buf))
            ops.append(d.P2POp(d.irecv, buf, p, group=self.group))
        for w in d.batch_isend_irecv(ops):
            w.wait()
        if self.stage:
            for blk, buf in landing:
                blk.copy_(buf)


class NativeHaloExchange(_HaloExchange):
    """The halo blocks through the C-ABI: `gpd_p2p_group` -- ncclSend / ncclRecv of RCCL in one group on the process's one
    communicator (`dist.NativeComm`), asynchronous on the current stream, capturable in a hipGraph with the sub-step that
    produced the positions.  The two tiny all-gathers of a plan go through `gpd_allgather_obs` on the same communicator."""

    def __init__(self, margin: float = 2.0, comm=None, device=None):
        super().__init__(margin)
        from ..dist import NativeComm
        self.nc = comm if comm is not None else NativeComm.shared(device=device)

    def _allgather_small(self, t):
        out = torch.empty((self.nc.world, t.numel()), dtype=torch.float32, device=t.device)
        src = t.to(torch.float32).contiguous()
        with torch.cuda.device(t.device):
            rc = self.nc.lib.gpd_allgather_obs(self.nc.comm, _ptr(src), _ptr(out), src.numel(),
                                               ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream))
        _native.check(rc, "gpd_allgather_obs (halo plan)")
        return out

    def _plan_changed(self, env):
        P = self.plan
        mk = lambda ops: (_native.GpdP2P * max(1, len(ops)))(*[_native.GpdP2P(peer=p, ptr=b.data_ptr(), count=b.numel()) for p, b in ops])  # noqa: E731
        self._S, self._R = mk(P.sends), mk(P.recvs)

    def _move(self, env):
        P = self.plan
        with torch.cuda.device(env.device):
            rc = self.nc.lib.gpd_p2p_group(self.nc.comm, self._S, len(P.sends), self._R, len(P.recvs),
                                           ctypes.c_void_p(torch.cuda.current_stream(env.device).cuda_stream))
        _native.check(rc, "gpd_p2p_group")


class SwarmAviary:
    """One world, `num_drones` drones, explicit integrator + the selected force models over the whole swarm; this object holds
    the drones of rank `rank` of `world_size`."""

    def __init__(self, num_drones: int, drone_model: DroneModel = DroneModel.CF2X, initial_xyzs=None, initial_rpys=None,
                 physics: Physics = Physics.PYB_DW, pyb_freq: int = 240, ctrl_freq: int = 240, act="raw_rpm",
                 world_min=None, world_max=None, cell: float = 10.5, zbin: float = 1.0, nz: int = 1, device=None,
                 pyb_like: bool = None, world_size: int = 1, rank: int = 0, exchange=None, rebin_every: int = None,
                 wake_lists: bool = True, list_cap: int = 48, partition: str = "spatial", expected_speed: float = None,
                 adaptive_lists: bool = True):
        if pyb_freq % ctrl_freq != 0:
            raise ValueError("[ERROR] in SwarmAviary.__init__(), pyb_freq is not divisible by ctrl_freq.")
        if act not in ("raw_rpm", ActionType.RPM, ActionType.PID):
            raise ValueError("SwarmAviary supports act = 'raw_rpm', ActionType.RPM or ActionType.PID")
        if not 0 <= rank < world_size:
            raise ValueError("need 0 <= rank < world_size")
        if world_size > 1 and exchange is None:
            raise ValueError("a world shared by several ranks needs an `exchange` (TorchSlabExchange / NativeSlabExchange)")
        self.TOTAL_DRONES = N = int(num_drones)
        self.WORLD_SIZE, self.RANK, self.exchange = int(world_size), int(rank), exchange
        self.per, self.slab, counts = swarm_partition(N, self.WORLD_SIZE)
        self.FIRST_DRONE = swarm_first_drone(N, self.WORLD_SIZE, self.RANK)       # global index of this rank's first drone
        self.NUM_DRONES = n = counts[self.RANK]                  # drones of THIS rank
        self.n_rows = self.slab * self.WORLD_SIZE
        self.DRONE_MODEL, self.PHYSICS, self.ACT_TYPE = drone_model, physics, act
        warn_if_pyb(physics)
        self.PYB_FREQ, self.CTRL_FREQ = pyb_freq, ctrl_freq
        self.PYB_STEPS_PER_CTRL = pyb_freq // ctrl_freq
        self.CTRL_TIMESTEP, self.PYB_TIMESTEP = 1. / ctrl_freq, 1. / pyb_freq
        P = DroneParams(drone_model)
        self.HOVER_RPM, self.MAX_RPM = P.HOVER_RPM, P.MAX_RPM
        if initial_xyzs is None:
            initial_xyzs = P.default_init_xyzs(N)
        xyz_all = np.asarray(initial_xyzs, dtype=np.float64).reshape(N, 3)            # the WHOLE world (every rank passes the same)
        rpy_all = np.zeros((N, 3)) if initial_rpys is None else np.asarray(initial_rpys, dtype=np.float64).reshape(N, 3)
        # Which drones a rank owns.  Forces and trajectories do not depend on it (integer sums: bit for bit the same), the WORK a
        # rank does does: its force launch runs the groups of 64 sorted drones that hold one of ITS drones.  "spatial" (default):
        # the drones are dealt in the row-major cell order of their INITIAL positions, so a rank's block is a stripe of the world
        # and ~1/W of the groups (a swarm that mixes thoroughly in flight degrades towards "index"); "index": block r = drones
        # r*per ... in the caller's numbering -- with a spatially random numbering every group holds drones of every rank and
        # every rank runs every group (1/W of its lanes busy).  `GLOBAL_IDS[i]` is the caller's index of this rank's drone i.
        if partition not in ("spatial", "index"):
            raise ValueError("partition must be 'spatial' or 'index'")
        self.cell = float(cell)
        if expected_speed is not None:
            # a skin wide enough for drones of that speed to stay inside list_delta = 0.49 (cell - 10 m) between two binnings:
            # nothing depends on it for correctness, it only keeps the cheap replay launches (16 us) from falling back to
            # sweeps (34 us) -- at the price of ~7 % more candidates per 0.5 m of skin
            every = int(rebin_every) if rebin_every is not None else 16
            self.cell = max(self.cell, 10.0 + float(expected_speed) * every / pyb_freq / 0.485)
        if not self.cell >= 10.0:
            raise ValueError("cell must be >= 10 m (the downwash model's lateral cut-off)")
        order_all = swarm_spatial_order(xyz_all, self.cell) if partition == "spatial" and self.WORLD_SIZE > 1 else np.arange(N)
        self._deal_order = order_all                             # position in the deal -> the caller's drone index
        self.GLOBAL_IDS = np.ascontiguousarray(order_all[self.FIRST_DRONE:self.FIRST_DRONE + n])
        xyz, rpy = xyz_all[self.GLOBAL_IDS].reshape(n, 1, 3), rpy_all[self.GLOBAL_IDS].reshape(n, 1, 3)
        self.INIT_XYZS, self.INIT_RPYS = xyz[:, 0], rpy[:, 0]
        # the kernel runs n single-drone lanes, one physics sub-step per launch.  The action -> RPM mapping is the kernel's own
        # (GPD_ACT_RAW_RPM: clip to [0, MAX_RPM], envs/CtrlAviary.py:140; GPD_ACT_RPM: HOVER_RPM (1 + 0.05 a),
        # envs/BaseRLAviary.py:191-192); waypoint actions go through the batched DSLPID kernel and arrive as RPMs.
        act_code = {"raw_rpm": ACT_RAW_RPM, ActionType.RPM: ActionType.RPM.code, ActionType.PID: ACT_DIRECT_RPM}[act]
        self.core = engine.SimCore(drone_model=drone_model, num_envs=n, drones_per_env=1, physics=physics, pyb_freq=pyb_freq,
                                   ctrl_freq=pyb_freq, act_code=act_code, task=engine.TASK_NONE, initial_xyzs=xyz,
                                   initial_rpys=rpy, auto_reset=False, track_rpm=True, device=device, pyb_like=pyb_like)
        self.device = dev = self.core.device
        self.flags = self.core.physics_flags
        self.ctrl = DSLPIDControlBatch(n, drone_model, device=dev) if act == ActionType.PID else None
        # ---- downwash grid (the same on every rank: laid over the whole world) --------------------
        lo = xyz_all[:, :2].min(axis=0) - 2 * self.cell if world_min is None else np.asarray(world_min, dtype=np.float64)
        hi = xyz_all[:, :2].max(axis=0) + 2 * self.cell if world_max is None else np.asarray(world_max, dtype=np.float64)
        self.x0, self.y0 = float(lo[0]), float(lo[1])
        self.nx = max(3, int(np.ceil((hi[0] - lo[0]) / self.cell)))
        self.ny = max(3, int(np.ceil((hi[1] - lo[1]) / self.cell)))
        while self.nx * self.ny > 65536:          # coarser cells keep the search exact (cell >= 10 m), only less selective
            self.cell *= 2
            self.nx, self.ny = max(3, int(np.ceil((hi[0] - lo[0]) / self.cell))), max(3, int(np.ceil((hi[1] - lo[1]) / self.cell)))
        cells = self.nx * self.ny
        # optional height bins inside every cell (sort key = cell * nz + bin): ordering only, any value gives the same forces
        self.zbin = float(zbin)
        self.z0 = float(xyz_all[:, 2].min() - self.zbin)
        self.nz = int(max(1, min(nz, 65536 // cells)))
        # How often the drones are re-binned.  Between two binnings the search radius is R = ceil((10 m + 2 dmax) / cell) cells
        # (dmax: the largest displacement since the binning) and the wake lists hold while dmax <= list_delta = 0.49 skin, so the
        # skin cell - 10 m buys cheap sub-steps: with no skin every sub-step re-bins (R = 1 always, no lists: the round-2
        # behaviour); otherwise every 16th by default -- with the default 0.5 m skin a drone slower than 3.7 m/s stays inside
        # list_delta for 1/15 s; faster ones make the launches sweep (and, beyond half the skin, widen the search) until the
        # next binning, they never break anything.  Measured at 65 536 drones (profiles/r03_swarm_*): 51 us per sub-step with a
        # binning every sub-step, 34.7 / 30.1 / 27.5 with one every 4th / 8th / 16th (25.7 with the replay's list reads four batches ahead).
        self.rebin_every = int(rebin_every) if rebin_every is not None else (1 if self.cell <= 10.0 else 16)
        if self.rebin_every < 1:
            raise ValueError("rebin_every must be >= 1")
        i32 = dict(dtype=torch.int32, device=dev)
        keys = cells * self.nz
        self._count, self._start = torch.zeros(2 * (keys + 1), **i32), torch.zeros(keys + 1, **i32)
        self._order = torch.zeros(self.n_rows, **i32)          # sorted slot -> row of the latest binning (what the force kernel reads)
        # ... and two copies of it in turn: the one the latest binning wrote is what the next one visits the rows in
        # (both start as the identity: whatever runs or is merely CAPTURED in between -- a captured binning has not written its
        # copy yet when an eager one reads it -- a visit buffer always holds a permutation of the rows)
        self._visit_in, self._visit_out = torch.arange(self.n_rows, **i32), torch.arange(self.n_rows, **i32)
        self._slot_key = torch.zeros(self.n_rows, **i32)
        self.pos4 = torch.full((self.n_rows, 4), float("nan"), dtype=torch.float32, device=dev)
        self._bin_pos = torch.full((self.n_rows, 4), float("nan"), dtype=torch.float32, device=dev)
        # a rank that holds the whole world keeps the positions by sorted slot as well (no exchange in row order to serve)
        self._slot_of = torch.full((self.n_rows,), -1, **i32) if self.WORLD_SIZE == 1 else None
        self._pos_sorted = torch.full((self.n_rows, 4), float("nan"), dtype=torch.float32, device=dev) if self.WORLD_SIZE == 1 else None
        self.dw_force = torch.zeros(self.core.ld, dtype=torch.float32, device=dev)
        if self.flags & PHYS_DW:
            self.core._state.dw_force = self.dw_force.data_ptr()
        # Wake lists: the pairs the force launch after a binning evaluates (with a margin of `list_delta` per drone), replayed by
        # the launches until the next binning instead of sweeping all candidates.  Pointless without a skin (every sub-step bins).
        groups = (self.n_rows + 63) // 64
        self.list_delta = 0.49 * (self.cell - 10.0)
        self.wake_lists = bool(wake_lists) and self.rebin_every > 1 and self.list_delta > 0
        # adaptive_lists: `list_delta` is the most the skin allows; every binning then picks the margin of ITS lists from how
        # far the drones moved (relative to the swarm's drift) since the previous one -- three times that, at least 1 cm.  A
        # hovering swarm lists 34 pairs per drone instead of 47 at cell = 10.5 m; a swarm that outruns the guess sweeps (exactly)
        # until the next binning.  False: always the full margin.
        self.adaptive_lists = bool(adaptive_lists)
        if self.wake_lists and not 4 <= int(list_cap) <= 65535:
            raise ValueError("list_cap must be in 4..65535 (a replay launch requests a wave's first four batches before it knows how many there are)")
        u16 = dict(dtype=torch.int16, device=dev)
        # (32-bit entries: 6 bits drone of the group + 26 bits the candidate's slot / row -- include/gpd.h)
        self._pair_list = torch.zeros((groups, 4, int(list_cap) * 64), dtype=torch.int32, device=dev) if self.wake_lists else None
        self._pair_nb = torch.zeros((groups, 4, 16), **u16) if self.wake_lists else None
        self._list_ok = torch.zeros(groups, **i32) if self.wake_lists else None
        self._drift = torch.zeros(4, dtype=torch.float32, device=dev)          # the swarm's common lateral drift since the binning
        self._sw = _native.GpdSwarm(n_rows=self.n_rows, slab=self.slab, world_size=self.WORLD_SIZE, rank=self.RANK, own_count=n,
                                    nx=self.nx, ny=self.ny, nz=self.nz, cell=self.cell, x0=self.x0, y0=self.y0, z0=self.z0,
                                    zbin=self.zbin, meta_rows=self.slab - self.per, pos4=self.pos4.data_ptr(), bin_pos=self._bin_pos.data_ptr(),
                                    cell_count=self._count.data_ptr(), cell_start=self._start.data_ptr(),
                                    order=self._order.data_ptr(), visit=None, visit_out=self._visit_out.data_ptr(),
                                    slot_key=self._slot_key.data_ptr(),
                                    dw_force=self.dw_force.data_ptr(),
                                    slot_of=self._slot_of.data_ptr() if self._slot_of is not None else None,
                                    pos_sorted=self._pos_sorted.data_ptr() if self._pos_sorted is not None else None,
                                    pair_list=self._pair_list.data_ptr() if self.wake_lists else None,
                                    pair_nb=self._pair_nb.data_ptr() if self.wake_lists else None,
                                    list_ok=self._list_ok.data_ptr() if self.wake_lists else None,
                                    list_cap=int(list_cap), list_delta=self.list_delta, drift=self._drift.data_ptr(), total_drones=N,
                                    list_adapt=int(bool(adaptive_lists)))
        self.step_counter = 0
        self._since_bin = 0                          # sub-steps since the last binning
        self._dw_version = -1                        # core.state_version the forces in dw_force were computed for

    # ---- the pieces of a sub-step (LocalSwarmGroup drives several ranks of one process through them) --------------------
    def _pack(self, vectors=None):
        """pos4 rows of this rank from the state block (after a reset / an outside change); the sort is stale afterwards"""
        c = self.core
        with torch.cuda.device(self.device):
            rc = c.lib.gpd_swarm_pack(ctypes.byref(c._state), ctypes.byref(self._sw), _ptr(c.obs12), _ptr(vectors), c._stream())
        _native.check(rc, "gpd_swarm_pack")
        self._since_bin = self.rebin_every           # (forces a binning)
        if getattr(self.exchange, "halo", False):    # ... and a new halo plan that holds nothing against the jump
            self.exchange.plan.forget()

    def _refs(self):
        """ctypes references of the four structs and the address of the observation block: built once (an eager sub-step is three
        C calls; the host side of each is what an eager loop waits for)"""
        r = self.__dict__.get("_ref_cache")
        if r is None:
            c = self.core
            r = self._ref_cache = (ctypes.byref(c._params), ctypes.byref(c._state), ctypes.byref(c._cfg), ctypes.byref(self._sw), _ptr(c.obs12))
        return r

    def _guard(self):
        """the device guard, or nothing when this world's device is the current one already"""
        return _NO_GUARD if torch.cuda.current_device() == self.device.index else torch.cuda.device(self.device)

    def _substep(self, rpm, vectors=None):
        c = self.core
        c.state_version += 1
        r = self._refs()
        with self._guard():
            rc = c.lib.gpd_swarm_step(r[0], r[1], r[2], r[3], _ptr(rpm), r[4], _ptr(vectors), c._stream())
        if rc:
            _native.check(rc, "gpd_swarm_step")
        self._since_bin += 1

    def _exchange(self):
        if self.WORLD_SIZE > 1:
            if getattr(self.exchange, "halo", False):       # blocks from the neighbouring stripes (re-planned with every binning)
                self.exchange.exchange(self, replan=self._since_bin >= self.rebin_every)
            else:                                           # every position to every rank
                self.exchange(self.pos4, self.RANK, self.slab)

    def all_positions(self) -> torch.Tensor:
        """(TOTAL_DRONES, 3) positions of the WHOLE world in the caller's drone order, on every rank -- a collective through
        `torch.distributed` (diagnostics / checks: a rank of a halo-exchanging world holds only its own and its neighbours'
        border drones)."""
        own = torch.full((self.per, 3), float("nan"), dtype=torch.float32, device=self.device)
        own[:self.NUM_DRONES] = self.core.positions(self.NUM_DRONES)
        if self.WORLD_SIZE == 1:
            return own[:self.NUM_DRONES].clone()
        import torch.distributed as dist
        stage = dist.get_backend() == "gloo"
        out = torch.empty((self.WORLD_SIZE * self.per, 3), dtype=torch.float32, device=None if stage else self.device)
        dist.all_gather_into_tensor(out, own.cpu() if stage else own)
        out = out.to(self.device).view(self.WORLD_SIZE, self.per, 3)
        _, _, counts = swarm_partition(self.TOTAL_DRONES, self.WORLD_SIZE)
        rows = torch.cat([out[r, :counts[r]] for r in range(self.WORLD_SIZE)])          # the order the drones were dealt in
        res = torch.empty_like(rows)
        res[torch.as_tensor(self._deal_order, dtype=torch.long, device=self.device)] = rows
        return res

    def _forces(self):
        """(binning when one is due,) the downwash forces of this rank's drones for the positions in pos4"""
        c = self.core
        binned = False
        r = self._refs()
        with self._guard():
            stream = c._stream()
            if self._since_bin >= self.rebin_every:
                binned = True
                self._visit_in, self._visit_out = self._visit_out, self._visit_in      # the last binning's copy is this one's visit order
                self._sw.visit = self._visit_in.data_ptr()
                self._sw.visit_out = self._visit_out.data_ptr()
                _native.check(c.lib.gpd_swarm_bin(r[3], stream), "gpd_swarm_bin")
                self._since_bin = 0
            rc = c.lib.gpd_swarm_forces(r[0], r[3], int(binned), stream)
            if rc:
                _native.check(rc, "gpd_swarm_forces")
        self._dw_version = c.state_version           # (the forces belong to this state)

    def _check_vectors(self, vectors):
        if vectors is not None and (vectors.device != self.device or vectors.dtype != torch.float32 or not vectors.is_contiguous()
                                    or tuple(vectors.shape) != (self.NUM_DRONES, 20)):
            raise ValueError(f"vectors must be a contiguous float32 ({self.NUM_DRONES}, 20) tensor on {self.device}")

    # ------------------------------------------------------------------------------------------
    def downwash(self, vectors: torch.Tensor = None) -> torch.Tensor:
        """Body-z downwash force of this rank's drones for the current positions -> [n] view (re-packs, exchanges, re-bins:
        the from-scratch path a reset takes; collective when the world is shared).  `vectors`: an (n, 20) tensor that
        receives the state vectors on the way."""
        self._check_vectors(vectors)
        self._pack(vectors)
        self._exchange()
        self._forces()
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
        RPMs of the embedded DSLPID controllers (`gpd_pid`, one launch)."""
        n = self.NUM_DRONES
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        if self.ctrl is None:
            return a.reshape(n, 4)
        c = self.core
        rpm, _, _ = self.ctrl.computeControl(self.CTRL_TIMESTEP, c.positions(n), c.quaternions(n), c.velocities(n), None, a.reshape(n, 3))
        return rpm

    def step(self, action):
        """One control step = PYB_STEPS_PER_CTRL × { one physics sub-step, [exchange,] downwash of the snapshot it left }.

        The forces a sub-step uses are computed right AFTER the sub-step before it, on the snapshot it left (the same
        positions: `envs/BaseAviary.py:346-347, 785-811`).  After a reset, a `set_state` or any other change of the state behind
        this class's back (call `invalidate()` then -- on every rank of a shared world), the first sub-step computes its own."""
        rpm = self._kernel_action(action).contiguous()
        vectors = torch.empty((self.NUM_DRONES, 20), dtype=torch.float32, device=self.device)
        dw = bool(self.flags & PHYS_DW)
        for s in range(self.PYB_STEPS_PER_CTRL):
            if dw and self._dw_version != self.core.state_version:
                self.downwash()
            self._substep(rpm, vectors if s == self.PYB_STEPS_PER_CTRL - 1 else None)
            if dw:
                self._exchange()
                self._forces()
        self.step_counter += self.PYB_STEPS_PER_CTRL
        return vectors, -1, False, False, {"answer": 42}

    def invalidate(self):
        """Tell the aviary that the state was changed without going through `reset()` / `core.set_state()` (e.g. by writing
        through the plane views `core.kin_P / kin_Q / kin_V / kin_W`, `core.positions()` ...; `core.kin` itself is a copy and rejects
        writes): the next step re-packs, re-bins and recomputes the downwash forces first."""
        self._dw_version = -1

    def state_vectors(self) -> torch.Tensor:
        """(n, 20) `_getDroneStateVector` rows (envs/BaseAviary.py:559-561)."""
        return self.core.state_vectors()

    # ---- checkpoint / resume (absent upstream) ----------------------------------------------------------
    def get_state(self) -> dict:
        """Snapshot of this rank's drones: the core's complete state (`SimCore.get_state`) + the embedded controllers' members +
        the step counter.  The sort, the wake lists and the halo plan are NOT part of it: forces are exact functions of the
        positions (order-independent integer sums), so `set_state` simply makes the next step re-pack, re-bin and recompute
        them -- the trajectory that follows is bit for bit the one that followed the snapshot (every rank of a shared world
        restores its own snapshot)."""
        out = dict(self.core.get_state(), step_counter_py=self.step_counter)
        if self.ctrl is not None:
            out["ctrl"] = self.ctrl._state.clone()
        return out

    def set_state(self, state: dict):
        state = dict(state)
        self.step_counter = int(state.pop("step_counter_py", self.step_counter))
        ctrl = state.pop("ctrl", None)
        if ctrl is not None and self.ctrl is not None:
            self.ctrl._state.copy_(ctrl)
        self.core.set_state(**state)
        self.invalidate()

    def downwash_oneshot(self) -> torch.Tensor:
        """The forces through the C-ABI's one-call entry `gpd_downwash_global` (count + scan/scatter + force on the state block,
        no persistent sort): single-rank worlds only; the cross-check of the persistent path."""
        if self.WORLD_SIZE != 1:
            raise ValueError("gpd_downwash_global works on one rank's state block")
        c, n = self.core, self.NUM_DRONES
        keys = self.nx * self.ny * self.nz
        i32 = dict(dtype=torch.int32, device=self.device)
        count, start, order = torch.zeros(2 * (keys + 1), **i32), torch.zeros(keys + 1, **i32), torch.zeros(n, **i32)
        srt, out = torch.zeros((n, 4), dtype=torch.float32, device=self.device), torch.zeros(n, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = c.lib.gpd_downwash_global(ctypes.byref(c._params), _ptr(c.kin_store), c.ld, n, self.cell, self.x0, self.y0, self.nx, self.ny,
                                           self.z0, self.zbin, self.nz, None, _ptr(count), _ptr(start), _ptr(order), _ptr(srt), _ptr(out),
                                           None, None, None, c._stream())
        _native.check(rc, "gpd_downwash_global")
        return out

    def close(self):
        pass


class LocalSwarmGroup:
    """The ranks of ONE shared world inside one process (all on one device): the exchange is a device-to-device copy of every
    rank's slab into every rank's position array.  What the bitwise tests run on a single-GPU box; a deployment runs one
    process per GPU with `NativeSlabExchange` / `TorchSlabExchange` instead."""

    def __init__(self, num_drones: int, world_size: int, exchange: str = "allgather", halo_margin: float = 2.0, **kw):
        if exchange not in ("allgather", "halo"):
            raise ValueError("exchange must be 'allgather' or 'halo'")
        self.ranks = [SwarmAviary(num_drones, world_size=world_size, rank=r, exchange=self._noop, **kw) for r in range(world_size)]
        self.W = world_size
        # "halo": every rank plans (HaloPlan, the same three phases a multi-process world runs, the two all-gathers being a
        # torch.stack here) and receives only its neighbours' border drones
        self.plans = [HaloPlan(halo_margin) for _ in self.ranks] if exchange == "halo" else None
        self.plans_made = 0
        dev = self.ranks[0].device
        self._ids = [torch.as_tensor(e.GLOBAL_IDS, dtype=torch.long, device=dev) for e in self.ranks]
        self._all_ids = torch.cat(self._ids)

    @staticmethod
    def _noop(pos4, rank, slab):
        raise RuntimeError("LocalSwarmGroup exchanges for all its ranks at once")

    def _exchange(self):
        if self.plans is not None:
            return self._exchange_halo()
        for src in self.ranks:
            sl = slice(src.RANK * src.slab, (src.RANK + 1) * src.slab)
            for dst in self.ranks:
                if dst is not src:
                    dst.pos4[sl].copy_(src.pos4[sl])

    def _exchange_halo(self):
        e0 = self.ranks[0]
        if e0._since_bin >= e0.rebin_every or not self.plans[0].ready:
            bounds = torch.stack([P.phase1(e) for P, e in zip(self.plans, self.ranks)])
            counts = torch.stack([P.phase2(e, bounds) for P, e in zip(self.plans, self.ranks)])
            for P, e in zip(self.plans, self.ranks):
                P.phase3(e, counts)
            self.plans_made += 1
        for P, e in zip(self.plans, self.ranks):
            P.gather(e)
        for d, P in enumerate(self.plans):              # rank d's k-th receive from s <- rank s's k-th send to d
            for s_rank in range(self.W):
                if s_rank == d:
                    continue
                out = [b for p, b in self.plans[s_rank].sends if p == d]
                inn = [b for p, b in P.recvs if p == s_rank]
                assert len(out) == len(inn)
                for a, b in zip(out, inn):
                    b.copy_(a)

    @property
    def bytes_per_substep(self):
        """bytes every rank sends per sub-step: halo blocks + meta rows (halo), or what the all-gather moves"""
        if self.plans is not None:
            return [P.bytes_sent for P in self.plans] if self.plans[0].ready else None
        return [(self.W - 1) * e.slab * 16 for e in self.ranks]

    def reset(self):
        out = []
        for P in self.plans or ():
            P.forget()                  # the jump back to the initial poses is not a margin violation
        for e in self.ranks:
            e.core.reset()
            if e.ctrl is not None:
                e.ctrl.reset()
            e.step_counter = 0
            v = torch.empty((e.NUM_DRONES, 20), dtype=torch.float32, device=e.device)
            e._pack(v)
            out.append(v)
        self._exchange()
        for e in self.ranks:
            e._forces()
        return self._global(torch.cat(out))

    def _global(self, per_rank: torch.Tensor) -> torch.Tensor:
        """rows in rank order (rank 0's drones, rank 1's ...) -> rows in the caller's drone order"""
        out = torch.empty_like(per_rank)
        out[self._all_ids] = per_rank
        return out

    def step(self, action):
        """`action`: the whole world's (N, A) actions -> the whole world's (N, 20) state vectors"""
        a = torch.as_tensor(action, dtype=torch.float32, device=self.ranks[0].device)
        rpm = [e._kernel_action(a[ids]).contiguous() for e, ids in zip(self.ranks, self._ids)]
        vec = [torch.empty((e.NUM_DRONES, 20), dtype=torch.float32, device=e.device) for e in self.ranks]
        S = self.ranks[0].PYB_STEPS_PER_CTRL
        for s in range(S):
            for e, r, v in zip(self.ranks, rpm, vec):
                e._substep(r, v if s == S - 1 else None)
            self._exchange()
            for e in self.ranks:
                e._forces()
        for e in self.ranks:
            e.step_counter += S
        return self._global(torch.cat(vec))

    def forces(self):
        return self._global(torch.cat([e.dw_force[:e.NUM_DRONES] for e in self.ranks]))
