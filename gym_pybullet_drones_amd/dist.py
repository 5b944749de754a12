"""Multi-GPU: shard aviaries across ranks (one process per GPU), optional obs all-gather over RCCL.

Aviaries are independent (the only cross-drone coupling, downwash and the MultiHover reductions,
is inside one aviary), so the physics needs NO collective: rank r owns a contiguous block of envs
and runs the same kernel on its own device.  The one optional exchange step is an all-gather of
the `(E_local*D, 12)` observation shards into the concatenated `(E*D, 12)` tensor a centralised
learner would consume.  Two implementations of that one collective:

* `NativeObsAllGather` -- the C-ABI entry `gpd_allgather_obs` (`ncclAllGather` of RCCL over xGMI on the
  caller's stream, capturable in one hipGraph with the `gpd_step` launch that produced the shard, no torch
  in the data path) on ONE communicator per process (`NativeComm`); `torch.distributed` is only used once,
  to hand rank 0's communicator id to the others;
* `ObsAllGather` -- `torch.distributed.all_gather_into_tensor` (RCCL with the "nccl" backend on ROCm; with
  the gloo backend -- CPU tests, or the single-device test hook of bench.py -- device shards are staged
  through host memory).
"""
import ctypes
import os

import torch
import torch.distributed as dist


def env_shard(total_envs: int, rank: int, world_size: int):
    """Contiguous block [start, stop) of envs owned by `rank` (sizes differ by at most one; hand
    `shard_rows_per_rank()` to `ObsAllGather` when `total_envs % world_size != 0`)."""
    base, rem = divmod(total_envs, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_rows_per_rank(total_envs: int, world_size: int, rows_per_env: int = 1):
    """Observation rows each rank contributes under `env_shard` (rows_per_env = drones per aviary)."""
    return [(env_shard(total_envs, r, world_size)[1] - env_shard(total_envs, r, world_size)[0]) * rows_per_env
            for r in range(world_size)]


def init_from_env(backend: str = None, device=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (torchrun); returns (rank, world, local_rank).  `device`: the
    device this rank was given already (default with RCCL: device LOCAL_RANK)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local if device is None else device)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _backend(group=None) -> str:
    return dist.get_backend(group) if dist.is_initialized() else "none"


class ObsAllGather:
    """Pre-allocated all-gather of observation shards through `torch.distributed`.

    `shard_rows`: rows of THIS rank's shard.  Equal shards on every rank (the default, weak scaling) use ONE
    `all_gather_into_tensor`; `rows_per_rank` (e.g. `shard_rows_per_rank(E, world, D)` for an `env_shard` split
    of a total that the world size does not divide) pads every shard to the largest one for the collective and
    compacts the result, so that `full` is always the plain concatenation of the shards in rank order."""

    def __init__(self, shard_rows: int, cols: int = 12, device=None, dtype=torch.float32, group=None,
                 rows_per_rank=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group = group
        self.rows = [int(shard_rows)] * self.world if rows_per_rank is None else [int(r) for r in rows_per_rank]
        if len(self.rows) != self.world or self.rows[self.rank] != int(shard_rows):
            raise ValueError(f"rows_per_rank {self.rows} does not describe a world of {self.world} ranks in which "
                             f"rank {self.rank} owns {shard_rows} rows")
        self.uniform = len(set(self.rows)) == 1
        self.cols = cols
        self.full = torch.empty((sum(self.rows), cols), dtype=dtype, device=device)
        self._pad = None if self.uniform else torch.zeros((self.world, max(self.rows), cols), dtype=dtype, device=device)
        # gloo moves host memory only: device shards are staged (tests / bench.py's single-device hook)
        self._stage = self.world > 1 and _backend(group) == "gloo" and self.full.is_cuda

    def sized(self, shard_elems: int):
        """A gather of `shard_elems` elements per rank (equal shards), same group -- torch's collectives take any count."""
        if shard_elems % self.cols:
            raise ValueError(f"shard_elems must be a multiple of {self.cols}")
        if self.uniform and shard_elems == self.rows[self.rank] * self.cols:
            return self
        return ObsAllGather(shard_elems // self.cols, self.cols, device=self.full.device, dtype=self.full.dtype, group=self.group)

    def __call__(self, shard: torch.Tensor, async_op: bool = False):
        """Gather `shard` (rows, cols) from every rank into `self.full` (sum of rows, cols)."""
        shard = shard.reshape(-1, self.cols)
        if self.world == 1:
            self.full.copy_(shard)
            return self.full if not async_op else (self.full, None)
        if self._stage or not self.uniform:
            mine = shard if self.uniform else self._pad[self.rank]
            if not self.uniform:
                mine[:shard.shape[0]].copy_(shard)
            out = self.full if self.uniform else self._pad
            if self._stage:
                host = torch.empty(out.shape, dtype=out.dtype)
                dist.all_gather_into_tensor(host.view(-1, self.cols), mine.reshape(-1, self.cols).cpu().contiguous(),
                                            group=self.group)
                out.copy_(host)
            else:
                dist.all_gather_into_tensor(out.view(-1, self.cols), mine.reshape(-1, self.cols).contiguous(), group=self.group)
            if not self.uniform:
                torch.cat([self._pad[r, :n] for r, n in enumerate(self.rows)], out=self.full)
            return (self.full, None) if async_op else self.full
        work = dist.all_gather_into_tensor(self.full, shard.contiguous(), group=self.group, async_op=async_op)
        return (self.full, work) if async_op else self.full


class NativeComm:
    """ONE RCCL communicator of this process for the C-ABI collectives (`gpd_comm_*`): `ncclCommInitRank` runs once, every
    `NativeObsAllGather` -- whatever its element count -- shares it (`ncclAllGather` takes the count per call).
    `torch.distributed` (any backend) carries the 128-byte communicator id from rank 0 to the other ranks, once."""

    _shared = None

    def __init__(self, device=None, group=None):
        from . import _native
        self._native = _native
        self.lib = _native.lib()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        ident = (ctypes.c_uint8 * _native.COMM_ID_BYTES)()
        blob = [None]
        self.comm = None
        self.group = group
        with torch.cuda.device(self.device):
            if self.rank == 0:
                # a failure here (no RCCL on this box) must still reach the other ranks, which sit in the broadcast below:
                # rank 0 broadcasts the error text instead of the id and EVERY rank raises (callers then agree on a
                # fallback with all_ranks_ok instead of dead-locking between a broadcast and an all-reduce)
                try:
                    _native.check(self.lib.gpd_comm_unique_id(ident), "gpd_comm_unique_id")
                    blob[0] = bytes(ident)
                except Exception as e:      # noqa: BLE001
                    blob[0] = f"rank 0: {type(e).__name__}: {e}"
            if self.world > 1:
                dist.broadcast_object_list(blob, src=0, group=group)
            if isinstance(blob[0], str):
                raise _native.GpdError(blob[0])
            ident = (ctypes.c_uint8 * _native.COMM_ID_BYTES).from_buffer_copy(blob[0])
            comm = ctypes.c_void_p()
            _native.check(self.lib.gpd_comm_init(ctypes.byref(comm), ident, self.rank, self.world), "gpd_comm_init")
            self.comm = comm
        n = ctypes.c_int32(0)
        _native.check(self.lib.gpd_comm_count(self.comm, ctypes.byref(n)), "gpd_comm_count")
        self.ranks_seen = int(n.value)              # what RCCL itself says the communicator spans (ncclCommCount)

    @classmethod
    def shared(cls, device=None, group=None):
        """The process-wide communicator (created on first use; collective: every rank must call it)."""
        if cls._shared is None or cls._shared.comm is None:
            cls._shared = cls(device=device, group=group)
        have = cls._shared
        if (device is not None and torch.device(device) != have.device and torch.device(device).index is not None) or \
                (group is not None and group is not have.group):
            raise ValueError(f"the process-wide communicator lives on {have.device} / group {have.group}: a second device or group "
                             f"needs its own NativeComm(device=..., group=...)")
        return have

    def close(self):
        if getattr(self, "comm", None):
            self.lib.gpd_comm_destroy(self.comm)
            self.comm = None
        if NativeComm._shared is self:
            NativeComm._shared = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeObsAllGather:
    """The same collective through the C-ABI (`gpd_allgather_obs`): `ncclAllGather` of RCCL on the current stream, on the
    process's ONE communicator (`NativeComm.shared()` unless `comm` is given).  Equal shards only (what `ncclAllGather`
    offers).  `sized(n)` returns a gather of another element count on the same communicator."""

    def __init__(self, shard_rows: int, cols: int = 12, device=None, group=None, comm: "NativeComm" = None):
        self.nc = comm if comm is not None else NativeComm.shared(device=device, group=group)
        self._native, self.lib = self.nc._native, self.nc.lib
        self.world, self.rank, self.device = self.nc.world, self.nc.rank, self.nc.device
        self.cols = cols
        self.count = int(shard_rows) * cols
        self.full = torch.empty((self.world * int(shard_rows), cols), dtype=torch.float32, device=self.device)

    @property
    def comm(self):
        return self.nc.comm

    def sized(self, shard_elems: int):
        """A gather of `shard_elems` floats per rank (a multiple of `cols`) on the SAME communicator."""
        if shard_elems % self.cols:
            raise ValueError(f"shard_elems must be a multiple of {self.cols}")
        return self if shard_elems == self.count else NativeObsAllGather(shard_elems // self.cols, self.cols, comm=self.nc)

    def __call__(self, shard: torch.Tensor):
        if shard.numel() != self.count or shard.dtype != torch.float32 or not shard.is_contiguous():
            raise ValueError(f"shard must be a contiguous float32 tensor of {self.count} elements")
        with torch.cuda.device(self.device):
            rc = self.lib.gpd_allgather_obs(self.nc.comm, ctypes.c_void_p(shard.data_ptr()), ctypes.c_void_p(self.full.data_ptr()),
                                            self.count, ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._native.check(rc, "gpd_allgather_obs")
        return self.full

    def close(self):
        """(the communicator is the process's: `NativeComm.shared().close()` ends it)"""


def all_ranks_ok(ok: bool, device=None) -> bool:
    """True iff `ok` holds on EVERY rank (a MIN all-reduce through torch.distributed): lets all ranks take the same branch
    after a step that may fail on some of them only, e.g. fall back together from the native collective to torch's."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=device if _backend() != "gloo" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a python float over all ranks (timing)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if _backend() != "gloo" else None)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device=None) -> list:
    """`value` of every rank, in rank order, on every rank (per-GPU figures next to the whole-job one)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(value)]
    w = dist.get_world_size()
    t = torch.zeros(w, dtype=torch.float64, device=device if _backend() != "gloo" else None)
    t[dist.get_rank()] = value
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


# ---- bringing a multi-process job up, and saying what it found ------------------------------------------------------------------
def rccl_debug_tail(limit: int = 1500):
    """What RCCL wrote at NCCL_DEBUG=WARN into this process's NCCL_DEBUG_FILE (see `bring_up`), for the report of a failed job."""
    try:
        text = open(os.environ["NCCL_DEBUG_FILE"].replace("%p", str(os.getpid())).replace("%h", os.uname().nodename)).read()
        return text[-limit:] or None
    except Exception:   # noqa: BLE001
        return None


def device_record(rank: int, local_rank: int, device) -> dict:
    """This rank's line of the rank -> device map (`device` None: a host-only rank, as in the CPU tests)."""
    if device is None or not torch.cuda.is_available():
        return {"rank": rank, "local_rank": local_rank, "device": None, "name": "host", "pci": None, "uuid": None, "pid": os.getpid(), "visible": {}}
    props = torch.cuda.get_device_properties(device)
    return {"rank": rank, "local_rank": local_rank, "device": torch.device(device).index, "name": props.name,
            "pci": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None, "pid": os.getpid(),
            "visible": {k: os.environ[k] for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES") if k in os.environ}}


def bring_up(backend: str, device, topology: dict = None) -> dict:
    """Initialise the process group from the launcher's environment and say what the job spans, BEFORE anything is timed:
    the rank -> device map of all ranks (two ranks on one device under RCCL is an error with that map in its message, not a
    hang), an all-reduce of ones over the group (`ranks_in_process_group`), and the C-ABI's own communicator
    (`gpd_comm_*`: ncclCommInitRank / ncclCommCount -> `n_ranks_seen_by_rccl`; when it fails on ANY rank every rank drops it
    together and the native collectives fall back to torch.distributed).  RCCL's warnings go to a per-process file
    (`rccl_debug_tail`).  Fills and returns `topology` (the caller keeps a reference: a watchdog can print it while this hangs)."""
    topo = {} if topology is None else topology
    world = int(os.environ.get("WORLD_SIZE", "1"))
    topo.update(world_size=world, backend=backend if world > 1 else None)
    if world > 1 and backend == "nccl":
        os.environ.setdefault("NCCL_DEBUG", "WARN")
        os.environ.setdefault("NCCL_DEBUG_FILE", "/tmp/gpd_rccl_%h_%p.log")
    rank, world, local = init_from_env(backend if world > 1 else None, device=device)
    mine = device_record(rank, local, device)
    topo.update(rank=rank, local_rank=local, devices_on_node=torch.cuda.device_count(), rank_device_map=[mine],
                ranks_in_process_group=None, n_ranks_seen_by_rccl=None, native_comm_note=None)
    if world > 1:
        # through the rendezvous store, not through a collective: the map must exist BEFORE the first RCCL call can fail or hang
        import json
        store = dist.distributed_c10d._get_default_store()
        store.set(f"gpd/device/{rank}", json.dumps(mine))
        everyone = [json.loads(store.get(f"gpd/device/{r}")) for r in range(world)]
        topo["rank_device_map"] = everyone
        # (a duplicate has the same index, bus id AND uuid: ranks that were each given ONE visible device all say "device 0")
        if backend == "nccl" and len({(e["device"], e["pci"], e["uuid"]) for e in everyone}) < world:
            raise RuntimeError("two ranks of this job sit on the same device (see rank_device_map): RCCL refuses duplicate GPUs -- "
                               "launch one rank per GPU (LOCAL_RANK = device index)")
        one = torch.ones(1, dtype=torch.float32, device=device if backend == "nccl" else None)
        dist.all_reduce(one)
        topo["ranks_in_process_group"] = int(one.item())
        if backend == "nccl":
            err = None
            try:
                topo["n_ranks_seen_by_rccl"] = NativeComm.shared(device=device).ranks_seen
            except Exception as e:      # noqa: BLE001 -- reported, not fatal
                err = f"{type(e).__name__}: {e}"[:300]
            if not all_ranks_ok(err is None, device=device):
                topo["n_ranks_seen_by_rccl"], topo["native_comm_note"] = None, f"no native RCCL communicator ({err or 'failed on another rank'})"
                if NativeComm._shared is not None:
                    NativeComm._shared.close()
    return topo


def dry_run_exchange(topo: dict, backend: str, device):
    """The smallest real exchange: every rank contributes 12 floats (its rank), through the process group and -- when the native
    communicator exists -- through `gpd_allgather_obs`.  -> (ok, note)"""
    world, rank = topo["world_size"], topo["rank"]
    if world == 1:
        return True, "one rank: nothing to exchange"
    row = torch.full((1, 12), float(rank), device=device if backend == "nccl" else None)
    rows = [torch.empty_like(row) for _ in range(world)]
    dist.all_gather(rows, row)
    ok = [float(r[0, 0]) for r in rows] == [float(r) for r in range(world)]
    note = "torch.distributed.all_gather carried 12 floats per rank"
    if topo.get("n_ranks_seen_by_rccl"):
        try:
            full = NativeObsAllGather(1, 12, device=device)(torch.full((1, 12), float(rank), device=device))
            torch.cuda.synchronize()
            ok = ok and [float(x) for x in full[:, 0].cpu()] == [float(r) for r in range(world)]
            note = "gpd_allgather_obs (ncclAllGather through the C-ABI) carried 12 floats per rank"
        except Exception as e:      # noqa: BLE001
            ok, note = False, f"gpd_allgather_obs failed: {type(e).__name__}: {e}"[:300]
    return bool(ok), note
