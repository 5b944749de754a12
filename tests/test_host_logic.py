"""CPU-side checks: constants, URDF parsing, the C-ABI library's exports, sharding, gloo all-gather."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ASSETS, REPO, urdf


def test_constants_match_survey_appendix_d():
    from gym_pybullet_drones_amd.params import DroneParams
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    x = DroneParams(DroneModel.CF2X)
    assert x.M == 0.027 and x.L == 0.0397 and x.KF == 3.16e-10 and x.KM == 7.94e-12
    assert x.GRAVITY == pytest.approx(0.2646, rel=1e-12)
    assert x.HOVER_RPM == pytest.approx(14468.429183500699, rel=1e-14)
    assert x.MAX_RPM == pytest.approx(21702.64377525105, rel=1e-14)
    assert x.MAX_THRUST == pytest.approx(0.59535, rel=1e-12)
    assert x.MAX_XY_TORQUE == pytest.approx(0.00835637404, rel=1e-8)
    assert x.MAX_Z_TORQUE == pytest.approx(0.00747955538, rel=1e-8)
    assert x.GND_EFF_H_CLIP == pytest.approx(0.03776371349, rel=1e-8)
    assert x.SPEED_LIMIT == pytest.approx(0.25)
    np.testing.assert_allclose(x.default_init_xyzs(2), [[0, 0, 0.1125], [0.1588, 0.1588, 0.1125]], atol=1e-12)
    p = DroneParams(DroneModel.CF2P)
    assert p.MAX_XY_TORQUE == pytest.approx(0.00590884875, rel=1e-8) and p.J[0, 0] == 2.3951e-5
    r = DroneParams(DroneModel.RACE)
    assert r.HOVER_RPM == pytest.approx(15494.600499144828, rel=1e-14)
    assert r.MAX_RPM == pytest.approx(31640.869585066295, rel=1e-14)
    assert r.GND_EFF_H_CLIP == pytest.approx(0.20730637885, rel=1e-8)
    np.testing.assert_allclose(r.PROP_OFFSETS[:, :2], [[.085, .0675], [-.085, .0675], [-.085, -.0675], [.085, -.0675]])


@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
def test_name_based_and_positional_parsers_agree(model):
    """The product parser (by tag name) and the oracle's positional parser (the reference's way) read the same file."""
    from gym_pybullet_drones_amd.params import DroneParams
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    from oracle.aviary_oracle import UrdfConstants
    a, b = DroneParams(DroneModel(model)), UrdfConstants(urdf(model), model)
    for k in ("M", "L", "KF", "KM", "THRUST2WEIGHT_RATIO", "COLLISION_H", "COLLISION_R", "COLLISION_Z_OFFSET", "MAX_SPEED_KMH",
              "GND_EFF_COEFF", "PROP_RADIUS", "DW_COEFF_1", "DW_COEFF_2", "DW_COEFF_3", "GRAVITY", "HOVER_RPM", "MAX_RPM",
              "GND_EFF_H_CLIP", "MAX_XY_TORQUE", "MAX_Z_TORQUE", "SPEED_LIMIT"):
        assert getattr(a, k) == getattr(b, k), k
    np.testing.assert_array_equal(a.J, b.J)
    np.testing.assert_array_equal(a.DRAG_COEFF, b.DRAG_COEFF)
    np.testing.assert_array_equal(a.PROP_OFFSETS, b.PROP_OFFSETS)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout not present")
@pytest.mark.parametrize("model", ["cf2x", "cf2p", "racer"])
def test_shipped_assets_carry_the_reference_constants(model):
    from oracle.aviary_oracle import UrdfConstants
    mine = UrdfConstants(urdf(model), model)
    ref = UrdfConstants(f"/root/reference/gym_pybullet_drones/assets/{model}.urdf", model)
    for k, v in vars(ref).items():
        np.testing.assert_array_equal(getattr(mine, k), v, err_msg=k)


def test_enums_are_value_compatible():
    from gym_pybullet_drones_amd.utils.enums import ActionType, DroneModel, ObservationType, Physics
    assert [m.value for m in DroneModel] == ["cf2x", "cf2p", "racer"]
    assert [m.value for m in Physics] == ["pyb", "dyn", "pyb_gnd", "pyb_drag", "pyb_dw", "pyb_gnd_drag_dw"]
    assert [m.value for m in ActionType] == ["rpm", "pid", "vel", "one_d_rpm", "one_d_pid"]
    assert [m.value for m in ObservationType] == ["kin", "rgb"]
    assert [a.dim for a in ActionType] == [4, 3, 4, 1, 1]
    assert Physics.PYB_GND_DRAG_DW.flags == 7 and Physics.DYN.flags == 0
    # PYB* = the add-on models + the ground plane (8); Bullet's default damping (16; restated, unpinned) is opt-in since round 5;
    # pyb_like=False: exactly the reference's explicit integrator + the add-on models
    assert Physics.PYB.mask(True) == 8 and Physics.PYB_GND_DRAG_DW.mask(True) == 15 and Physics.DYN.mask(True) == 0
    assert Physics.PYB.mask("damped") == 24 and Physics.PYB_GND_DRAG_DW.mask("damped") == 31 and Physics.DYN.mask("damped") == 0
    assert Physics.PYB.mask(False) == 0 and Physics.PYB_DW.mask(False) == 4
    from gym_pybullet_drones_amd.utils import enums
    assert enums._pyb_mode("0") == "off" and enums._pyb_mode("1") == "ground" and enums._pyb_mode("damped") == "damped"
    with pytest.raises(ValueError):
        enums._pyb_mode("dampd")                 # a typo is not silently "ground"
    assert Physics.PYB.mask() == {"off": 0, "ground": 8, "damped": 24}[enums._pyb_like]      # (the process default: GPD_PYB_LIKE)
    keep = enums._pyb_like
    try:
        enums.set_pyb_like("damped")
        assert Physics.PYB.mask() == 24
        enums.set_pyb_like(False)
        assert Physics.PYB_GND.mask() == 1
    finally:
        enums.set_pyb_like(keep)


def test_trunc_counter_is_the_float64_threshold():
    from gym_pybullet_drones_amd.params import trunc_counter
    assert trunc_counter(8, 240) == 1920
    for L, f in ((8, 240), (8, 48), (0.1, 240), (7.3, 1000), (1 / 3, 240)):
        c = trunc_counter(L, f)
        assert not (c / f > L) and ((c + 1) / f > L)


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads (no GPU needed) and exports exactly the functions include/gpd.h declares."""
    from gym_pybullet_drones_amd import _native
    _native.build()
    L = _native.lib()
    hdr = open(os.path.join(REPO, "include", "gpd.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(gpd_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_native.exported_symbols())
    for name in declared:
        assert hasattr(L, name)
    assert L.gpd_abi_version() == int(re.search(r"#define GPD_ABI_VERSION (\d+)", hdr).group(1))
    # argument validation happens before any device work, so it is testable without a GPU
    assert L.gpd_step(None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert b"NULL" in L.gpd_last_error()
    assert L.gpd_pid(None, None, 0, ctypes.c_float(0.0), None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert L.gpd_pid_sync(None, None, 0, ctypes.c_float(0.0), None, None, None, None, None, None, None, None, None, None, 0, None) == -1


def test_struct_mirrors_match_the_header_layout():
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.params import DroneParams, GpdParams
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    sizes = (ctypes.c_int32 * 3)()
    _native.lib().gpd_struct_sizes(sizes)
    assert tuple(sizes) == (ctypes.sizeof(GpdParams), ctypes.sizeof(_native.GpdState), ctypes.sizeof(_native.GpdStepCfg))
    s = DroneParams(DroneModel.CF2P).to_struct(pid_model=DroneModel.CF2P)
    assert s.drone_model == 1 and list(s.mixer)[:3] == [0.0, -1.0, -1.0]
    assert s.pid_kf == pytest.approx(3.16e-10, rel=1e-6) and s.hover_rpm == pytest.approx(14468.429, rel=1e-6)
    r = DroneParams(DroneModel.RACE).to_struct(pid_model=DroneModel.RACE)
    assert r.pid_kf == 0.0        # no controller: the kernel entry points refuse PID action types


def test_product_path_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.envs import HoverAviary, VectorHoverAviary
    with pytest.raises(_native.GpdError):
        HoverAviary()
    with pytest.raises(_native.GpdError):
        VectorHoverAviary(16)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "gym_pybullet_drones_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(root, f)


def test_env_shard_partitions():
    from gym_pybullet_drones_amd.dist import env_shard
    for total, world in ((524288, 8), (10, 3), (7, 8), (131072, 4)):
        blocks = [env_shard(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and blocks[-1][1] == total
        assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_obs_allgather_world2_gloo(tmp_path):
    """The N>1 exchange path (all-gather of observation shards) on CPU with gloo, world_size 2."""
    script = tmp_path / "w.py"
    script.write_text(f"""
import sys; sys.path.insert(0, {REPO!r})
import torch, torch.distributed as dist
from gym_pybullet_drones_amd import dist as gdist
rank, world, local = gdist.init_from_env("gloo")
assert world == 2
a, b = gdist.env_shard(10, rank, world)
shard = torch.arange((b - a) * 12, dtype=torch.float32).reshape(b - a, 12) + 1000 * rank
ag = gdist.ObsAllGather(b - a, 12, device="cpu")
full = ag(shard)
assert full.shape == (10, 12)
assert torch.equal(full[:5], torch.arange(60, dtype=torch.float32).reshape(5, 12))
assert torch.equal(full[5:], torch.arange(60, dtype=torch.float32).reshape(5, 12) + 1000)
full2, work = ag(shard, async_op=True); work.wait()
assert torch.equal(full2, full)
assert gdist.max_over_ranks(float(rank)) == 1.0
# a total the world size does not divide: env_shard hands out 6 + 5 aviaries of 2 drones, the gather pads and compacts
a, b = gdist.env_shard(11, rank, world)
rows = gdist.shard_rows_per_rank(11, world, rows_per_env=2)
assert rows == [12, 10] and rows[rank] == (b - a) * 2
shard = (torch.arange(rows[rank] * 12, dtype=torch.float32).reshape(rows[rank], 12) + 1000 * rank)
ag2 = gdist.ObsAllGather(rows[rank], 12, device="cpu", rows_per_rank=rows)
full = ag2(shard)
assert full.shape == (22, 12)
assert torch.equal(full[:12], torch.arange(144, dtype=torch.float32).reshape(12, 12))
assert torch.equal(full[12:], torch.arange(120, dtype=torch.float32).reshape(10, 12) + 1000)
try:
    gdist.ObsAllGather(rows[rank] + 1, 12, device="cpu", rows_per_rank=rows)
    raise SystemExit("a rows_per_rank list that contradicts this rank's shard must be refused")
except ValueError:
    pass
dist.barrier(); dist.destroy_process_group()
open({str(tmp_path)!r} + f"/ok{{rank}}", "w").write("ok")
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_bring_up_reports_the_topology_world2_gloo(tmp_path):
    """`dist.bring_up` (what bench.py's Job and its `--dry-run-topology` stand on), world_size 2 over gloo on CPU: the rank -> device
    map travels through the rendezvous store (it exists before any collective runs), the group's own all-reduce counts two ranks,
    and the 12-float exchange carries every rank's row."""
    script = tmp_path / "up.py"
    script.write_text(f"""
import json, sys
sys.path.insert(0, {REPO!r})
import torch.distributed as dist
from gym_pybullet_drones_amd import dist as gdist
topo = gdist.bring_up("gloo", None)
ok, note = gdist.dry_run_exchange(topo, "gloo", None)
assert topo["world_size"] == 2 and topo["ranks_in_process_group"] == 2 and topo["n_ranks_seen_by_rccl"] is None
assert [e["rank"] for e in topo["rank_device_map"]] == [0, 1] and len({{e["pid"] for e in topo["rank_device_map"]}}) == 2
assert ok and "12 floats" in note
dist.barrier(); dist.destroy_process_group()
open({str(tmp_path)!r} + f"/up{{topo['rank']}}", "w").write(json.dumps(topo))
""")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29537", str(script)],
                         capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    assert (tmp_path / "up0").exists() and (tmp_path / "up1").exists()


def test_package_installs_with_pip_and_imports_from_anywhere(tmp_path):
    """`pip install .` (setup.cfg metadata, setup.py's build hook; GPD_SKIP_NATIVE_BUILD=1 here: the hook's hipcc step
    is `_native.build()`, which `__graft_entry__.build()` exercises) into an empty directory, then -- from another working
    directory, without the source tree on the path -- the package imports under its real name, carries its URDFs, kernels' sources
    and the C header, and registers the reference's four environment ids."""
    site = tmp_path / "site"
    env = dict(os.environ, GPD_SKIP_NATIVE_BUILD="1")
    res = subprocess.run([sys.executable, "-m", "pip", "install", "--no-build-isolation", "--no-deps", "--quiet", "--target", str(site), REPO],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    pkg = site / "gym_pybullet_drones_amd"
    for rel in ("__init__.py", "engine.py", "envs/HoverAviary.py", "assets/cf2x.urdf", "csrc/step_rollout.hip", "csrc/gpd_common.inc", "include/gpd.h"):
        assert (pkg / rel).exists(), rel
    code = ("import gym_pybullet_drones_amd as g, gym_pybullet_drones_amd.envs as e, gym_pybullet_drones_amd.control as c, importlib.metadata as m;"
            "from gym_pybullet_drones_amd import _gym_shim as s;"
            "ids = set(s._REGISTRY) if not s.HAVE_GYMNASIUM else {i for i in __import__('gymnasium').envs.registration.registry};"
            "assert {'hover-aviary-v0', 'multihover-aviary-v0', 'ctrl-aviary-v0', 'velocity-aviary-v0'} <= ids;"
            "print(g.__file__, m.version('gym-pybullet-drones-amd'), e.HoverAviary.__name__, c.DSLPIDControl.__name__)")
    run = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(tmp_path),
                         env={k: v for k, v in dict(os.environ, PYTHONPATH=str(site)).items()})
    assert run.returncode == 0, run.stdout + run.stderr
    assert str(site) in run.stdout and "0.1.0" in run.stdout and "HoverAviary DSLPIDControl" in run.stdout
    import shutil
    for junk in ("build", "gym_pybullet_drones_amd.egg-info", "gym_pybullet_drones_amd/include"):      # (what the in-tree build leaves behind)
        shutil.rmtree(os.path.join(REPO, junk), ignore_errors=True)
    assert not os.path.exists(os.path.join(REPO, "gym_pybullet_drones_amd", "__init__.py")) or \
        "exec(" not in open(os.path.join(REPO, "gym_pybullet_drones_amd", "__init__.py")).read()      # the alias hack is gone


def test_placement_layout_candidates_are_disjoint_and_spread():
    """`placement.layout_candidates`: every (head, tail) pair on the grid lies inside the arena with the two units disjoint, no pair
    twice, and the probing order is spread out (the first eight probes already span most of the arena in both coordinates)."""
    from gym_pybullet_drones_amd.placement import GiB, layout_candidates
    A, H, T, G = 88 * GiB, 12 * GiB, int(5.6 * GiB), int(2.2 * GiB)
    c = layout_candidates(A, H, T, G)
    assert len(c) == len(set(c)) > 500
    assert all(0 <= h and h + H <= A and 0 <= t and t + T <= A and (t + T <= h or t >= h + H) for h, t in c)
    first = c[:8]
    assert max(h for h, _ in first) - min(h for h, _ in first) > A / 2 and max(t for _, t in first) - min(t for _, t in first) > A / 2
    assert layout_candidates(H + T, H, T, G) == [(0, H)] or len(layout_candidates(H + T + G, H, T, G)) >= 2      # (a tight arena still has a layout)


def test_logger_layout_and_files(tmp_path):
    """Reference layout: states (N,16,T) = [pos, vel, rpy, ang_vel, rpm] re-ordered from the 20-float state vector
    (utils/Logger.py:117), growing arrays when duration_sec = 0, one CSV per signal and drone."""
    from gym_pybullet_drones_amd.utils.Logger import Logger
    from gym_pybullet_drones_amd.utils.utils import str2bool
    lg = Logger(logging_freq_hz=10, output_folder=str(tmp_path / "r"), num_drones=2)
    for k in range(5):
        for j in range(2):
            lg.log(j, k / 10, np.arange(20, dtype=float) + 100 * j + k, control=np.arange(12, dtype=float))
    assert lg.states.shape == (2, 16, 5) and lg.controls.shape == (2, 12, 5)
    np.testing.assert_array_equal(lg.states[1, :, 2], np.r_[0:3, 10:13, 7:10, 13:20] + 102.0)
    d = lg.save_as_csv("x")
    names = sorted(os.listdir(d))
    assert len(names) == 2 * 23 and "rpm3-1.csv" in names and "yar0.csv" in names
    pre = Logger(logging_freq_hz=10, output_folder=str(tmp_path / "p"), num_drones=1, duration_sec=1)
    assert pre.states.shape == (1, 16, 10)
    pre.log(0, 0.0, np.zeros(20)); pre.log(0, 0.1, np.ones(20))
    assert pre.counters[0] == 2 and pre.states[0, 0, 1] == 1
    assert str2bool("yes") is True and str2bool("0") is False


def test_swarm_partition_and_slab_exchange_world_size_2():
    """One world dealt to ranks (`swarm_partition`: contiguous blocks, one meta row per slab) and the all-gather of the ranks'
    position slabs through torch.distributed (gloo, two processes, CPU tensors): after the exchange every rank holds the same
    array, rank r's slab (its drones, then rows without one, the last ceil(per / 256) its meta rows) at rows r*slab."""
    import torch.multiprocessing as mp
    from gym_pybullet_drones_amd.envs.SwarmAviary import swarm_partition
    assert swarm_partition(10, 1) == (10, 11, [10])
    from gym_pybullet_drones_amd.envs.SwarmAviary import swarm_first_drone
    assert swarm_partition(10, 4) == (3, 4, [3, 3, 2, 2])          # balanced: sizes differ by at most one
    assert swarm_partition(65536, 8) == (8192, 8224, [8192] * 8)
    assert swarm_partition(9, 4) == (3, 4, [3, 2, 2, 2])            # (blocks of ceil(N / W) would leave the fourth rank empty)
    assert swarm_partition(4, 4) == (1, 2, [1, 1, 1, 1])
    for n, w in ((10, 4), (9, 4), (1501, 3), (65536, 8), (7, 7)):
        per, slab, counts = swarm_partition(n, w)
        firsts = [swarm_first_drone(n, w, r) for r in range(w)]
        assert firsts == [sum(counts[:r]) for r in range(w)] and sum(counts) == n and max(counts) == per and min(counts) >= 1
    with pytest.raises(ValueError):
        swarm_partition(3, 4)
    # the spatial deal: a permutation, row-major over the cells of the initial positions, index order inside a cell
    from gym_pybullet_drones_amd.envs.SwarmAviary import swarm_spatial_order
    rng = np.random.default_rng(0)
    xyz = np.concatenate([rng.uniform(0, 42, size=(500, 2)), np.ones((500, 1))], axis=1)
    o = swarm_spatial_order(xyz, 10.5)
    assert sorted(o.tolist()) == list(range(500))
    cx, cy = np.floor((xyz[o, 0] - xyz[:, 0].min()) / 10.5), np.floor((xyz[o, 1] - xyz[:, 1].min()) / 10.5)
    key = cy * 100 + cx
    assert np.all(np.diff(key) >= 0) and np.all(np.diff(o)[np.diff(key) == 0] > 0)
    port = 29500 + os.getpid() % 2000
    mp.spawn(_slab_exchange_worker, args=(2, port), nprocs=2, join=True)


def _slab_exchange_worker(rank, world, port):
    import torch
    import torch.distributed as dist
    from gym_pybullet_drones_amd.envs.SwarmAviary import TorchSlabExchange, swarm_partition
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N = 7
        per, slab, counts = swarm_partition(N, world)
        pos4 = torch.full((slab * world, 4), float("nan"))
        own = torch.arange(counts[rank] * 4, dtype=torch.float32).view(-1, 4) + 100 * rank
        pos4[rank * slab:rank * slab + counts[rank]] = own
        pos4[(rank + 1) * slab - 1, 3] = 0.5 + rank                     # the meta row's dmax^2
        TorchSlabExchange()(pos4, rank, slab)
        for r in range(world):
            want = torch.arange(counts[r] * 4, dtype=torch.float32).view(-1, 4) + 100 * r
            assert torch.equal(pos4[r * slab:r * slab + counts[r]], want)
            assert pos4[r * slab + counts[r]:(r + 1) * slab, :3].isnan().all()
            assert float(pos4[(r + 1) * slab - 1, 3]) == 0.5 + r
    finally:
        dist.destroy_process_group()


def _stub_rccl(tmp):
    """tests/stubs/rccl_stub.c (a stand-in for librccl.so that moves host memory between processes) -> a shared object"""
    out = os.path.join(str(tmp), "librccl_stub.so")
    subprocess.run(["gcc", "-O1", "-shared", "-fPIC", os.path.join(REPO, "tests", "stubs", "rccl_stub.c"), "-o", out, "-lpthread", "-lrt"], check=True)
    return out


def _comm_worker(rank, world, stub, idfile):
    """one rank of the C-ABI's collective glue: unique id (rank 0) -> init -> count -> all-gather (out of place, and in place like
    the position slabs of a shared swarm world) -> destroy"""
    import time
    os.environ["GPD_RCCL_LIB"] = stub
    from gym_pybullet_drones_amd import _native
    L = _native.lib()
    ident = (ctypes.c_uint8 * _native.COMM_ID_BYTES)()
    if rank == 0:
        _native.check(L.gpd_comm_unique_id(ident), "gpd_comm_unique_id")
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.rename(idfile + ".tmp", idfile)
    else:
        for _ in range(2000):
            if os.path.exists(idfile):
                break
            time.sleep(0.005)
        ident = (ctypes.c_uint8 * _native.COMM_ID_BYTES).from_buffer_copy(open(idfile, "rb").read())
    comm = ctypes.c_void_p()
    _native.check(L.gpd_comm_init(ctypes.byref(comm), ident, rank, world), "gpd_comm_init")
    n = ctypes.c_int32(0)
    _native.check(L.gpd_comm_count(comm, ctypes.byref(n)), "gpd_comm_count")
    assert n.value == world
    rows = 5
    shard = (np.arange(rows * 12, dtype=np.float32) + 1000 * rank).reshape(rows, 12)
    full = np.zeros((world * rows, 12), dtype=np.float32)
    for _ in range(3):                           # (one communicator, several calls, two counts: what bench.py's gathers do)
        _native.check(L.gpd_allgather_obs(comm, shard.ctypes.data_as(ctypes.c_void_p), full.ctypes.data_as(ctypes.c_void_p), rows * 12, None),
                      "gpd_allgather_obs")
        for r in range(world):
            assert np.array_equal(full[r * rows:(r + 1) * rows], (np.arange(rows * 12, dtype=np.float32) + 1000 * r).reshape(rows, 12))
    # in place: every rank's slab already sits at its place in the receive buffer (gpd.h, GpdSwarm)
    slab = 7
    pos4 = np.full((world * slab, 4), np.nan, dtype=np.float32)
    pos4[rank * slab:(rank + 1) * slab] = np.arange(slab * 4, dtype=np.float32).reshape(slab, 4) + 100 * rank
    send = ctypes.c_void_p(pos4.ctypes.data + rank * slab * 16)
    _native.check(L.gpd_allgather_obs(comm, send, pos4.ctypes.data_as(ctypes.c_void_p), slab * 4, None), "gpd_allgather_obs (in place)")
    for r in range(world):
        assert np.array_equal(pos4[r * slab:(r + 1) * slab], np.arange(slab * 4, dtype=np.float32).reshape(slab, 4) + 100 * r)
    # grouped point-to-point (gpd_p2p_group: the halo exchange): rank r sends two blocks of different sizes to every other rank --
    # 3 + r + 2 p floats, then 2 -- and the blocks of a pair arrive in the order they were listed
    def block(src, dst, k):
        n = 3 + src + 2 * dst if k == 0 else 2
        return (np.arange(n, dtype=np.float32) + 100 * src + 10 * dst + 1000 * k)
    sends, recvs, keep = [], [], []
    for p in range(world):
        if p == rank:
            continue
        for k in range(2):
            out = np.ascontiguousarray(block(rank, p, k)); keep.append(out)
            sends.append(_native.GpdP2P(peer=p, ptr=out.ctypes.data, count=out.size))
            buf = np.full(block(p, rank, k).size, -1, dtype=np.float32); keep.append(buf)
            recvs.append((p, k, buf))
    S = (_native.GpdP2P * len(sends))(*sends)
    Rv = (_native.GpdP2P * len(recvs))(*[_native.GpdP2P(peer=p, ptr=b.ctypes.data, count=b.size) for p, k, b in recvs])
    for _ in range(2):
        _native.check(L.gpd_p2p_group(comm, S, len(sends), Rv, len(recvs), None), "gpd_p2p_group")
        for p, k, b in recvs:
            assert np.array_equal(b, block(p, rank, k)), (rank, p, k, b)
            b[:] = -1
    assert L.gpd_p2p_group(comm, None, 1, Rv, len(recvs), None) == _native.GPD_EINVAL
    assert L.gpd_p2p_group(comm, S, 0, Rv, 0, None) == 0                     # nothing to do
    # argument errors come back as codes, not crashes
    assert L.gpd_allgather_obs(comm, None, full.ctypes.data_as(ctypes.c_void_p), 12, None) == _native.GPD_EINVAL
    assert L.gpd_comm_count(None, ctypes.byref(n)) == _native.GPD_EINVAL
    _native.check(L.gpd_comm_destroy(comm), "gpd_comm_destroy")


@pytest.mark.parametrize("world", [2, 3])
def test_comm_glue_with_several_ranks_over_a_stub_rccl(tmp_path, world):
    """The C-ABI's multi-rank entries (`gpd_comm_unique_id / _init / _count`, `gpd_allgather_obs`, `gpd_p2p_group`, `gpd_comm_destroy`) with
    rank > 0 and world_size > 1 -- which no single-GPU box can run against RCCL itself (it refuses two ranks on one device):
    `GPD_RCCL_LIB` points libgpd.so's dlopen at tests/stubs/rccl_stub.c, a stand-in that moves HOST buffers between processes
    through shared memory.  What it pins: the id hand-over, init on every rank, ONE communicator serving several calls and
    counts, the shard order of the gathered tensor, the in-place form the swarm slabs use, and the error codes."""
    import torch.multiprocessing as mp
    stub = _stub_rccl(tmp_path)
    mp.spawn(_comm_worker, args=(world, stub, os.path.join(str(tmp_path), "id.bin")), nprocs=world, join=True)


def test_header_is_plain_c_and_a_c_program_links(tmp_path):
    """include/gpd.h is the drop-in boundary: plain C (C11, -pedantic, no warnings), and a C program built against it links
    libgpd.so, finds the struct sizes of the header in the library and gets an error code (not a crash) for a NULL call --
    `examples/c/abi_check.c`, no GPU needed."""
    from gym_pybullet_drones_amd import _native
    _native.lib()
    exe = os.path.join(str(tmp_path), "abi_check")
    res = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(REPO, "include"),
                          os.path.join(REPO, "examples", "c", "abi_check.c"), "-o", exe, "-L", _native.CSRC, "-lgpd",
                          "-Wl,-rpath," + _native.CSRC], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ABI 9" in run.stdout and "-> -1:" in run.stdout


def test_every_entry_answers_null_pointers_with_a_code_not_a_crash():
    """Every function of include/gpd.h called with NULL for every pointer (and 1, -1, INT_MAX for every integer): a negative
    code and a message that names the entry -- the reference's convention is `print` + `exit()`, the
    boundary's is an error code; no GPU needed, nothing is launched."""
    from gym_pybullet_drones_amd import _native
    L = _native.lib()
    plain = {"gpd_abi_version", "gpd_last_error", "gpd_struct_sizes", "gpd_sizeof_swarm", "gpd_comm_destroy"}     # (no failure mode / NULL is fine)
    ints = (ctypes.c_int, ctypes.c_int32, ctypes.c_int64, ctypes.c_size_t)
    tried = 0
    for ival in (1, -1, 2 ** 31 - 1):
        for name in sorted(set(_native.exported_symbols()) - plain):
            f = getattr(L, name)
            args = [(0 if (t is ctypes.c_size_t and ival < 0) else ival) if t in ints else 1.0 if t is ctypes.c_float else None for t in f.argtypes]
            rc = f(*args)
            msg = L.gpd_last_error().decode()
            assert rc < 0 and msg.startswith(name), (name, ival, rc, msg)
            tried += 1
    assert tried >= 3 * 20
    assert L.gpd_comm_destroy(None) == 0          # (destroying nothing is not an error)


def test_swarm_entries_reject_bad_arguments_before_touching_a_device():
    """The `GpdSwarm` entries validate their arguments before the first HIP call: codes and messages, no crash, no GPU needed."""
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd.params import DroneParams
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    L = _native.lib()
    P = DroneParams(DroneModel.CF2X).to_struct(pid_model=DroneModel.CF2X)
    buf = (ctypes.c_float * 4096)()
    ptr = ctypes.cast(buf, ctypes.c_void_p).value
    ok = dict(n_rows=80, slab=40, world_size=2, rank=1, own_count=30, nx=3, ny=3, nz=1, cell=10.5, x0=0.0, y0=0.0, z0=0.0, zbin=1.0,
              meta_rows=1, pos4=ptr, bin_pos=ptr, cell_count=ptr, cell_start=ptr, order=ptr, visit=None, visit_out=ptr + 1024,
              slot_key=ptr, dw_force=ptr, slot_of=None, pos_sorted=None, pair_list=None, pair_nb=None, list_ok=None, list_cap=0,
              list_delta=0.0, drift=ptr, total_drones=60, list_adapt=1)

    def rc(entry, **change):
        sw = _native.GpdSwarm(**{**ok, **change})
        if entry == "bin":
            return L.gpd_swarm_bin(ctypes.byref(sw), None)
        return L.gpd_swarm_forces(ctypes.byref(P), ctypes.byref(sw), 0, None)

    for entry in ("bin", "forces"):
        assert rc(entry, cell=9.0) == _native.GPD_EINVAL and b"cell must be >= 10" in L.gpd_last_error()
        assert rc(entry, n_rows=81) == _native.GPD_EINVAL                    # != world_size * slab
        assert rc(entry, rank=2) == _native.GPD_EINVAL
        assert rc(entry, own_count=40) == _native.GPD_EINVAL                 # no room for the meta rows
        assert rc(entry, meta_rows=0) == _native.GPD_EINVAL
        assert rc(entry, nx=2) == _native.GPD_ERANGE and rc(entry, nx=300, ny=300) == _native.GPD_ERANGE
        assert rc(entry, pos4=None) == _native.GPD_EINVAL and rc(entry, drift=None) == _native.GPD_EINVAL
        assert rc(entry, total_drones=0) == _native.GPD_EINVAL
        assert rc(entry, visit=ptr) == _native.GPD_EINVAL                    # aliases order
        assert rc(entry, slot_of=ptr) == _native.GPD_EINVAL                  # ... comes with pos_sorted
        assert rc(entry, slot_of=ptr, pos_sorted=ptr) == _native.GPD_EINVAL  # ... and only when this rank holds the whole world
    assert rc("forces", pair_list=ptr) == _native.GPD_EINVAL                 # lists need pair_nb / list_ok / list_cap
    assert L.gpd_swarm_forces(None, ctypes.byref(_native.GpdSwarm(**ok)), 0, None) == _native.GPD_EINVAL
    # gpd_swarm_step: the step configuration it accepts
    st = _native.GpdState(kin=ptr, last_rpm=ptr, pid=None, step_counter=ptr, ld=64)
    cfg = dict(num_envs=30, drones_per_env=1, act_type=5, substeps=1, physics_flags=7, pyb_dt=1 / 240, ctrl_dt=1 / 240, inv_ctrl_dt=240.0,
               lanes_per_wave=64, task=0, xy_bound=1.5, z_bound=2.0, tilt_bound=0.4, term_dist=1e-4, trunc_counter=1920, target_per_env=0,
               init_per_env=0, auto_reset=0)
    sw = _native.GpdSwarm(**ok)

    def step_rc(**change):
        c = _native.GpdStepCfg(**{**cfg, **change})
        return L.gpd_swarm_step(ctypes.byref(P), ctypes.byref(st), ctypes.byref(c), ctypes.byref(sw), ptr, ptr, None, None)

    assert step_rc(substeps=2) == _native.GPD_ENOTSUP and step_rc(auto_reset=1) == _native.GPD_ENOTSUP and step_rc(task=1) == _native.GPD_ENOTSUP
    assert step_rc(act_type=1) == _native.GPD_ENOTSUP                        # waypoints: gpd_pid first
    assert step_rc(num_envs=31) == _native.GPD_EINVAL and step_rc(drones_per_env=2) == _native.GPD_EINVAL
    assert step_rc(physics_flags=64) == _native.GPD_EINVAL
    assert L.gpd_swarm_pack(None, ctypes.byref(sw), None, None, None) == _native.GPD_EINVAL


# ---- halo exchange of a shared world: the plan on CPU tensors, the torch transport over gloo ------------------
class _FakeRank:
    """what HaloPlan reads of a SwarmAviary: the partition numbers and the packed position array (here: CPU tensors)"""
    def __init__(self, N, W, r, xyz_dealt):
        import torch
        from gym_pybullet_drones_amd.envs.SwarmAviary import swarm_first_drone, swarm_partition
        self.WORLD_SIZE, self.RANK = W, r
        self.per, self.slab, counts = swarm_partition(N, W)
        self.NUM_DRONES = counts[r]
        self.first = swarm_first_drone(N, W, r)
        self.pos4 = torch.full((W * self.slab, 4), float("nan"))
        self.device = self.pos4.device
        self.set_own(xyz_dealt)

    def set_own(self, xyz_dealt):
        import torch
        own = torch.as_tensor(xyz_dealt[self.first:self.first + self.NUM_DRONES], dtype=torch.float32)
        self.pos4[self.RANK * self.slab:self.RANK * self.slab + self.NUM_DRONES, :3] = own
        self.pos4[self.RANK * self.slab:self.RANK * self.slab + self.NUM_DRONES, 3] = 0
        meta = self.slab - self.per
        self.pos4[(self.RANK + 1) * self.slab - meta:(self.RANK + 1) * self.slab, 1:] = 7.0 + self.RANK       # (x stays NaN: no drone)


def _plan_all(plans, ranks):
    import torch
    bounds = torch.stack([P.phase1(e) for P, e in zip(plans, ranks)])
    counts = torch.stack([P.phase2(e, bounds) for P, e in zip(plans, ranks)])
    for P, e in zip(plans, ranks):
        P.phase3(e, counts)
    for P, e in zip(plans, ranks):
        P.gather(e)
    for d, P in enumerate(plans):
        for s in range(len(plans)):
            if s != d:
                out, inn = [b for p, b in plans[s].sends if p == d], [b for p, b in P.recvs if p == s]
                assert len(out) == len(inn)
                for a, b in zip(out, inn):
                    b.copy_(a)


@pytest.mark.parametrize("W", [2, 3, 8])
def test_halo_plan_covers_every_pair_the_model_couples(W):
    """`HaloPlan` on CPU tensors: a 200 m x 150 m world dealt to W ranks in stripes.  After the exchange every rank holds, for
    each of its drones, EVERY drone of another rank within the model's 10 m lateral cut-off (envs/BaseAviary.py:801) -- now, and
    after every drone has moved up to margin / 2 in any direction; stripes make the far ranks send nothing but their meta rows;
    the bytes are a fraction of the all-gather's; a drone that outruns the margin is reported on every rank by the next plan."""
    import torch
    from gym_pybullet_drones_amd.envs.SwarmAviary import HaloPlan, swarm_spatial_order
    rng = np.random.default_rng(W)
    N, margin = 4000, 2.0
    xyz = np.concatenate([rng.uniform(0, 200, size=(N, 1)), rng.uniform(0, 150, size=(N, 1)), rng.uniform(1, 12, size=(N, 1))], axis=1)
    order = swarm_spatial_order(xyz, 10.5)
    dealt = xyz[order]
    ranks = [_FakeRank(N, W, r, dealt) for r in range(W)]
    plans = [HaloPlan(margin) for _ in ranks]
    _plan_all(plans, ranks)
    owner = np.concatenate([np.full(e.NUM_DRONES, e.RANK) for e in ranks])

    def held_by(e):
        p = e.pos4[:, :3].numpy()
        return p[np.isfinite(p).all(axis=1)]

    def check(now):
        for e in ranks:
            mine = now[owner == e.RANK]
            have = held_by(e)
            others = now[owner != e.RANK]
            d = np.hypot(others[:, None, 0] - mine[None, :, 0], others[:, None, 1] - mine[None, :, 1]).min(axis=1)
            need = others[d < 10.0]
            # every needed drone's CURRENT position is among the rows the rank holds (the same rows travel every sub-step)
            assert len(need) > 0
            key = {tuple(np.float32(v)) for v in have}
            assert all(tuple(np.float32(v)) in key for v in need), e.RANK

    check(dealt)
    # the same plan, every drone displaced by up to margin / 2 (in y; any amount in x and z): re-send current positions, still covered
    moved = dealt + np.concatenate([rng.uniform(-30, 30, size=(N, 1)), rng.uniform(-0.49 * margin, 0.49 * margin, size=(N, 1)),
                                    rng.uniform(-1, 1, size=(N, 1))], axis=1)
    for e in ranks:
        e.set_own(moved)
    for P, e in zip(plans, ranks):
        P.gather(e)
    for d, P in enumerate(plans):
        for s in range(W):
            if s != d:
                for a, b in zip([b for p, b in plans[s].sends if p == d], [b for p, b in P.recvs if p == s]):
                    b.copy_(a)
    check(moved)
    # stripes: far ranks exchange nothing but meta rows; every rank got everybody's meta rows; the halo is a fraction of the all-gather
    for e, P in zip(ranks, plans):
        for s in range(W):
            if abs(s - e.RANK) >= 2 and W > 3:
                assert P.recv_cnt[s] == 0 or abs(s - e.RANK) == 2, (e.RANK, s, P.recv_cnt)
            if s != e.RANK:
                meta = e.pos4.view(W, e.slab, 4)[s, e.per:]
                assert torch.isnan(meta[:, 0]).all() and (meta[:, 1:] == 7.0 + s).all()
        if W > 2:
            assert P.bytes_sent < 0.7 * P.bytes_allgather, (P.bytes_sent, P.bytes_allgather)
    # a drone that outran the margin: the next plan says so, on every rank
    far = moved.copy()
    far[order.tolist().index(int(order[5])), 1] += 3 * margin
    for e in ranks:
        e.set_own(far)
    bounds = torch.stack([P.phase1(e) for P, e in zip(plans, ranks)])
    assert float(bounds[:, 2].sum()) >= 1
    for P, e in zip(plans, ranks):
        with pytest.raises(RuntimeError, match="halo exchange"):
            P.phase2(e, bounds)
    # ... ONCE: the violation is reported, the plan and the positions it was made from are dropped, and the next exchange plans
    # afresh from where the drones are (`y_plan` used to survive the raise, so every later call raised too)
    assert all(P.y_plan is None and not P.ready for P in plans)
    _plan_all(plans, ranks)
    check(far)
    # a legitimate teleport (reset() to the initial poses, set_state(), invalidate(): everything that re-packs calls forget()):
    # every drone jumps by much more than margin / 2 -- not a violation, the new plan covers the new positions
    for P, e in zip(plans, ranks):
        P.forget()
        e.set_own(dealt)
    _plan_all(plans, ranks)
    check(dealt)
    assert all(P.ready for P in plans)


def _halo_gloo_worker(rank, world, port, N):
    import torch
    import torch.distributed as dist
    from gym_pybullet_drones_amd.envs.SwarmAviary import TorchHaloExchange, swarm_spatial_order
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)
        xyz = np.concatenate([rng.uniform(0, 100, size=(N, 1)), rng.uniform(0, 90, size=(N, 1)), rng.uniform(1, 5, size=(N, 1))], axis=1)
        dealt = xyz[swarm_spatial_order(xyz, 10.5)]
        e = _FakeRank(N, world, rank, dealt)
        e._since_bin, e.rebin_every = 1, 1
        ex = TorchHaloExchange(margin=1.0)
        ex.exchange(e, replan=True)
        for step in range(3):                    # the same plan, new positions every sub-step
            e.set_own(dealt + 0.1 * (step + 1))
            ex.exchange(e, replan=False)
        assert ex.plans_made == 1 and ex.bytes_per_substep > 0
        now = dealt + 0.3
        first = [_FakeRank(N, world, r, now) for r in range(world)]
        mine = now[first[rank].first:first[rank].first + e.NUM_DRONES]
        have = {tuple(np.float32(v)) for v in e.pos4[:, :3].numpy() if np.isfinite(v).all()}
        for r in range(world):
            if r == rank:
                continue
            theirs = now[first[r].first:first[r].first + first[r].NUM_DRONES]
            d = np.hypot(theirs[:, None, 0] - mine[None, :, 0], theirs[:, None, 1] - mine[None, :, 1]).min(axis=1)
            assert all(tuple(np.float32(v)) in have for v in theirs[d < 10.0]), (rank, r)
            meta = e.pos4.view(world, e.slab, 4)[r, e.per:]
            assert (meta[:, 1:] == 7.0 + r).all()
        ex.check(e)                              # (collective; nobody outran the margin)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_over_torch_distributed_gloo(world):
    """`TorchHaloExchange` between real processes (gloo, CPU tensors): the two tiny all-gathers of the plan and the batched
    point-to-point blocks; every rank ends up with the current positions of the drones it needs and everybody's meta rows."""
    import torch.multiprocessing as mp
    mp.spawn(_halo_gloo_worker, args=(world, 29600 + os.getpid() % 300 + world, 1500), nprocs=world, join=True)


def test_documents_name_every_entry_and_the_current_abi():
    """DESIGN.md's boundary table and INTEGRATION.md's mapping name every function include/gpd.h declares, and quote the ABI
    version the header defines (a stale entry list was a round-3 finding)."""
    hdr = open(os.path.join(REPO, "include", "gpd.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(gpd_\w+)\s*\(", hdr, flags=re.M))
    abi = re.search(r"#define GPD_ABI_VERSION (\d+)", hdr).group(1)
    for doc in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(REPO, doc)).read()
        assert not [n for n in sorted(declared) if n not in text], doc
    assert f"C-ABI (v{abi})" in open(os.path.join(REPO, "DESIGN.md")).read()
    assert f"gpd_abi_version() == {abi}" in open(os.path.join(REPO, "INTEGRATION.md")).read()


def test_parse_urdf_parameters_keeps_the_reference_tuple():
    """`_parseURDFParameters` (envs/BaseAviary.py:985-1017): 17 values in the reference's order; needs no device."""
    import types
    from gym_pybullet_drones_amd.envs.BaseAviary import BaseAviary
    from gym_pybullet_drones_amd.utils.enums import DroneModel
    t = BaseAviary._parseURDFParameters(types.SimpleNamespace(DRONE_MODEL=DroneModel.CF2X))
    assert len(t) == 17
    M, L, T2W, J, J_INV, KF, KM, CH, CR, CZ, VMAX, GND, PROP, DRAG, DW1, DW2, DW3 = t
    assert (M, L, T2W, KF, KM) == (0.027, 0.0397, 2.25, 3.16e-10, 7.94e-12)
    assert np.allclose(np.diag(J), [1.4e-5, 1.4e-5, 2.17e-5]) and np.allclose(J @ J_INV, np.eye(3))
    assert (CH, CR, VMAX, GND, PROP) == (0.025, 0.06, 30.0, 11.36859, 2.31348e-2) and CZ == 0.0
    assert np.allclose(DRAG, [9.1785e-7, 9.1785e-7, 10.311e-7]) and (DW1, DW2, DW3) == (2267.18, 0.16, -0.11)


def test_kin_planes_round_trip_on_the_host():
    """ABI 9: `GpdState.kin` is four planes (P: pos xyz + body rate x | Q: quat xyzw | V: vel xyz + body rate y | W: body rate z);
    `engine.kin_rows_from_planes` is the inverse the tests and the C host use to get the logical [13][ld] rows back."""
    from gym_pybullet_drones_amd.engine import kin_rows_from_planes
    ld = 8
    rows = np.arange(13 * ld, dtype=np.float32).reshape(13, ld)
    P = np.stack([rows[0], rows[1], rows[2], rows[10]], axis=1)
    Q = np.stack([rows[3], rows[4], rows[5], rows[6]], axis=1)
    V = np.stack([rows[7], rows[8], rows[9], rows[11]], axis=1)
    store = np.concatenate([P.ravel(), Q.ravel(), V.ravel(), rows[12]])
    assert store.size == 13 * ld
    np.testing.assert_array_equal(kin_rows_from_planes(store, ld), rows)
    import torch
    np.testing.assert_array_equal(kin_rows_from_planes(torch.as_tensor(store), ld).numpy(), rows)


def test_state_vector_aviaries_spaces_and_stock_action_code():
    """`CtrlAviary` / `VelocityAviary` (reference envs/CtrlAviary.py:83-144, envs/VelocityAviary.py:87-127): the spaces the
    reference advertises, and the rule that only a class whose `_preprocessAction` is still the stock one lets the kernel map the
    action (a subclass that overrides the hook gets its own mapping called instead).  No device: the hooks are called on bare objects."""
    from gym_pybullet_drones_amd.envs import CtrlAviary, VelocityAviary
    from gym_pybullet_drones_amd.utils.enums import ACT_RAW_RPM, ActionType

    def bare(cls, n=3, max_rpm=21702.64):
        o = object.__new__(cls)
        o.NUM_DRONES, o.MAX_RPM = n, max_rpm
        return o
    c, v = bare(CtrlAviary), bare(VelocityAviary)
    for env in (c, v):
        box = env._observationSpace()
        assert box.shape == (3, 20) and box.dtype == np.float32
        inf, pi = np.inf, np.pi
        np.testing.assert_array_equal(box.low[1], np.float32([-inf, -inf, 0, -1, -1, -1, -1, -pi, -pi, -pi] + [-inf] * 6 + [0] * 4))
        np.testing.assert_array_equal(box.high[2], np.float32([inf] * 3 + [1] * 4 + [pi] * 3 + [inf] * 6 + [env.MAX_RPM] * 4))
        assert env._computeReward() == -1 and env._computeTerminated() is False and env._computeTruncated() is False
        assert env._computeInfo() == {"answer": 42}
    a = c._actionSpace()
    assert a.shape == (3, 4) and (a.low == 0).all() and np.allclose(a.high, c.MAX_RPM)
    a = v._actionSpace()
    np.testing.assert_array_equal(a.low, np.float32([[-1, -1, -1, 0]] * 3))
    np.testing.assert_array_equal(a.high, np.ones((3, 4), np.float32))
    np.testing.assert_array_equal(c._preprocessAction(np.array([[-5.0, 1e9, 3.0, 4.0]] * 3)), [[0, c.MAX_RPM, 3, 4]] * 3)
    assert c._fusedActionCode() == ACT_RAW_RPM and v._fusedActionCode() == ActionType.VEL.code

    class Plain(CtrlAviary):
        pass

    class Mine(CtrlAviary):
        def _preprocessAction(self, action):
            return np.asarray(action) * 2
    assert bare(Plain)._fusedActionCode() == ACT_RAW_RPM and bare(Mine)._fusedActionCode() is None
