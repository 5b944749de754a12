"""The multi-rank paths on ONE GPU box: bench.py under torch.distributed.run with two ranks (gloo rendezvous, both ranks
on device 0 -- the test hook; RCCL itself refuses two ranks on one device), and the C-ABI's RCCL all-gather with a
one-rank communicator, alone and captured in one hipGraph together with the step that produces the shard."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO

pytestmark = pytest.mark.gpu


def _bench(args, world, port, timeout=900, script="bench.py"):
    env = dict(os.environ, GPD_DIST_BACKEND="gloo", GPD_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, script), "--gpus", str(world)] + args
    res = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def _bench_self(args, timeout=1200, gpus=2, script="bench.py", expect_rc=0, **extra_env):
    """`python bench.py --gpus N ...` the way the driver types it -- NO torch.distributed.run around it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GPD_DIST_BACKEND="gloo", GPD_BENCH_SINGLE_DEVICE="1")
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(REPO, script), "--gpus", str(gpus)] + args, cwd=REPO, env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert (res.returncode == expect_rc) if expect_rc is not None else (res.returncode != 0), res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]          # ONE JSON line, whatever ran
    return json.loads(lines[0])


def test_bench_gpus_2_launches_its_two_ranks_itself(gpu_device):
    j = _bench_self(["--steps", "20", "--warmup", "5", "--min-time", "0.05", "--no-cpu-baseline", "--no-suite", "--no-hbm-leg"])
    assert j["n_gpus"] == 2 and j["config"]["total_drones"] == 2 * 65536 and j["steps"] == 20 and j["warmup"] == 5
    assert j["config"]["ranks_in_process_group"] == 2 and "self-launch" in j["config"]["launcher"]
    pg = j["per_gpu"]["values"]
    assert len(pg) == 2 and all(v > 1e8 for v in pg) and j["value"] <= sum(pg) * 1.0001
    assert j["metric"].startswith("env steps/sec (whole node), HoverAviary N=65536")


def test_bench_gpus_2_default_run_carries_the_suite_and_the_hbm_leg(gpu_device):
    """What the driver's SCALE run gets from one command: the headline with the topology the ranks found, the HBM-saturating leg
    beside it, and BASELINE configs 4 and 5 under `suite`."""
    j = _bench_self(["--steps", "20", "--warmup", "5", "--min-time", "0.05", "--no-cpu-baseline", "--hbm-leg-time", "0.3", "--hbm-leg-reallocations", "1"], timeout=2400)
    assert j["n_gpus"] == 2 and j["config"]["workload"] == "hover65536_240hz"
    h = j["hbm_saturating"]
    assert h["workload"] == "hover4m_240hz" and h["bytes_per_launch"] > 4 * 256 * 2 ** 20 and 0.2 < h["frac"] < 1.1
    su = j["suite"]
    assert "error" not in su and not [n for n, r in su.items() if "error" in r], su
    ag = su["hover65536x8_allgather"]
    assert ag["config"]["obs_allgather"] is True and ag["without_allgather"]["value"] >= ag["value"] * 0.5 and ag["n_gpus"] == 2
    mh = su["multihover2x16384x8"]
    assert mh["config"]["total_drones"] == 2 * 2 * 16384 and mh["value"] > 1e8
    # the line says what the ranks found before anything was timed: the rank -> device map, the group's own count
    topo = j["config"]["topology"]
    assert topo["world_size"] == 2 and topo["ranks_in_process_group"] == 2 and [e["rank"] for e in topo["rank_device_map"]] == [0, 1]
    assert all(e["device"] == 0 and e["name"] for e in topo["rank_device_map"]) and topo["backend"] == "gloo"


def test_bench_gpus_8_rehearsal_of_the_driver_command(gpu_device):
    """The command the driver's SCALE run types at N = 8 -- `python bench.py --gpus 8 --steps K --warmup W`, no launcher around it --
    rehearsed end to end on ONE device: eight self-launched ranks (gloo rendezvous, every rank on device 0, 1/8 of every workload's
    aviaries: `GPD_BENCH_E_DIV=8`).  What is shown is plumbing, not performance (VERDICT r04 "next" #4; no 8-GPU node is available to
    this build: no hardware scaling curve exists): eight per-GPU values, eight ranks in the process group, the HBM leg, and the suite
    with BASELINE config 4 (with AND without the all-gather) and config 5.  (The one-world workload split over eight ranks:
    `test_bench_extra_eight_ranks_share_the_1m_world`.)"""
    j = _bench_self(["--steps", "20", "--warmup", "5", "--min-time", "0.02", "--hbm-leg-time", "0.05", "--hbm-leg-reallocations", "1", "--no-cpu-baseline", "--suite-timeout", "900"],
                    timeout=2400, gpus=8, GPD_BENCH_E_DIV="8")
    assert j["n_gpus"] == 8 and len(j["per_gpu"]["values"]) == 8 and j["config"]["ranks_in_process_group"] == 8
    assert j["config"]["rehearsal_divisor"] == 8 and j["config"]["total_drones"] == 8 * 8192 and "self-launch" in j["config"]["launcher"]
    assert j["scaling"] == "weak" and j["value"] <= sum(j["per_gpu"]["values"]) * 1.0001 and "error" not in j["hbm_saturating"]
    su = j["suite"]
    assert "error" not in su and not [n for n, r in su.items() if "error" in r], su
    ag = su["hover65536x8_allgather"]
    assert ag["n_gpus"] == 8 and ag["config"]["obs_allgather"] is True and ag["without_allgather"]["value"] > 0 and len(ag["per_gpu"]["values"]) == 8
    assert su["multihover2x16384x8"]["n_gpus"] == 8 and su["multihover2x16384x8"]["config"]["total_drones"] == 8 * 2 * 2048
    assert len(j["config"]["topology"]["rank_device_map"]) == 8


def test_bench_extra_eight_ranks_share_the_1m_world(gpu_device):
    """bench_extra.py's strong-scaling workload self-launched with eight ranks on ONE device (1/8 size): the world split over eight
    ranks, positions exchanged by halo, the margin check "ok"."""
    sw = _bench_self(["--workload", "swarm1m_ext_240hz", "--mode", "graph", "--steps", "20", "--warmup", "5", "--min-time", "0.02", "--no-cpu-baseline",
                      "--no-parity"], timeout=2400, gpus=8, script="bench_extra.py", GPD_BENCH_E_DIV="8")
    assert sw["n_gpus"] == 8 and sw["scaling"] == "strong" and sw["config"]["swarm"]["ranks"] == 8 and sw["config"]["swarm"]["exchange"] == "halo"
    assert sw["config"]["swarm"]["halo_margin_check"] == "ok" and sw["config"]["rehearsal_divisor"] == 8


def test_dry_run_topology_and_two_rccl_ranks_on_one_device_fail_fast(gpu_device):
    """VERDICT r05 #7.  (a) `--dry-run-topology` brings the job up, all-gathers 12 floats per rank, prints the topology block and exits
    0 in seconds (gloo test hook: two ranks on device 0).  (b) The same two ranks on ONE device under the REAL backend ("nccl" =
    RCCL, which refuses duplicate GPUs): a readable line -- the rank -> device map, what is wrong -- and a non-zero exit code within the
    minute, not a hang until somebody's timeout."""
    import time
    t0 = time.perf_counter()
    j = _bench_self(["--dry-run-topology"], timeout=300)
    assert j["dry_run_topology"] and j["ok"] and j["n_gpus"] == 2 and len(j["topology"]["rank_device_map"]) == 2
    assert time.perf_counter() - t0 < 120
    t0 = time.perf_counter()
    j = _bench_self(["--dry-run-topology", "--init-timeout", "60"], timeout=300, expect_rc=None, GPD_DIST_BACKEND="nccl")
    assert time.perf_counter() - t0 < 150
    assert j["value"] is None and "error" in j and j["n_gpus"] == 2
    assert "same device" in j["error"] and len(j["topology"]["rank_device_map"]) == 2, j
    assert {e["device"] for e in j["topology"]["rank_device_map"]} == {0}


def test_bench_watchdog_prints_the_headline_when_the_suite_hangs(gpu_device):
    """A collective of a workload that has never met the node hangs: staged by tests/helpers/bench_with_a_hanging_suite.py (bench.py with
    a suite whose first workload sleeps forever on every rank).  The watchdog prints the headline line it holds, with the suite's entry
    saying what happened, and every rank leaves: return code 0, ONE JSON line."""
    j = _bench_self(["--steps", "20", "--warmup", "5", "--min-time", "0.02", "--no-cpu-baseline", "--no-hbm-leg", "--suite-timeout", "25"],
                    timeout=900, gpus=4, script=os.path.join("tests", "helpers", "bench_with_a_hanging_suite.py"), GPD_BENCH_E_DIV="8")
    assert j["n_gpus"] == 4 and j["value"] > 0 and len(j["per_gpu"]["values"]) == 4
    assert "not finished after 25 s" in j["suite"]["error"]


def test_bench_two_ranks_with_obs_allgather(gpu_device):
    j = _bench(["--steps", "20", "--warmup", "5", "--allgather", "--min-time", "0.02", "--no-second-leg", "--no-cpu-baseline", "--no-hbm-leg", "--no-parity"], 2, 29541)
    assert j["n_gpus"] == 2 and j["config"]["total_drones"] == 2 * 65536 and j["config"]["obs_allgather"] is True
    assert j["steps"] == 20 and j["warmup"] == 5 and j["timed_steps"] == 20 * j["repeats"] and j["scaling"] == "weak"
    # (the gloo test hook stages every 20-step observation block, 63 MB per rank, through host memory: slow by design)
    assert j["metric"].startswith("env steps/sec") and j["unit"] == "drone-steps/s" and j["value"] > 1e6
    assert j["ms_per_step"] == pytest.approx(j["timed_region_ms"] / j["timed_steps"])
    assert j["roofline"]["bound"] == "hbm" and 0 < j["roofline"]["frac"] < 1.5 and "cpu_baseline" not in j


def test_bench_two_ranks_named_config5_workload(gpu_device):
    """BASELINE config 5 per GPU (16 384 two-drone aviaries, pairwise downwash) through the same two-rank path, both legs."""
    j = _bench(["--steps", "20", "--warmup", "5", "--workload", "multihover2x16384x8", "--min-time", "0.05", "--no-cpu-baseline"], 2, 29542)
    assert j["n_gpus"] == 2 and j["config"]["total_drones"] == 2 * 2 * 16384 and j["config"]["drones_per_env"] == 2
    assert "DW" in j["config"]["physics"] and j["steps"] == 20 and j["warmup"] == 5
    assert j["one_launch_per_step"]["value"] > 0 and j["one_launch_per_step"]["roofline"]["kernel"] == "gpd_step_kernel"


def test_bench_two_ranks_share_one_swarm_world(gpu_device):
    """ONE world of 65 536 drones shared by two ranks (both on device 0, positions exchanged through torch.distributed / gloo --
    the test hook; on a node RCCL carries them): rank r steps its block of drones, all-gathers, evaluates its own forces."""
    j = _bench(["--workload", "swarm65536_ext_240hz", "--mode", "eager", "--steps", "24", "--warmup", "4", "--min-time", "0.02",
                "--no-cpu-baseline"], 2, 29543, script="bench_extra.py")
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["total_drones"] == 65536
    sw = j["config"]["swarm"]
    assert sw["ranks"] == 2 and sw["total_drones"] == 65536 and "torch.distributed" in sw["note"]
    assert j["value"] > 1e5 and j["steps"] == 24


def test_native_allgather_one_rank_and_inside_a_graph(gpu_device):
    from gym_pybullet_drones_amd import dist as gdist
    from gym_pybullet_drones_amd.envs import VectorHoverAviary
    from gym_pybullet_drones_amd.utils.enums import ActionType
    E = 4096
    env = VectorHoverAviary(E, act=ActionType.RPM, ctrl_freq=240, device=gpu_device)
    ag = gdist.NativeObsAllGather(env.core.N, 12, device=gpu_device)
    rng = np.random.default_rng(0)
    acts = torch.as_tensor(rng.uniform(-1, 1, size=(4, E, 1, 4)).astype(np.float32), device=gpu_device)
    env.step(acts[0])
    full = ag(env.core.obs12)
    torch.cuda.synchronize()
    assert full.shape == (E, 12) and torch.equal(full, env.core.obs12)
    # step + gather captured together: one graph launch = env.step() of every aviary + the collective, no torch op inside
    stream = torch.cuda.Stream(gpu_device)
    stream.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for k in range(1, 4):
                env.step(acts[k])
                ag(env.core.obs12)
    torch.cuda.current_stream(gpu_device).wait_stream(stream)
    ref = VectorHoverAviary(E, act=ActionType.RPM, ctrl_freq=240, device=gpu_device)
    for k in range(4):
        ref.step(acts[k])
    ag.full.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(ag.full, ref.core.obs12)
    with pytest.raises(ValueError):
        ag(env.core.obs12[:10])
    ag.close()


def _swarm_rank(rank, world, port, out_dir, N, halo=False):
    """one PROCESS = one rank of a shared swarm world (both on device 0; positions exchanged through torch.distributed / gloo)"""
    import torch.distributed as dist
    from gym_pybullet_drones_amd.envs import SwarmAviary, TorchHaloExchange, TorchSlabExchange
    from gym_pybullet_drones_amd.utils.enums import Physics
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        d = np.load(os.path.join(out_dir, "scene.npz"))
        env = SwarmAviary(N, initial_xyzs=d["xyz"], initial_rpys=d["rpy"], physics=Physics.PYB_GND_DRAG_DW, pyb_freq=240, ctrl_freq=120,
                          device=dev, world_size=world, rank=rank, exchange=TorchHaloExchange(margin=1.0) if halo else TorchSlabExchange(),
                          rebin_every=4)
        ids = torch.as_tensor(env.GLOBAL_IDS, dtype=torch.long, device=dev)
        rpm = torch.as_tensor(d["rpm"], device=dev)
        vec, _ = env.reset()
        for k in range(rpm.shape[0]):
            vec, *_ = env.step(rpm[k][ids])
        torch.cuda.synchronize()
        everything = env.all_positions()            # (collective: the whole world's positions on every rank, whatever the exchange)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), ids=env.GLOBAL_IDS, vec=vec.cpu().numpy(), force=env.dw_force[:env.NUM_DRONES].cpu().numpy(),
                 everything=everything.cpu().numpy(), sent=np.array(getattr(env.exchange, "bytes_per_substep", 0) or 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("halo", [False, True])
def test_two_processes_share_one_swarm_world_bitwise(gpu_device, tmp_path, halo):
    """The multi-PROCESS form of the sharded world (what a node runs, one process per GPU): two processes, each holding its
    stripe of the drones, exchanging position slabs through torch.distributed every sub-step.  State vectors and forces of all
    drones after 10 control steps, bit for bit those of one process holding the whole world."""
    import torch.multiprocessing as mp
    from gym_pybullet_drones_amd.envs import SwarmAviary
    from gym_pybullet_drones_amd.utils.enums import Physics
    rng = np.random.default_rng(9)
    N = 1300
    sites = np.array([(x, y) for x in np.arange(-28, 29, 4.0) for y in np.arange(-18, 19, 4.0)])
    idx = rng.permutation(len(sites) * 12)[:N]
    xyz = np.concatenate([sites[idx % len(sites)] + rng.uniform(-0.3, 0.3, size=(N, 2)), (1.0 + idx // len(sites))[:, None]], axis=1)
    rpy = rng.uniform(-0.05, 0.05, size=(N, 3))
    one = SwarmAviary(N, initial_xyzs=xyz, initial_rpys=rpy, physics=Physics.PYB_GND_DRAG_DW, pyb_freq=240, ctrl_freq=120, device=gpu_device)
    rpm = (one.HOVER_RPM * (1 + 0.03 * rng.uniform(-1, 1, size=(10, N, 4)))).astype(np.float32)
    np.savez(os.path.join(str(tmp_path), "scene.npz"), xyz=xyz, rpy=rpy, rpm=rpm)
    mp.spawn(_swarm_rank, args=(2, 29551 + int(halo), str(tmp_path), N, halo), nprocs=2, join=True)
    v1, _ = one.reset()
    for k in range(10):
        v1, *_ = one.step(torch.as_tensor(rpm[k], device=gpu_device))
    v1, f1 = v1.cpu().numpy(), one.dw_force[:N].cpu().numpy()
    seen = np.zeros(N, dtype=bool)
    for r in range(2):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(d["vec"], v1[d["ids"]]) and np.array_equal(d["force"], f1[d["ids"]]), r
        assert np.array_equal(d["everything"], v1[:, :3])                 # SwarmAviary.all_positions(): the caller's drone order
        assert (int(d["sent"]) > 0) == halo
        seen[d["ids"]] = True
    assert seen.all() and np.abs(f1).max() > 1e-3


def test_native_p2p_group_one_rank_self_send_and_inside_a_graph(gpu_device):
    """`gpd_p2p_group` (the halo exchange's transport: ncclSend / ncclRecv of RCCL inside one group) on the REAL library with the
    one-rank communicator a single GPU allows: the rank sends two blocks of different sizes to itself and receives them into two
    other buffers -- eagerly, and captured in a hipGraph behind the kernel that produces the blocks (what a captured sub-step of a
    shared world replays).  The multi-rank form of the same call runs against the stub in tests/test_host_logic.py; RCCL with more
    than one rank has never run (no multi-GPU node was available to this build)."""
    import ctypes
    from gym_pybullet_drones_amd import _native
    from gym_pybullet_drones_amd import dist as gdist
    nc = gdist.NativeComm.shared(device=gpu_device)
    assert nc.world == 1 and nc.ranks_seen == 1
    a = torch.arange(1000, dtype=torch.float32, device=gpu_device)
    b = torch.arange(37, dtype=torch.float32, device=gpu_device) * -2.0
    ra, rb = torch.zeros_like(a), torch.zeros_like(b)
    mk = lambda ops: (_native.GpdP2P * len(ops))(*[_native.GpdP2P(peer=0, ptr=t.data_ptr(), count=t.numel()) for t in ops])  # noqa: E731
    S, R = mk([a, b]), mk([ra, rb])
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream(gpu_device).cuda_stream)  # noqa: E731
    with torch.cuda.device(gpu_device):
        _native.check(nc.lib.gpd_p2p_group(nc.comm, S, 2, R, 2, st()), "gpd_p2p_group")
    torch.cuda.synchronize()
    assert torch.equal(ra, a) and torch.equal(rb, b)
    # captured: producer kernel + the grouped exchange, replayed with new contents
    stream = torch.cuda.Stream(gpu_device)
    stream.wait_stream(torch.cuda.current_stream(gpu_device))
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            a.mul_(3.0)
            b.add_(1.0)
            with torch.cuda.device(gpu_device):
                _native.check(nc.lib.gpd_p2p_group(nc.comm, S, 2, R, 2, ctypes.c_void_p(stream.cuda_stream)), "gpd_p2p_group (captured)")
    torch.cuda.current_stream(gpu_device).wait_stream(stream)
    ra.zero_(); rb.zero_()
    a0, b0 = a.clone(), b.clone()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(a, a0 * 9.0) and torch.equal(ra, a) and torch.equal(rb, b0 + 2.0)
    # argument errors come back as codes
    assert nc.lib.gpd_p2p_group(nc.comm, None, 1, R, 2, st()) == _native.GPD_EINVAL
