"""The headline kernel's instruction stream, checked on the cross-compiled ISA (no GPU needed).

DESIGN.md §3 argues from issue slots: at one wave per SIMD every instruction of the rollout step costs ~5.4 cycles and a
taken branch ~60, and the wait for the prefetched action rows has to be an exact `vmcnt(18)`.  None of that is visible to
a numerical test -- a compiler update or an innocent-looking edit can put a `vmcnt(0)` or a skip-branch back into the
loop and cost 10-40 % without changing a bit of the results.  This test pins the properties on the assembly hipcc
produces for gfx950."""
import os
import re
import subprocess
import tempfile
from collections import Counter

import pytest

from conftest import REPO

HEADLINE = "gpd_rollout1_kernelILb0ELb0ELi4ELi0ELb1ELb0ELb0ELb0ELi0ELin1ELb0EE"  # <PID=0, EXT=0, AW=4, ACT=RPM, S1=1, MULTI=0, NT_OBS=0, RING=0, DC=0, FL=-1, HI=0>


POLICY = "gpd_rollout_policy_kernelILb0ELi4ELi0ELi5ELb0E"  # <PID=0, AW=4, ACT=RPM, NK1=5 (72-float rows), tanh>


def _asm(unit, extra):
    from gym_pybullet_drones_amd import _native
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    flags = [f for f in _native.COMMON_FLAGS if f != "-fPIC"] + extra
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "unit.s")
        subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(REPO, "include"),
                                          os.path.join(REPO, "gym_pybullet_drones_amd", "csrc", unit), "-o", out],
                       check=True, capture_output=True)
        return open(out).read().split("\n")


_UNIT_ASM = {}


def _unit_asm(index):
    """The assembly of unit `index`; the first request compiles all four units side by side (the largest takes ~2 min on its own)."""
    from concurrent.futures import ThreadPoolExecutor
    from gym_pybullet_drones_amd import _native
    if not _UNIT_ASM:
        if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            pytest.skip("no hipcc")
        with ThreadPoolExecutor(len(_native.UNITS)) as pool:
            for i, lines in enumerate(pool.map(lambda u: _asm(*u), _native.UNITS)):
                _UNIT_ASM[i] = lines
    return _UNIT_ASM[index]


@pytest.fixture(scope="module")
def gpd_asm():
    return _unit_asm(0)


@pytest.fixture(scope="module")
def policy_asm():
    return _unit_asm(1)


@pytest.fixture(scope="module")
def swarm_asm():
    from gym_pybullet_drones_amd import _native
    assert _native.UNITS[2][0] == "swarm.hip"
    return _unit_asm(2)


@pytest.fixture(scope="module")
def abi_asm():
    from gym_pybullet_drones_amd import _native
    assert _native.UNITS[3][0] == "abi.hip"
    return _unit_asm(3)


def _kernel(lines, name):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + name + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end], "\n".join(lines[end:end + 80])


@pytest.fixture(scope="module")
def headline_isa(gpd_asm):
    return _kernel(gpd_asm, HEADLINE)


def _ops(lines):
    for l in lines:
        s = l.strip()
        if s and not s.startswith((";", ".")) and not s.endswith(":"):
            yield s.split()[0], s


def test_rollout_step_is_straight_line_and_within_its_slot_budget(headline_isa):
    body, meta = headline_isa
    assert re.search(r"ScratchSize: 0\b", meta), "the kernel spills to scratch"
    # the loop holds three copies of the step; a copy starts at the first of its three ds_write_b128 (the row patch)
    writes = [i for i, l in enumerate(body) if "ds_write_b128" in l]
    assert len(writes) == 9
    hot = list(_ops(body[writes[3]:writes[6]]))            # the second copy, row write to row write (the rare blocks are
                                                           # laid out behind the loop, not inside this range)
    kinds = Counter("branch" if op.startswith(("s_cbranch", "s_branch")) else "nop" if op == "s_nop" else
                    "packed" if op.startswith("v_pk_") else "other" for op, _ in hot)
    assert len(hot) <= 285, f"{len(hot)} issue slots in a rollout step (272 when this test was written)"
    assert kinds["branch"] <= 4 and kinds["nop"] <= 3 and kinds["packed"] >= 40, kinds
    # the branches of the hot path are the rare-case tests and the step count: none of them is an exec-mask skip
    assert not [s for op, s in hot if op in ("s_cbranch_execz", "s_cbranch_execnz")]


def test_action_rows_are_claimed_with_an_exact_count(headline_isa):
    body, _ = headline_isa
    writes = [i for i, l in enumerate(body) if "ds_write_b128" in l]
    loop = body[writes[0]:]
    waits = [s for op, s in _ops(loop) if op == "s_waitcnt" and "vmcnt" in s]
    assert any("vmcnt(18)" in s for s in waits), waits      # everything but this iteration's 3 x 6 stores
    first = next(i for i, l in enumerate(loop) if "vmcnt(18)" in l)
    assert not [l for l in loop[:first] if "s_waitcnt" in l and "vmcnt" in l], "a memory wait inside the three steps"
    # stores of the loop use <uniform base in SGPRs> + <32-bit lane offset>
    stores = [s for op, s in _ops(loop[:first]) if op.startswith("global_store")]
    assert len(stores) == 18 and all(re.search(r"s\[\d+:\d+\]", s) for s in stores), stores[:3]


def test_policy_kernel_runs_its_layers_on_the_matrix_cores(policy_asm):
    """gpd_rollout_policy for the reference's 72-float rows: 44 operand tiles x 3 bf16 hi/lo products = 132 MFMAs per step, both
    column tiles' layer-1 operands from lane-half swaps (no LDS round trip for activations), no scratch."""
    body, meta = _kernel(policy_asm, POLICY)
    assert re.search(r"ScratchSize: 0\b", meta), "the policy kernel spills to scratch"
    ops = Counter(op for op, _ in _ops(body))
    assert ops["v_mfma_f32_32x32x16_bf16"] == 132, ops["v_mfma_f32_32x32x16_bf16"]
    assert ops["v_permlane32_swap_b32_e32"] >= 40 and ops["v_exp_f32_e32"] == 128 and ops["v_cvt_pk_bf16_f32"] >= 140
    assert not [op for op in ops if op.startswith("scratch_")]


TORN = """_Z4tornv:
; %bb.0:
	s_load_dwordx8 s[48:55], s[0:1], 0x19c
	s_load_dwordx4 s[8:11], s[0:1], 0x228
	s_waitcnt lgkmcnt(0)
	v_cmp_gt_u32_e32 vcc, s8, v0
	s_and_saveexec_b64 s[22:23], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v1, s9
.LBB0_2:
	s_or_b64 exec, exec, s[22:23]
	s_load_dwordx16 s[{lo}:{hi}], s[0:1], 0x18
	s_waitcnt lgkmcnt(0)
	v_writelane_b32 v164, s48, 29
	s_nop 1
	v_writelane_b32 v164, s49, 30
	v_writelane_b32 v164, s50, 31
	v_writelane_b32 v164, s51, 32
	v_writelane_b32 v164, s52, 33
	v_writelane_b32 v164, s53, 34
	v_writelane_b32 v164, s54, 35
	v_writelane_b32 v164, s55, 36
	v_mul_f32_e32 v2, s{hi}, v1
	v_readlane_b32 s12, v164, 29
	s_cmp_lg_u32 s12, 0
	s_cbranch_scc1 .LBB0_2
; %bb.3:
	s_endpgm
.Lfunc_end0:
"""


def test_spill_checker_recognises_the_miscompile_it_was_written_for():
    """The shape hipcc (ROCm 7.2) produced for gpd_rollout_policy_kernel<PID, 4, VEL, 1, tanh> under the default scheduler
    (DESIGN.md section 3.7): GpdStepCfg loaded into s[48:55], GpdParams loaded over s[48:51] one block later, s[48:55] spilled
    after that as if intact.  The same code with the second load elsewhere is clean."""
    import isa_spill_check as chk
    (name, body), = chk.kernels(TORN.format(lo=36, hi=51))
    found = chk.torn_spills(body)
    assert len(found) == 1 and found[0][2] == [48, 49, 50, 51] and found[0][3]["lanes"] == list(range(29, 37))
    (name, body), = chk.kernels(TORN.format(lo=56, hi=71))
    assert chk.torn_spills(body) == []


def test_kernarg_exemption_needs_the_kernarg_pointer_as_base():
    """The exemption for a host-only argument word covers loads off s[0:1] or an s_mov_b64 copy of it, nothing else: the torn shape
    at the same immediate offset from another base register is still a finding."""
    import isa_spill_check as chk
    torn = TORN.format(lo=36, hi=51)
    (name, body), = chk.kernels(torn)
    first = chk.torn_spills(body)[0]
    imm = int(re.search(r",\s*(0x[0-9a-fA-F]+|\d+)\s*$", first[1]).group(1), 0)
    dst_lo = min(chk.sregs(chk.split_ops(first[1].split(None, 1)[1])[0]))
    offs = {imm + 4 * (r - dst_lo) for r in first[2]}
    base = re.match(r"\S+\s+[^,]+,\s*(s\[\d+:\d+\])", first[1]).group(1)
    exempt = chk.torn_spills(body, unused_kernarg_offsets=offs)
    if base == "s[0:1]":
        assert exempt == []
        moved = [l.replace(first[1], first[1].replace("s[0:1]", "s[90:91]")) for l in body]
        assert len(chk.torn_spills(moved, unused_kernarg_offsets=offs)) == 1
    else:
        assert len(exempt) == 1


def test_no_kernel_saves_a_half_overwritten_argument_tuple(gpd_asm, policy_asm, swarm_asm, abi_asm):
    """Every kernel of the four units, every instantiation: no scalar-load destination tuple is spilled after part of it was
    overwritten (backward SGPR liveness over the kernel's control-flow graph, tests/isa_spill_check.py).  All of these kernels
    run with their 106 SGPRs full and spill kernel arguments to VGPR lanes, so the register allocator's handling of exactly
    this is load-bearing for every GpdParams / GpdStepCfg field they read."""
    import isa_spill_check as chk
    n = 0
    for asm in (gpd_asm, policy_asm, swarm_asm, abi_asm):
        text = "\n".join(asm)
        preload = chk.preload_lengths(text)
        for name, body in chk.kernels(text):
            n += 1
            # GpdParams.pid_kf (word 36 of the struct) is read by the HOST only (the "no controller for this airframe" check): its three
            # neighbours hover_resid / km_over_kf / pid_gravity are fetched as an x4 whose fourth register is reused at once.  The struct
            # sits at byte 0 of the argument block -- or at byte 56, behind the fourteen preloaded dwords, in the kernels whose
            # descriptor says so (gpd_step_kernel / gpd_rollout1_kernel / the one-world kernels); only THAT word of THAT kernel, and
            # only in a load off the kernel-argument pointer, is exempt
            unused = {56 + 36 * 4} if preload.get(name, 0) == 14 else {36 * 4}
            found = chk.torn_spills(body, unused_kernarg_offsets=unused)
            assert not found, (name, [(l, dead, run["lanes"]) for _, l, dead, run in found])
    assert n >= 180, n          # (every kernel of the four units, the preloaded-argument ones included)


def test_no_kernel_needs_scratch_memory_or_reads_the_dispatch_packet(gpd_asm, policy_asm, swarm_asm, abi_asm):
    """Two ways a kernel of this library got 1.6 - 10 us slower in round 5 without changing a bit of its results, both invisible to
    every numerical test (profiles/r05_ab_step_kernel_round1.log, r05_bench_*): (1) a struct the kernel builds for itself stays an
    alloca (a select between two of its members became a load from a selected address) -> scratch memory, and a launch that needs
    scratch pays for its set-up; (2) twelve field reads combined into vector loads of a struct -> the alloca is promoted to LDS and
    indexed by a flat thread id computed from the DISPATCH PACKET, which lives in host memory: one PCIe read per wave.  No kernel of
    the four units may use either."""
    n = 0
    for unit, asm in (("step_rollout", gpd_asm), ("policy", policy_asm), ("swarm", swarm_asm), ("abi", abi_asm)):
        text = "\n".join(asm)
        scratch = re.findall(r"; ScratchSize: (\d+)", text)
        n += len(scratch)
        assert scratch and all(v == "0" for v in scratch), (unit, [v for v in scratch if v != "0"])
        assert ".amdhsa_user_sgpr_dispatch_ptr 1" not in text, unit
        assert not re.search(r"^\s*scratch_(load|store)", text, flags=re.M), unit
    assert n >= 150, n


@pytest.mark.parametrize("sched", ["default", "max-ilp"])
def test_dslpid_policy_kernels_in_the_policy_unit_under_both_schedulers(sched):
    """The configuration the miscompile was found in: the DSLPID policy kernels instantiated in the
    POLICY unit (`-DGPD_PID_POLICY_IN_POLICY_TU`), compiled under the default scheduler -- and, for the A/B, under max-ilp.
    Whatever the allocator does with today's source, no kernel of that unit may save a half-overwritten argument tuple; the VEL
    shape that broke at commit 7333ea5 (<PID, 4, VEL, 1, tanh>) must be among the kernels checked.  (Its run-time twin:
    tests/test_gpu_policy.py::test_vel_policy_kernel_is_right_under_both_schedulers.)"""
    import isa_spill_check as chk
    extra = ["-DGPD_PID_POLICY_IN_POLICY_TU"] + (["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if sched == "max-ilp" else [])
    asm = "\n".join(_asm("policy.hip", extra))
    names = []
    for name, body in chk.kernels(asm):
        names.append(name)
        found = chk.torn_spills(body)
        assert not found, (sched, name, [(l, dead, run["lanes"]) for _, l, dead, run in found])
    assert any("gpd_rollout_policy_kernelILb1ELi4ELi2ELi1ELb0E" in n for n in names), names[:5]
    assert sum("gpd_rollout_policy_kernelILb1" in n for n in names) == 12          # 3 action types x {12-float, full row} x {tanh, relu}


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_swarm_force_kernel_tests_pairs_packed_and_evaluates_them_compacted(swarm_asm, mode):
    """dwg_force_kernel<MODE> (MODE 0: no wake lists, 1: the launch after a binning builds them, 2: the
    launches in between replay them): phase A is packed and branch-free (v_pk_* + v_and + v_alignbit, no compare in the sweep;
    the build's margins add scalar min / abs, still no branch), the compaction is a DPP prefix sum into an LDS queue of 16-bit
    pairs, the evaluation gathers its drone by ds_bpermute and adds 64-bit integers with LDS atomics; the build writes its
    batches as 16-bit stores, the replay reads them back; no scratch, at most 128 VGPRs (four workgroups per CU)."""
    body, meta = _kernel(swarm_asm, f"dwg_force_kernelILi{mode}E")
    assert re.search(r"ScratchSize: 0\b", meta)
    assert int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) <= 128
    ops = Counter(op for op, _ in _ops(body))
    sites = 6 if mode == 2 else 2                          # evaluate() call sites: full batches, the flush (, the replay: four batches in flight)
    assert ops["v_alignbit_b32"] >= 16 and ops["v_pk_fma_f32"] >= 8 and ops["v_pk_add_f32"] >= 24
    # (12 more ds_bpermute: the two sums of the common drift in the extra workgroup; the wave reduction of the displacement
    # maxima, on the set-up's critical path, is four DPP steps + four v_readlane: wave_allreduce)
    assert ops["v_mov_b32_dpp"] == 10 and ops["ds_write_b16"] >= 1 and ops["ds_bpermute_b32"] == 3 * sites + 12 and ops["ds_add_u64"] == sites
    assert ops["v_exp_f32_e32"] == sites and not [op for op in ops if op.startswith("scratch_")]
    # build: the wave's batch count is its one 16-bit store; replay: >= 8 entry loads + >= 8 position gathers (prologue + loop), and
    # its own loop holds no LDS tile read between the gathers and the model (the three ds_bpermute per pair are the drone's position)
    assert (ops["global_store_short"] >= 1) == (mode == 1) and (ops["global_load_ushort"] >= 1) == (mode == 2), ops
    if mode == 2:
        assert ops["global_load_dwordx3"] >= 8, ops          # x, y, z of the candidates (prologue + loop)
    # the sweep: from the first alignbit to the last, only packed arithmetic, bit operations, LDS reads and their waits
    first = next(i for i, l in enumerate(body) if "v_alignbit_b32" in l)
    last = max(i for i, l in enumerate(body) if "v_alignbit_b32" in l)
    sweep = Counter(op for op, _ in _ops(body[first:last + 1]))
    assert not [op for op in sweep if op.startswith(("v_cmp", "s_cbranch", "v_rcp", "v_exp"))], sweep


def test_history_rows_are_streamed_out_in_16_byte_pieces(abi_asm):
    """gpd_hist_rows_kernel: whole rows per workgroup, written as ONE contiguous block with non-temporal 16-byte stores."""
    body, meta = _kernel(abi_asm, "gpd_hist_rows_kernel")
    assert re.search(r"ScratchSize: 0\b", meta)
    stores = [s for op, s in _ops(body) if op.startswith("global_store")]
    assert any(s.startswith("global_store_dwordx4") and s.endswith(" nt") for s in stores) and all(s.endswith(" nt") for s in stores), stores


def test_two_drone_aviaries_exchange_through_dpp(gpd_asm):
    """MULTI rollout kernel, RPM, all force terms (BASELINE config 5's kernel): the mate's position (3 values) and its reward /
    distance / out-of-bounds terms (3) arrive by `quad_perm:[1,0,3,2]` moves -- in each of the three copies of the step."""
    body, _ = _kernel(gpd_asm, "gpd_rollout1_kernelILb0ELb1ELi4ELi0ELb0ELb1ELb1ELb0ELi0ELin1ELb0E")      # any aviary size, any flags
    assert sum("quad_perm:[1,0,3,2]" in l for l in body) == 18
    # compiled for pairs, PYB_DW's flags, one sub-step per step (what BASELINE config 5 at 240 Hz runs): the same moves, and nothing of the other sizes
    # is left -- no LDS exchange of positions (the only LDS traffic is the observation patch: 3 writes + 3 reads per step copy)
    body, meta = _kernel(gpd_asm, "gpd_rollout1_kernelILb0ELb1ELi4ELi0ELb1ELb1ELb1ELb0ELi2ELi4ELb0E")
    assert sum("quad_perm:[1,0,3,2]" in l for l in body) == 18 and re.search(r"ScratchSize: 0\b", meta)
    assert sum(op.startswith("ds_") for op, _ in _ops(body)) == 18


def test_stacks_of_eight_run_their_exchange_straight_line(gpd_asm):
    """gpd_rollout1_kernel compiled for aviaries of eight (BASELINE config 3 ii, RPM, all force terms, one sub-step per step): in each
    of the three step copies the six 16-byte reads of the mates' positions stand in ONE run (issued together: the second group's LDS
    latency passes under the first group's arithmetic); no scratch."""
    body, meta = _kernel(gpd_asm, "gpd_rollout1_kernelILb0ELb1ELi4ELi0ELb1ELb1ELb1ELb0ELi8ELi7ELb0E")
    assert re.search(r"ScratchSize: 0\b", meta)
    ops = [op for op, _ in _ops(body) if op.startswith("ds_") or op.startswith(("s_cbranch", "s_branch"))]
    runs, cur = [], 0
    for op in ops + ["end"]:
        if op == "ds_read_b128":
            cur += 1
        else:
            if cur:
                runs.append(cur)
            cur = 0
    assert sum(r >= 6 for r in runs) >= 3, runs


def test_one_world_step_kernel_reduces_without_the_lds_crossbar_and_bursts_in_two_runs(swarm_asm):
    """gpd_swarm_step_kernel (profiles/r04_swarm_step_timeline.txt): the tail's three wave reductions are DPP
    steps + v_readlane (`wave_allreduce`: no ds_bpermute left in the kernel), and the row bursts of a full wave are LDS reads in a
    run followed by their stores -- somewhere in the kernel five ds_read_b128 stand next to each other (the state vectors' fast
    path), which the per-store test of the ragged path never produces; one sub-step's physics fits 96 VGPRs, no scratch."""
    body, meta = _kernel(swarm_asm, "gpd_swarm_step_kernelILi5E")
    assert re.search(r"ScratchSize: 0\b", meta) and int(re.search(r"; NumVgprs: (\d+)", meta).group(1)) <= 96
    ops = [op for op, _ in _ops(body)]
    c = Counter(ops)
    assert c["ds_bpermute_b32"] == 0 and c["v_readlane_b32"] >= 12
    assert sum(1 for op, s in _ops(body) if "_dpp" in op and ("row_mirror" in s or "row_half_mirror" in s)) >= 6
    lds = [op for op in ops if op.startswith(("ds_read_b128", "global_store", "ds_write"))]
    runs, best = 0, 0
    for op in lds:
        runs = runs + 1 if op == "ds_read_b128" else 0
        best = max(best, runs)
    assert best >= 5, best
