"""Static check of gfx950 assembly (hipcc -S / -save-temps) for one miscompile signature: a scalar load whose destination
dword is DEAD -- overwritten on every path before anything reads it.  The register allocator never produces that on its own
(an unused kernarg dword is simply not loaded), but clang 20 (ROCm 7.2) did in `gpd_rollout_policy_kernel<true,4,VEL,1,false>`
under the default scheduler: `s_load_dwordx8 s[48:55]` (GpdStepCfg) in the entry block, `s_load_dwordx16 s[36:51]` (GpdParams)
in the next block, and only afterwards the SGPR spill `v_writelane_b32 v164, s48..s55, 29..36` that was meant to save the
FIRST load -- four step-configuration words were replaced by drone parameters.

usage: isa_spill_check.py file.s [kernel-name-substring]   (tests/test_kernel_isa.py runs `torn_spills` over both units)
"""
import re
import sys

NO_DEF = ("s_cmp", "s_bitcmp", "s_store", "s_waitcnt", "s_branch", "s_cbranch", "s_setpc", "s_barrier", "s_nop", "s_endpgm", "s_sleep",
          "s_setprio", "s_sendmsg", "s_dcache", "s_icache", "s_trap", "s_sethalt", "s_setreg", "s_set_gpr", "s_code_end", "s_ttrace",
          "s_incperflevel", "s_decperflevel", "s_buffer_store", "s_scratch_store", "s_atc", "s_wakeup", "s_version", "s_rfe", "s_cbranch_g_fork")
RMW = ("s_cmov", "s_cmovk", "s_bitset", "s_addk", "s_mulk", "s_bfm")                  # destination is also read (or conditionally kept)
SDST2 = ("v_add_co_", "v_sub_co_", "v_subrev_co_", "v_addc_co_", "v_subb_co_", "v_subbrev_co_", "v_div_scale_", "v_mad_u64_u32", "v_mad_i64_i32")
SREG = re.compile(r"(?<![\w.])s\[(\d+):(\d+)\]|(?<![\w.])s(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def split_ops(rest):
    return [o.strip() for o in rest.split(",")] if rest else []


def def_use(mn, ops):
    """SGPRs written / read by one instruction (conservative: when unsure an operand is a read)."""
    d, u = set(), set()
    if mn.startswith("s_") and not mn.startswith(NO_DEF):
        d = sregs(ops[0]) if ops else set()
        u = sregs(",".join(ops[1:]))
        if mn.startswith(RMW):
            u |= d
    elif mn.startswith(("v_readlane", "v_readfirstlane")) or (mn.startswith(("v_cmp", "v_cmpx")) and mn.endswith("_e64")):
        d = sregs(ops[0])
        u = sregs(",".join(ops[1:]))
    elif mn.startswith(SDST2) and len(ops) > 1:
        d = sregs(ops[1])
        u = sregs(",".join(ops[2:]))
    else:
        u = sregs(",".join(ops))
    return d, u


def kernels(text):
    lines = text.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        # (a kernel's label is followed by its entry block: "%bb.0", or -- kernels whose leading arguments are preloaded into SGPRs,
        # -amdgpu-kernarg-preload-count -- the numbered block of the backward-compatibility prologue that loads them and branches on)
        if m and i + 1 < len(lines) and re.search(r"%bb\.\d+", lines[i + 1]):
            j = next(k for k in range(i, len(lines)) if lines[k].startswith(".Lfunc_end"))
            yield m.group(1), lines[i + 1:j]
            i = j
        i += 1


def parse(body):
    """(instructions [(line index in body, mnemonic, operands, text)], successors per instruction, (defs, uses) per instruction)"""
    ins, labels = [], {}
    for n, raw in enumerate(body):
        l = raw.split(";")[0].strip()
        if not l:
            continue
        m = re.match(r"^(\.L\w+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if l.startswith("."):
            continue
        parts = l.split(None, 1)
        ins.append((n, parts[0], split_ops(parts[1] if len(parts) > 1 else ""), l))
    n_ins = len(ins)
    succ = []
    for k, (_, mn, ops, _) in enumerate(ins):
        if mn == "s_endpgm":
            succ.append(())
        elif mn == "s_branch":
            succ.append((labels[ops[0]],))
        elif mn.startswith("s_cbranch"):
            succ.append(tuple(x for x in (k + 1, labels[ops[0]]) if x < n_ins))
        elif mn.startswith(("s_setpc", "s_swappc")):
            raise ValueError("indirect branch: not handled")
        else:
            succ.append((k + 1,) if k + 1 < n_ins else ())
    du = [def_use(mn, ops) for _, mn, ops, _ in ins]
    return ins, succ, du


def check(body):
    """[(line index within the kernel body, text, dead registers)] for every scalar load with a dead destination dword."""
    ins, succ, du = parse(body)
    n_ins = len(ins)
    live_in = [set() for _ in range(n_ins)]
    changed = True
    while changed:                                   # backward liveness, iterated to the fixed point (instruction granularity)
        changed = False
        for k in range(n_ins - 1, -1, -1):
            out = set()
            for s in succ[k]:
                out |= live_in[s]
            new = (out - du[k][0]) | du[k][1]
            if new != live_in[k]:
                live_in[k] = new
                changed = True
    bad = []
    for k, (n, mn, ops, l) in enumerate(ins):
        if mn.startswith(("s_load_", "s_buffer_load_")):
            out = set()
            for s in succ[k]:
                out |= live_in[s]
            dead = du[k][0] - out
            if dead:
                bad.append((n, l, sorted(dead)))
    return bad


def spill_runs(body):
    """Runs of `v_writelane_b32 vS, s(r+i), (l+i)`: one SGPR tuple saved to consecutive lanes of a spill VGPR."""
    runs, cur = [], None
    for n, raw in enumerate(body):
        l = raw.split(";")[0].strip()
        m = re.match(r"v_writelane_b32 (v\d+), s(\d+), (\d+)$", l)
        if m:
            v, r, lane = m.group(1), int(m.group(2)), int(m.group(3))
            if cur and cur["v"] == v and cur["regs"][-1] + 1 == r and cur["lanes"][-1] + 1 == lane:
                cur["regs"].append(r); cur["lanes"].append(lane); cur["end"] = n
            else:
                cur = dict(v=v, regs=[r], lanes=[lane], start=n, end=n)
                runs.append(cur)
        elif l and not l.startswith(("s_nop", "v_", "ds_", "global_", "s_waitcnt")):
            cur = None
    return [r for r in runs if len(r["regs"]) > 1]


def preload_lengths(text):
    """kernel name -> `.amdhsa_user_sgpr_kernarg_preload_length` (dwords of the argument block that arrive in SGPRs) of every kernel
    descriptor in an assembly file."""
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)([\s\S]*?)\.end_amdhsa_kernel", text):
        n = re.search(r"\.amdhsa_user_sgpr_kernarg_preload_length (\d+)", m.group(2))
        out[m.group(1)] = int(n.group(1)) if n else 0
    return out


def kernarg_pointer_pairs(body):
    """SGPR pairs that hold the kernel-argument segment pointer: s[0:1] (these kernels enable no dispatch / queue pointer, which
    tests/test_kernel_isa.py asserts) and every pair an `s_mov_b64` copies it to, in textual order, until something else defines
    the pair.  -> {line index: set of pairs valid at that line}"""
    cur, out = {(0, 1)}, {}
    for i, line in enumerate(body):
        l = line.split(";")[0].strip()
        out[i] = set(cur)
        if not l or l.endswith(":"):
            continue
        mn, _, rest = l.partition(" ")
        ops = split_ops(rest.strip())
        d, _u = def_use(mn, ops)
        m = re.match(r"s_mov_b64\s+s\[(\d+):(\d+)\],\s*s\[(\d+):(\d+)\]$", l)
        if m and (int(m.group(3)), int(m.group(4))) in cur:
            cur.add((int(m.group(1)), int(m.group(2))))
            continue
        cur = {pr for pr in cur if not (set(pr) & d)}
    return out


def torn_spills(body, unused_kernarg_offsets=()):
    """The miscompile proper: a scalar load A keeps some destination dwords alive and loses others to a later definition,
    and afterwards ONE spill run saves both kinds together as if A's tuple were intact (reachability over the kernel's control-flow graph).
    `unused_kernarg_offsets`: byte offsets of kernel-argument words NO device code reads (a host-only struct member): there is no
    s_load_dwordx3, so three used words next to such a member are fetched as an x4 whose fourth register the allocator is free to
    reuse at once -- the same shape, and harmless; a finding all of whose dead dwords sit at such offsets is not reported, PROVIDED
    the load's base is the kernel-argument pointer (s[0:1] or a copy of it, `kernarg_pointer_pairs`): a load of the same offset
    from any other base -- weights, a struct in memory -- is reported as ever."""
    karg = kernarg_pointer_pairs(body) if unused_kernarg_offsets else {}
    dead_of = {n: (l, set(dead)) for n, l, dead in check(body)}
    out = []
    runs = spill_runs(body)
    ins, succ, du = parse(body)
    for n, (l, dead) in dead_of.items():
        mn, rest = l.split(None, 1)
        dst = sregs(split_ops(rest)[0])
        alive = dst - dead
        if not alive:
            continue
        m_imm = re.search(r",\s*(0x[0-9a-fA-F]+|\d+)\s*$", rest)
        m_base = re.match(r"[^,]+,\s*s\[(\d+):(\d+)\]", rest)
        from_kernarg = m_base is not None and (int(m_base.group(1)), int(m_base.group(2))) in karg.get(n, set())
        if (unused_kernarg_offsets and from_kernarg and m_imm
                and all(int(m_imm.group(1), 0) + 4 * (r - min(dst)) in unused_kernarg_offsets for r in dead)):
            continue
        for run in runs:
            if run["start"] < n:
                continue
            regs = set(run["regs"])
            if regs != dst:          # the signature: A's whole destination tuple, and nothing else, saved as one value
                continue
            # A's surviving dwords must still be A's at the spill: the load's value of at least one of them REACHES the first
            # instruction of the run along the control-flow graph without a redefinition (a latch block laid out in front of its
            # loop is reached through the loop body, which sits textually behind it: textual order is not the path)
            k_load = next(k for k, (nn, *_r) in enumerate(ins) if nn == n)
            k_run = next(k for k, (nn, *_r) in enumerate(ins) if nn >= run["start"])
            for r in sorted(regs & alive):
                seen, todo, hit = set(), list(succ[k_load]), False
                while todo:
                    k = todo.pop()
                    if k in seen:
                        continue
                    seen.add(k)
                    if k == k_run:
                        hit = True
                        break
                    if r in du[k][0]:
                        continue
                    todo.extend(succ[k])
                if hit:
                    out.append((n, l, sorted(dead), run))
                    break
    return out


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for name, body in kernels(text):
        if want not in name:
            continue
        for n, l, dead, run in torn_spills(body):
            total += 1
            print(f"{name[:90]}: line {n}: {l}: s{dead} lost before the spill at line {run['start']} "
                  f"(s{run['regs'][0]}..s{run['regs'][-1]} -> {run['v']} lanes {run['lanes'][0]}..{run['lanes'][-1]})")
    print(f"{total} torn spill(s)")
