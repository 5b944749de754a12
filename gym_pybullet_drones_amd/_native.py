"""ctypes binding of `csrc/libgpd.so` (the C-ABI declared in include/gpd.h).

There is NO fallback: if the library is missing or does not match the header, importing the
product path raises.  `build()` compiles it in-tree with hipcc for gfx950 (cross-compiles on a
machine without a GPU).
"""
import ctypes
import os
import subprocess

from .params import GpdParams

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")        # the source tree's include/ ...
if not os.path.exists(os.path.join(INCLUDE, "gpd.h")):            # ... or the copy setup.py puts inside an installed package
    INCLUDE = os.path.join(_HERE, "include")
LIB_PATH = os.path.join(CSRC, "libgpd.so")
ABI_VERSION = 9

# Four translation units over csrc/gpd_common.inc (the shared physics), one library:
#   step_rollout.hip  gpd_step / gpd_rollout*            -mllvm -amdgpu-sched-strategy=max-ilp: interleaves independent dependency chains, which
#                     fills the one-wait-state hazard behind every packed-fp32 result with useful work instead of s_nops (13 of 287 issue
#                     slots of a rollout step)
#   policy.hip        gpd_rollout_policy                 the default scheduler: 10 % faster on its MFMA + activation mix (round-2 A/B)
#   swarm.hip         the one-world kernels              max-ilp, as in rounds 2-4
#   abi.hip           small kernels, RCCL, library-level entries
#   -mllvm -amdgpu-kernarg-preload-count=14: the first 14 argument dwords of a kernel arrive in SGPRs with the wave (gfx942+ command
#   processor) instead of through a scalar load -- gpd_step_kernel's argument list starts with what its load section needs
COMMON_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC"]
MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
KERNARG_PRELOAD = ["-mllvm", "-amdgpu-kernarg-preload-count=14"]
HIPCC_FLAGS = COMMON_FLAGS + MAX_ILP + ["-shared"]        # (kept under this name for the ISA tests)
UNITS = (("step_rollout.hip", MAX_ILP + KERNARG_PRELOAD), ("policy.hip", []), ("swarm.hip", MAX_ILP + KERNARG_PRELOAD), ("abi.hip", MAX_ILP))
HEADERS = ("gpd_common.inc", "policy_kernel.inc")


class GpdError(RuntimeError):
    pass


class GpdState(ctypes.Structure):
    """mirror of `struct GpdState`"""
    _fields_ = [("kin", ctypes.c_void_p), ("last_rpm", ctypes.c_void_p), ("pid", ctypes.c_void_p),
                ("step_counter", ctypes.c_void_p), ("ld", ctypes.c_int64), ("dw_force", ctypes.c_void_p),
                ("act_ring", ctypes.c_void_p), ("ring_pos", ctypes.c_void_p), ("hist_len", ctypes.c_int32),
                ("pad_", ctypes.c_int32), ("bad", ctypes.c_void_p)]


class GpdStepCfg(ctypes.Structure):
    """mirror of `struct GpdStepCfg`"""
    _fields_ = [("num_envs", ctypes.c_int32), ("drones_per_env", ctypes.c_int32), ("act_type", ctypes.c_int32),
                ("substeps", ctypes.c_int32), ("physics_flags", ctypes.c_uint32), ("pyb_dt", ctypes.c_float),
                ("ctrl_dt", ctypes.c_float), ("inv_ctrl_dt", ctypes.c_float), ("lanes_per_wave", ctypes.c_int32),
                ("task", ctypes.c_int32), ("xy_bound", ctypes.c_float),
                ("z_bound", ctypes.c_float), ("tilt_bound", ctypes.c_float), ("term_dist", ctypes.c_float),
                ("trunc_counter", ctypes.c_int32), ("target_per_env", ctypes.c_int32),
                ("init_per_env", ctypes.c_int32), ("auto_reset", ctypes.c_int32)]


class GpdP2P(ctypes.Structure):
    """mirror of `struct GpdP2P` (one block of gpd_p2p_group)"""
    _fields_ = [("peer", ctypes.c_int32), ("pad_", ctypes.c_int32), ("ptr", ctypes.c_void_p), ("count", ctypes.c_int64)]


class GpdSwarm(ctypes.Structure):
    """mirror of `struct GpdSwarm`"""
    _fields_ = [("n_rows", ctypes.c_int32), ("slab", ctypes.c_int32), ("world_size", ctypes.c_int32), ("rank", ctypes.c_int32),
                ("own_count", ctypes.c_int32), ("nx", ctypes.c_int32), ("ny", ctypes.c_int32), ("nz", ctypes.c_int32),
                ("cell", ctypes.c_float), ("x0", ctypes.c_float), ("y0", ctypes.c_float), ("z0", ctypes.c_float),
                ("zbin", ctypes.c_float), ("meta_rows", ctypes.c_int32),
                ("pos4", ctypes.c_void_p), ("bin_pos", ctypes.c_void_p), ("cell_count", ctypes.c_void_p),
                ("cell_start", ctypes.c_void_p), ("order", ctypes.c_void_p), ("visit", ctypes.c_void_p), ("visit_out", ctypes.c_void_p),
                ("slot_key", ctypes.c_void_p), ("dw_force", ctypes.c_void_p), ("slot_of", ctypes.c_void_p),
                ("pos_sorted", ctypes.c_void_p), ("pair_list", ctypes.c_void_p), ("pair_nb", ctypes.c_void_p),
                ("list_ok", ctypes.c_void_p), ("list_cap", ctypes.c_int32), ("list_delta", ctypes.c_float),
                ("drift", ctypes.c_void_p), ("total_drones", ctypes.c_int32), ("list_adapt", ctypes.c_int32)]


DEBUG_LIB_PATH = os.path.join(CSRC, "libgpd_debug.so")


def build(force: bool = False, verbose: bool = False, debug: bool = False) -> str:
    """Compile the four units of csrc/ -> csrc/libgpd.so for gfx950 (side by side).  Returns the library path.
    `debug=True`: the debug-bounds build (-DGPD_DEBUG_BOUNDS, include/gpd.h `gpd_debug_status`) -> csrc/libgpd_debug.so; use it by
    setting GPD_LIB to that path before the package is imported."""
    srcs = [os.path.join(CSRC, u) for u, _ in UNITS]
    hdr = os.path.join(INCLUDE, "gpd.h")
    LIB_PATH = DEBUG_LIB_PATH if debug else globals()["LIB_PATH"]
    if not force and os.path.exists(LIB_PATH):
        newest = max([os.path.getmtime(f) for f in srcs] + [os.path.getmtime(hdr)] + [os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS])
        if os.path.getmtime(LIB_PATH) >= newest:
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for (unit, extra), src in zip(UNITS, srcs):          # the units compile side by side
        obj = os.path.join(CSRC, unit.replace(".hip", ".dbg.o" if debug else ".o"))
        cmd = [hipcc] + COMMON_FLAGS + extra + (["-DGPD_DEBUG_BOUNDS"] if debug else []) + ["-I", INCLUDE, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for cmd, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise GpdError("hipcc failed: " + " ".join(cmd) + "\n" + out)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise GpdError("hipcc (link) failed:\n" + res.stdout + res.stderr)
    for o in objs:
        os.remove(o)
    return LIB_PATH


_lib = None

_P = ctypes.c_void_p
_SIGNATURES = {
    "gpd_abi_version": (ctypes.c_int, []),
    "gpd_last_error": (ctypes.c_char_p, []),
    "gpd_struct_sizes": (None, [ctypes.POINTER(ctypes.c_int32)]),
    "gpd_step": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg),
                                _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gpd_step_sync": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg),
                                     _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gpd_rollout": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg),
                                   ctypes.c_int32, _P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P, _P, _P,
                                   ctypes.c_int64, _P, _P]),
    "gpd_rollout_history": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg),
                                           ctypes.c_int32, _P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P, _P, _P,
                                           ctypes.c_int64, _P]),
    "gpd_rollout_policy": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg), _P,
                                          ctypes.c_int32, _P, _P, _P, _P, _P, ctypes.c_int64, _P, _P, _P, ctypes.c_int64, _P,
                                          ctypes.POINTER(ctypes.c_float), _P, _P, _P]),
    "gpd_hist_rows": (ctypes.c_int, [ctypes.POINTER(GpdState), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P]),
    "gpd_full_obs": (ctypes.c_int, [ctypes.POINTER(GpdState), ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P,
                                    ctypes.c_int64, _P, ctypes.c_int64, _P, ctypes.c_int64, _P]),
    "gpd_downwash_global": (ctypes.c_int, [ctypes.POINTER(GpdParams), _P, ctypes.c_int64, ctypes.c_int32, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_int32, _P, _P, _P, _P, _P, _P, ctypes.POINTER(GpdState), _P, _P, _P]),
    "gpd_sizeof_swarm": (ctypes.c_int, []),
    "gpd_swarm_step": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdState), ctypes.POINTER(GpdStepCfg),
                                      ctypes.POINTER(GpdSwarm), _P, _P, _P, _P]),
    "gpd_swarm_pack": (ctypes.c_int, [ctypes.POINTER(GpdState), ctypes.POINTER(GpdSwarm), _P, _P, _P]),
    "gpd_swarm_bin": (ctypes.c_int, [ctypes.POINTER(GpdSwarm), _P]),
    "gpd_swarm_forces": (ctypes.c_int, [ctypes.POINTER(GpdParams), ctypes.POINTER(GpdSwarm), ctypes.c_int32, _P]),
    "gpd_reset": (ctypes.c_int, [ctypes.POINTER(GpdState), _P, ctypes.c_int32, _P, ctypes.c_int32, ctypes.c_int32,
                                 ctypes.c_int32, _P, _P]),
    "gpd_pid": (ctypes.c_int, [ctypes.POINTER(GpdParams), _P, ctypes.c_int64, ctypes.c_float, _P, _P, _P, _P, _P, _P,
                               _P, _P, _P, _P, ctypes.c_int32, _P]),
    "gpd_pid_sync": (ctypes.c_int, [ctypes.POINTER(GpdParams), _P, ctypes.c_int64, ctypes.c_float, _P, _P, _P, _P, _P, _P,
                                    _P, _P, _P, _P, ctypes.c_int32, _P]),
    "gpd_state_vectors": (ctypes.c_int, [ctypes.POINTER(GpdState), _P, _P, ctypes.c_int32, _P]),
    "gpd_comm_unique_id": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint8)]),
    "gpd_comm_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint8), ctypes.c_int32,
                                     ctypes.c_int32]),
    "gpd_comm_count": (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32)]),
    "gpd_comm_destroy": (ctypes.c_int, [_P]),
    "gpd_allgather_obs": (ctypes.c_int, [_P, _P, _P, ctypes.c_size_t, _P]),
    "gpd_p2p_group": (ctypes.c_int, [_P, ctypes.POINTER(GpdP2P), ctypes.c_int32, ctypes.POINTER(GpdP2P), ctypes.c_int32, _P]),
    "gpd_debug_status": (ctypes.c_int, [ctypes.POINTER(ctypes.c_uint32), ctypes.c_int32, _P]),
    "gpd_clock_probe": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), _P]),
}
COMM_ID_BYTES = 128
GPD_EINVAL, GPD_ERANGE, GPD_ENOTSUP = -1, -2, -3


def lib() -> ctypes.CDLL:
    """Load (once) and verify the shared library.  Raises GpdError when it is absent or stale."""
    global _lib
    if _lib is not None:
        return _lib
    try:        # PyTorch's HIP runtime first: libgpd.so then binds to the copy torch mapped (one runtime per process; loaded the other way
        import torch  # noqa: F401 -- round a kernel launch fails with "no ROCm-capable device is detected", gpurun_out/smoke3.log, round 5)
    except ImportError:
        pass
    path = os.environ.get("GPD_LIB", LIB_PATH)     # GPD_LIB: A/B-test another build of the same ABI
    if not os.path.exists(path):
        raise GpdError(f"{path} not found: the HIP extension has not been built "
                       "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
                       "There is no CPU fallback for the simulator's hot path.")
    try:
        L = ctypes.CDLL(path)
    except OSError as e:
        raise GpdError(f"cannot load {path}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError as e:
            raise GpdError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if L.gpd_abi_version() != ABI_VERSION:
        raise GpdError(f"libgpd ABI {L.gpd_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
    sizes = (ctypes.c_int32 * 3)()
    L.gpd_struct_sizes(sizes)
    want = (ctypes.sizeof(GpdParams), ctypes.sizeof(GpdState), ctypes.sizeof(GpdStepCfg))
    if tuple(sizes) != want:
        raise GpdError(f"struct layout mismatch: library {tuple(sizes)} vs binding {want}")
    if L.gpd_sizeof_swarm() != ctypes.sizeof(GpdSwarm):
        raise GpdError(f"struct layout mismatch: GpdSwarm is {L.gpd_sizeof_swarm()} bytes in the library, {ctypes.sizeof(GpdSwarm)} in the binding")
    _lib = L
    return L


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().gpd_last_error().decode("utf-8", "replace")
        raise GpdError(f"{what} failed (code {rc}): {msg}")


def exported_symbols():
    return list(_SIGNATURES)
