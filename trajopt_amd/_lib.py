"""ctypes binding of libtrajopt_amd.so - the C ABI declared in include/trajopt_amd.h.

The library is built in-tree by `make -C trajopt_amd/csrc` (or __graft_entry__.build()).  There is no
CPU fallback: if the shared object is missing, or no HIP device is present when a compute entry
point is called, a TrajoptError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("TRAJOPT_AMD_SO") or os.path.join(_HERE, "libtrajopt_amd.so")

LAYOUT_PROBLEM_MAJOR = 0
LAYOUT_PROBLEM_MINOR = 1
MAX_ALPHAS = 16
ABI_VERSION = 5          # include/trajopt_amd.h::TRAJOPT_ABI_VERSION
MAX_STATE, MAX_ACT = 12, 4

ENV_PENDULUM, ENV_CARTPOLE, ENV_DOUBLE_CARTPOLE, ENV_LQR, ENV_QUADROTOR, ENV_DOUBLE_PENDULUM, ENV_ROBOT, ENV_QUAD_PENDULUM = 1, 2, 3, 4, 5, 6, 7, 8


class TrajoptError(RuntimeError):
    pass


class EnvStruct(C.Structure):
    """include/trajopt_env.h::trajopt_env_t"""
    _fields_ = [("model", C.c_int32), ("dm_state", C.c_int32), ("dm_act", C.c_int32),
                ("slew_rate", C.c_int32), ("periodic", C.c_int32), ("cost_times_dt", C.c_int32),
                ("nb_features", C.c_int32), ("features", C.c_int32 * MAX_STATE), ("reserved0", C.c_int32),
                ("dt", C.c_double),
                ("umax", C.c_double * MAX_ACT), ("xmax", C.c_double * MAX_STATE),
                ("goal", C.c_double * MAX_STATE), ("gw", C.c_double * MAX_STATE),
                ("uw", C.c_double * MAX_ACT),
                ("lin_A", C.c_double * (MAX_STATE * MAX_STATE)),
                ("lin_B", C.c_double * (MAX_STATE * MAX_ACT)),
                ("lin_c", C.c_double * MAX_STATE),
                ("phys", C.c_double * 16)]


class LambdaSchedule(C.Structure):
    _fields_ = [("min_lmbda", C.c_double), ("max_lmbda", C.c_double), ("mult_lmbda", C.c_double)]


_dp = C.c_void_p  # device pointers travel as integers


class BackwardIO(C.Structure):
    """include/trajopt_amd.h::trajopt_backward_io_t"""
    _fields_ = ([(k, _dp) for k in
                 ("Cxx", "cx", "Cuu", "cu", "Cxu", "A", "B", "lmbda", "dlmbda",
                  "Qxx", "Qux", "Quu", "qx", "qu", "V", "v", "dV", "K", "kff", "diverge",
                  "uref", "g_norm", "active", "cost_w")] +
                [("cost_gxx", C.c_double * MAX_STATE), ("cost_guu", C.c_double * MAX_ACT),
                 ("kernel", C.c_int32), ("pad_", C.c_int32), ("out_bank", _dp)])


class GpsBackwardIO(C.Structure):
    """include/trajopt_amd.h::trajopt_gps_backward_io_t"""
    _fields_ = [(k, _dp) for k in
                ("Cxx", "cx", "Cuu", "cu", "Cxu", "c0", "A", "B", "c", "sigma_dyn", "alpha",
                 "Qxx", "Qux", "Quu", "qx", "qu", "q0", "V", "v", "v0", "K", "kff", "sigma_ctl", "diverge", "status")]


class BspBackwardIO(C.Structure):
    """include/trajopt_amd.h::trajopt_bsp_backward_io_t"""
    _fields_ = [(k, _dp) for k in
                ("Q", "q", "R", "r", "P", "p", "F", "G", "T", "U", "V", "X", "Y", "Z", "lmbda",
                 "S", "s", "tau", "dS", "K", "kff", "diverge")]


class ILQROptions(C.Structure):
    _fields_ = [("sched", LambdaSchedule), ("tolfun", C.c_double), ("tolgrad", C.c_double),
                ("min_imp", C.c_double), ("reg", C.c_int32), ("nb_alphas", C.c_int32),
                ("alphas", C.c_double * MAX_ALPHAS), ("argmin_alpha", C.c_int32), ("keep_value", C.c_int32),
                ("staged_line_search", C.c_int32), ("no_fusion", C.c_int32)]


class ILQRPlan(C.Structure):
    """include/trajopt_amd.h::trajopt_ilqr_plan_t"""
    _fields_ = [("structured_cost", C.c_int32), ("fused_quad_bwd", C.c_int32), ("staged_line_search", C.c_int32),
                ("backward_kernel", C.c_int32), ("rollout_straight_line", C.c_int32), ("pad_", C.c_int32)]


KERNEL_AUTO, KERNEL_WORKGROUP, KERNEL_MFMA16, KERNEL_LANE, KERNEL_MFMA4, KERNEL_COLUMNS = 0, 1, 2, 3, 4, 5


class ILQRWorkspace(C.Structure):
    _fields_ = ([("nb_steps", C.c_int32), ("gains_layout", C.c_int32), ("nb_problems", C.c_int64)] +
                [(k, _dp) for k in
                 ("x0", "weighting", "xref", "uref", "K", "kff", "V", "v", "Qxx", "Qux", "Quu", "qx", "qu",
                  "last_return", "lmbda", "dlmbda", "alpha", "status", "iters", "trace_len", "trace")] +
                [("trace_cap", C.c_int32), ("pad2_", C.c_int32)] +
                [(k, _dp) for k in
                 ("A", "B", "Cxx", "cx", "Cuu", "cu", "Cxu", "cand_K", "cand_kff", "cand_V", "cand_v", "cand_Qxx",
                  "cand_Qux", "cand_Quu", "cand_qx", "cand_qu", "dV", "g_norm", "diverge", "active", "sel",
                  "cand_state", "cand_action", "cand_ret", "cand_finite", "pending_list", "pending_count",
                  "ref_plane", "gains_bank", "out_bank")])


# every symbol include/trajopt_amd.h declares
EXPORTS = (
    "trajopt_abi_version", "trajopt_error_string", "trajopt_last_hip_error", "trajopt_device_count",
    "trajopt_ilqr_backward_pass_host", "trajopt_ilqr_backward_pass_batched",
    "trajopt_ilqr_backward_pass_host_noalloc", "trajopt_ilqr_backward_pass_host_noalloc_doubles",
    "trajopt_backward_supports_structured_cost",
    "trajopt_gps_backward_supported", "trajopt_gps_backward_pass_batched", "trajopt_gps_backward_pass_host",
    "trajopt_gps_forward_pass_batched", "trajopt_gps_augment_cost_batched", "trajopt_gps_kl_divergence_batched",
    "trajopt_gps_quad_expectation_batched", "trajopt_gps_forward_pass_host", "trajopt_gps_augment_cost_host",
    "trajopt_gps_kl_divergence_host", "trajopt_gps_quad_expectation_host",
    "trajopt_bspilqr_backward_supported", "trajopt_bspilqr_backward_pass_batched", "trajopt_bspilqr_backward_pass_host",
    "trajopt_riccati_backward_pass_host", "trajopt_riccati_backward_pass_batched", "trajopt_riccati_forward_pass_batched",
    "trajopt_env_step_batched", "trajopt_ilqr_forward_pass_batched", "trajopt_quadratize_batched",
    "trajopt_ilqr_init_rollout", "trajopt_ilqr_iterate", "trajopt_ilqr_plan", "trajopt_convert_layout",
    "trajopt_profile_begin", "trajopt_profile_end", "trajopt_profile_select",
)

# what the rollout kernel's arithmetic depends on (the stamp of profiles/*_rollout_isa_census.json)
ROLLOUT_SOURCES = ("rollout.hip", "sys.h", "env_models.h", "common.h", "trajopt_env.h")


def source_stamp(only=None):
    """Hash of the kernel sources (csrc/*.hip, *.h, include/*.h): measurements kept under profiles/ carry the stamp
    of the sources they were taken on, and bench.py only quotes one whose stamp is the current one.
    only: restrict to these file names (the ISA census of one kernel depends on its own sources only)."""
    import glob
    import hashlib
    h = hashlib.sha1()
    root = os.path.dirname(_HERE)
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        if only is not None and os.path.basename(f) not in only:
            continue
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise TrajoptError(
                "%s is missing: build it with `make -C trajopt_amd/csrc` (there is no CPU fallback)" % SO_PATH)
        # torch ships its own libamdhip64; it must be the first HIP runtime in the process: loaded after the
        # system one (which this library would pull in) torch.cuda.is_available() turns False
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(SO_PATH)
        L.trajopt_error_string.restype = C.c_char_p
        L.trajopt_last_hip_error.restype = C.c_char_p
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        L = lib()
        msg = L.trajopt_error_string(rc).decode()
        hip = L.trajopt_last_hip_error().decode()
        raise TrajoptError("trajopt_amd: %s (%d)%s" % (msg, rc, (" - " + hip) if hip and rc == -3 else ""))


def device_count():
    return lib().trajopt_device_count()


def require_gpu():
    n = device_count()
    if n <= 0:
        raise TrajoptError("trajopt_amd needs a HIP device; none is visible (there is no CPU fallback)")
    return n


def dptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
