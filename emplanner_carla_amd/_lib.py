"""ctypes binding of include/emplanner.h (the C-ABI shared library built by build.py).

There is NO fallback: if ``libemplanner.so`` is missing or cannot be loaded, importing a
planner function raises ``RuntimeError``; if no gfx950 GPU is visible, creating a context
raises ``EmpError``.  Nothing in this package computes planner results on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libemplanner.so")


def configure_hw_queues(n: int = 8) -> bool:
    """Lane mode (emp_set_pipeline with n >= 2) wants one hardware queue per stream; the HIP runtime maps all streams of a
    process onto GPU_MAX_HW_QUEUES queues (default 4) and reads the variable ONCE, when it initialises.  Neither the
    library nor this package touches the process environment on its own: a program that owns its process (bench.py does)
    calls this before anything initialises HIP - before the first torch.cuda call and the first Planner.  Returns False,
    and changes nothing, if the variable is already set."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(int(n))
    return True


def hw_queues() -> int:
    """The queue count the HIP runtime will use / has used, as far as the environment tells."""
    try:
        return int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    except ValueError:
        return 4


EMP_HOST, EMP_DEVICE, EMP_HOST_PINNED = 0, 1, 2
EMP_EDGE_CANONICAL, EMP_EDGE_TILED = 0, 1
EMP_DP_FUSED, EMP_DP_TWO_KERNEL = 0, 1
EMP_PIPELINE_AUTO, EMP_PIPELINE_STAGED, EMP_PIPELINE_MAX = -1, 1, 8

# emp_option (include/emplanner.h; ABI 11: twelve of them): per-context tuning / A-B / test-hook values
OPTIONS = {"path_qp_form": 0, "cartesian_form": 1, "smooth_force_fallback": 2, "edge_block": 3, "edge_form": 4, "sweep_exclusive": 5,
           "edge_after_enrich": 6, "lane_edge_order": 7, "cycle_graph": 8, "sweep_clock_probe": 9, "edge_clock_probe": 10,
           "foreign_streams": 11}
#: the values a fresh context holds (everything else is 0)
OPTION_DEFAULTS = {"edge_after_enrich": 1, "lane_edge_order": 2, "foreign_streams": 1}

ST_DP_INFEASIBLE = 1
ST_S_OUT_OF_RANGE = 2
ST_BOUND_INDEX = 4
ST_QP_FAILED = 8
ST_SMOOTH_FAILED = 16
ST_TRUNCATED = 32


class EmpError(RuntimeError):
    """A C-ABI call returned a negative emp_error."""


class DpParams(C.Structure):
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("sample_s", C.c_double), ("sample_l", C.c_double),
                ("sampling_res", C.c_double), ("w_collision", C.c_double), ("w_smooth", C.c_double * 3),
                ("w_ref", C.c_double)]


class QpParams(C.Structure):
    _fields_ = [("ds", C.c_double), ("w_l", C.c_double), ("w_dl", C.c_double), ("w_ddl", C.c_double),
                ("w_dddl", C.c_double), ("w_centre", C.c_double), ("w_end_l", C.c_double),
                ("w_end_dl", C.c_double), ("w_end_ddl", C.c_double), ("host_d1", C.c_double),
                ("host_d2", C.c_double), ("host_w", C.c_double), ("obs_length", C.c_double),
                ("obs_width", C.c_double), ("decimate", C.c_int32), ("midpoint", C.c_int32),
                ("use_qp", C.c_int32), ("reserved", C.c_int32)]


class SmoothParams(C.Structure):
    _fields_ = [("w_smooth", C.c_double), ("w_length", C.c_double), ("w_ref", C.c_double),
                ("x_thre", C.c_double), ("y_thre", C.c_double)]


class MpcParams(C.Structure):
    _fields_ = [("a", C.c_double), ("b", C.c_double), ("Cf", C.c_double), ("Cr", C.c_double), ("m", C.c_double),
                ("Iz", C.c_double), ("q_diag", C.c_double * 4), ("f_diag", C.c_double * 4), ("r", C.c_double)]


class SpeedQpParams(C.Structure):
    """emp_speed_qp_params: keyword arguments of the reference's speed_QP (speed_planning_test.py:410-411)."""
    _fields_ = [("w_cost_s_dot2", C.c_double), ("w_cost_v_ref", C.c_double), ("w_cost_jerk", C.c_double),
                ("reference_speed", C.c_double)]


class SpeedDpParams(C.Structure):
    _fields_ = [("reference_speed", C.c_double), ("w_cost_ref_speed", C.c_double), ("w_cost_accel", C.c_double),
                ("w_cost_obs", C.c_double)]


_vp, _i32, _f64, _u64 = C.c_void_p, C.c_int32, C.c_double, C.c_uint64


class CycleIO(C.Structure):
    _fields_ = [(n, _vp) for n in (
        "ref_line", "n_ref", "origin_xy", "start_xy", "start_v", "start_a", "obs_xy", "n_obs",
        "dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status",
        "dyn_dis_speed",
        # the optional front end (ABI 11): the cycle starts from the global path
        "global_path", "n_global", "pre_match_index", "match_index", "ref_status")] + [("max_global", C.c_int32), ("reserved_io", C.c_int32)]


# name -> (restype, argtypes); data pointers are void* so numpy arrays and raw device addresses both fit
PROTOTYPES = {
    "emp_abi_version": (C.c_int, []),
    "emp_dp_params_default": (None, [C.POINTER(DpParams)]),
    "emp_qp_params_default": (None, [C.POINTER(QpParams)]),
    "emp_smooth_params_default": (None, [C.POINTER(SmoothParams)]),
    "emp_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "emp_destroy": (None, [_vp]),
    "emp_last_error": (C.c_char_p, [_vp]),
    "emp_synchronize": (C.c_int, [_vp]),
    "emp_stream": (_vp, [_vp]),
    "emp_host_alloc": (C.c_int, [_vp, _u64, C.POINTER(_vp)]),
    "emp_host_free": (C.c_int, [_vp, _vp]),
    "emp_wait_cycle": (C.c_int, [_vp, _i32]),
    "emp_cycle_ticket": (_u64, [_vp]),
    "emp_wait_ticket": (C.c_int, [_vp, _u64]),
    "emp_device_alloc": (C.c_int, [_vp, _u64, C.POINTER(_vp)]),
    "emp_device_free": (C.c_int, [_vp, _vp]),
    "emp_copy_to_device": (C.c_int, [_vp, _vp, _vp, _u64]),
    "emp_copy_to_host": (C.c_int, [_vp, _vp, _vp, _u64]),
    "emp_set_timing": (C.c_int, [_vp, C.c_int]),
    "emp_set_timing_filter": (C.c_int, [_vp, C.c_char_p]),
    "emp_set_pipeline": (C.c_int, [_vp, C.c_int]),
    "emp_result_stream": (_vp, [_vp]),
    "emp_pipeline_depth": (C.c_int, [_vp]),
    "emp_pipeline_form": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "emp_set_fence": (C.c_int, [_vp, C.c_int]),
    "emp_set_option": (C.c_int, [_vp, _i32, _i32]),
    "emp_get_option": (C.c_int, [_vp, _i32, C.POINTER(_i32)]),
    "emp_sweep_clock_mhz": (_f64, [_vp, C.POINTER(_f64), C.POINTER(_f64)]),
    "emp_cycle_graph_replays": (C.c_int64, [_vp]),
    "emp_edge_probe": (C.c_int, [_vp, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_i32)]),
    "emp_edge_clock_mhz": (_f64, [_vp]),
    "emp_sweep_probe_spans": (C.c_int, [_vp, C.POINTER(_f64), C.POINTER(_f64)]),
    "emp_pack_records": (C.c_int, [_vp, _i32, _i32, _i32, _i32] + [_vp] * 8 + [C.c_int, C.c_int]),
    "emp_pack_trajectory_records": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 4 + [C.c_int, C.c_int]),
    "emp_kernel_ms": (_f64, [_vp, C.c_char_p]),
    "emp_kernel_launches": (C.c_int, [_vp, C.c_char_p]),
    "emp_kernel_samples": (C.c_int, [_vp, C.c_char_p, _vp, _i32]),
    "emp_edge_tensor_elems": (_u64, [C.POINTER(DpParams), _i32, C.c_int]),
    "emp_dp_edge_costs": (C.c_int, [_vp, C.POINTER(DpParams), _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int,
                                    C.c_int]),
    "emp_dp_plan": (C.c_int, [_vp, C.POINTER(DpParams), _i32, _i32, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp,
                              C.c_int]),
    "emp_dp_sweep": (C.c_int, [_vp, C.POINTER(DpParams), _i32, _vp, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_dp_enrich": (C.c_int, [_vp, C.POINTER(DpParams), _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_enrich_nodes": (C.c_int, [_vp, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_frenet_project": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 13 + [C.c_int]),
    "emp_match_projection": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 6 + [C.c_int]),
    "emp_find_match_points": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 8 + [C.c_int]),
    "emp_heading_kappa": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_lmin_lmax": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 6 + [_f64, _f64, _vp, _vp, _vp, C.c_int]),
    "emp_path_qp": (C.c_int, [_vp, C.POINTER(QpParams), _i32, _i32] + [_vp] * 9 + [C.c_int]),
    "emp_smooth_line": (C.c_int, [_vp, C.POINTER(SmoothParams), _i32, _i32] + [_vp] * 5 + [C.c_int]),
    "emp_frenet_path_to_xy": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 10 + [C.c_int]),
    "emp_plan_cycle": (C.c_int, [_vp, C.POINTER(DpParams), C.POINTER(QpParams), C.POINTER(SmoothParams), _i32, _i32,
                                 _i32, _i32, C.c_int, C.POINTER(CycleIO), C.c_int]),
    "emp_s_map": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_s_l": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 8 + [C.c_int]),
    "emp_s_l_deri": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 8 + [C.c_int]),
    "emp_proj_point": (C.c_int, [_vp, _i32, _i32] + [_vp] * 8 + [C.c_int]),
    "emp_trajectory_index2s": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, C.c_int]),
    "emp_frenet2cartesian": (C.c_int, [_vp, _i32, _i32, _i32] + [_vp] * 7 + [_i32, C.c_int]),
    "emp_dy_obs_deri": (C.c_int, [_vp, _i32, _vp, _vp, C.c_int]),
    "emp_quintic_coefficients": (C.c_int, [_vp, _i32, _vp, _vp, C.c_int]),
    "emp_obs_cost": (C.c_int, [_vp, _i32, _f64, _f64, _f64, _vp, _vp, C.c_int]),
    "emp_obs_cost_n": (C.c_int, [_vp, _i32, _i32, _f64, _f64, _f64, _vp, _vp, C.c_int]),
    "emp_free_edge_costs": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _f64, C.POINTER(_f64), _f64, _vp, C.c_int]),
    "emp_reference_line": (C.c_int, [_vp, C.POINTER(SmoothParams), _i32, _i32] + [_vp] * 10 + [C.c_int]),
    "emp_mpc_params_default": (None, [C.POINTER(MpcParams)]),
    "emp_mpc_lateral": (C.c_int, [_vp, C.POINTER(MpcParams), _i32, _i32] + [_vp] * 15 + [C.c_int]),
    "emp_lqr_params_default": (None, [C.POINTER(MpcParams)]),
    "emp_lqr_lateral": (C.c_int, [_vp, C.POINTER(MpcParams), _i32, _i32] + [_vp] * 13 + [C.c_int]),
    "emp_speed_dp_params_default": (None, [C.POINTER(SpeedDpParams)]),
    "emp_st_graph": (C.c_int, [_vp, _i32, _i32] + [_vp] * 8 + [C.c_int]),
    "emp_speed_dp": (C.c_int, [_vp, C.POINTER(SpeedDpParams), _i32, _i32] + [_vp] * 11 + [C.c_int]),
    "emp_st_edge_costs": (C.c_int, [_vp, C.POINTER(SpeedDpParams), _i32, _i32, _i32] + [_vp] * 7 + [C.c_int]),
    "emp_st_collision_cost": (C.c_int, [_vp, _i32, _f64, _vp, _vp, C.c_int]),
    "emp_speed_start_condition": (C.c_int, [_vp, _i32] + [_vp] * 7 + [C.c_int]),
    "emp_speed_qp_params_default": (None, [C.POINTER(SpeedQpParams)]),
    "emp_speed_convex_space": (C.c_int, [_vp, _i32, _i32, _i32, _f64] + [_vp] * 14 + [C.c_int]),
    "emp_speed_qp": (C.c_int, [_vp, C.POINTER(SpeedQpParams), _i32] + [_vp] * 14 + [C.c_int]),
    "emp_speed_increase_points": (C.c_int, [_vp, _i32] + [_vp] * 9 + [C.c_int]),
    "emp_path_speed_merge": (C.c_int, [_vp, _i32, _i32] + [_vp] * 13 + [C.c_int]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (soname libamdhip64.so.7, the
    # same as /opt/rocm's).  If torch is imported AFTER this library, the loader maps a second copy and that
    # copy finds no GPU.  Importing torch first makes our DT_NEEDED resolve to the already-loaded runtime.
    if "torch" not in sys.modules and os.environ.get("EMP_SKIP_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m emplanner_carla_amd.build` "
            "(hipcc, gfx950).  This package has no CPU implementation to fall back to.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)   # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.emp_abi_version() != 11:
        raise RuntimeError("libemplanner.so ABI version mismatch")
    _lib = lib
    return lib
