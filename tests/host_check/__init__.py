"""Builds and loads the CPU logic-check library (g++).  Test tool only - see host_check.cpp."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "host_check.cpp")
OUT = os.path.join(HERE, "_build", "libhostcheck.so")
CSRC = os.path.join(HERE, "..", "..", "emplanner_carla_amd", "csrc")


def load():
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("emp_core.h", "emp_frenet_core.h", "emp_qp_core.h", "emp_st_core.h", "emp_st_backend_core.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", SRC, "-o", OUT],
                       check=True)
    lib = C.CDLL(OUT)
    d, i, p = C.c_double, C.c_int, C.c_void_p
    lib.hc_segment_cost.restype = d
    lib.hc_segment_cost.argtypes = [d, d, d, d, d, d, p, p, i, d, d, d, d, d]
    lib.hc_neighbour_cost.restype = d
    lib.hc_neighbour_cost.argtypes = [d, d, d, d, p, p, i, d, d, d, d, d]
    lib.hc_path_qp.restype = i
    lib.hc_path_qp.argtypes = [i, p, p, d, d, d, p, p, p, p, p]
    lib.hc_path_qp_gi.restype = i
    lib.hc_path_qp_gi.argtypes = [i, p, p, d, d, d, p, p, p, p, p]
    lib.hc_box_qp.restype = i
    lib.hc_box_qp.argtypes = [i, p, i, d, d, d, d, p, p]
    lib.hc_heading_kappa.restype = None
    lib.hc_heading_kappa.argtypes = [p, i, p, p]
    lib.hc_s_map.restype = None
    lib.hc_s_map.argtypes = [p, i, d, d, p]
    lib.hc_dot2.restype = None
    lib.hc_dot2.argtypes = [i, p, p, p]
    lib.hc_match.restype = i
    lib.hc_match.argtypes = [p, i, d, d, i, i, i]
    lib.hc_st_edge_cost.restype = d
    lib.hc_st_edge_cost.argtypes = [p, p, i, p, p, p, p, p]
    lib.hc_st_reach.restype = None
    lib.hc_st_reach.argtypes = [i, d, p, p, p, p, p, p]
    lib.hc_st_graph.restype = None
    lib.hc_st_graph.argtypes = [i] + [p] * 8
    lib.hc_st_grid.restype = None
    lib.hc_st_grid.argtypes = [p, p]
    lib.hc_st_terminal.restype = i
    lib.hc_st_terminal.argtypes = [p, p, p]
    lib.hc_stb_convex_space.restype = i
    lib.hc_stb_convex_space.argtypes = [p, p, p, p, i, p, p, p, p, i, d, p, p, p, p]
    lib.hc_stb_speed_qp.restype = i
    lib.hc_stb_speed_qp.argtypes = [p, p, d, d, p, p, p, p, p, p, p, p, p, p]
    lib.hc_stb_increase_points.restype = i
    lib.hc_stb_increase_points.argtypes = [p] * 8
    lib.hc_stb_np_interp.restype = d
    lib.hc_stb_np_interp.argtypes = [p, p, i, d]
    return lib
