"""Build the HIP shared library (and nothing else) for gfx950, in-tree.

    python -m emplanner_carla_amd.build            # -> emplanner_carla_amd/libemplanner.so

hipcc cross-compiles without a GPU.  Floating-point contraction is OFF for the whole library:
the DP arithmetic is specified operation by operation (csrc/emp_core.h) and compared bit for
bit with oracle/exact.py.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libemplanner.so")
SOURCES = ["emp_api.hip"]
HEADERS = ["emp_core.h", "emp_context.h", "emp_dp_kernels.h", "emp_qp_core.h", "emp_frenet_core.h",
           "emp_qp_wave.h", "emp_qp_rows.h", "emp_smooth_rows.h", "emp_tail_kernels.h", "emp_st_core.h", "emp_st_kernels.h", "emp_mpc_kernels.h",
           "emp_st_backend_core.h", "emp_st_backend_kernels.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X planner library cannot be built")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "emplanner.h"))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [hipcc()] + FLAGS + os.environ.get("EMP_EXTRA_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
