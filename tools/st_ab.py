"""Development check: the speed DP kernel of this build against another build of the library (EMP_AB_LIB, e.g. the
previous commit's), bit for bit on B scenes.  Usage: EMP_AB_LIB=path/to/other.so python tools/st_ab.py [B]"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def run(B):
    from emplanner_carla_amd import _lib
    if os.environ.get("EMP_DBG_LIB"):
        _lib.LIB_PATH = os.path.abspath(os.environ["EMP_DBG_LIB"])
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, speed_dp_params
    o = S.make_dynamic_batch(range(B), 16)
    pl = Planner(0)
    sets = pl.st_graph(*o[:4])
    r = pl.speed_dp(speed_dp_params(), *sets, o[4])
    return dict(cost=r.cost, s_dot=r.s_dot, node=r.node, end=r.end_node, ss=r.speed_s, tt=r.speed_t)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if os.environ.get("EMP_AB_CHILD"):
        np.savez(os.environ["EMP_AB_CHILD"], **run(B))
        sys.exit(0)
    out = []
    for tag, env in (("a", {}), ("b", {"EMP_DBG_LIB": os.environ["EMP_AB_LIB"]})):
        f = f"/tmp/st_ab_{tag}.npz"
        subprocess.run([sys.executable, __file__, str(B)], check=True, env={**os.environ, **env, "EMP_AB_CHILD": f})
        out.append(np.load(f))
    bad = 0
    for k in out[0].files:
        a, b = out[0][k], out[1][k]
        same = (a == b) | (np.isnan(a.astype(float)) & np.isnan(b.astype(float)))
        print(f"{k}: {same.size - same.sum()} of {same.size} entries differ")
        bad += int(same.size - same.sum())
    print("IDENTICAL" if bad == 0 else "DIFFERENT")
    sys.exit(1 if bad else 0)
