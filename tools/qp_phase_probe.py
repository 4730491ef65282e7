"""Development probe (needs -DEMP_DEV_HOOKS=1 builds under variants/): what an interior-point iteration of the path-QP kernel costs and
where - the kernel's duration on 4096 benchmark scenes with the iteration count of EVERY scene capped at K (emp_qp_params.reserved =
10 + K, read by development builds only), for builds with the factorisation and / or the two substitutions compiled out.  Results of
capped or gutted solves mean nothing; durations do.  Usage: python tools/qp_phase_probe.py variants/lib_dev.so [more libs...]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child(lib):
    sys.path.insert(0, ROOT)
    from emplanner_carla_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    import torch
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params
    cfg = S.CFG2
    b = S.make_batch(range(4096), cfg, start_ahead=S.BENCH_START_AHEAD)
    B, P = b.ref.shape[:2]
    dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in dict(
        ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v,
        start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs).items()}
    pl = Planner(0)
    p, sp = dp_params_from_cfg(cfg), smooth_params()
    out = []
    for K in ("launch", "loads", "bounds only", "set-up", "start point", 0, 1, 2, 4, 8, 12, None):
        q = qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width)
        if K == "launch":
            q.reserved = 4               # (builds with -DEMP_QP_PROBE_EMPTY) the kernel returns at once
        elif K == "loads":
            q.reserved = 5               # ... behind the loads of the DP path
        elif K == "bounds only":
            q.reserved = 1               # debug stage 1: every group idles after cal_lmin_lmax (load + bounds + the kernel's skeleton)
        elif K == "set-up":
            q.reserved = 2               # ... after the B-spline problem is built
        elif K == "start point":
            q.reserved = 3               # ... after the unconstrained minimiser
        elif K is not None:
            q.reserved = 10 + K
        for _ in range(3):
            pl.plan_cycle(p, q, sp, **dev)
        pl.synchronize()
        pl.set_timing(True, only="path_qp")
        for _ in range(10):
            pl.plan_cycle(p, q, sp, **dev)
        pl.synchronize()
        out.append((K, pl.kernel_ms("path_qp") * 1e3))
        pl.set_timing(False)
    ks = [(k, t) for k, t in out if isinstance(k, int)]
    slope = (ks[-1][1] - ks[2][1]) / (ks[-1][0] - ks[2][0])
    print(f"{os.path.basename(lib):28s}", "  ".join(f"K={k}: {t:6.1f} us" for k, t in out), f"  per iteration {slope:.2f} us", flush=True)
    pl.close()

if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
