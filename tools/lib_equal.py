"""Bit-identity of two builds of the library on whole planning cycles (development aid for changes that must not change a bit:
storage layouts, launch geometry): python tools/lib_equal.py A.so B.so [cfg2|cfg5] [scenes].  Each library plans the same scenes in
a child process of its own; every output array must be equal bit for bit (padding beyond each scene's length excluded)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")


def child(lib, cfg_name, scenes, out):
    sys.path.insert(0, ROOT)
    from emplanner_carla_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params
    cfg = {"cfg2": S.CFG2, "cfg5": S.CFG5, "default": S.CFG_DEFAULT}[cfg_name]
    res = {}
    pl = Planner(0)
    for tag, kw in (("bench", dict(start_ahead=S.BENCH_START_AHEAD)), ("tight", dict(per_seed=S.survey_geometry_kwargs))):
        b = S.make_batch(range(scenes), cfg, **kw)
        B, P = b.ref.shape[:2]
        r = pl.plan_cycle(dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params(),
                          ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v,
                          start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)
        for k in OUT:
            a = np.array(getattr(r, k))
            for arr, ln in (("dp_s", "dp_len"), ("dp_l", "dp_len"), ("path_s", "path_len"), ("path_l", "path_len"), ("traj", "traj_len")):
                if k == arr:
                    a[np.arange(a.shape[1])[None, :] >= np.asarray(getattr(r, ln))[:, None]] = 0.0
            res[f"{tag}_{k}"] = a
    pl.close()
    np.savez(out, **res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
        sys.exit(0)
    a, b = sys.argv[1], sys.argv[2]
    pos = [x for x in sys.argv[3:] if not x.startswith("--")]
    cfg = pos[0] if len(pos) > 0 else "cfg2"
    n = int(pos[1]) if len(pos) > 1 else 4096
    outs = []
    for k, lib in enumerate((a, b)):
        out = f"/tmp/lib_equal_{k}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, cfg, str(n), out], check=True)
        outs.append(np.load(out))
    bad = [k for k in outs[0].files if not np.array_equal(outs[0][k], outs[1][k], equal_nan=True)]
    ok = ((outs[0]["bench_status"] & ~1) == 0).mean()
    if "--diff" in sys.argv:            # builds that may differ by rounding (solver arithmetic): how far apart, scene statuses first
        for tag in ("bench", "tight"):
            sa, sb = outs[0][f"{tag}_status"], outs[1][f"{tag}_status"]
            both = ((sa & ~1) == 0) & ((sb & ~1) == 0)
            print(f"  {tag}: status differs on {(sa != sb).sum()} of {len(sa)} scenes; planned by both {both.sum()}")
            for k in ("dp_rows", "dp_s", "dp_l", "path_s", "path_l", "traj"):
                x, y = outs[0][f"{tag}_{k}"][both], outs[1][f"{tag}_{k}"][both]
                d = np.abs(x - y)
                print(f"    {k:8s} max |a-b| {d.max():.3e}   max |a-b| / max(|b|, 1) {(d / np.maximum(np.abs(y), 1.0)).max():.3e}")
    print(f"{cfg} {n} scenes x 2 geometries: {'BIT-IDENTICAL' if not bad else 'DIFFERENT: ' + ', '.join(bad)} ({a} vs {b}; planned {ok:.3f})")
    sys.exit(1 if bad else 0)
