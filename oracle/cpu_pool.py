"""Worker of bench.py's multi-core CPU baseline ("plus a multiprocessing.Pool(nproc) run").

TEST / MEASUREMENT INFRASTRUCTURE ONLY, like the rest of ``oracle/``: it runs ``oracle/ref_port.plan_cycle`` (the
reference-structured NumPy path) on a list of scene seeds inside a pool process.  Kept in its own module so that a
spawned worker imports NumPy and the oracle, not bench.py's torch.
"""
from __future__ import annotations

import contextlib
import io
import time


def warm(_):
    from emplanner_carla_amd import scenes  # noqa: F401
    from oracle import ref_port  # noqa: F401
    return 1


def plan_seeds(job):
    """job = (lattice config name, seeds[, scenes.make_scene options]) -> (scenes done, seconds)."""
    cfg_name, seeds = job[0], job[1]
    scene_kw = job[2] if len(job) > 2 else {}
    from emplanner_carla_amd import scenes as S
    from oracle import ref_port as op
    cfg = getattr(S, cfg_name)
    kw = dict(sampling_res=cfg.sampling_res, row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    t0 = time.perf_counter()
    for sd in seeds:
        sc = S.make_scene(int(sd), cfg, **scene_kw)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                op.plan_cycle([tuple(r) for r in sc.ref], sc.origin_xy, sc.start_xy, sc.start_v, sc.start_a, sc.obs_xy,
                              dp_kwargs=kw, obs_length=cfg.obs_length, obs_width=cfg.obs_width, verbose=False)
        except IndexError:
            pass
    return len(seeds), time.perf_counter() - t0
