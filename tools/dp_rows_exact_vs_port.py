"""CPU-only check (round-2 review): DP rows of oracle/exact.py - the kernels' arithmetic, closed-form quirk term included,
to which the GPU is bit-identical on 65 536 scenes - against oracle/ref_port.py, the reference's own floating-point route
(per-edge 6x6 inverse, per-sample sums), scene by scene.  A near-tie in the DP argmin could in principle fall the other way
between the two; every scene whose rows differ is listed with the cost gap that decided it.
Usage: python tools/dp_rows_exact_vs_port.py [N] [processes] [first_seed]  -> profiles/r03_dp_rows_exact_vs_port.json"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NPROC = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, len(os.sched_getaffinity(0)))
SEED0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0


def one(seed):
    from emplanner_carla_amd import scenes as S
    from oracle import exact as ex
    from oracle import ref_port as op
    cfg = S.CFG2
    b = S.make_batch([seed], cfg)
    k = int(b.n_obs[0])
    st = b.sl_start[0]
    xrows, xfeas, _ = ex.dp_plan(b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start, cfg.row, cfg.col, cfg.sample_s, cfg.sample_l, cfg.sampling_res)
    cost, pre = op.dp_tables(list(b.sl_obs_s[0, :k]), list(b.sl_obs_l[0, :k]), st[0], st[1], st[2], st[3], row=cfg.row, col=cfg.col,
                             sample_s=cfg.sample_s, sample_l=cfg.sample_l)
    rows, feas = op.dp_backtrack(cost, pre, verbose=False)
    same = list(xrows[0].astype(int)) == [int(r) for r in rows]
    last = np.sort(cost[:, -1])
    return seed, same, bool(xfeas[0]) == bool(feas), float(last[1] - last[0]) / max(float(last[0]), 1.0), [int(r) for r in rows], [int(r) for r in xrows[0]]


if __name__ == "__main__":
    t0 = time.time()
    flips, feas_bad = [], 0
    with mp.get_context("spawn").Pool(NPROC) as pool:
        for seed, same, feas_ok, gap, rows, xrows in pool.imap_unordered(one, range(SEED0, SEED0 + N), chunksize=4):
            feas_bad += not feas_ok
            if not same:
                flips.append(dict(seed=seed, relative_gap_of_the_two_best_terminal_costs=gap, port_rows=rows, exact_rows=xrows))
    rep = dict(config="cfg2_40x9_8obs", scenes=N, first_seed=SEED0, scenes_with_different_rows=len(flips), feasibility_mismatch=feas_bad,
               flips=sorted(flips, key=lambda d: d["seed"]), seconds=round(time.time() - t0, 1), processes=NPROC)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_dp_rows_exact_vs_port.json")
    json.dump(rep, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k != "flips"}))
    for f in rep["flips"]:
        print("   flip:", json.dumps(f))
