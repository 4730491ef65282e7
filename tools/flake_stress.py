"""Hunt for the rare mismatch of tests/test_gpu_fullsize.py::test_pipelined_cycles_equal_plain_cycles: batch 1 of that test
planned in pipelined mode from torch's default stream and read right away, many times, against the plain result."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, qp_params, smooth_params
R = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfg = S.CFG2
dev = torch.device("cuda:0")
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
batches = []
for k in range(3):
    n = 1536 + 64 * k
    b = S.make_batch(range(1000 * k, 1000 * k + n), cfg)
    batches.append(dict(ref_line=t(b.ref), n_ref=t(np.full(n, b.ref.shape[1], np.int32)), origin_xy=t(b.origin_xy),
                        start_xy=t(b.start_xy), start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs)))
FIELDS = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")
bad = 0
for outer in range(4):
    pl = Planner(0)
    plain = []
    for ins in batches:
        r = pl.plan_cycle(p, q, sp, **ins)
        pl.synchronize()
        plain.append({k: getattr(r, k).cpu().numpy() for k in FIELDS})
    pl.set_pipeline(True)
    for rnd in range(R):
        for k, ins in enumerate(batches):
            r = pl.plan_cycle(p, q, sp, **ins)
            got = {kk: getattr(r, kk).cpu().numpy() for kk in ("path_len", "status", "path_s", "traj_len")}
            for name in got:
                if not np.array_equal(plain[k][name], got[name], equal_nan=True):
                    rows = np.nonzero(np.any(plain[k][name].reshape(len(got[name]), -1) != got[name].reshape(len(got[name]), -1), axis=1))[0]
                    bad += 1
                    i = int(rows[0])
                    print(f"outer {outer} round {rnd} batch {k} {name}: scenes {rows[:6]}; scene {i}: status {plain[k]['status'][i]} vs {got['status'][i]}, "
                          f"path_len {plain[k]['path_len'][i]} vs {got['path_len'][i]}, path_s {plain[k]['path_s'][i][:6]} vs {got['path_s'][i][:6]}")
                    break
    pl.set_pipeline(False)
    del pl
print("mismatches:", bad)
