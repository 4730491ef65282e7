"""Stress the two-batches-in-flight mode: many rounds of six different batches planned back to back, every output compared
bit for bit with the plain (one batch in flight) results.  Usage: python tools/pipeline_stress.py [rounds] [noise]"""
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params

R = int(sys.argv[1]) if len(sys.argv) > 1 else 50
noise = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = S.CFG2
dev = torch.device("cuda:0")
pl = Planner(0)
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
batches = []
for k in range(6):
    n = 1536 + 64 * k
    b = S.make_batch(range(1000 * k, 1000 * k + n), cfg)
    batches.append(dict(ref_line=t(b.ref), n_ref=t(np.full(n, b.ref.shape[1], np.int32)), origin_xy=t(b.origin_xy),
                        start_xy=t(b.start_xy), start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy),
                        n_obs=t(b.n_obs)))
torch.cuda.synchronize()
fields = lambda r: {k: v.cpu().numpy() for k, v in vars(r).items() if isinstance(v, torch.Tensor)}
plain = []
for ins in batches:
    r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
    pl.synchronize()
    plain.append(fields(r))
# plain again: is the plain mode itself reproducible?
for k, ins in enumerate(batches):
    r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
    pl.synchronize()
    f = fields(r)
    for name in f:
        if not np.array_equal(plain[k][name].view(np.uint8), f[name].view(np.uint8)):
            print("PLAIN mode not reproducible:", k, name)
side = torch.cuda.Stream(device=dev)
junk = torch.randn(4096, 4096, device=dev)
bad = 0
pl.set_pipeline(True)
for rnd in range(R):
    if noise:
        with torch.cuda.stream(side):
            for _ in range(noise):
                junk = (junk @ junk).tanh()
    with torch.cuda.stream(pl.torch_stream()):
        res = [pl.plan_cycle(p, q, sp, max_pts=M, **ins) for ins in batches]
    pl.synchronize()
    for k, r in enumerate(res):
        f = fields(r)
        for name in f:
            a, b = plain[k][name], f[name]
            if not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
                n = a.shape[0]
                rows = np.nonzero(np.any(a.reshape(n, -1).view(np.uint8) != b.reshape(n, -1).view(np.uint8), axis=1))[0]
                bad += 1
                with np.errstate(all="ignore"):
                    mag = np.nanmax(np.abs(a.reshape(n, -1)[rows].astype(np.float64) - b.reshape(n, -1)[rows].astype(np.float64)))
                print(f"round {rnd} batch {k} field {name}: {rows.size} scenes differ, first {rows[:6]}, max |diff| {mag:.3e}, "
                      f"status plain {plain[k]['status'][rows[:6]]} now {f['status'][rows[:6]]}")
    # the caller on torch's default stream, reading every result right away, dropping it at once; allocator churn between
    churn = []
    for k, ins in enumerate(batches):
        r = pl.plan_cycle(p, q, sp, max_pts=M, **ins)
        f = fields(r)
        del r
        churn.append(torch.empty((rnd * 7919 + k * 104729) % 3000000 + 1, device=dev))
        if len(churn) > 3:
            churn.pop(0)
        for name in f:
            if not np.array_equal(plain[k][name].view(np.uint8), f[name].view(np.uint8)):
                bad += 1
                print(f"round {rnd} batch {k} field {name}: differs (default stream, read at once)")
    # another entry point right behind a pipelined cycle
    r = pl.plan_cycle(p, q, sp, max_pts=M, **batches[2])
    sm, _, _, bsl, _ = pl.frenet_project(**batches[2])
    tgt = pl.frenet_path_to_xy(batches[2]["ref_line"], sm, batches[2]["n_ref"], bsl, r.path_s, r.path_l, r.path_len)[0].cpu().numpy()
    if rnd == 0:
        tgt_first = tgt
    elif not np.array_equal(tgt.view(np.uint8), tgt_first.view(np.uint8)):
        bad += 1
        print(f"round {rnd}: frenet_path_to_xy behind a pipelined cycle differs")
pl.set_pipeline(False)
print(f"{R} rounds, {bad} differing (batch, field) pairs")
