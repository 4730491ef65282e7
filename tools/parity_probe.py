"""Stage-by-stage look at single scenes of tools/parity_sweep.py: where does the device leave the port?
Usage: python tools/parity_probe.py SEED [SEED ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
from oracle import ref_port as op

cfg = S.CFG2
p = dp_params_from_cfg(cfg)
M = max_path_points(p)
pl = Planner(0)
np.set_printoptions(precision=6, linewidth=200)
for seed in [int(x) for x in sys.argv[1:]]:
    b = S.make_batch([seed], cfg)
    nk = int(b.n_obs[0])
    P = b.ref.shape[1]
    nref = np.full(1, P, np.int32)
    out = op.plan_cycle(b.ref[0], tuple(b.origin_xy[0]), tuple(b.start_xy[0]), tuple(b.start_v[0]), tuple(b.start_a[0]),
                        [tuple(o) for o in b.obs_xy[0, :nk]],
                        dp_kwargs=dict(row=cfg.row, col=cfg.col, sample_s=cfg.sample_s, sample_l=cfg.sample_l, sampling_res=cfg.sampling_res),
                        obs_length=cfg.obs_length, obs_width=cfg.obs_width, verbose=False)
    r = pl.plan_cycle(p, qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params(), max_pts=M, ref_line=b.ref,
                      n_ref=nref, origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v, start_a=b.start_a, obs_xy=b.obs_xy,
                      n_obs=b.n_obs)
    print(f"== seed {seed}: device status {int(r.status[0])}, port qp {out.get('qp_status')} smooth {out.get('smooth_status')}")
    sm, os_, ol_, bsl, st = pl.frenet_project(b.ref, nref, b.origin_xy, b.start_xy, b.start_v, b.start_a, b.obs_xy, b.n_obs)
    print("  s_map max |diff|", np.abs(sm[0] - np.asarray(out["s_map"])).max(), " begin_sl diff", bsl[0] - np.array([out["begin_s"], out["begin_l"]]))
    k = int(r.path_len[0])
    ps, pll = np.asarray(out["path_s"]), np.asarray(out["path_l"])
    print("  path_len", k, len(ps), " path_l max |diff|", np.abs(r.path_l[0, :k] - pll[:k]).max() if k == len(ps) else None)
    # Cartesian targets: device on the device's path and on the port's path
    tgt_port = np.array([(q[0], q[1]) for q in out["target_xy"]])
    pad = lambda a: np.concatenate([a, np.zeros(M - len(a))])[None, :]
    for name, s_in, l_in in (("device path", r.path_s[0, :k], r.path_l[0, :k]), ("port path", ps, pll)):
        t, no, st2 = pl.frenet_path_to_xy(b.ref, sm, nref, bsl, pad(s_in), pad(l_in), np.array([len(s_in)], np.int32))
        n = int(no[0])
        d = np.abs(t[0, :n] - tgt_port[:n]).max(axis=1) if n == len(tgt_port) else None
        print(f"  target_xy from {name}: n {n} vs {len(tgt_port)}; max |diff| {None if d is None else d.max():.3e} at point {None if d is None else int(d.argmax())}")
        if d is not None and d.max() > 1e-9:
            print("     per point", d)
    # smoothing of the PORT's targets on the device
    tp = np.zeros((1, M + 1, 2))
    tp[0, :len(tgt_port)] = tgt_port
    so, it, st3 = pl.smooth_line(smooth_params(), tp, np.array([len(tgt_port)], np.int32))
    want = np.asarray(out["trajectory"])
    m = len(want)
    print("  smooth_line(port targets): iters", int(it[0]), "status", int(st3[0]), " max |xy diff|", np.abs(so[0, :m, :2] - want[:, :2]).max(),
          " theta", np.abs(so[0, :m, 2] - want[:, 2]).max(), " kappa", np.abs(so[0, :m, 3] - want[:, 3]).max())
    mt = int(r.traj_len[0])
    if mt == m:
        d = np.abs(r.traj[0, :m] - want)
        print("  cycle trajectory vs port: max |diff| x,y,theta,kappa", d.max(axis=0), " at points", d.argmax(axis=0))
        print("    port kappa[:4]", want[:4, 3], " device", r.traj[0, :4, 3])
