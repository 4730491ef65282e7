"""Who shares a wavefront in the edge-cost kernel, priced on the CPU: the kernel's lane mapping is
wave = (tile of S = 7 scenes, column j), lane = (scene, destination row i), loop over the source rows k and, inside, over the
obstacles within longitudinal reach of the column; a scan is executed by the wavefront when ANY lane's obstacle passes the box
test, and the lanes that fail it idle.  For a permutation of the scenes: executed wave-level scans and the active-lane
fraction of those scans.  Compared: input order, scenes sorted by several keys, a greedy grouping on the per-column reach
signature, and the bound of seven IDENTICAL scenes per wavefront.
Usage: python tools/edge_grouping_sim.py [scenes] [cfg2|cfg5]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import scenes as S
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = S.CFG5 if (len(sys.argv) > 2 and sys.argv[2] == "cfg5") else S.CFG2
b = S.make_batch(range(B), cfg)
row, col, ss, sl = cfg.row, cfg.col, cfg.sample_s, cfg.sample_l
Sx = 64 // row
lat = ((row + 1) / 2 - 1 - np.arange(row)) * sl
llo, lhi = np.minimum.outer(lat, lat), np.maximum.outer(lat, lat)          # [k][i]
t9 = 9 * ss / 10
ps = b.sl_start[:, 0]
s0 = ps[:, None] + np.arange(1, col)[None, :] * ss                            # B, col-1
os_, ol_ = b.sl_obs_s, b.sl_obs_l                                             # B, K
K = os_.shape[1]
valid = np.arange(K)[None, :] < b.n_obs[:, None]
near = (os_[:, None, :] > s0[:, :, None] - 6.5) & (os_[:, None, :] < s0[:, :, None] + t9 + 6.5) & valid[:, None, :]   # B,J,K
dx = np.maximum(np.maximum(s0[:, :, None] - os_[:, None, :], os_[:, None, :] - (s0 + t9)[:, :, None]), 0)             # B,J,K
dy = np.maximum(np.maximum(llo[None, None] - ol_[:, :, None, None], ol_[:, :, None, None] - lhi[None, None]), 0)      # B,K,k,i
J = col - 1
# rank of obstacle m among the near ones of (scene, column): the loop iteration it is scanned in
rank = np.cumsum(near, -1) - 1                                                # B,J,K
pop = near.sum(-1)                                                            # B,J
maxit = int(pop.max())
# passes[b, j, t, k, i]: lane (b, i) passes the box test in iteration t of source row k
passes = np.zeros((B, J, maxit, row, row), bool)
for m in range(K):
    box = (dx[:, :, m, None, None] ** 2 + dy[:, None, m, :, :] ** 2) < 36.5   # B,J,k,i
    box &= near[:, :, m, None, None]
    for t in range(maxit):
        sel = near[:, :, m] & (rank[:, :, m] == t)                            # B,J
        passes[:, :, t] |= box & sel[:, :, None, None]
lane_scans = passes.sum()

def price(order):
    n = (len(order) // Sx) * Sx
    p = passes[order[:n]].reshape(n // Sx, Sx, J, maxit, row, row)
    executed = p.any(axis=(1, 5))                                             # tile, J, t, k
    ls = p.sum()
    return int(executed.sum()), float(ls / (executed.sum() * 64.0))

res = {}
res["input order"] = price(np.arange(B))
# bound: every wavefront holds seven copies of one scene
ex1 = passes.any(axis=4)                                                      # B,J,t,k  (any lane of the scene's row group)
res["bound: 7 identical scenes per wavefront"] = (int(ex1.sum() / Sx), float(passes.sum() * Sx / (ex1.sum() * 64.0)) / Sx * 1.0)
res["bound: 7 identical scenes per wavefront"] = (int(round(ex1.sum() / Sx)), float(lane_scans / (ex1.sum() / Sx * 64.0)))
# keys
work = passes.any(axis=4).sum(axis=(1, 2, 3))                                 # executed scans of a scene alone
res["sorted by the scene's own scan count"] = price(np.argsort(-work, kind="stable"))
first = np.where(valid, os_ - ps[:, None], 1e9).min(1)
res["sorted by the nearest obstacle's station"] = price(np.argsort(first, kind="stable"))
res["sorted by the obstacle count"] = price(np.argsort(-b.n_obs, kind="stable"))
# greedy grouping on the per-column executed-scan profile: seed = heaviest free scene, then the six free scenes whose profile
# adds the fewest extra wave-level scans (cost = sum_j max(profile) - the seed's)
prof = passes.any(axis=4).sum(axis=3)                                         # B,J,t -> scans per (column, iteration) over k
prof = prof.reshape(B, -1).astype(np.int32)
free = np.ones(B, bool)
order = []
idx_sorted = np.argsort(-work, kind="stable")
cap = 2048                                                                     # candidates looked at per pick (heaviest first)
for seed in idx_sorted:
    if not free[seed]:
        continue
    free[seed] = False
    group, cur = [seed], prof[seed].copy()
    for _ in range(Sx - 1):
        cand = np.flatnonzero(free)
        if len(cand) == 0:
            break
        cand = cand[np.argsort(-work[cand], kind="stable")[:cap]] if len(cand) > cap else cand
        extra = np.maximum(prof[cand], cur[None, :]).sum(1) - cur.sum() - 0.5 * prof[cand].sum(1)     # added scans, minus a bonus for absorbing heavy scenes
        pick = cand[int(np.argmin(extra))]
        free[pick] = False
        group.append(pick)
        cur = np.maximum(cur, prof[pick])
    order.extend(group)
res["greedy grouping on the (column, iteration) scan profile"] = price(np.array(order))
base = res["input order"][0]
print(f"{B} scenes, {cfg.name}: lane-level scans {lane_scans}")
for k, (ex, util) in res.items():
    print(f"  {k:62s} wave-level scans {ex:9d} ({ex / base:5.3f} of input order)   active lanes of a scan {util:5.3f}")
json.dump({k: {"wave_scans": v[0], "active_lane_frac": v[1]} for k, v in res.items()}, open("/tmp/edge_grouping_sim.json", "w"), indent=1)
