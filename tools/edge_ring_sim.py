"""Prices, on the CPU, the edge-cost kernel's WORK-RING form against the lane mapping that shipped until round 4
(development aid; the numbers quoted in DESIGN.md 3.1 and profiles/r05_edge/README.md).

Shipped form: wavefront = (tile of S scenes, column j), lane = (scene, destination row i); loop over the source rows k, inside
it over the obstacles within longitudinal reach of the column; a scan is executed by the wavefront when ANY lane's obstacle
passes the box test - lanes whose box test fails, and lanes of scenes with fewer obstacles near this column, idle.

Ring form: the dense part (base cost, box tests) stays as it is; an edge with at least one obstacle in reach is pushed as ONE
entry (edge, obstacle mask) into the wavefront's LDS ring; whenever the ring holds 64 entries the wavefront pops them, one per
lane, and every lane scans ITS entry's obstacles in ascending order (the reference's order, path_planning.py:573-582).  A
wave-level scan then idles a lane only when its entry has fewer obstacles than the round's maximum (and in the last, partial
round of a wavefront).  `split`: two rings, entries with one obstacle and entries with more.

Usage: python tools/edge_ring_sim.py [cfg2|cfg5] [scenes] [waves_per_tile]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import scenes as S

cfg = {"cfg2": S.CFG2, "cfg5": S.CFG5}[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (700 if cfg is S.CFG2 else 48)
WPT = int(sys.argv[3]) if len(sys.argv) > 3 else 2          # wavefronts that share a tile's columns (block size)
kw = dict(start_ahead=S.BENCH_START_AHEAD)
b = S.make_batch(range(B), cfg, **kw)
row, col = cfg.row, cfg.col
ss, sl = cfg.sample_s, cfg.sample_l
Sx = 64 // row
tiles = B // Sx
lat = ((row + 1) / 2 - 1 - np.arange(row)) * sl
llo = np.minimum.outer(lat, lat)          # [k][i]
lhi = np.maximum.outer(lat, lat)
t9 = 9 * ss / 10
ps = b.sl_start[:, 0]
J = np.arange(1, col)
s0 = ps[:, None] + J[None, :] * ss                      # B, col-1
s9 = s0 + t9
os_, ol_ = b.sl_obs_s, b.sl_obs_l                       # B, n_obs
near = (os_[:, None, :] > s0[:, :, None] - 6.5) & (os_[:, None, :] < s9[:, :, None] + 6.5)        # B, col-1, m
dx = np.maximum(np.maximum(s0[:, :, None] - os_[:, None, :], os_[:, None, :] - s9[:, :, None]), 0)    # B, col-1, m
wave_scans_old = lane_scans = 0
ring_rounds = ring_scans = ring_scans_split = entries_total = 0
for tl in range(tiles):
    sl_ = slice(tl * Sx, (tl + 1) * Sx)
    dy = np.maximum(np.maximum(llo[None, None] - ol_[sl_, :, None, None], ol_[sl_, :, None, None] - lhi[None, None]), 0)   # S, m, k, i
    box = (dx[sl_][:, :, :, None, None] ** 2 + dy[:, None] ** 2 < 36.5) & near[sl_][:, :, :, None, None]     # S, col-1, m, k, i
    cnt = box.sum(2)                                       # S, col-1, k, i : obstacles the edge scans
    lane_scans += int(cnt.sum())
    # shipped form: per (j, k): the wavefront walks max over scenes of the near count; a scan executes if any lane passes
    nn = near[sl_]                                         # S, col-1, m
    # rank of each near obstacle within its scene/column (the t-th set bit): scans at iteration t execute if any lane's t-th near obstacle passes
    rank = np.cumsum(nn, axis=2) - 1                       # S, col-1, m
    tmax = int(nn.sum(2).max())
    for t in range(tmax):
        sel = nn & (rank == t)                             # S, col-1, m (one m per scene/col at most)
        passes = (box & sel[:, :, :, None, None]).any(2)   # S, col-1, k, i
        wave_scans_old += int(passes.any(axis=(0, 3)).sum())       # per (col, k): any lane of the wavefront
    # ring form: per wavefront (columns dealt round-robin to WPT wavefronts), entries in (j, k, lane) order
    for w in range(WPT):
        c = cnt[:, w::WPT]                                  # S, cols_w, k, i
        order = np.transpose(c, (1, 2, 0, 3)).reshape(-1)   # (j, k, s, i)
        e = order[order > 0]
        entries_total += len(e)
        n = len(e)
        pad = (-n) % 64
        r = np.concatenate([e, np.zeros(pad, int)]).reshape(-1, 64)
        ring_rounds += len(r)
        ring_scans += int(r.max(1).sum())
        one, more = e[e == 1], e[e > 1]
        for q in (one, more):
            if len(q):
                pad = (-len(q)) % 64
                ring_scans_split += int(np.concatenate([q, np.zeros(pad, int)]).reshape(-1, 64).max(1).sum())
edges = tiles * Sx * (col - 1) * row * row
print(f"{cfg.name}: {tiles * Sx} scenes, S = {Sx}, {WPT} wavefronts per tile")
print(f"  lane-scans per edge {lane_scans / edges:.3f}; edges with an obstacle in reach {entries_total / edges:.3f}")
print(f"  shipped mapping : wave-scans {wave_scans_old}, active lanes of a scan {lane_scans / (64 * wave_scans_old):.3f}")
print(f"  ring            : wave-scans {ring_scans} ({ring_scans / wave_scans_old:.3f} of shipped), active lanes {lane_scans / (64 * ring_scans):.3f}, rounds {ring_rounds}")
print(f"  ring, split 1|2+: wave-scans {ring_scans_split} ({ring_scans_split / wave_scans_old:.3f} of shipped), active lanes {lane_scans / (64 * ring_scans_split):.3f}")
