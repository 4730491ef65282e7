"""Pricing of VERDICT r05 item 2(a) before building it: path-QP wavefronts whose 8-lane groups PULL the next scene when they converge
("persistent groups"), and the alternative of a second pass over the compacted stragglers.  Input: the per-scene interior-point
iteration counts of a 4096-scene batch (tools/qp_iters_probe.py on the GPU box writes gpurun_out/qp_cases.npz).  Cost model, in units
of one interior-point iteration of a WAVEFRONT (all 64 lanes execute it whether one group or eight are live): an iteration costs 1, a
refill (load the DP path, QP bounds, B-spline set-up, start point: executed by the whole wavefront whenever ANY of its groups needs a
scene) costs s - 0.8-0.9 by the kernel's instruction counts (ISA of cycle_qp_rows_kernel<8,3>: ~3350 dynamic instructions an iteration,
~2900 before the loop), 0.3-0.4 if the start point's factor-and-solve were folded into the first iteration.

    python tools/qp_persistent_sim.py [gpurun_out/qp_cases.npz]

Result (profiles/r06_qp/persistent_groups_sim.txt): with s = 0.8 no policy saves anything (refills cost what the imbalance costs); with
s = 0.2 the best is -7..-10 % of the kernel's instructions while the kernel ALONE gets longer (fewer wavefronts, 25-41 iterations
each instead of 17); a second pass over compacted stragglers saves 8-9 %.  The 25 % between a wavefront's 10.0 iterations and its
scenes' 7.5 is not reachable by scheduling - what is left is the cost of an iteration."""
import heapq
import sys

import numpy as np


def simulate(it, waves, refill_at, s):
    B, nxt = len(it), 0
    W = [{"t": 0.0, "rem": [0] * 8, "cost": 0.0} for _ in range(waves)]
    pq = [(0.0, w) for w in range(waves)]
    heapq.heapify(pq)
    done = []
    while pq:
        _, w = heapq.heappop(pq)
        wv = W[w]
        idle = [g for g in range(8) if wv["rem"][g] == 0]
        live = 8 - len(idle)
        if nxt < B and (len(idle) >= refill_at or live == 0):
            for g in idle:
                if nxt < B:
                    wv["rem"][g] = it[nxt]
                    nxt += 1
            wv["t"] += s
            wv["cost"] += s
            heapq.heappush(pq, (wv["t"], w))
            continue
        if live == 0:
            done.append(wv["t"])
            continue
        wv["rem"] = [max(r - 1, 0) for r in wv["rem"]]
        wv["t"] += 1
        wv["cost"] += 1
        heapq.heappush(pq, (wv["t"], w))
    return sum(w["cost"] for w in W), max(done)


def main():
    it = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qp_cases.npz")["iters"].astype(int)
    B = len(it)
    cur = [int(it[i:i + 8].max()) for i in range(0, B, 8)]
    print(f"{B} scenes: mean iterations of a scene {it.mean():.2f}, of a wavefront of eight {np.mean(cur):.2f}, longest {max(cur)}; "
          f"ideal wave-iterations {it.sum() / 8:.0f}, today {sum(cur)}")
    for s in (0.2, 0.4, 0.8):
        base = sum(cur) + len(cur) * s
        print(f"refill cost s = {s}: today {base:.0f} units, longest wavefront {max(cur) + s:.1f}")
        for waves in (256, 192, 128):
            for k in (1, 2, 4):
                tot, longest = simulate(it, waves, k, s)
                print(f"   {waves} persistent wavefronts, refill when {k} group(s) idle: {tot:.0f} units ({tot / base:.3f} of today), "
                      f"longest wavefront {longest:.1f}")
    s, s2 = 0.8, 0.3
    base = sum(cur) + len(cur) * s
    for K in (7, 8, 9, 10):
        p1 = sum(min(c, K) for c in cur) + len(cur) * s
        rem = np.maximum(it - K, 0)
        rem = rem[rem > 0]
        p2 = sum(int(rem[i:i + 8].max()) for i in range(0, len(rem), 8)) + (len(rem) + 7) // 8 * s2
        print(f"two passes, first capped at {K} iterations: {len(rem)} stragglers, {p1 + p2:.0f} units ({(p1 + p2) / base:.3f} of today)")


if __name__ == "__main__":
    main()
