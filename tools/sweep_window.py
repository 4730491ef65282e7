"""What runs beside the DP sweep in the staged step: from a rocprofv3 --kernel-trace CSV of staged steps, for every
steady-state launch of dp_sweep_kernel its duration and the kernels of the OTHER queue whose execution overlaps its window
(fraction of the window each covers).  Usage: python tools/sweep_window.py <kernel_trace.csv> [skip_first] [out.json]"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("void ", "").replace("emp::", "").split("(")[0].split("<")[0]
rows.sort(key=lambda r: r["s"])
sweeps = [r for r in rows if r["n"].startswith("dp_sweep_kernel")][skip:]
others = [r for r in rows if not r["n"].startswith("dp_sweep_kernel")]
stats = defaultdict(lambda: {"launches": 0, "dur_us": 0.0, "cover": defaultdict(float)})
per = []
j0 = 0
for sw in sweeps:
    while j0 < len(others) and others[j0]["e"] < sw["s"] - 2_000_000:
        j0 += 1
    cover = defaultdict(float)
    j = j0
    while j < len(others) and others[j]["s"] < sw["e"]:
        o = others[j]
        ov = min(o["e"], sw["e"]) - max(o["s"], sw["s"])
        if ov > 0:
            cover[o["n"]] += ov / (sw["e"] - sw["s"])
        j += 1
    dur = (sw["e"] - sw["s"]) / 1e3
    key = "+".join(sorted(k for k, v in cover.items() if v > 0.2)) or "nothing"
    st = stats[key]
    st["launches"] += 1
    st["dur_us"] += dur
    for k, v in cover.items():
        st["cover"][k] += v
    per.append(dur)
out = {"sweep_launches": len(sweeps), "mean_us": sum(per) / max(len(per), 1), "by_neighbour": {}}
print(f"{len(sweeps)} sweep launches, mean {out['mean_us']:.2f} us")
for key, st in sorted(stats.items(), key=lambda kv: -kv[1]["launches"]):
    n = st["launches"]
    cov = {k: round(v / n, 2) for k, v in st["cover"].items()}
    out["by_neighbour"][key] = {"launches": n, "mean_us": round(st["dur_us"] / n, 2), "mean_window_cover": cov}
    print(f"  beside {key:60s} {n:5d} launches  {st['dur_us'] / n:6.2f} us   cover {cov}")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
