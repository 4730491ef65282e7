"""Turn gpurun_out/prof_<TAG>/ (written by tools/profile.sh on the GPU box) into the tracked evidence under
profiles/: the rocprofv3 --kernel-trace --stats table, the per-kernel PMC means (FETCH_SIZE / WRITE_SIZE, each
from its own pass) and the dp_sweep_traffic entry of profiles/counters.json, which bench.py reads for roofline.traffic.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on
gfx950 counts 128-B requests as 64 B for wide coalesced streaming reads, so read bytes = 2 x FETCH_SIZE x 1024;
WRITE_SIZE is taken as is.  Usage: python tools/summarize_profile.py TAG [scenes_per_gpu] [steps] [warmup] [config name]"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
warmup = int(sys.argv[4]) if len(sys.argv) > 4 else 10
config = sys.argv[5] if len(sys.argv) > 5 else "cfg2_40x9_8obs"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _counters  # noqa: E402
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
if os.path.exists(os.path.join(src, "bench_plain.json")):
    shutil.copy(os.path.join(src, "bench_plain.json"), os.path.join(dst, f"{tag}_bench.json"))


def short(name):
    name = name.replace("void ", "").replace("emp::", "")
    return name.split("(")[0]


pmc = collections.defaultdict(dict)
for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    with open(os.path.join(src, f"pmc_{kind}", f"{tag}_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k][counter] = sum(v) / len(v)
        pmc[k]["launches"] = len(v)

# The sweep inside the timed region, from the kernel trace of the same command: its launches in start order are
# [untimed steps][the K timed steps][2 + min(K, 5) of the diagnostic pass, one batch in flight] (bench.py --no-legs).
region = None
trace_csv = os.path.join(src, "trace", f"{tag}_kernel_trace.csv")
if os.path.exists(trace_csv):
    sw = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace_csv))
                 if short(r["Kernel_Name"]).startswith("dp_sweep_kernel")))
    untimed = None
    try:
        line = [ln for ln in open(os.path.join(src, "bench_trace.log")) if ln.startswith("{")][-1]
        untimed = int(json.loads(line)["untimed_steps_before_the_timed_region"])
    except Exception:
        pass
    diag = 2 + min(steps, 5)
    extra = 0              # round 5: one more untimed pass per input batch behind the diagnostic pass (planned fractions)
    blocks, between, pipeline = 1, 0, None   # round 6: the timed block is repeated (timed_blocks); the clock probe's steps follow it
    try:
        d = json.loads(line)
        extra = int(d["config"].get("input_batches", 0))
        blocks = int(d.get("timed_blocks", 1))
        between = int(d.get("untimed_steps_between_the_timed_region_and_the_diagnostic_pass", 0))
        pipeline = d["config"].get("pipeline")
    except Exception:
        pass
    n_timed = steps * blocks
    if untimed is not None and len(sw) == untimed + n_timed + between + diag + extra:
        dur = [(e - b) / 1e3 for b, e in sw]
        timed = dur[untimed:untimed + n_timed]
        alone = dur[untimed + n_timed + between + 2:untimed + n_timed + between + diag]
        ts = sorted(timed)
        region = {"launches_in_trace": len(sw), "untimed": untimed, "timed_blocks": blocks, "timed_launches": n_timed,
                  "timed_mean_us": sum(timed) / len(timed), "timed_median_us": ts[len(ts) // 2],
                  "timed_min_us": min(timed), "timed_max_us": max(timed), "alone_mean_us": sum(alone) / len(alone),
                  "all_mean_us": sum(dur) / len(dur)}
        _counters.upsert("dp_sweep_trace", {"config": config, "scenes_per_gpu": scenes, "pipeline": pipeline,
                                            "timed_launch_us_min_median_max": [round(min(timed), 2), round(ts[len(ts) // 2], 2), round(max(timed), 2)],
                                            "timed_launch_us_mean": round(sum(timed) / len(timed), 2), "timed_launches": n_timed,
                                            "alone_launch_us_mean": round(sum(alone) / len(alone), 2),
                                            "source": f"profiles/{tag}_sweep_regions.json (rocprofv3 --kernel-trace of python bench.py --steps {steps} "
                                                      f"--warmup {warmup}: the sweep's launches inside the timed region)"},
                         ("config", "scenes_per_gpu", "pipeline"))
    else:
        region = {"launches_in_trace": len(sw), "untimed": untimed, "note": "launch count does not match untimed + steps + diagnostic pass"}

stats = list(csv.DictReader(open(os.path.join(dst, f"{tag}_kernel_stats.csv"))))
lines = [f"# rocprofv3 summary `{tag}` - `python bench.py --steps {steps} --warmup {warmup}` on one MI355X ({scenes} scenes/GPU)", "",
         f"`rocprofv3 --kernel-trace --stats` (all launches: {warmup} warm-up steps and the untimed settling steps, {steps} timed, the per-kernel diagnostic pass):", "",
         "| kernel | calls | mean us | % of GPU time |", "|---|---|---|---|"]
for r in stats:
    lines.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.2f} |")
if region and "timed_mean_us" in region:
    E_bytes = None
    lines += ["", f"dp_sweep launches of the same trace by region (start order: {region['untimed']} untimed, {region['timed_blocks']} x {steps} timed, the clock probe's, the diagnostic pass): "
                  f"timed region mean {region['timed_mean_us']:.2f} us (min {region['timed_min_us']:.2f}, median {region['timed_median_us']:.2f}, max {region['timed_max_us']:.2f}); "
                  f"diagnostic pass, one batch in flight: {region['alone_mean_us']:.2f} us; all launches {region['all_mean_us']:.2f} us"]
    json.dump(region, open(os.path.join(dst, f"{tag}_sweep_regions.json"), "w"), indent=1)
elif region:
    lines += ["", f"dp_sweep launches by region: {region}"]
lines += ["", "PMC passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, separate runs), per-launch means:", "",
          "| kernel | FETCH_SIZE KiB | read MB (x2 gfx950 correction) | WRITE_SIZE KiB | write MB |", "|---|---|---|---|---|"]
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w") as f:
    f.write("kernel,launches,FETCH_SIZE_KiB,WRITE_SIZE_KiB,read_bytes_corrected,write_bytes\n")
    for k, v in sorted(pmc.items()):
        fs, ws = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        f.write(f"{k},{v.get('launches', 0)},{fs:.1f},{ws:.1f},{2 * fs * 1024:.0f},{ws * 1024:.0f}\n")
        lines.append(f"| `{k}` | {fs:.0f} | {2 * fs * 1024 / 1e6:.1f} | {ws:.0f} | {ws * 1024 / 1e6:.1f} |")
sweep = next((k for k in pmc if k.startswith("dp_sweep_kernel")), None)
if sweep:
    t = {"config": config, "scenes_per_gpu": scenes,
         "hbm_bytes_per_launch": int(2 * pmc[sweep].get("FETCH_SIZE", 0) * 1024 + pmc[sweep].get("WRITE_SIZE", 0) * 1024),
         "source": f"profiles/{tag}_pmc.csv: 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 of {sweep} (KiB counters, separate rocprofv3 "
                   f"--pmc passes; gfx950 FETCH_SIZE counts 128-B requests as 64 B, MI355X_MICROARCH.md)"}
    _counters.upsert("dp_sweep_traffic", t, ("config", "scenes_per_gpu"))
    lines += ["", f"dp_sweep HBM traffic per launch: {t['hbm_bytes_per_launch'] / 1e6:.1f} MB"]
open(os.path.join(dst, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
