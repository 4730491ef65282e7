"""Per-kernel means of the SQ counter passes written by tools/pmc_sq.sh -> <dir>/summary.csv and a table on stdout.

Derived columns (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles per wave; lanes_active = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) is the mean fraction of a
wavefront's lanes enabled while a VALU instruction executes (both count instruction issue quad-cycles; the
thread counter weighs them by the exec mask: the min-plus sweep, 63 of 64 lanes live, reads 0.956)."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
# optional: TAG CONFIG SCENES SCENE_DIST -> copy the summary to profiles/<TAG>_sq.csv and record the edge kernel's counters
record = sys.argv[2:6] if len(sys.argv) >= 6 else None
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
dur = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("emp::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"], r["Workgroup_Size"], r["Grid_Size"])
for path in sorted(glob.glob(os.path.join(root, "p1", "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("void ", "").replace("emp::", "").split("(")[0]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
names = sorted({c for v in agg.values() for c in v})
with open(os.path.join(root, "summary.csv"), "w") as f:
    f.write("kernel,launches,mean_us_under_pmc,vgpr,agpr,sgpr,lds_bytes,wg_size,grid," + ",".join(names) + ",lanes_active_frac,valu_insts_per_wave\n")
    for k in sorted(agg):
        v = {c: sum(x) / len(x) for c, x in agg[k].items()}
        n = max(len(x) for x in agg[k].values())
        la = ""
        if v.get("SQ_ACTIVE_INST_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
            la = f"{v['SQ_THREAD_CYCLES_VALU'] / (64.0 * v['SQ_ACTIVE_INST_VALU']):.4f}"
        ipw = f"{v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.1f}" if v.get("SQ_WAVES") and v.get("SQ_INSTS_VALU") else ""
        d = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        f.write(",".join([k.replace(",", ";"), str(n), f"{d:.2f}", *meta[k]] + [f"{v.get(c, float('nan')):.6g}" for c in names] + [la, ipw]) + "\n")
print(open(os.path.join(root, "summary.csv")).read())
if record:
    import shutil
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _counters
    tag, config, scenes, scene_dist = record
    dst = os.path.join(_counters.ROOT, "profiles", f"{tag}_sq.csv")
    shutil.copy(os.path.join(root, "summary.csv"), dst)
    STEP = {"frenet_project_wave_kernel": "project", "dp_edge": "dp_edge", "dp_sweep_kernel": "dp_sweep", "dp_enrich_wave_kernel": "dp_enrich",
            "cycle_qp_rows_kernel": "path_qp", "cycle_cartesian_rows_kernel": "to_cartesian"}
    kernels = {}
    for k in sorted(agg):
        for prefix, name in STEP.items():
            if k.startswith(prefix) and "wide" not in k:
                v = {c: sum(x) / len(x) for c, x in agg[k].items()}
                kernels[name] = {"kernel": k.split("<")[0], "insts_valu": int(v["SQ_INSTS_VALU"]), "valu_busy_quad_cycles": int(v["SQ_ACTIVE_INST_VALU"]),
                                 "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4),
                                 "mean_us_under_pmc": round(sum(dur[k]) / len(dur[k]), 1) if dur.get(k) else None}
    if len(kernels) == 6:      # the whole planning cycle was in the pass: what bench.py's roofline_step adds up
        _counters.upsert("step_valu_counters", {
            "config": config, "scenes_per_gpu": int(scenes), "scene_dist": scene_dist, "kernels": kernels,
            "source": f"profiles/{tag}_sq.csv (rocprofv3 --pmc SQ passes of bench.py --no-pipeline, tools/pmc_sq.sh: every kernel alone)"},
            ("config", "scenes_per_gpu", "scene_dist"))
    for k in sorted(agg):
        if k.startswith("dp_edge_kernel") or k.startswith("dp_edge_ring_kernel"):
            v = {c: sum(x) / len(x) for c, x in agg[k].items()}
            _counters.upsert("dp_edge_counters", {
                "config": config, "scenes_per_gpu": int(scenes), "scene_dist": scene_dist,
                "insts_valu": int(v["SQ_INSTS_VALU"]), "insts_salu": int(v["SQ_INSTS_SALU"]),
                "valu_busy_quad_cycles": int(v["SQ_ACTIVE_INST_VALU"]),
                "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4),
                "valu_issue_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / (v["SQ_BUSY_CYCLES"] / 32.0), 4),
                "source": f"profiles/{tag}_sq.csv (rocprofv3 --pmc SQ passes of bench.py --no-pipeline, tools/pmc_sq.sh)"},
                ("config", "scenes_per_gpu", "scene_dist"))
        if k.startswith("speed_dp_kernel"):
            v = {c: sum(x) / len(x) for c, x in agg[k].items()}
            se_cycles = v["SQ_BUSY_CYCLES"] / 32.0            # SQ_BUSY_CYCLES sums over the 32 shader engines
            _counters.upsert("speed_dp_counters", {
                "scenes_per_gpu": int(scenes), "obstacle_slots": 16,
                "insts_valu": int(v["SQ_INSTS_VALU"]), "insts_salu": int(v["SQ_INSTS_SALU"]),
                "valu_busy_quad_cycles": int(v["SQ_ACTIVE_INST_VALU"]),
                "lanes_active_frac": round(v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"]), 4),
                "valu_issue_busy_frac": round(v["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / se_cycles, 4),
                "mean_waves_per_simd": round(v["SQ_WAVE_CYCLES"] * 4 / 1024.0 / se_cycles, 2),
                "mean_us_under_pmc": round(sum(dur[k]) / len(dur[k]), 1) if dur.get(k) else None,
                "source": f"profiles/{tag}_sq.csv (rocprofv3 --pmc SQ passes of tools/st_microbench.py, tools/pmc_sq_cmd.sh)"},
                ("scenes_per_gpu", "obstacle_slots"))

