"""Print the kernel timeline of a few steady-state steps from a rocprofv3 kernel trace CSV (which kernel ran when, on
which queue): python tools/timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
count = int(sys.argv[3]) if len(sys.argv) > 3 else 26
t0 = int(rows[first]["Start_Timestamp"])
for r in rows[first:first + count]:
    name = r["Kernel_Name"].replace("void ", "").replace("emp::", "").split("(")[0][:30]
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print(f"{name:32s} q{r['Queue_Id']} {s:9.1f} {e:9.1f} {e - s:7.1f}")
