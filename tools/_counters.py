"""profiles/counters.json: per-workload figures taken from committed rocprofv3 counter passes, read by bench.py for
`roofline.traffic` and `roofline_dp_edge.executed_*` (never measured inside bench.py itself: PMC passes serialise the
kernels and cannot run inside the timed region).  Each entry names the profile it came from."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "profiles", "counters.json")


def upsert(kind, entry, keys):
    try:
        data = json.load(open(PATH))
    except Exception:
        data = {}
    rows = [e for e in data.get(kind, []) if not all(e.get(k) == entry.get(k) for k in keys)]
    rows.append(entry)
    data[kind] = sorted(rows, key=lambda e: json.dumps([e.get(k) for k in keys]))
    json.dump(data, open(PATH, "w"), indent=1, sort_keys=True)
    open(PATH, "a").write("\n")
