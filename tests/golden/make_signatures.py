"""The reference's public surface on the hot path, as data: for every module the drop-in package mirrors, every
function (argument names in order, defaults as source text) and every class (its methods the same way), read from
the reference's source with ``ast`` - nothing is imported or executed, and no source text is kept beyond names and
default literals.  tests/test_cabi_and_host.py compares the drop-ins against the committed file; the live test in
tests/test_reference_live.py re-derives it when /root/reference is present.

Run:  python tests/golden/make_signatures.py        (writes tests/golden/signatures.json)
"""
from __future__ import annotations

import ast
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = "/root/reference"
#: reference module (path under the reference root) -> (drop-in module, scope).  "full": every name of the reference module
#: must exist in the drop-in (the planner modules).  "input_side": the module is a downstream
#: consumer that SURVEY.md section 2 row 6 marks OUT OF SCOPE (it reads live carla.Vehicle state); section 8f row 3 mirrors
#: only the lateral controllers' constructor + cal_vehicle_info + _control, and the test holds what the drop-in DOES define
#: to the reference's names and signatures.
MODULES = {
    "planner/path_planning.py": ("emplanner_carla_amd.planner.path_planning", "full"),
    "planner/planning_utils.py": ("emplanner_carla_amd.planner.planning_utils", "full"),
    "planner/speed_planning_test.py": ("emplanner_carla_amd.planner.speed_planning_test", "full"),
    "controller/controller.py": ("emplanner_carla_amd.controller.controller", "input_side"),
}


def _args(fn: ast.FunctionDef):
    a = fn.args
    pos = [x.arg for x in a.posonlyargs + a.args]
    defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
    out = [[n, d] for n, d in zip(pos, defaults)]
    if a.vararg:
        out.append(["*" + a.vararg.arg, None])
    for k, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append([k.arg, None if d is None else ast.unparse(d)])
    if a.kwarg:
        out.append(["**" + a.kwarg.arg, None])
    return out


def surface_of_source(text: str):
    tree = ast.parse(text)
    out = {"functions": {}, "classes": {}}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            out["functions"][node.name] = _args(node)
        elif isinstance(node, ast.ClassDef):
            out["classes"][node.name] = {m.name: _args(m) for m in node.body if isinstance(m, ast.FunctionDef)}
    return out


def reference_surface():
    return {rel: dict(dropin=mod, scope=scope,
                      **surface_of_source(open(os.path.join(REFERENCE_ROOT, rel), encoding="utf-8").read()))
            for rel, (mod, scope) in MODULES.items()}


def main():
    out = reference_surface()
    path = os.path.join(HERE, "signatures.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(m["functions"]) + sum(len(c) for c in m["classes"].values()) for m in out.values())
    print(path, n, "callables")


if __name__ == "__main__":
    sys.exit(main())
