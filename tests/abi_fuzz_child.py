"""Child process of tests/test_gpu_abi_fuzz.py: raw ctypes calls into libemplanner.so with host pointers and hostile
arguments - counts that are negative, zero or beyond the documented limits, NULL where a pointer is required and where
it is optional, capacities of 1.  Contract (include/emplanner.h): a call returns EMP_OK or a negative emp_error and
emp_last_error() has the text; it never crashes, never corrupts the context (a clean call afterwards still yields the
clean answer), and per-scene trouble is a status bit, not an error.  Prints one line per probe and 'ABI-FUZZ-OK'."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from emplanner_carla_amd import _lib as L  # noqa: E402
from emplanner_carla_amd import scenes as S  # noqa: E402
from emplanner_carla_amd.api import dp_params_from_cfg, max_path_points, qp_params, smooth_params  # noqa: E402

lib = L.load()
h = C.c_void_p()
assert lib.emp_create(0, C.byref(h)) == 0
cfg = S.CFG2
b = S.make_batch(range(24), cfg)
B, P, MO = len(b), b.ref.shape[1], b.obs_xy.shape[1]
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=5, obs_width=5), smooth_params()
M = max_path_points(p)
ptr = lambda a: C.c_void_p(a.ctypes.data) if a is not None else None
probes = errors = 0


def expect(rc, what, allow_ok=False):
    global probes, errors
    probes += 1
    msg = lib.emp_last_error(h)
    if rc < 0:
        errors += 1
        assert msg, f"{what}: error {rc} without a message"
    else:
        assert allow_ok and rc == 0, f"{what}: accepted (rc {rc})"
    print(f"{what}: rc {rc} {msg.decode()[:70] if rc < 0 and msg else ''}")


# ---- emp_dp_plan
obs_s, obs_l, n_obs, start = [np.ascontiguousarray(x) for x in (b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)]
rows, mc, st = np.zeros((B, p.col)), np.zeros(B), np.zeros(B, np.int32)


def dp(pp=p, nB=B, mo=MO, os_=obs_s, ol_=obs_l, no=n_obs, s0=start, mode=1, r=rows, m=mc, s=st):
    return lib.emp_dp_plan(h, C.byref(pp) if pp is not None else None, nB, mo, ptr(os_), ptr(ol_), ptr(no), ptr(s0), mode, ptr(r),
                           ptr(m), ptr(s), L.EMP_HOST)


assert dp() == 0
clean = rows.copy()
for field, bad in (("row", 0), ("row", -3), ("row", 1025), ("col", 0), ("col", -1), ("col", 5000), ("sample_s", 0.0),
                   ("sample_s", -2.5), ("sample_l", 0.0), ("sampling_res", 0.0)):
    pp = dp_params_from_cfg(cfg)
    setattr(pp, field, bad)
    expect(dp(pp=pp), f"emp_dp_plan {field}={bad}")
pp = dp_params_from_cfg(cfg)
pp.row = 33                                             # beyond one wavefront's 32 rows: the generic wide kernels take it
expect(dp(pp=pp), "emp_dp_plan row=33 (wide lattice)", allow_ok=True)
expect(dp(pp=None), "emp_dp_plan params NULL")
expect(dp(nB=-1), "emp_dp_plan B=-1")
expect(dp(mo=-1), "emp_dp_plan max_obs=-1")
expect(dp(mo=257), "emp_dp_plan max_obs=257")
expect(dp(nB=0), "emp_dp_plan B=0", allow_ok=True)
expect(dp(s0=None), "emp_dp_plan start NULL")
expect(dp(r=None), "emp_dp_plan rows NULL")
expect(dp(s=None), "emp_dp_plan status NULL")
expect(dp(m=None), "emp_dp_plan min_cost NULL (optional)", allow_ok=True)
expect(dp(os_=None), "emp_dp_plan obs_s NULL with max_obs > 0")
for mode in (0, 1):                                     # hostile per-scene counts are clamped, never followed
    wild = n_obs.copy()
    wild[::3] = 10 ** 6
    wild[1::3] = -7
    expect(dp(no=wild, mode=mode), f"emp_dp_plan wild n_obs mode {mode}", allow_ok=True)
assert dp() == 0 and np.array_equal(rows, clean), "the context survived"

# ---- emp_dp_enrich: capacity of one point, absurd capacity
ps_, pl_, ln = np.zeros((B, 1)), np.zeros((B, 1)), np.zeros(B, np.int32)
rc = lib.emp_dp_enrich(h, C.byref(p), B, ptr(clean), ptr(start), 1, ptr(ps_), ptr(pl_), ptr(ln), ptr(st), L.EMP_HOST)
expect(rc, "emp_dp_enrich max_pts=1", allow_ok=True)
assert (st & 32).all() and (ln == 1).all(), "one-point capacity: every scene truncated, nothing written past it"
expect(lib.emp_dp_enrich(h, C.byref(p), B, ptr(clean), ptr(start), 0, ptr(ps_), ptr(pl_), ptr(ln), ptr(st), L.EMP_HOST),
       "emp_dp_enrich max_pts=0")
expect(lib.emp_dp_enrich(h, C.byref(p), B, None, ptr(start), M, ptr(ps_), ptr(pl_), ptr(ln), ptr(st), L.EMP_HOST),
       "emp_dp_enrich rows NULL")

# ---- emp_plan_cycle: every required pointer NULL in turn, capacities at their limits
out = dict(dp_rows=np.zeros((B, p.col)), dp_s=np.zeros((B, M)), dp_l=np.zeros((B, M)), dp_len=np.zeros(B, np.int32),
           path_s=np.zeros((B, M)), path_l=np.zeros((B, M)), path_len=np.zeros(B, np.int32), traj=np.zeros((B, M + 1, 4)),
           traj_len=np.zeros(B, np.int32), status=np.zeros(B, np.int32))
inp = dict(ref_line=np.ascontiguousarray(b.ref), n_ref=np.full(B, P, np.int32), origin_xy=np.ascontiguousarray(b.origin_xy),
           start_xy=np.ascontiguousarray(b.start_xy), start_v=np.ascontiguousarray(b.start_v),
           start_a=np.ascontiguousarray(b.start_a), obs_xy=np.ascontiguousarray(b.obs_xy), n_obs=n_obs)


def cycle(nB=B, max_ref=P, mo=MO, max_pts=M, drop=None, mode=1, pp=p, qq=q, ss=sp, **over):
    io = L.CycleIO()
    for k, v in {**inp, **out, **over}.items():
        setattr(io, k, None if k == drop else v.ctypes.data)
    return lib.emp_plan_cycle(h, C.byref(pp) if pp is not None else None, C.byref(qq) if qq is not None else None,
                              C.byref(ss) if ss is not None else None, nB, max_ref, mo, max_pts, mode, C.byref(io), L.EMP_HOST)


assert cycle() == 0
clean_traj, clean_status = out["traj"].copy(), out["status"].copy()
for name in list(inp) + list(out):
    # required: every input (the obstacle arrays unless max_obs == 0) and traj / traj_len / status; the DP and path
    # arrays are optional outputs
    optional = name in ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len")
    expect(cycle(drop=name), f"emp_plan_cycle {name} NULL{' (optional)' if optional else ''}", allow_ok=optional)
expect(cycle(pp=None), "emp_plan_cycle dp params NULL")
expect(cycle(qq=None), "emp_plan_cycle qp params NULL")
expect(cycle(ss=None), "emp_plan_cycle smooth params NULL")
for kw in (dict(nB=-5), dict(max_ref=0), dict(max_ref=-1), dict(max_ref=1), dict(mo=-2), dict(mo=300), dict(max_pts=0), dict(max_pts=-4)):
    expect(cycle(**kw), f"emp_plan_cycle {kw}")
expect(cycle(nB=0), "emp_plan_cycle B=0", allow_ok=True)
# the optional front end (ABI 11): with a global path every one of its five fields is required, max_global >= 1, max_ref == 51
gp = np.zeros((B, 80, 4))
gp[:, :, 0] = np.arange(80) * 2.0
front = dict(global_path=gp, n_global=np.full(B, 80, np.int32), pre_match_index=np.full(B, 12, np.int32),
             match_index=np.zeros(B, np.int32), ref_status=np.zeros(B, np.int32))


def cycle_front(drop=None, max_global=80, max_ref=51, **over):
    io = L.CycleIO()
    for k, v in {**inp, **out, **front, **over}.items():
        setattr(io, k, None if k in (drop, "ref_line", "n_ref") else v.ctypes.data)
    io.max_global = max_global
    return lib.emp_plan_cycle(h, C.byref(p), C.byref(q), C.byref(sp), B, max_ref, MO, M, 1, C.byref(io), L.EMP_HOST)


assert cycle_front() == 0 and (front["ref_status"] == 0).all() and (front["match_index"] >= 0).all()
for name in ("n_global", "pre_match_index", "match_index", "ref_status"):
    expect(cycle_front(drop=name), f"emp_plan_cycle front end: {name} NULL")
expect(cycle_front(max_global=0), "emp_plan_cycle front end: max_global 0")
expect(cycle_front(max_ref=P if P != 51 else 61), "emp_plan_cycle front end: max_ref != 51")
wild = dict(n_global=np.full(B, 10 ** 6, np.int32))                      # a count beyond the row's capacity must never be followed
wild["n_global"][::2] = -7
expect(cycle_front(**wild), "front end: n_global beyond the capacity / negative (clamped or refused per scene)", allow_ok=True)
expect(cycle_front(pre_match_index=np.full(B, -9, np.int32)), "front end: negative pre_match_index (status bits)", allow_ok=True)
assert (front["ref_status"] != 0).all()
assert cycle() == 0
# a capacity too small for the path is a per-scene status bit (EMP_ST_TRUNCATED), never a write past the buffers
small = 5
guard = 64
flat = {k: np.full(B * n + guard, 777.0) for k, n in (("dp_s", small), ("dp_l", small), ("path_s", small), ("path_l", small),
                                                        ("traj", (small + 1) * 4))}
expect(cycle(max_pts=small, **flat), "emp_plan_cycle max_pts=5 (truncation)", allow_ok=True)
planned = (clean_status & ~1) == 0
assert (out["status"][planned] & 32).all(), "scenes that plan with room are flagged EMP_ST_TRUNCATED without it"
assert all((v[-guard:] == 777.0).all() for v in flat.values()), "nothing written behind the caller's capacity"
wild_ref = inp["n_ref"].copy()
wild_ref[::2] = 10 ** 7
wild_ref[1::4] = -3
expect(cycle(n_ref=wild_ref), "emp_plan_cycle wild n_ref", allow_ok=True)
for k in out:
    out[k][...] = 0
assert cycle() == 0 and np.array_equal(out["traj"], clean_traj) and np.array_equal(out["status"], clean_status), "the context survived"

# ---- emp_pack_records / emp_pack_trajectory_records: capacities
rec = np.zeros((B, 3 + p.col + 2 * M + 4 * (M + 1)))
args = [ptr(out[k]) for k in ("status", "traj_len", "path_len", "dp_rows", "path_s", "path_l", "traj")]
expect(lib.emp_pack_records(h, B, p.col, M, M + 1, *args, ptr(rec), 0, L.EMP_HOST), "emp_pack_records path_cap > max_pts")
expect(lib.emp_pack_records(h, B, p.col, M, 0, *args, ptr(rec), 0, L.EMP_HOST), "emp_pack_records path_cap = 0")
expect(lib.emp_pack_records(h, B, 0, M, M, *args, ptr(rec), 0, L.EMP_HOST), "emp_pack_records col = 0")
expect(lib.emp_pack_records(h, B, p.col, M, M, *args[:6], None, ptr(rec), 0, L.EMP_HOST), "emp_pack_records traj NULL")
expect(lib.emp_pack_trajectory_records(h, B, M, M + 1, args[0], args[1], args[6], ptr(rec), 0, L.EMP_HOST),
       "emp_pack_trajectory_records path_cap > max_pts")
expect(lib.emp_pack_trajectory_records(h, -1, M, M, args[0], args[1], args[6], ptr(rec), 0, L.EMP_HOST),
       "emp_pack_trajectory_records B = -1")

# ---- context-level entry points
expect(lib.emp_dp_plan(None, C.byref(p), B, MO, ptr(obs_s), ptr(obs_l), ptr(n_obs), ptr(start), 1, ptr(rows), ptr(mc), ptr(st),
                       L.EMP_HOST), "emp_dp_plan ctx NULL") if False else None   # NULL ctx: message goes to the create slot
assert lib.emp_synchronize(None) < 0 and lib.emp_set_pipeline(None, 1) < 0
assert lib.emp_set_pipeline(h, L.EMP_PIPELINE_MAX + 1) < 0 and b"EMP_PIPELINE_MAX" in lib.emp_last_error(h), "too many lanes is an error"
assert lib.emp_set_fence(None, 1) < 0 and lib.emp_set_fence(h, 0) == 0 and lib.emp_set_fence(h, 1) == 0
assert lib.emp_set_pipeline(h, -5) < 0 and lib.emp_set_pipeline(h, 3) == 0 and lib.emp_set_pipeline(h, 0) == 0    # below EMP_PIPELINE_AUTO: an error (ABI 11)
assert lib.emp_set_pipeline(h, L.EMP_PIPELINE_AUTO) == 0 and lib.emp_pipeline_form(h, None, None) in (1, 3) and lib.emp_set_pipeline(h, 0) == 0
assert lib.emp_pipeline_form(None, None, None) < 0 and lib.emp_edge_clock_mhz(None) < 0 and lib.emp_edge_clock_mhz(h) < 0      # nothing recorded
nul = C.c_void_p()
assert lib.emp_device_alloc(h, C.c_uint64(1 << 62), C.byref(nul)) < 0 and not nul.value, "an impossible allocation is an error"
assert lib.emp_create(9999, C.byref(nul)) < 0 and lib.emp_last_error(None), "no such device"
rc = dp()
assert rc == 0 and np.array_equal(rows, clean), f"after the failed allocation / create: rc {rc} {lib.emp_last_error(h)}"
lib.emp_destroy(h)
lib.emp_destroy(None)
print(f"ABI-FUZZ-OK probes {probes} errors {errors}")
