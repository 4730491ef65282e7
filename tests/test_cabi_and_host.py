"""CPU suite: the C-ABI library builds, loads and exports every symbol include/emplanner.h declares
(no compute calls - there is no GPU here); host-side logic (scene generator, sharding, record packing,
the world_size-2 gather over gloo)."""
import ctypes
import datetime
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "emplanner.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(emp_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from emplanner_carla_amd import build
    lib_path = build.build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    names = _declared_functions()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in emplanner.h but not exported: {missing}"
    # and the ctypes binding covers the same set
    from emplanner_carla_amd import _lib
    assert sorted(_lib.PROTOTYPES) == names


def test_loader_fails_loudly_without_gpu_or_library(tmp_path):
    from emplanner_carla_amd import _lib
    from emplanner_carla_amd.api import Planner
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.EmpError, match="no HIP device|No such|device"):
        Planner(0)
    # a missing library is an error, not a silent CPU path
    code = ("import emplanner_carla_amd._lib as L; L.LIB_PATH = '/nonexistent/libemplanner.so'\n"
            "try:\n    L.load()\nexcept RuntimeError as e:\n    print('RAISED', 'no CPU implementation' in str(e))\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT,
                         env=dict(os.environ, EMP_SKIP_TORCH_PRELOAD="1"))
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_package_never_imports_the_oracle():
    """The product path must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "emplanner_carla_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "ref_port" not in text and "qp_dense" not in text, f


def test_scene_generator_is_deterministic_and_shaped():
    from emplanner_carla_amd import scenes as S
    a = S.make_batch(range(5), S.CFG2)
    b = S.make_batch(range(5), S.CFG2)
    for f in ("ref", "origin_xy", "start_xy", "start_v", "start_a", "obs_xy", "n_obs", "sl_obs_s", "sl_start"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert a.ref.shape == (5, 61, 4) and a.obs_xy.shape == (5, 8, 2) and (a.n_obs == 8).all()
    # arc: consecutive reference points 2 m apart, curvature constant, heading consistent with the chord
    d = np.hypot(np.diff(a.ref[0, :, 0]), np.diff(a.ref[0, :, 1]))
    assert np.allclose(d, 2.0, atol=1e-5) and np.ptp(a.ref[0, :, 3]) == 0
    c = S.make_batch(range(3), S.CFG1)
    assert c.obs_xy.shape == (3, 1, 2) and (c.n_obs == 0).all()


def test_shard_ranges_cover_everything():
    from emplanner_carla_amd.dist import shard_range
    for total, world in ((32768, 8), (4096, 1), (10, 4), (3, 8), (0, 2)):
        blocks = [shard_range(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
        for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
            assert s1 == s0 + c0


def _free_port():
    """A port nobody listens on right now (bind to 0, read it back): a pid-derived port can collide with a stranger's, and a
    rendezvous on a taken port waits out torch's 30-minute default with the workers blocking pytest's exit."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_ranks(ctx, target, args_of_rank, world=2):
    """Start one daemon process per rank, collect one queue item from each, always reap the processes."""
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(*args_of_rank(r), q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    try:
        got = dict(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    return got


def _worker(rank, world, port, total, width, q):
    import torch
    import torch.distributed as dist
    from emplanner_carla_amd.dist import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    start, count = shard_range(total, rank, world)
    local = (torch.arange(start, start + count, dtype=torch.float64).reshape(-1, 1)
             * torch.ones(1, width, dtype=torch.float64) + torch.arange(width, dtype=torch.float64) * 1e-3)
    out = gather_records(local, total)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_gather_records_world_size_2_gloo(total):
    """The N>1 result collection (equal and ragged shards) on CPU with the gloo backend."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port, width = _free_port(), 5
    got = _run_ranks(ctx, _worker, lambda r: (r, 2, port, total, width))
    want = np.arange(total, dtype=np.float64).reshape(-1, 1) * np.ones((1, width)) + np.arange(width) * 1e-3
    for r in range(2):
        assert np.array_equal(got[r], want)


def _worker_dst(rank, world, port, total, width, q):
    import torch
    import torch.distributed as dist
    from emplanner_carla_amd.dist import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    start, count = shard_range(total, rank, world)
    local = torch.arange(start, start + count, dtype=torch.float64).reshape(-1, 1).repeat(1, width)
    local = local + torch.arange(width, dtype=torch.float64) * 1e-3
    out = gather_records(local, total, dst=0)
    q.put((rank, None if out is None else out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_gather_records_to_rank_0_world_size_2_gloo(total):
    """The gather proper (BASELINE: 'RCCL gather only'): rank 0 receives every block in scene order, rank 1 nothing."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    port, width = _free_port(), 5
    got = _run_ranks(ctx, _worker_dst, lambda r: (r, 2, port, total, width))
    want = np.arange(total, dtype=np.float64).reshape(-1, 1) * np.ones((1, width)) + np.arange(width) * 1e-3
    assert got[1] is None and np.array_equal(got[0], want)


class _StubPlanner:
    """Stands in for emplanner_carla_amd.api.Planner in the CPU test of the multi-GPU step loop: `plan_cycle` returns a
    CycleResult that encodes (scene index, step) in every field, so the gathered records can be checked exactly."""

    def __init__(self, col, max_pts):
        self.col, self.max_pts = col, max_pts

    def plan_cycle(self, scene_ids, step):
        import torch
        from emplanner_carla_amd.api import CycleResult
        B, M = len(scene_ids), self.max_pts
        ids = torch.as_tensor(scene_ids, dtype=torch.float64)
        f = lambda *shape: (ids.reshape(-1, *([1] * (len(shape)))) * 1000.0 + step + torch.arange(
            int(np.prod(shape)), dtype=torch.float64).reshape(1, *shape) * 1e-3).contiguous()
        return CycleResult(dp_rows=f(self.col), dp_s=None, dp_l=None, dp_len=None, path_s=f(M), path_l=-f(M),
                           path_len=torch.full((B,), M // 2 + 1, dtype=torch.int32), traj=f(M + 1, 4),
                           traj_len=torch.full((B,), M // 2 + 2, dtype=torch.int32),
                           status=(torch.as_tensor(scene_ids) % 3).to(torch.int32))


def _worker_steps(rank, world, port, total, fields, dst, q):
    import torch.distributed as dist
    from emplanner_carla_amd.dist import StepGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    col, M = 6, 9
    start, count = shard_range(total, rank, world)
    stub = _StubPlanner(col, M)
    sg = StepGather(col, M, total, planner=None, fields=fields, dst=dst, depth=2)
    last = None
    for step in range(5):                                # more steps than the in-flight ring holds
        last = sg.submit(stub.plan_cycle(list(range(start, start + count)), step))
    sg.drain()
    if last is None:
        q.put((rank, None))
    else:
        u = sg.unpack(last)
        q.put((rank, {k: v.numpy() for k, v in u.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total,fields,dst", [(10, "full", 0), (9, "trajectory", 0), (9, "full", None)],
                         ids=["equal_full_rank0", "ragged_trajectory_rank0", "ragged_full_all"])
def test_step_loop_of_the_many_scene_mode_world_size_2_gloo(total, fields, dst):
    """bench.py's per-step code for N > 1 (emplanner_carla_amd.dist.StepGather: pack -> gather -> in-flight ring -> unpack)
    on two gloo ranks with a stub planner, equal and ragged shards, full and trajectory-only records, gather to rank 0 and
    all_gather: what arrives is every scene's last-step record, in scene order."""
    import torch.multiprocessing as mp
    from emplanner_carla_amd.dist import path_capacity
    ctx = mp.get_context("spawn")
    port = _free_port()
    got = _run_ranks(ctx, _worker_steps, lambda r: (r, 2, port, total, fields, dst))
    col, M = 6, 9
    cap = path_capacity(M)
    want = _StubPlanner(col, M).plan_cycle(list(range(total)), 4)
    for r in range(2):
        if dst is not None and r != dst:
            assert got[r] is None
            continue
        u = got[r]
        assert np.array_equal(u["status"], want.status.numpy()) and np.array_equal(u["traj_len"], want.traj_len.numpy())
        assert np.array_equal(u["traj"], want.traj.numpy()[:, :cap + 1])
        if fields == "full":
            assert np.array_equal(u["dp_rows"], want.dp_rows.numpy()) and np.array_equal(u["path_l"], want.path_l.numpy()[:, :cap])
        else:
            assert set(u) == {"status", "traj_len", "traj"}


def _worker_steps_big(rank, world, port, total, q):
    """One rank of the 8-rank step loop: a few steps on its ragged shard of `total` scenes; rank 0 reports a digest of
    what it gathered (the whole matrix is 4099 x 63 doubles - small enough to send, but only rank 0 has it)."""
    import torch.distributed as dist
    from emplanner_carla_amd.dist import StepGather, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    col, M = 6, 9
    start, count = shard_range(total, rank, world)
    stub = _StubPlanner(col, M)
    sg = StepGather(col, M, total, planner=None, fields="full", dst=0, depth=2)
    last = None
    for step in range(4):
        last = sg.submit(stub.plan_cycle(list(range(start, start + count)), step))
    sg.drain()
    q.put((rank, None if last is None else last.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_step_loop_world_size_8_ragged_shards_gather_to_rank_0():
    """The 8-GPU shape of BASELINE configs[3] on CPU (gloo): eight ranks, 4099 scenes (ragged: three ranks hold 513, five
    512), the whole per-step loop (pack -> gather to rank 0 -> in-flight ring) for four steps.  Rank 0's matrix must equal
    the ONE-PROCESS result - the records of all 4099 scenes of the last step, in scene order - and no other rank receives
    anything.  (No 2/4/8-GPU node has been available in any round: this and the shard == slice GPU tests are what stands
    in for the scaling run; README says so.)"""
    import torch.multiprocessing as mp
    from emplanner_carla_amd.dist import pack_records, path_capacity, shard_range
    ctx = mp.get_context("spawn")
    port, total, world = _free_port(), 4099, 8
    assert sorted(shard_range(total, r, world)[1] for r in range(world)) == [512] * 5 + [513] * 3
    got = _run_ranks(ctx, _worker_steps_big, lambda r: (r, world, port, total), world=world)
    col, M = 6, 9
    want = pack_records(_StubPlanner(col, M).plan_cycle(list(range(total)), 3), col, M, path_cap=path_capacity(M)).numpy()
    assert got[0].shape == want.shape and np.array_equal(got[0], want)
    assert all(got[r] is None for r in range(1, world))


def test_pack_unpack_records_roundtrip():
    import torch
    from emplanner_carla_amd.api import CycleResult
    from emplanner_carla_amd.dist import pack_records, record_width, unpack_records
    B, col, M = 4, 6, 9
    rng = np.random.default_rng(0)
    res = CycleResult(dp_rows=rng.integers(0, 12, (B, col)).astype(np.float64), dp_s=None, dp_l=None, dp_len=None,
                      path_s=rng.normal(size=(B, M)), path_l=rng.normal(size=(B, M)),
                      path_len=np.array([9, 8, 0, 9], np.int32), traj=rng.normal(size=(B, M + 1, 4)),
                      traj_len=np.array([10, 9, 0, 10], np.int32), status=np.array([0, 1, 8, 0], np.int32))
    rec = pack_records(res, col, M)
    assert tuple(rec.shape) == (B, record_width(col, M))
    back = unpack_records(rec, col, M)
    assert np.array_equal(back["status"].numpy(), res.status) and np.array_equal(back["traj_len"].numpy(), res.traj_len)
    assert np.array_equal(back["traj"].numpy(), res.traj) and np.array_equal(back["path_l"].numpy(), res.path_l)
    assert np.array_equal(back["dp_rows"].numpy(), res.dp_rows)
    # trimmed to the stations a cycle can fill (what the multi-GPU bench gathers)
    from emplanner_carla_amd.dist import path_capacity
    assert path_capacity(41) == 22 and path_capacity(9) == 6 and path_capacity(9, decimate=1, midpoint=False) == 9
    cap = 6
    rec = pack_records(res, col, M, path_cap=cap)
    assert tuple(rec.shape) == (B, record_width(col, M, cap)) and rec.shape[1] < record_width(col, M)
    back = unpack_records(rec, col, M, path_cap=cap)
    assert np.array_equal(back["traj"].numpy(), res.traj[:, :cap + 1]) and np.array_equal(back["path_s"].numpy(), res.path_s[:, :cap])
    assert np.array_equal(back["status"].numpy(), res.status)
    # trajectory-only records (what a controller-side consumer needs)
    rec = pack_records(res, col, M, path_cap=cap, fields="trajectory")
    assert tuple(rec.shape) == (B, 2 + 4 * (cap + 1)) == (B, record_width(col, M, cap, "trajectory"))
    back = unpack_records(rec, col, M, path_cap=cap, fields="trajectory")
    assert np.array_equal(back["traj"].numpy(), res.traj[:, :cap + 1]) and np.array_equal(back["traj_len"].numpy(), res.traj_len)


def test_graft_entry_build_runs_on_cpu():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()


def test_only_the_cycle_call_is_ordered_on_the_result_stream():
    """Host logic of the two-batches-in-flight mode (emplanner_carla_amd/api.py): `_Args.done` makes a caller on a
    foreign torch stream wait for the RESULT stream only for the call whose back stage runs there - `plan_cycle` and
    nothing else marks itself so.  (The mark once sat in `frenet_project`: `plan_cycle` then waited for the front stage
    only, and a tensor read right after the call raced with the back stage.)"""
    import inspect
    from emplanner_carla_amd import api
    marked = [name for name, fn in inspect.getmembers(api.Planner, inspect.isfunction)
              if re.search(r"^\s*a\.cycle\s*=\s*True", inspect.getsource(fn), re.M)]
    assert marked == ["plan_cycle"]
    done = inspect.getsource(api._Args.done)
    assert "self.planner.pipelined and self.cycle" in done and "torch_result_stream()" in done
    code = "\n".join(line.split("#")[0] for line in done.split('"""')[2].splitlines())
    assert ".record_stream(" not in code, "outputs are kept alive by Planner.plan_cycle, not tied to a stream"


def test_planning_loop_policy_for_refused_requests():
    """service.answer_refused: what the planning process sends when a request has no plan (no GPU involved).  IndexError
    statuses raise as in the reference; an infeasible QP repeats the previous trajectory by default (an unmodified
    reference driver hands element 0 of the reply to its controller, test_9.py:395-399), raises or sends the sentinel on
    request."""
    from emplanner_carla_amd.service import answer_refused
    prev = ([(0.0, 0.0, 0.0, 0.0)], [7], [1.0], [0.5])
    good = ([(1.0, 1.0, 0.0, 0.0)], [9], [2.0], [0.25])
    assert answer_refused(good, 0, 9, prev, "previous") is good
    assert answer_refused(None, 8, 11, prev, "previous") == (prev[0], [11], prev[2], prev[3])
    assert answer_refused(None, 16, 11, prev, "sentinel") == (None, [11], [], [])
    for status in (2, 4, 2 | 8):
        with pytest.raises(IndexError):
            answer_refused(None, status, 11, prev, "previous")
    with pytest.raises(ValueError):
        answer_refused(None, 8, 11, prev, "raise")
    with pytest.raises(ValueError, match="no previous"):
        answer_refused(None, 8, 11, None, "previous")
    with pytest.raises(ValueError, match="on_infeasible"):
        answer_refused(None, 8, 11, prev, "whatever")


def test_repeated_trajectories_are_logged_and_bounded(caplog):
    """service.RefusalPolicy: every repeat of the previous trajectory is a WARNING with its count, a valid reply resets the
    count, and the request after max_repeats consecutive repeats raises instead of sending a stale trajectory again."""
    import logging
    from emplanner_carla_amd.service import RefusalPolicy
    good = ([(1.0, 1.0, 0.0, 0.0)], [9], [2.0], [0.25])
    pol = RefusalPolicy("previous", max_repeats=3)
    assert pol.answer(good, 0, 9) is good
    with caplog.at_level(logging.WARNING, logger="emplanner_carla_amd.service"):
        for k in range(3):
            assert pol.answer(None, 8, 20 + k) == (good[0], [20 + k], good[2], good[3])
        assert [r.getMessage().count(f"repeat {k + 1} of at most 3") for k, r in enumerate(caplog.records)] == [1, 1, 1]
        with pytest.raises(ValueError, match="4 requests in a row"):
            pol.answer(None, 8, 30)
    assert pol.answer(good, 0, 31) is good and pol.repeats == 0
    assert pol.answer(None, 16, 32)[1] == [32] and pol.repeats == 1
    with pytest.raises(ValueError):
        RefusalPolicy("raise").answer(None, 8, 1)


def _callable_args(fn):
    """[name, default] pairs of a Python callable in the fixture's form (defaults as VALUES here)."""
    import inspect
    out = []
    for name, p in inspect.signature(fn).parameters.items():
        if p.kind is p.VAR_POSITIONAL:
            name = "*" + name
        elif p.kind is p.VAR_KEYWORD:
            name = "**" + name
        out.append((name, p.default))
    return out


def _same_default(got, want_src):
    import inspect
    if want_src is None:
        return got is inspect.Parameter.empty
    if got is inspect.Parameter.empty:
        return False
    want = eval(want_src, {"np": np, "math": __import__("math")})      # literals and arithmetic of literals only
    if isinstance(want, float) or isinstance(got, float):
        return float(got) == float(want)
    return got == want and type(got) is type(want)


def test_dropin_surface_matches_the_reference_signature_fixture():
    """tests/golden/signatures.json (written from the reference's source by tests/golden/make_signatures.py): every function
    and every class method the reference's planner / controller modules define exists in the drop-in module under the same
    name, with the same argument names in the same order and the same defaults.  A drop-in may ADD trailing keyword
    arguments with defaults (speed_DP's reference_behaviour); it may not rename, reorder or drop anything.  The controller
    module is scope "input_side" (out of scope; section 8f row 3 mirrors the lateral controllers'
    constructor, cal_vehicle_info and _control): there, whatever the drop-in defines must be the reference's."""
    import importlib
    import json
    sig = json.load(open(os.path.join(ROOT, "tests", "golden", "signatures.json")))
    assert set(sig) == {"planner/path_planning.py", "planner/planning_utils.py", "planner/speed_planning_test.py",
                        "controller/controller.py"}
    problems, checked = [], 0

    def compare(where, fn, want):
        got = _callable_args(fn)
        if len(got) < len(want):
            problems.append(f"{where}: {len(got)} arguments, the reference has {len(want)}")
            return
        for (gn, gd), (wn, wd) in zip(got, want):
            if gn != wn:
                problems.append(f"{where}: argument {gn!r} where the reference has {wn!r}")
            elif not _same_default(gd, wd):
                problems.append(f"{where}: default of {gn!r} is {gd!r}, the reference's is {wd}")
        import inspect
        for gn, gd in got[len(want):]:
            if gd is inspect.Parameter.empty and not gn.startswith("*"):
                problems.append(f"{where}: extra argument {gn!r} without a default")

    import inspect
    for rel, entry in sig.items():
        mod = importlib.import_module(entry["dropin"])
        full = entry["scope"] == "full"
        for name, want in entry["functions"].items():
            fn = getattr(mod, name, None)
            if fn is None:
                problems.append(f"{rel}: {name} is missing")
                continue
            compare(f"{rel}:{name}", fn, want)
            checked += 1
        for cname, methods in entry["classes"].items():
            cls = getattr(mod, cname, None)
            if cls is None:
                if full:
                    problems.append(f"{rel}: class {cname} is missing")
                continue
            for mname, want in methods.items():
                m = getattr(cls, mname, None)
                if m is None or (not full and mname not in vars(cls)):
                    if full:
                        problems.append(f"{rel}: {cname}.{mname} is missing")
                    continue
                compare(f"{rel}:{cname}.{mname}", m, want)
                checked += 1
        if not full:      # nothing public in the drop-in that the reference does not have
            for cname, cls in vars(mod).items():
                if inspect.isclass(cls) and cls.__module__ == mod.__name__:
                    assert cname in entry["classes"], f"{rel}: the drop-in defines a class {cname} the reference lacks"
                    extra = [m for m in vars(cls) if callable(vars(cls)[m]) and m not in entry["classes"][cname]]
                    assert not extra, f"{rel}: {cname} defines {extra}, which the reference lacks"
    assert not problems, "\n".join(problems)
    assert checked >= 45


def test_bare_bench_command_relaunches_itself_as_n_ranks(tmp_path):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment must not run ONE rank: it
    re-executes itself under torch.distributed.run with N processes on the loopback address, passes its own arguments through
    and prints rank 0's one JSON line.  CPU check of the argv and of the pass-through (a stand-in interpreter prints what it
    was started with and a line of noise - no GPU, no rendezvous here)."""
    sys.path.insert(0, ROOT)
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5", "--total-scenes", "32768"]
    cmd = bench.self_launch_argv(argv, 8, 29123, python="py")
    assert cmd[:3] == ["py", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29123"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv
    # the pass-through: a fake `python` that ignores torch.distributed.run and answers like N ranks would
    fake = tmp_path / "fakepy"
    fake.write_text("#!/bin/sh\necho 'NCCL version banner'\necho '{\"n_gpus\": 2, \"argv\": \"'\"$*\"'\", \"ws\": \"'\"$WORLD_SIZE\"'\"}'\n")
    fake.chmod(0o755)
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); import bench; sys.executable = {str(fake)!r}; "
            f"sys.argv = ['bench.py', '--gpus', '2', '--steps', '3']; sys.exit(bench.main() or 0)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert len(lines) == 1, out.stdout                        # ONE line on stdout: the banner went to stderr
    import json
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "--nproc-per-node 2" in d["argv"] and d["argv"].endswith("--gpus 2 --steps 3")
    assert "NCCL version banner" in out.stderr and "launching 2 ranks" in out.stderr
    # with WORLD_SIZE set (the driver's own launch) bench.py does NOT relaunch: it goes on to initialise its rank
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:' in src


def test_failing_submit_does_not_shrink_the_cycle_stream_ring():
    """service.CycleStream owns a ring slot from submit() to result().  A request that fails in between - a refused argument,
    a HIP error, a shape that does not fit - must hand its slot back: the wire server answers the exception with an error
    frame and KEEPS SERVING, so a leaked slot per failure would leave every session waiting in _take_slot after `depth`
    failures.  Host logic only: a stand-in planner whose plan_cycle fails on demand."""
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import dp_params

    class Slot:
        def __init__(self):
            self.B, self.max_ref, self.max_pts = 8, 51, 32
            self.inputs = {"obs_xy": np.zeros((8, 4, 2)), "global_path": np.zeros((8, 64, 4))}
            self.outputs = {k: np.zeros(8, np.int32) for k in ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len",
                                                                "traj", "traj_len", "status", "match_index", "ref_status")}
            self._ticket, self.waits, self.use_dyn = None, 0, False

        def load(self, **arrays):
            if FakePlanner.fail == "load":
                raise ValueError("could not broadcast input array")
            return self

        def wait(self):
            self.waits += 1
            if self._ticket == "poisoned":
                self._ticket = None
                raise RuntimeError("emp_wait_ticket failed")
            self._ticket = None
            return self

    class Ring:
        def __init__(self):
            self.slots = [Slot() for _ in range(4)]

        def close(self):
            pass

    class FakePlanner:
        fail = None

        def set_pipeline(self, mode):
            pass

        def set_fence(self, on):
            pass

        def host_ring(self, *a, **kw):
            return Ring()

        def plan_cycle(self, *a, slot=None, **kw):
            if self.fail == "plan_cycle":
                raise RuntimeError("EMP_REQUIRE: LDS too large for the layout")
            slot._ticket = "poisoned" if self.fail == "wait" else 7

    pl = FakePlanner()
    stream = service.CycleStream(pl, capacity=8, max_static=4)
    a = dict(global_path=np.zeros((2, 60, 4)), n_global=np.full(2, 60, np.int32), pred=np.zeros((2, 2)), veh=np.zeros((2, 2)),
             v=np.zeros((2, 2)), a=np.zeros((2, 2)), pre_match=np.zeros(2, np.int32), obs_xy=np.zeros((2, 4, 2)),
             n_obs=np.zeros(2, np.int32), dyn=np.full((2, 2), np.nan))
    dp = dp_params()
    ring = None
    for how in ("load", "plan_cycle") * 5:                        # ten failures on a ring of four slots
        FakePlanner.fail = how
        with pytest.raises((RuntimeError, ValueError)):
            stream.submit(a, dp=dp)
        ring = next(iter(stream._rings.values()))
        assert len(ring.free) == 4, f"a failing submit ({how}) leaked a slot"
    FakePlanner.fail = "wait"                                      # the call is issued, the wait in result() fails
    for _ in range(6):
        h = stream.submit(a, dp=dp)
        assert len(ring.free) == 3
        with pytest.raises(RuntimeError):
            stream.result(h)
        assert len(ring.free) == 4 and h["slot"] is None
    FakePlanner.fail = None
    for _ in range(6):                                             # and the ring still works
        st_ref, match, res, M = stream.plan_arrays(a, dp=dp)
        assert res.status.shape == (2,) and len(ring.free) == 4
    assert sorted(id(s) for s in ring.free) == sorted(id(s) for s in ring.slots)     # every slot exactly once
    stream.close()


def test_no_kernel_addresses_memory_through_a_select_of_address_spaces(tmp_path):
    """Round 6 found a kernel whose reads went through `cond ? device_pointer : lds_pointer`: the compiler turns that into ONE flat
    load behind a select of two address spaces, it worked, and builds that differed only in where an unrelated value was loaded read
    the address wrong (an aperture violation on the GPU box).  The device code is compiled to assembly here
    (hipcc cross-compiles, ~30 s) and no kernel may contain a flat_* instruction - except heading_kappa_wave, a real out-of-line
    function whose pointer arguments are generic by construction."""
    from emplanner_carla_amd import build
    out = tmp_path / "emp_api.s"
    cmd = [build.hipcc()] + [f for f in build.FLAGS if f not in ("-shared", "-fPIC")] + \
          ["--cuda-device-only", "-S", os.path.join(build.CSRC, "emp_api.hip"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    flat, cur = {}, None
    for line in open(out):
        m = re.match(r"^(_Z\S+):", line)
        if m:
            cur = m.group(1)
        tok = line.split()
        if cur and tok and tok[0].startswith("flat_"):
            flat[cur] = flat.get(cur, 0) + 1
    offenders = {k: v for k, v in flat.items() if "heading_kappa_wave" not in k}
    assert not offenders, f"flat memory instructions in {offenders}"


def test_reference_line_conversion_cache_follows_node_identity():
    """planner/_runtime.line_array converts the reference line once while the SAME immutable node tuples are handed in (the
    reference's loop passes one list to six functions) and converts again the moment any node is another object - a replaced node,
    another list of equal length, a list of lists (mutable nodes are never cached)."""
    from emplanner_carla_amd.planner import _runtime as R
    path = [(float(i), 2.0 * i, 0.1, 0.0) for i in range(51)]
    a, n = R.line_array(path)
    b, _ = R.line_array(path)
    assert a is b and n[0] == 51 and a.shape == (1, 51, 4)
    c, _ = R.line_array(list(path))                      # another list object, the same node objects: still the same line
    assert c is a
    path[3] = (9.0, 9.0, 9.0, 9.0)                       # a node replaced in place
    d, _ = R.line_array(path)
    assert d is not a and d[0, 3, 0] == 9.0 and a[0, 3, 0] == 3.0
    other = [(float(i), 2.0 * i, 0.1, 0.0) for i in range(51)]     # equal values, other objects
    e, _ = R.line_array(other)
    assert e is not d and np.array_equal(e[0, :3], a[0, :3])
    m1, _ = R.line_array([[1.0, 2.0, 3.0, 4.0]] * 3)
    m2, _ = R.line_array([[1.0, 2.0, 3.0, 4.0]] * 3)
    assert m1 is not m2                                   # mutable nodes: converted every time
    short, n0 = R.line_array([])
    assert short.shape == (1, 0, 4) and n0[0] == 0
