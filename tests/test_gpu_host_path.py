"""The overlapped host path of the planning cycle (EMP_HOST_PINNED, api.HostRing): page-locked input / output rings, inputs
on a copy stream one call ahead, outputs on a stream of their own, no host wait inside the call when a pipeline is set
(reference counterpart: one request per Pipe message, test_9.py:92-96, 220, 390-395).

What must hold: the outputs that arrive in the slot's host arrays are bit for bit those of the same scenes planned with
device-resident inputs - in every pipeline mode, over more calls than the ring has slots, with a different batch (and a
different batch size) per call, and while the caller overwrites a slot's inputs as soon as the ring hands it out again."""
from __future__ import annotations

import numpy as np
import pytest

from emplanner_carla_amd import scenes as S

pytestmark = pytest.mark.gpu
OUTPUTS = ("dp_rows", "dp_s", "dp_l", "dp_len", "path_s", "path_l", "path_len", "traj", "traj_len", "status")


def _params(cfg):
    from emplanner_carla_amd.api import dp_params_from_cfg, qp_params, smooth_params
    return dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()


def _host(b):
    B, P = b.ref.shape[:2]
    return dict(ref_line=b.ref, n_ref=np.full(B, P, np.int32), origin_xy=b.origin_xy, start_xy=b.start_xy, start_v=b.start_v,
                start_a=b.start_a, obs_xy=b.obs_xy, n_obs=b.n_obs)


def _masked(out):
    res = dict(out)
    for arr, ln in (("dp_s", "dp_len"), ("dp_l", "dp_len"), ("path_s", "path_len"), ("path_l", "path_len"), ("traj", "traj_len")):
        a = np.array(out[arr], copy=True)
        a[np.arange(a.shape[1])[None, :] >= np.asarray(out[ln])[:, None]] = 0.0
        res[arr] = a
    return res


@pytest.mark.parametrize("pipe", [0, 1, 3], ids=["no_pipeline", "staged", "lanes3"])
def test_host_ring_equals_the_resident_path(pipe):
    import torch
    from emplanner_carla_amd.api import Planner
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    B = 1536
    batches = [S.make_batch(range(900 * k, 900 * k + B - 64 * (k % 3)), cfg, start_ahead=S.BENCH_START_AHEAD) for k in range(7)]
    ref_pl = Planner(0)
    want = []
    for b in batches:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in _host(b).items()}
        r = ref_pl.plan_cycle(p, q, sp, **dev)
        ref_pl.synchronize()
        want.append(_masked({k: getattr(r, k).cpu().numpy() for k in OUTPUTS}))
    ref_pl.close()
    pl = Planner(0)
    try:
        pl.set_pipeline(pipe)
        ring = pl.host_ring(p, B, batches[0].ref.shape[1], cfg.n_obs)
        assert len(ring.slots) == max(pl._lib.emp_pipeline_depth(pl._h), 1)
        slots = []
        for k, b in enumerate(batches):
            n = len(b)
            slot = ring.next()                                   # waits for the call that used the slot last
            if k >= len(ring.slots):                             # ... whose outputs must therefore be complete: check them NOW,
                j = k - len(ring.slots)                          # before this call's inputs overwrite the slot
                got = _masked({f: np.array(slot.outputs[f][:len(batches[j])]) for f in OUTPUTS})
                _same(got, want[j], f"call {j} (slot reused by call {k})")
            slot.load(**_host(b))
            slot.B = n                                           # a smaller batch in the same slot: plan only its scenes
            pl.plan_cycle(p, q, sp, None, None, None, None, None, None, None, None, slot=slot)
            slot.B = B
            slots.append(slot)
        ring.wait_all()
        for j in range(len(batches) - len(ring.slots), len(batches)):
            got = _masked({f: np.array(slots[j].outputs[f][:len(batches[j])]) for f in OUTPUTS})
            _same(got, want[j], f"call {j}")
        ring.close()
    finally:
        pl.set_pipeline(0)
        pl.close()


def _same(a, b, what):
    ok = (b["status"] & ~1) == 0
    for f in OUTPUTS:
        x, y = (a[f], b[f]) if f in ("status", "dp_rows") else (a[f][ok], b[f][ok])
        assert np.array_equal(x, y), f"{what}: {f} differs"


def test_every_entry_point_accepts_page_locked_arrays():
    """EMP_HOST_PINNED is ordinary host memory to every other entry point: a page-locked array goes where a NumPy array goes."""
    from emplanner_carla_amd.api import Planner, dp_params_from_cfg
    cfg = S.CFG2
    b = S.make_batch(range(64), cfg)
    pl = Planner(0)
    try:
        a = {k: pl.pinned_empty(v.shape, v.dtype) for k, v in dict(s=b.sl_obs_s, l=b.sl_obs_l, n=b.n_obs, st=b.sl_start).items()}
        for k, v in dict(s=b.sl_obs_s, l=b.sl_obs_l, n=b.n_obs, st=b.sl_start).items():
            a[k][...] = v
        r1 = pl.dp_plan(dp_params_from_cfg(cfg), a["s"], a["l"], a["n"], a["st"])
        r2 = pl.dp_plan(dp_params_from_cfg(cfg), b.sl_obs_s, b.sl_obs_l, b.n_obs, b.sl_start)
        for x, y in zip(r1, r2):
            assert np.array_equal(x, y)
        pl.pinned_free(a["s"])
        with pytest.raises(Exception):
            pl._check(pl._lib.emp_host_free(pl._h, 12345))
    finally:
        pl.close()


class _Arrays:
    """What Planner._plan_cycle_pinned needs of a slot, over arbitrary page-locked arrays."""

    def __init__(self, inputs, outputs, B, P, mo, M):
        self.inputs, self.outputs, self.B, self.max_ref, self.max_obs, self.max_pts = inputs, outputs, B, P, mo, M
        self.use_dyn, self._ticket = False, None


@pytest.mark.parametrize("pipe", [0, 1, 3], ids=["no_pipeline", "staged", "lanes3"])
@pytest.mark.parametrize("layout", ["one_by_one", "interleaved", "reversed_block"])
def test_pinned_arrays_of_any_layout_and_nothing_else_is_written(layout, pipe):
    """The layout contract of EMP_HOST_PINNED (include/emplanner.h; ADVICE r05: Stage::place used to GUESS from addresses that
    the arrays were one block and copied the whole span).  one_by_one: every array its own emp_host_alloc (virtually adjacent,
    page granular - never one copy across allocations).  interleaved: ONE allocation carved as output / foreign bytes / output
    / input / foreign ... - the foreign bytes (kilobytes, not padding) must survive the call untouched.  reversed_block: one
    allocation, arrays in descending address order with 256-byte padding (the block path, whatever the argument order).
    Outputs bit for bit those of the resident path."""
    import torch
    from emplanner_carla_amd.api import Planner, max_path_points
    cfg = S.CFG2
    p, q, sp = _params(cfg)
    B = 192
    b = S.make_batch(range(4000, 4000 + B), cfg, start_ahead=S.BENCH_START_AHEAD)
    host = _host(b)
    P, mo, M = b.ref.shape[1], cfg.n_obs, max_path_points(p)
    pl = Planner(0)
    try:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in host.items()}
        r = pl.plan_cycle(p, q, sp, **dev)
        pl.synchronize()
        want = _masked({k: getattr(r, k).cpu().numpy() for k in OUTPUTS})
        pl.set_pipeline(pipe)
        f, i = np.float64, np.int32
        ins = [(k, v.shape, v.dtype) for k, v in host.items()]
        outs = [("dp_rows", (B, p.col), f), ("dp_s", (B, M), f), ("dp_l", (B, M), f), ("dp_len", (B,), i), ("path_s", (B, M), f),
                ("path_l", (B, M), f), ("path_len", (B,), i), ("traj", (B, M + 1, 4), f), ("traj_len", (B,), i), ("status", (B,), i)]
        guards, blocks = [], []
        if layout == "one_by_one":
            inputs = {k: pl.pinned_empty(shp, dt) for k, shp, dt in ins}
            outputs = {k: pl.pinned_empty(shp, dt) for k, shp, dt in outs}
            blocks = list(inputs.values()) + list(outputs.values())
        else:
            # one allocation; interleaved: [out0][guard 3000 B][in0][out1][guard]...; reversed_block: outputs then inputs, each group
            # contiguous with 256-byte alignment, handed out from the TOP of the block downwards
            order = []
            if layout == "interleaved":
                for k in range(max(len(ins), len(outs))):
                    if k < len(outs):
                        order.append(("out",) + outs[k])
                        order.append(("guard", f"g{k}", (3000,), np.uint8))
                    if k < len(ins):
                        order.append(("in",) + ins[k])
            else:
                order = [("out",) + o for o in reversed(outs)] + [("guard", "g", (4096,), np.uint8)] + [("in",) + x for x in reversed(ins)]
            sizes = [-(-int(np.prod(shp)) * np.dtype(dt).itemsize // 256) * 256 for _, _, shp, dt in order]
            block = pl.pinned_empty((sum(sizes),), np.uint8)
            blocks = [block]
            inputs, outputs, off = {}, {}, 0
            for (kind, name, shp, dt), sz in zip(order, sizes):
                n = int(np.prod(shp)) * np.dtype(dt).itemsize
                view = block[off:off + n].view(dt).reshape(shp)
                if kind == "guard":
                    view[...] = 0xA5
                    guards.append(view)
                else:
                    (inputs if kind == "in" else outputs)[name] = view
                off += sz
        for k, v in host.items():
            inputs[k][...] = v
        slot = _Arrays(inputs, outputs, B, P, mo, M)
        for rep in range(5):                               # more calls than the pipeline is deep: pools are taken over
            for o in outputs.values():
                o[...] = 0
            pl.plan_cycle(p, q, sp, None, None, None, None, None, None, None, None, slot=slot)
            pl.synchronize()
            got = _masked({k: np.array(outputs[k]) for k in OUTPUTS})
            _same(got, want, f"{layout} call {rep}")
            for g in guards:
                assert (g == 0xA5).all(), f"{layout}: bytes between two arrays of the call were overwritten"
        pl.set_pipeline(0)
        for blk in blocks:
            pl.pinned_free(blk)
    finally:
        pl.set_pipeline(0)
        pl.close()


def test_more_batches_in_flight_than_the_pipeline_is_deep_from_several_threads():
    """service.CycleStream keeps one ring per layout / capacity, so several rings x four slots may be in flight on a pipeline
    that is four pools deep, and sessions wait for their results (emp_wait_ticket) from other threads while submits take pools
    over.  A waiter must never return before ITS batch's outputs have landed (the pool's ticket used to change
    before the new call had waited for the old one).  Eight threads, three capacities, every result against the serial path."""
    import threading
    from emplanner_carla_amd import service
    from emplanner_carla_amd.api import Planner, dp_params
    from tests.conftest import load_golden
    from tests.test_wire import _driver_request
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    dp = dp_params(sample_s=14.7)
    plain = Planner(0)
    want = service.plan_requests(plain, reqs, dp=dp)
    plain.close()
    planner = Planner(0)
    stream = service.CycleStream(planner, capacity=4, max_static=4)      # capacities 4, 8, 16, 32: a ring each
    errors = []

    def session(k):
        try:
            for r in range(40):
                n = (3, 7, 13, 29)[(k + r) % 4]
                pick = [(k * 7 + r * 3 + j) % len(reqs) for j in range(n)]
                a = service.pack_requests([reqs[c] for c in pick])
                h = stream.submit(a, dp=dp)
                st_ref, match, res, M = stream.result(h)
                for j, c in enumerate(pick):
                    status = int(st_ref[j]) | int(res.status[j])
                    assert status == want[c][1], f"session {k} round {r} request {c}: status"
                    if want[c][0] is None:
                        continue
                    m = int(res.traj_len[j])
                    assert m == len(want[c][0][0]) and np.array_equal(res.traj[j, :m], np.asarray(want[c][0][0])), \
                        f"session {k} round {r} request {c}: trajectory"
        except Exception as exc:          # noqa: BLE001 - reported by the main thread
            errors.append(f"session {k}: {type(exc).__name__}: {exc}")

    try:
        threads = [threading.Thread(target=session, args=(k,)) for k in range(8)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        assert not errors, errors[:3]
        assert not any(t.is_alive() for t in threads)
        assert len(stream._rings) >= 3
        for ring in stream._rings.values():
            assert len(ring.free) == len(ring.slots)
    finally:
        stream.close()
        planner.close()
