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
