"""The planner service's wire format (emplanner_carla_amd/wire.py): fixed-stride request /
reply records over TCP in place of the reference's Pipe tuples (test_9.py:220, :390-392).  CPU: framing, record
round trips and the server loop with a stub in place of the GPU; GPU: the reference driver's 18 recorded requests
through a real server against the recorded replies."""
import threading

import numpy as np
import pytest

from tests.conftest import assert_rel, load_golden

RTOL = 1e-6


def _driver_request(g, c):
    ns, nd = int(g["n_static"][c]), int(g["n_dynamic"][c])
    return ([tuple(r) for r in g["static"][c, :ns]], [tuple(r) for r in g["dynamic"][c, :nd]], tuple(g["veh"][c]),
            tuple(g["pred"][c]), tuple(g["v"][c]), tuple(g["a"][c]), [tuple(r) for r in g["path"][c]], [int(g["pre_match"][c])])


def _start(plan_arrays, **kw):
    from emplanner_carla_amd import wire
    srv = wire.PlannerServer(plan_arrays, **kw)
    th = threading.Thread(target=srv.serve_forever, daemon=True)
    th.start()
    return srv


def test_request_records_decode_to_the_packed_arrays_of_the_pipe_path():
    """What the server decodes from the wire equals what service.pack_requests builds from the same tuples (the host
    logic - 30 m rule for static obstacles, first dynamic obstacle only - lives in both)."""
    from emplanner_carla_amd import service, wire
    from emplanner_carla_amd.api import dp_params
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    lay = wire.Layout(dp_params(sample_s=14.7), max_static=4, max_dynamic=2)
    paths, pids = {}, []
    for r in reqs:                                            # the drive used several global paths: one id per content
        key = np.asarray(r[6]).tobytes()
        pids.append(paths.setdefault(key, len(paths) + 1))
    by_id = {pid: np.frombuffer(key).reshape(-1, 4) for key, pid in paths.items()}
    payload = lay.encode_requests(reqs, pids, request_ids=list(range(100, 100 + len(reqs))))
    assert len(payload) == len(reqs) * lay.request_stride
    ids, a = lay.decode_requests(payload, len(reqs), by_id)
    want = service.pack_requests(reqs)
    assert list(ids) == list(range(100, 100 + len(reqs)))
    for k in ("n_global", "pred", "veh", "v", "a", "pre_match", "n_obs"):
        assert np.array_equal(a[k], want[k]), k
    assert np.array_equal(a["dyn"], want["dyn"], equal_nan=True)
    assert np.array_equal(a["global_path"], want["global_path"])
    K = want["obs_xy"].shape[1]
    assert np.array_equal(a["obs_xy"][:, :K], want["obs_xy"])
    with pytest.raises(wire.WireError):
        lay.decode_requests(payload[:-1], len(reqs), by_id)
    with pytest.raises(wire.WireError):
        lay.decode_requests(payload, len(reqs), {99: by_id[1]})
    with pytest.raises(wire.WireError):
        wire.Layout(dp_params(), max_static=1).encode_requests(reqs, pids)


def test_server_loop_with_a_stub_planner():
    """HELLO (stride agreement), SET_PATH once per distinct path, PLAN -> REPLY in request order, failed requests as
    ok = 0 records, a protocol error as an ERROR frame that leaves the session usable."""
    from emplanner_carla_amd import wire
    from emplanner_carla_amd.api import CycleResult, dp_params, max_path_points
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(6)]
    seen = []

    boom = []

    def stub(arrays, dp):
        if boom:
            raise RuntimeError("planner on fire")
        B, M = len(arrays["pred"]), max_path_points(dp)
        seen.append(arrays)
        k = np.arange(B, dtype=np.float64)
        traj = np.zeros((B, M + 1, 4))
        traj[:, :, 0] = k[:, None] * 100 + np.arange(M + 1)
        traj[:, :, 1] = arrays["pred"][:, 1:2]
        path_s = np.tile(np.arange(M, dtype=np.float64), (B, 1)) + k[:, None]
        status = np.where(np.arange(B) % 3 == 2, 8, 0).astype(np.int32)
        res = CycleResult(dp_rows=np.tile(np.arange(dp.col, dtype=np.float64), (B, 1)), dp_s=None, dp_l=None, dp_len=None,
                          path_s=path_s, path_l=-path_s, path_len=np.full(B, 5, np.int32), traj=traj,
                          traj_len=np.full(B, 6, np.int32), status=status)
        return np.zeros(B, np.int32), arrays["pre_match"] + 1, res, M

    srv = _start(stub)
    try:
        cl = wire.PlannerClient(*srv.address, dp=dp_params(sample_s=14.7), max_static=4, max_dynamic=2)
        out = cl.plan(reqs)
        n_paths = len(cl._paths)
        assert len(seen) == 1 and 1 <= n_paths <= 6 and seen[0]["global_path"].shape[0] == 6
        for k, (reply, status) in enumerate(out):
            if k % 3 == 2:
                assert reply is None and status == 8
                continue
            traj, match, ps, pl = reply
            assert status == 0 and match == [int(g["pre_match"][k]) + 1]
            assert len(traj) == 6 and traj[3] == (k * 100 + 3.0, float(g["pred"][k][1]), 0.0, 0.0)
            assert ps == [float(k + j) for j in range(5)] and pl == [-float(k + j) for j in range(5)]
        out2 = cl.plan(reqs[:2])                              # the path is not sent again
        assert len(cl._paths) == n_paths and len(out2) == 2 and out2[0][0] is not None
        # a malformed PLAN: the server answers ERROR and keeps the session
        wire.send_frame(cl.sock, wire.T_PLAN, b"x" * 10, 1)
        ftype, _, payload = wire.recv_frame(cl.sock)
        assert ftype == wire.T_ERROR and b"records" in payload
        assert cl.plan(reqs[:1])[0][0] is not None
        # so do a truncated SET_PATH (too short to hold its own header) and an error raised by the planner itself
        wire.send_frame(cl.sock, wire.T_SET_PATH, b"\x01\x00")
        ftype, _, payload = wire.recv_frame(cl.sock)
        assert ftype == wire.T_ERROR and payload
        boom.append(True)
        with pytest.raises(wire.WireError, match="planner on fire"):
            cl.plan(reqs[:1])
        boom.clear()
        assert cl.plan(reqs[:1])[0][0] is not None
        cl.close()
        # a client built for another lattice cannot talk to a server that derives other strides: HELLO carries both
        lay = wire.Layout(dp_params(col=7))
        bad = bytearray(lay.hello())
        bad[-8:-4] = (lay.cap + 1).to_bytes(4, "little")
        with pytest.raises(wire.WireError):
            wire.Layout.from_hello(bytes(bad))
        # a HELLO sizes every buffer of the session: lattices and capacities no planner means are refused before anything
        # is allocated from them, non-positive spacings before anything divides by them (round-2 review)
        import struct
        good = list(wire._HELLO.unpack(wire.Layout(dp_params()).hello()))
        for field, value in ((0, 10 ** 6), (0, 0), (1, 10 ** 7), (1, -3), (10, 2 ** 31), (11, 10 ** 5), (2, 0.0), (4, 0.0), (4, -1.0),
                             (3, float("nan")), (2, float("inf")), (4, 1e-9)):
            f = list(good)
            f[field] = value
            with pytest.raises(wire.WireError, match="HELLO"):
                wire.Layout.from_hello(wire._HELLO.pack(*f))
        # and a session survives a hostile HELLO: ERROR frame, then business as usual
        cl = wire.PlannerClient(*srv.address, dp=dp_params(sample_s=14.7), max_static=4, max_dynamic=2)
        f = list(good)
        f[1] = 10 ** 7
        wire.send_frame(cl.sock, wire.T_HELLO, wire._HELLO.pack(*f))
        ftype, _, payload = wire.recv_frame(cl.sock)
        assert ftype == wire.T_ERROR and b"HELLO" in payload
        assert cl.plan(reqs[:1])[0][0] is not None
        # per-session cap on the global paths held
        small = wire.PlannerServer(stub, max_path_bytes=2000)
        threading.Thread(target=small.serve_forever, daemon=True).start()
        try:
            c2 = wire.PlannerClient(*small.address, dp=dp_params(sample_s=14.7), max_static=4, max_dynamic=2)
            wire.send_frame(c2.sock, wire.T_SET_PATH, struct.pack("<II", 1, 50) + np.zeros((50, 4)).tobytes())      # 1600 B: fine
            wire.send_frame(c2.sock, wire.T_SET_PATH, struct.pack("<II", 2, 50) + np.zeros((50, 4)).tobytes())      # 3200 B: refused
            ftype, _, payload = wire.recv_frame(c2.sock)
            assert ftype == wire.T_ERROR and b"exceed" in payload
            c2.close()
        finally:
            small.shutdown()
        cl.close()
    finally:
        srv.shutdown()


@pytest.mark.gpu
def test_remote_planning_equals_the_reference_driver_run():
    """A real server (one GPU planner) and a client on a socket: the 18 requests of the reference's driver run, one PLAN
    frame, against the recorded replies at 1e-6 - the same bar as the Pipe path (tests/test_gpu_cycle.py)."""
    from emplanner_carla_amd import service, wire
    from emplanner_carla_amd.api import Planner, dp_params
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    planner = Planner(0)
    srv = _start(lambda arrays, dp: service.plan_arrays(planner, arrays, dp=dp))
    try:
        cl = wire.PlannerClient(*srv.address, dp=dp_params(sample_s=14.7), max_static=4, max_dynamic=2)
        replies = cl.plan(reqs)
        local = service.plan_requests(planner, reqs, dp=dp_params(sample_s=14.7))
        compared = 0
        for c, (reply, status) in enumerate(replies):
            assert status == local[c][1]
            if not g["qp_ok"][c]:
                assert reply is None and status & (8 | 16)
                continue
            traj, match, ps, pl = reply
            n, m = int(g["n_traj"][c]), int(g["n_path"][c])
            assert match == [int(g["match"][c])] and len(traj) == n and len(ps) == m
            assert reply == local[c][0], "the wire reply is the Pipe reply, bit for bit"
            assert_rel(np.asarray(traj)[:, :3], g["traj"][c, :n, :3], RTOL, f"request {c}: trajectory")
            assert_rel(np.asarray(pl), g["path_l"][c, :m], RTOL, f"request {c}: path_l")
            compared += 1
        assert compared >= 12
        cl.close()
    finally:
        srv.shutdown()
        planner.close()


@pytest.mark.gpu
def test_overlapped_server_with_concurrent_sessions_equals_the_serial_path():
    """What ``wire.serve`` runs: the server on ``service.CycleStream`` (page-locked rings, staged pipeline, no server-side
    lock).  Six client sessions - more than the ring has slots - fire PLAN frames of different sizes at once, twelve rounds each; every reply must be the
    reply of the serial path (``service.plan_requests`` on a plain planner), bit for bit, whatever overlapped with it."""
    import threading
    from emplanner_carla_amd import service, wire
    from emplanner_carla_amd.api import Planner, dp_params
    g = load_golden("driver_s147.npz")
    reqs = [_driver_request(g, c) for c in range(len(g["case"]))]
    dp = dp_params(sample_s=14.7)
    plain = Planner(0)
    want = service.plan_requests(plain, reqs, dp=dp)
    plain.close()
    planner = Planner(0)
    stream = service.CycleStream(planner, capacity=32, max_static=4)
    srv = _start(lambda arrays, dp_: stream.plan_arrays(arrays, dp=dp_), overlapped=True)
    errors = []

    def session(k):
        try:
            cl = wire.PlannerClient(*srv.address, dp=dp, max_static=4, max_dynamic=2)
            for r in range(12):
                pick = [(k * 5 + r * 3 + j) % len(reqs) for j in range(3 + 4 * (k % 4))]
                got = cl.plan([reqs[c] for c in pick])
                for c, (reply, status) in zip(pick, got):
                    assert status == want[c][1] and reply == want[c][0], f"session {k} round {r} request {c}"
            cl.close()
        except Exception as exc:          # noqa: BLE001 - reported by the main thread
            errors.append(f"session {k}: {type(exc).__name__}: {exc}")

    try:
        threads = [threading.Thread(target=session, args=(k,)) for k in range(6)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        assert not any(t.is_alive() for t in threads)
    finally:
        srv.shutdown()
        stream.close()
        planner.close()
