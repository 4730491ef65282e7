"""Does RCCL beside the planner's streams disturb the pipeline forms?  One rank (all a one-GPU box allows), a real ProcessGroupNCCL,
4096 scenes per step, every step's records packed on the result stream and all_gather_into_tensor'ed on a side stream - the
N > 1 per-step traffic of bench.py with RCCL's own streams and kernels in the process.  Prints ms per step for the pipeline
mode given (staged | 3 | ...), with and without the collective.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 tools/rccl_lanes_probe.py 3"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emplanner_carla_amd import _lib as L
L.configure_hw_queues(int(os.environ.get("EMP_PROBE_QUEUES", "8")))
import numpy as np, torch, torch.distributed as dist
from emplanner_carla_amd import dist as emp_dist, scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
mode = sys.argv[1] if len(sys.argv) > 1 else "3"
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local); dev = torch.device("cuda", local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
cfg, B = S.CFG2, 4096
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ring = []
for k in range(4):
    b = S.make_batch(range(k * B, (k + 1) * B), cfg, start_ahead=S.BENCH_START_AHEAD); P = b.ref.shape[1]
    ring.append(dict(ref_line=t(b.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy),
                     start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs)))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(local); pl.set_pipeline("staged" if mode == "staged" else int(mode)); ts = pl.torch_stream()
side = torch.cuda.Stream(device=dev)
keep = []
def step(i, collective):
    with torch.cuda.stream(ts):
        res = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **ring[i % 4])
    if not collective:
        return
    rs = pl.torch_result_stream()
    with torch.cuda.stream(rs):
        rec = emp_dist.pack_records(res, p.col, M, path_cap=emp_dist.path_capacity(M), planner=pl, fields="full")
    side.wait_stream(rs)
    with torch.cuda.stream(side):
        out = torch.empty((B * world, rec.shape[1]), dtype=rec.dtype, device=dev)
        dist.all_gather_into_tensor(out, rec)
    keep.append((rec, out))
    if len(keep) > 8: keep.pop(0)
def fence(): pl.synchronize(); torch.cuda.synchronize()
for collective in (False, True, False, True):
    for i in range(150): step(i, collective)
    fence()
    t0 = time.perf_counter()
    for i in range(100): step(i, collective)
    fence()
    print(f"pipeline {mode}: {'with' if collective else 'without'} pack + RCCL all_gather_into_tensor: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms per step", flush=True)
rec, out = keep[-1]
assert torch.equal(out, rec)
pl.close(); dist.destroy_process_group()
