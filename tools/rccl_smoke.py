"""RCCL on this stack, one rank: the exact torch.distributed calls of bench.py's N > 1 path (init with device_id, barrier,
all_reduce MAX, all_gather of per-rank times, dist.gather into views of one tensor on a side stream behind a pack kernel,
all_gather_into_tensor) with world_size = 1 - the only RCCL run a one-GPU box allows.  It cannot show scaling; it shows that the
calls, their argument forms and the stream usage are accepted by RCCL / ProcessGroupNCCL here.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_smoke.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from emplanner_carla_amd import _lib as L
from emplanner_carla_amd import dist as emp_dist, scenes as S
from emplanner_carla_amd.api import Planner, dp_params_from_cfg, max_path_points, qp_params, smooth_params
rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local); dev = torch.device("cuda", local)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")      # (torch.distributed.run sets both)
dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)
print("backend", dist.get_backend(), "world", dist.get_world_size(), flush=True)
dist.barrier(); torch.cuda.synchronize()
el = torch.tensor([1.25 + rank], dtype=torch.float64, device=dev)
dist.all_reduce(el, op=dist.ReduceOp.MAX); assert float(el.item()) == 1.25 + world - 1
mine = torch.tensor([0.5], dtype=torch.float64, device=dev); allr = [torch.zeros_like(mine) for _ in range(world)]
dist.all_gather(allr, mine); assert float(allr[0].item()) == 0.5
# the per-step exchange: plan, pack on the result stream, gather on a side stream (StepGather drives dist.gather itself only for
# world > 1, so the collective calls are issued here in the same form)
cfg, B = S.CFG2, 1024
b = S.make_batch(range(B), cfg); P = b.ref.shape[1]
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
inputs = dict(ref_line=t(b.ref), n_ref=t(np.full(B, P, np.int32)), origin_xy=t(b.origin_xy), start_xy=t(b.start_xy),
              start_v=t(b.start_v), start_a=t(b.start_a), obs_xy=t(b.obs_xy), n_obs=t(b.n_obs))
p, q, sp = dp_params_from_cfg(cfg), qp_params(obs_length=cfg.obs_length, obs_width=cfg.obs_width), smooth_params()
M = max_path_points(p); pl = Planner(local); pl.set_pipeline(1); ts = pl.torch_stream()
side = torch.cuda.Stream(device=dev)
outs = []
for step in range(6):
    with torch.cuda.stream(ts):
        res = pl.plan_cycle(p, q, sp, max_pts=M, mode=L.EMP_DP_TWO_KERNEL, **inputs)
    rs = pl.torch_result_stream()
    with torch.cuda.stream(rs):
        rec = emp_dist.pack_records(res, p.col, M, path_cap=emp_dist.path_capacity(M), planner=pl, fields="full")
    side.wait_stream(rs)
    with torch.cuda.stream(side):
        send = rec.contiguous()
        out = torch.empty((B * world, rec.shape[1]), dtype=rec.dtype, device=dev)
        dist.gather(send, list(out.split(B, dim=0)) if rank == 0 else None, dst=0)          # bench.py --gather rank0
        out2 = torch.empty_like(out)
        dist.all_gather_into_tensor(out2, send)                                             # --gather all
        done = torch.cuda.Event(); done.record(side)
    outs.append((rec, out, out2, done))
pl.synchronize(); torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
rec, out, out2, _ = outs[-1]
assert torch.equal(out, rec) and torch.equal(out2, rec), "gathered records differ from the packed ones"
st = emp_dist.unpack_records(out, p.col, M, path_cap=emp_dist.path_capacity(M))["status"].cpu().numpy()
print("rccl smoke ok: 6 steps, gather + all_gather_into_tensor on a side stream, records complete;", int(((st & ~1) == 0).sum()), "of", B, "scenes planned")
pl.close(); dist.destroy_process_group()
