"""Development probe: what ONE synchronous library call on one scene costs, by where its arrays live - ordinary (pageable) NumPy arrays
(EMP_HOST), page-locked arrays (emp_host_alloc) and device tensors - for emp_s_map (3 small inputs, 1 output, a 2 us kernel)."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from emplanner_carla_amd import _lib as L
from emplanner_carla_amd import scenes as S
from emplanner_carla_amd.api import Planner

b = S.make_batch([7], S.CFG2)
P = b.ref.shape[1]
pl = Planner(0)
line, n_ref, org = np.ascontiguousarray(b.ref), np.full(1, P, np.int32), np.ascontiguousarray(b.origin_xy)
out = np.zeros((1, P))
ptr = lambda a: C.c_void_p(a.ctypes.data)

def timeit(f, n=2000):
    for _ in range(50): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6

print("api.s_map, numpy arrays:            %.1f us" % timeit(lambda: pl.s_map(line, n_ref, org)))
print("raw ctypes call, pageable arrays:   %.1f us" % timeit(lambda: pl._lib.emp_s_map(pl._h, 1, P, ptr(line), ptr(n_ref), ptr(org), ptr(out), L.EMP_HOST)))
pin = {k: pl.pinned_empty(v.shape, v.dtype) for k, v in dict(line=line, n_ref=n_ref, org=org, out=out).items()}
for k, v in dict(line=line, n_ref=n_ref, org=org).items():
    pin[k][...] = v
print("raw ctypes call, page-locked arrays (EMP_HOST): %.1f us" % timeit(lambda: pl._lib.emp_s_map(pl._h, 1, P, ptr(pin["line"]), ptr(pin["n_ref"]), ptr(pin["org"]), ptr(pin["out"]), L.EMP_HOST)))
print("raw ctypes call, page-locked arrays (EMP_HOST_PINNED): %.1f us" % timeit(lambda: pl._lib.emp_s_map(pl._h, 1, P, ptr(pin["line"]), ptr(pin["n_ref"]), ptr(pin["org"]), ptr(pin["out"]), L.EMP_HOST_PINNED)))
dev = {k: torch.from_numpy(v).cuda() for k, v in dict(line=line, n_ref=n_ref, org=org, out=out).items()}
dp = lambda t: C.c_void_p(t.data_ptr())
def devcall():
    pl._lib.emp_s_map(pl._h, 1, P, dp(dev["line"]), dp(dev["n_ref"]), dp(dev["org"]), dp(dev["out"]), L.EMP_DEVICE)
    pl.synchronize()
print("raw ctypes call, device tensors + synchronize: %.1f us" % timeit(devcall))
def devcall_nosync():
    pl._lib.emp_s_map(pl._h, 1, P, dp(dev["line"]), dp(dev["n_ref"]), dp(dev["org"]), dp(dev["out"]), L.EMP_DEVICE)
print("raw ctypes call, device tensors, no wait: %.1f us" % timeit(devcall_nosync)); pl.synchronize()
assert np.array_equal(out, pin["out"]) and np.array_equal(out, dev["out"].cpu().numpy())
# copies alone
h = np.zeros(51 * 4)
d = torch.zeros(51 * 4, dtype=torch.float64, device="cuda")
st = torch.cuda.Stream()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
def cp(src_ptr, dst_ptr, kind):
    hip.hipMemcpyAsync(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), C.c_size_t(51 * 4 * 8), C.c_int(kind), C.c_void_p(st.cuda_stream))
print("hipMemcpyAsync H2D 1.6 KB pageable (issue only): %.1f us" % timeit(lambda: cp(h.ctypes.data, d.data_ptr(), 1))); st.synchronize()
ph = pl.pinned_empty((51 * 4,), np.float64)
print("hipMemcpyAsync H2D 1.6 KB page-locked (issue only): %.1f us" % timeit(lambda: cp(ph.ctypes.data, d.data_ptr(), 1))); st.synchronize()
def cps(srcp, dstp, kind):
    cp(srcp, dstp, kind); hip.hipStreamSynchronize(C.c_void_p(st.cuda_stream))
print("H2D 1.6 KB pageable + stream sync: %.1f us" % timeit(lambda: cps(h.ctypes.data, d.data_ptr(), 1)))
print("H2D 1.6 KB page-locked + stream sync: %.1f us" % timeit(lambda: cps(ph.ctypes.data, d.data_ptr(), 1)))
print("D2H 1.6 KB pageable + stream sync: %.1f us" % timeit(lambda: cps(d.data_ptr(), h.ctypes.data, 2)))
print("D2H 1.6 KB page-locked + stream sync: %.1f us" % timeit(lambda: cps(d.data_ptr(), ph.ctypes.data, 2)))
pl.close()
