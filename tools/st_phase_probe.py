"""Development probe (needs -DEMP_ST_PROBE=k builds under variants/, see csrc/emp_st_kernels.h): where the S-T speed DP kernel of
configs[4] spends its time - its duration alone on the 4096 benchmark scenes (16 dynamic-obstacle slots) for builds with the
obstacle work removed layer by layer.  Results of a gutted kernel mean nothing; durations do.
Usage: python tools/st_phase_probe.py variants/lib_st0.so variants/lib_st1.so ..."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib):
    sys.path.insert(0, ROOT)
    from emplanner_carla_amd import _lib
    _lib.LIB_PATH = os.path.abspath(lib)
    import torch
    from emplanner_carla_amd import scenes as S
    from emplanner_carla_amd.api import Planner, speed_dp_params
    dyn = S.make_dynamic_batch(range(4096), 16)
    dev = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in dyn[:5]]
    pl = Planner(0)
    sdp = speed_dp_params()
    sets = pl.st_graph(*dev[:4])
    live = int((~torch.isnan(sets[0])).sum().item())
    for _ in range(3):
        pl.speed_dp(sdp, *sets, dev[4], tables=False)
    pl.synchronize()
    pl.set_timing(True, only="speed_dp")
    for _ in range(10):
        pl.speed_dp(sdp, *sets, dev[4], tables=False)
    pl.synchronize()
    print(f"{os.path.basename(lib):16s} speed_dp {pl.kernel_ms('speed_dp') * 1e3:8.1f} us per 4096 scenes   ({live / 4096:.2f} S-T obstacles a scene)", flush=True)
    pl.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for lib in sys.argv[1:]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
