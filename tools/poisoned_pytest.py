"""Run pytest in a process whose free device memory was first filled with a garbage pattern and released (what a
previous tenant of the GPU leaves behind): python tools/poisoned_pytest.py <nan|random|huge> <GiB> <pytest args...>"""
import sys

import pytest
import torch

kind, gib = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda:0")
chunks = []
for _ in range(gib):
    c = torch.empty(1 << 27, dtype=torch.float64, device=dev)
    if kind == "nan":
        c.fill_(float("nan"))
    elif kind == "huge":
        c.fill_(-1e300)
    else:
        c.view(torch.int64).random_()
    chunks.append(c)
torch.cuda.synchronize()
del chunks, c
torch.cuda.empty_cache()
sys.exit(pytest.main(sys.argv[3:]))
