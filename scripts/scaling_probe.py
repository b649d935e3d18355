"""Diagnostic: solve time versus batch size and iteration cap."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench

dev = torch.device("cuda:0")
Q, R, Qt, xg = bench.cartpole_cost()
M = _hip.Model(100, Q, R, Qt, xg, None, [-5.0], [5.0])
for max_iter in (2, 20, 200):
    for B in (16384, 32768, 49152, 65536, 131072):
        x0_h, us_h = bench.make_batch(B, 0)
        x0 = torch.tensor(x0_h, device=dev); us = torch.tensor(us_h, device=dev)
        O = _hip.Opts(max_iter=max_iter)
        ws = torch.empty(_hip.solve_workspace_bytes(M, B), dtype=torch.uint8, device=dev)
        res = _hip.solve(M, O, x0, us, workspace=ws); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); res = _hip.solve(M, O, x0, us, workspace=ws); b.record(); torch.cuda.synchronize()
        print("max_iter %3d  B %6d  %8.2f ms   %.3f ms/iter" % (max_iter, B, a.elapsed_time(b), a.elapsed_time(b) / max_iter), flush=True)
        del ws, x0, us, res
