#!/usr/bin/env python3
"""Where a backward step of the large-state path spends its cycles (diagnostic hook)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ilqr_amd import _hip  # noqa: E402
from ilqr_amd.synthetic import SyntheticWorkload  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
if len(sys.argv) > 2:  # a build with -DILQR_LARGE_QP_PROF: also the parts of a boxQP pass
    _hip.LIB_PATH = os.path.abspath(sys.argv[2])
T = 200
dev = torch.device("cuda:0")
wl = SyntheticWorkload(T=T)
M = wl.device_model()
x0, us = wl.batch(B)
x0, us = torch.tensor(x0, device=dev), torch.tensor(us, device=dev)
xs, _ = _hip.large_rollout_cost(M, x0, us)
mu = torch.ones(B, dtype=torch.float64, device=dev)
buf = torch.zeros((B, 8), dtype=torch.int64, device=dev)
_hip.check(_hip.lib().ilqr_large_debug_profile(C.c_void_p(buf.data_ptr())))
qp_parts = hasattr(_hip.lib(), "ilqr_large_debug_qp_phases")
if qp_parts:
    _hip.lib().ilqr_large_debug_qp_phases(None, 1)
_hip.large_backward(M, _hip.Opts(), xs, us, mu, want_masks=False)
torch.cuda.synchronize()
if qp_parts:
    acc = (C.c_ulonglong * 8)()
    _hip.lib().ilqr_large_debug_qp_phases(acc, 0)
    names = ["gradient + classification", "factorisation", "gradient norm", "clamped right-hand side", "solve + search",
             "line search (64 trips at once)"]
    print("boxQP passes %d in %d calls; shader clocks per pass:" % (acc[6], acc[7]))
    for nm, v in zip(names, acc):
        print("   %-32s %7.0f" % (nm, v / max(1, acc[6])))
_hip.check(_hip.lib().ilqr_large_debug_profile(None))
p = buf.cpu().numpy().astype(np.float64)
per = p[:, :3] / p[:, 3:4]
print("cycles per backward step, mean over %d problems: matrix part 1 %.0f, boxQP + gains %.0f, matrix part 2 %.0f"
      % (B, per[:, 0].mean(), per[:, 1].mean(), per[:, 2].mean()))
print("max over problems: %.0f %.0f %.0f" % tuple(per.max(0)))
print("inside the second phase: warm start (factor + solve) %.0f, boxQP proper %.0f cycles per step; %.2f boxQP "
      "iterations and %.2f factorisations per step" % ((p[:, 4] / p[:, 3]).mean(), (p[:, 5] / p[:, 3]).mean(),
                                                       (p[:, 6] / p[:, 3]).mean(), (p[:, 7] / p[:, 3]).mean()))
