"""Diagnostic: which features of a wave (its problems' iteration counts, rejected iterations, boxQP work,
the XCD it ran on) go with its lifetime in the solve kernel, over three runs of the benchmark batch."""
import ctypes as C, os, sys
import numpy as np, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ilqr_amd import _hip
import bench
B=16384
dev=torch.device("cuda:0")
Q,R,Qt,xg=bench.cartpole_cost()
M=_hip.Model(100,Q,R,Qt,xg,None,[-5.0],[5.0])
x0_h,us_h=bench.make_batch(B,0)
x0=torch.tensor(x0_h,device=dev); us=torch.tensor(us_h,device=dev)
nw=B//64
O=_hip.Opts()
_hip.solve(M,O,x0,us); torch.cuda.synchronize()
lifes=[]
for rep in range(3):
    log=torch.zeros((nw,4),dtype=torch.int64,device=dev)
    _hip.check(_hip.lib().ilqr_debug_wave_log(C.c_void_p(log.data_ptr())))
    res=_hip.solve(M,O,x0,us,trace=True); torch.cuda.synchronize()
    _hip.check(_hip.lib().ilqr_debug_wave_log(None))
    L=log.cpu().numpy()
    lifes.append((L[:,1]-L[:,0])/1e5)
    xcc=L[:,3]; hw=L[:,2]
lifes=np.array(lifes)
print("run-to-run correlation of per-wave life:", np.corrcoef(lifes)[0,1], np.corrcoef(lifes)[0,2])
life=lifes.mean(0)
it=res["iters"].cpu().numpy().reshape(nw,64)
tr={k:v.cpu().numpy() for k,v in res["trace"].items()}   # [max_iter,B]
acc=tr["accepted"].reshape(200,nw,64); qpb=tr["qp_backtrack"].reshape(200,nw,64); ncl=tr["n_clamped"].reshape(200,nw,64); qpi=tr["qp_iter"].reshape(200,nw,64)
feat={
 "sum iters":it.sum(1), "lanes at 200":(it==200).sum(1),
 "rejected iterations (any lane)":((acc==0).any(2)).sum(0),
 "sum rejected":(acc==0).sum((0,2)),
 "max-lane qp_backtrack sum":np.where(qpb>=0,qpb,0).max(2).sum(0),
 "max-lane qp_iter sum":np.where(qpi>=0,qpi,0).max(2).sum(0),
 "max-lane n_clamped sum":np.where(ncl>=0,ncl,0).max(2).sum(0),
 "xcc":xcc,
}
for k,v in feat.items():
    print("%-34s corr with life %.2f   (mean %.1f)"%(k,np.corrcoef(life,v)[0,1],np.mean(v)))
for x in range(8):
    print("xcc",x,"mean life %.2f"%life[xcc==x].mean(), end="; ")
print()
