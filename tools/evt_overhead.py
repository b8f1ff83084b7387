import sys; sys.path.insert(0,'.')
import torch, numpy as np
import gromacs_cosmo_amd as g
from gromacs_cosmo_amd import abi
ip=g.system.ewald(1.0); s=g.system.water_box(5,5,6,model='spce')
nb=g.gpu.NbnxnHip(0); p=g.workload.prepare(s,ip,nb.min_ci_balanced()); g.workload.upload(nb,p,ip)
fl=abi.FORCE_VIRIAL
nb.compute(p.xq,fl)
st=torch.cuda.ExternalStream(nb.stream(abi.LOCAL))
def ev(): return torch.cuda.Event(enable_timing=True)
for K in (1,10,100):
    a,b=ev(),ev()
    nb.clear_outputs(fl); torch.cuda.synchronize()
    a.record(st)
    for _ in range(K): nb.launch_kernel_resident(fl)
    b.record(st); torch.cuda.synchronize()
    print("b2b K",K, a.elapsed_time(b)/K*1e3,"us per kernel")
# empty pair
ts=[]
for _ in range(50):
    a,b=ev(),ev(); a.record(st); b.record(st); torch.cuda.synchronize(); ts.append(a.elapsed_time(b)*1e3)
print("empty event pair us", np.median(ts))
# clear + kernel per step, events around kernel
es=[]
for _ in range(100):
    a,b=ev(),ev(); nb.clear_outputs(fl); a.record(st); nb.launch_kernel_resident(fl); b.record(st); es.append((a,b))
torch.cuda.synchronize()
print("per-step events around kernel us", np.mean([a.elapsed_time(b) for a,b in es])*1e3)
es=[]
for _ in range(100):
    a,b=ev(),ev(); a.record(st); nb.clear_outputs(fl); nb.launch_kernel_resident(fl); b.record(st); es.append((a,b))
torch.cuda.synchronize()
print("per-step events around clear+kernel us", np.mean([a.elapsed_time(b) for a,b in es])*1e3)
