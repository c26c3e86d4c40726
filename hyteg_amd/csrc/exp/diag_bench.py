import sys, time, json
sys.path.insert(0, '.')
import numpy as np, torch
from hyteg_amd import capi, host
level=8
st=host.Storage.from_gmsh('hyteg_amd/data/meshes/tet_1el.msh')
stream=torch.cuda.current_stream(); st.set_stream(stream.cuda_stream)
A=host.P1ConstantOperator(st,level,level)
n=capi.cell_size(level); nbuf=9
rng=np.random.default_rng(0)
srcs=[host.P1Function(st,'s%d'%k,level,level) for k in range(nbuf)]
dsts=[host.P1Function(st,'d%d'%k,level,level) for k in range(nbuf)]
for f in srcs: f.upload_cell(0,level,rng.random(n))
w=list(A.stencils(0,level)[0])
sp=[f.cell_pointer(0,level) for f in srcs]; dp=[f.cell_pointer(0,level) for f in dsts]
ts=[torch.rand(n,dtype=torch.float64,device='cuda') for _ in range(nbuf)]
td=[torch.zeros(n,dtype=torch.float64,device='cuda') for _ in range(nbuf)]
def timeit(name, fn, steps=2000):
    for k in range(50): fn(k)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    t0=time.perf_counter(); e0.record(stream)
    for k in range(steps): fn(k)
    t_issue=time.perf_counter()-t0
    e1.record(stream); torch.cuda.synchronize()
    print(f"{name:55s} dev {e0.elapsed_time(e1)*1e3/steps:7.2f} us/step   host-issue {t_issue*1e6/steps:7.2f} us/step", flush=True)
timeit("capi.p1_apply_cell on host-layer (hipMalloc) buffers", lambda k: capi.p1_apply_cell(dp[k%nbuf],sp[k%nbuf],level,w,0,stream.cuda_stream))
timeit("capi.p1_apply_cell on torch buffers", lambda k: capi.p1_apply_cell(td[k%nbuf].data_ptr(),ts[k%nbuf].data_ptr(),level,w,0,stream.cuda_stream))
timeit("host.P1ConstantOperator.apply (python loop)", lambda k: A.apply(srcs[k%nbuf],dsts[k%nbuf],level,host.Inner))
def cyc(k):
    if k % 100 == 0: A.apply_cycle(srcs,dsts,level,host.Inner,0,k,100)
timeit("host apply_cycle (C++ loop, 100 per call)", cyc)
