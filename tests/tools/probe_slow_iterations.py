import sys, time, numpy as np
sys.path.insert(0,'.')
from plssvm_amd import backend
from plssvm_amd._capi import Options
from plssvm_amd.parameter import Parameter
rng=np.random.default_rng(7)
n,d=50000,128
y=np.where(np.arange(n)%2==0,1.0,-1.0).astype(np.float32)
shift=(y[:,None]>0)*0.5
cases={"sparse01 linear bf16x6":((rng.random((n,d))<0.05+0.03*shift).astype(np.float32),"linear",1),
       "counts rbf default":(rng.poisson(3.0+shift,size=(n,d)).astype(np.float32),"rbf",3),
       "counts rbf bf16x6":(rng.poisson(3.0+shift,size=(n,d)).astype(np.float32),"rbf",1)}
for name,(X,kernel,gm) in cases.items():
    print("==",name)
    with backend.ResidentProblem(Parameter(kernel_type=kernel,gamma=1.0/d),X,options=Options(gram_mode=gm)) as prob:
        prob.cg_begin(y,1e-30)
        i_prev=prob.info()
        for it in range(30):
            t0=time.perf_counter(); prob.cg_step(1); prob.synchronize(); dt=(time.perf_counter()-t0)*1e3
            i=prob.info()
            print(f" it {it:2d} wall {dt:7.3f} ms  delta {i['residuum']:.4e}  timed {i['matvec_timed']} kern_total {i['matvec_kernel_ms_total']:.3f} gram {i['gram_mode']} direct {i['rbf_direct']} sym {i['symmetric']} launches/mv {i['tile_launches_per_matvec']} persistent {i['persistent_launches']}")
