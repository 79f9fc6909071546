import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pyhillfit_amd
from pyhillfit_amd import doseresponse as dr
from pyhillfit_amd.sampler import SingleLevelSampler
dr.setup("data/crumb_dataset.json"); dr.define_model(2)
names=[(d,c) for d in dr.drugs for c in dr.channels]
packed0=dr.pack_single_level(names)
cost=packed0.counts[:,0]*1.0+packed0.counts[:,1]*3.0+packed0.counts[:,2]*3.0
orders={"file":np.arange(len(names)), "lpt":np.argsort(-cost,kind="stable"), "spt":np.argsort(cost,kind="stable")}
dev=torch.device("cuda",0)
for tag,order in orders.items():
    nm=[names[i] for i in order]
    packed=dr.pack_single_level(nm)
    s=SingleLevelSampler(packed,2,list(range(len(nm))),[1.0]*len(nm),4096,thinning=5,seed=25,device=dev)
    s.init([6.0,0.8,8.0],cov_identity=False,cov_scale=0.05)
    I=2000; s.reserve(12*I)
    rows=torch.empty((s.rows_between(0,I),len(nm),4,4096),dtype=torch.float64,device=dev)
    for _ in range(4): s.advance(I,out=rows)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(6): s.advance(I,out=rows)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/6
    print(tag, "ms/launch %.2f"%(dt*1e3), "samples/s %.4g"%(len(nm)*4096*I/dt))
    del s, rows
