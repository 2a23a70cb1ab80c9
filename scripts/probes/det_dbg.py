import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tangram_amd.mapping_optimizer as mo
from tangram_amd.batched import train_many
from tangram_amd.synthetic import make_workload
dev="cuda:0"
C,K,V=18,250,9852
w=make_workload(C,K,V,dev,seed=1)
S,G,d=w["S"].cpu().numpy(),w["G"].cpu().numpy(),w["d"].cpu().numpy()
ds=np.full(C,1.0/C,np.float32)
b=lambda: mo.Mapper(S=S,G=G,d=d,d_source=ds,lambda_d=1,device=dev,random_state=3)
for E in (1,2,5,50,1000):
    seq=[b().train(num_epochs=E,learning_rate=0.1,print_each=None) for _ in range(3)]
    print(E,"sequential identical:",[np.array_equal(seq[0][0],x[0]) for x in seq[1:]], "hist", [seq[0][1]["main_loss"][-1]==x[1]["main_loss"][-1] for x in seq[1:]])
    res,_=train_many([b]*6,E,0.1,max_concurrent=6,device=dev)
    print(E,"concurrent vs seq:",[np.array_equal(seq[0][0],x[0]) for x in res], [float(np.abs(seq[0][0]-x[0]).max()) for x in res])
    # side stream but one at a time
    res1,_=train_many([b]*3,E,0.1,max_concurrent=1,device=dev)
    print(E,"side-stream serial vs seq:",[np.array_equal(seq[0][0],x[0]) for x in res1])
