import time, sys, os
t0=time.perf_counter()
sys.path.insert(0, "bayesian-coresets_amd")
import numpy as np
import bayesiancoresets_amd as bc
t1=time.perf_counter(); print("import %.3f" % (t1-t0))
X=np.random.RandomState(0).randn(1000, 32)
for i in range(3):
    t=time.perf_counter(); s=bc.snnls.GIGA(X.T, X.sum(axis=0)); t2=time.perf_counter(); s.build(10); t3=time.perf_counter()
    print("ctor %.3f s  build %.3f s" % (t2-t, t3-t2))
import torch
t=time.perf_counter(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); print("torch cuda init %.3f" % (time.perf_counter()-t))
