import numpy as np, sys
a,b=np.load(sys.argv[1]),np.load(sys.argv[2])
for k in a.files:
    if k in ("comm_kind",): print(k,a[k],b[k]); continue
    x,y=a[k].astype(np.float64),b[k].astype(np.float64)
    print(k, x.shape, "rel", np.linalg.norm(x-y)/max(np.linalg.norm(x),1e-300), x.ravel()[:6], y.ravel()[:6])
