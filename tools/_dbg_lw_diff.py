import os, sys, subprocess, numpy as np
here=os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"tests")
n,m,h1,h2,N=64,6,256,256,3037
res={}
for tag,val in (("general","0"),("persistent","1")):
    out="/tmp/%s.npz"%tag
    subprocess.run([sys.executable, os.path.join(here,"_lw_fvp_worker.py"), out, str(n),str(m),str(h1),str(h2),str(N)], check=True, env=dict(os.environ, MJX_LW_PERSIST=val))
    res[tag]=np.load(out)
for k in ("g","hv"):
    a,b=res["general"][k],res["persistent"][k]
    d=np.nonzero(a!=b)[0]
    print(k, "differing", d.size, "of", a.size, "first", d[:10], "max rel", np.max(np.abs(a-b)/(np.abs(a)+1e-30)) if d.size else 0)
    o=[0,n*h1,n*h1+h1,n*h1+h1+h1*h2,n*h1+h1+h1*h2+h2]
    for i in range(len(o)-1):
        print("   block",i,"diffs",np.count_nonzero((d>=o[i])&(d<o[i+1])))
    print("   tail diffs", np.count_nonzero(d>=o[-1]))
a,b=res["general"]["hv"].astype(np.float64),res["persistent"]["hv"].astype(np.float64)
W=a[:n*h1].reshape(h1,n); Wp=b[:n*h1].reshape(h1,n)
rd=np.linalg.norm(W-Wp,axis=1)/np.linalg.norm(W,axis=1)
print("gW1 row rel diffs (units 12..52):", np.round(rd[12:52],3))
b1=a[n*h1:n*h1+h1]; b1p=b[n*h1:n*h1+h1]
print("gb1 rel diffs > 1e-5:", np.nonzero(np.abs(b1-b1p)>1e-5*np.abs(b1).max())[0][:40])
print("feature-wise diff norms for unit 20:", np.round(np.abs(W[20]-Wp[20])/np.abs(W[20]).max(),3)[:16])
print("units with rel diff > 1e-3:", np.nonzero(rd > 1e-3)[0], np.round(rd[rd > 1e-3], 2))
print("units with any bit diff:", np.nonzero((W != Wp).any(axis=1))[0])
g_a,g_b=res["general"]["g"].astype(np.float64),res["persistent"]["g"].astype(np.float64)
Wg=g_a[:n*h1].reshape(h1,n); Wgp=g_b[:n*h1].reshape(h1,n)
rg=np.linalg.norm(Wg-Wgp,axis=1)/np.linalg.norm(Wg,axis=1)
print("g: units with rel diff > 1e-3:", np.nonzero(rg > 1e-3)[0], np.round(rg[rg > 1e-3], 2))
