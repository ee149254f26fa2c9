import numpy as np, subprocess, sys
n=100000;K=3
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B).astype(np.float64); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
rp=np.fromfile('ba100k.bin',np.int64,n+3)[2:]; deg=np.diff(rp).astype(np.float64)
nk=H.sum(axis=2).astype(np.float64); sm=(H*sup[None,None,:]).sum(axis=2)
tail2=H[:,1,:].astype(np.int64)[:,::-1].cumsum(axis=1)[:,::-1]
W=128
def tile_surv(perm_nodes,tau,T=128):
    nt=n//T
    pp=perm_nodes[:nt*T].reshape(nt,T)
    tmax=tau[pp].max(axis=1)
    lo=deg[pp].min(axis=1); hi=deg[pp].max(axis=1)
    LB=np.maximum(0,np.maximum(lo[:,None]-hi[None,:],lo[None,:]-hi[:,None]))
    for k in range(2):
        slo=sm[pp,k].min(axis=1); shi=sm[pp,k].max(axis=1); nmax=nk[pp,k].max(axis=1)
        gap=np.maximum(0,np.maximum(slo[:,None]-shi[None,:],slo[None,:]-shi[:,None]))
        LB+=gap/np.maximum(nmax[:,None],nmax[None,:])
    thr=np.maximum(tmax[:,None],tmax[None,:])
    iu=np.triu_indices(nt)
    return (LB<=thr)[iu].mean()
def gran_perm(key):
    g=key[:n//16*16].reshape(-1,16).max(axis=1)
    go=np.argsort(g,kind='stable')
    return (go[:,None]*16+np.arange(16)[None,:]).ravel()
def run(name,perm):
    perm=perm.astype(np.int32); perm.tofile('perm.bin')
    subprocess.check_call(['./band',str(n),'hist100k.bin','perm.bin',str(W),'20','1','tau.bin'])
    tau=np.fromfile('tau.bin',np.float64)
    print(f"{name}: tau0 median {np.median(tau):.2f}; tile pairs surviving: tau0 {tile_surv(perm,tau)*100:.1f}%, 0.8 tau0 {tile_surv(perm,0.8*tau)*100:.1f}%, 0.7 tau0 {tile_surv(perm,0.7*tau)*100:.1f}%",flush=True)
run('rank-sorted granules (ring-2 members above bin 56)',gran_perm(tail2[:,56].astype(np.float64)))
F=np.column_stack([deg,sm[:,0]/np.maximum(nk[:,0],1),sm[:,1]/np.maximum(nk[:,1],1)])
# granule-level kd order on the granule means of F
G=F[:n//16*16].reshape(-1,16,3).mean(axis=1)
def kdorder(idx,leaf):
    if len(idx)<=leaf: return [idx]
    ext=G[idx].max(axis=0)-G[idx].min(axis=0); d=int(np.argmax(ext))
    o=idx[np.argsort(G[idx,d],kind='stable')]; half=(len(o)//2+leaf-1)//leaf*leaf
    if half>=len(o): half=len(o)//2
    return kdorder(o[:half],leaf)+kdorder(o[half:],leaf)
sys.setrecursionlimit(10000)
go=np.concatenate(kdorder(np.arange(n//16),8))
run('granules in kd order of (deg, mean1, mean2)',(go[:,None]*16+np.arange(16)[None,:]).ravel())
# nodes fully sorted kd (not granule constrained), for reference
def kdn(idx,leaf):
    if len(idx)<=leaf: return [idx]
    ext=F[idx].max(axis=0)-F[idx].min(axis=0); d=int(np.argmax(ext))
    o=idx[np.argsort(F[idx,d],kind='stable')]; half=(len(o)//2+leaf-1)//leaf*leaf
    if half>=len(o): half=len(o)//2
    return kdn(o[:half],leaf)+kdn(o[half:],leaf)
run('nodes in kd order (no granule constraint)',np.concatenate(kdn(np.arange(n),128)))
