import numpy as np, time, subprocess, sys
n=100000;K=3
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B).astype(np.float64); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
rp=np.fromfile('ba100k.bin',np.int64,n+3)[2:]; deg=np.diff(rp).astype(np.float64)
nk=H.sum(axis=2).astype(np.float64); sm=(H*sup[None,None,:]).sum(axis=2); mean=sm/nk
variant=int(sys.argv[1]); W=int(sys.argv[2]) if len(sys.argv)>2 else 128
if variant==0: F=np.column_stack([deg,mean[:,0],mean[:,1],mean[:,2]])
else: F=np.column_stack([deg,mean[:,0],mean[:,1]])
def tile_stats(perm,tau,T):
    nt=(n+T-1)//T; pad=nt*T-n
    pp=np.concatenate([perm,np.full(pad,perm[-1],dtype=perm.dtype)]).reshape(nt,T)
    tmax=tau[pp].max(axis=1)
    if variant==0:
        lo=F[pp].min(axis=1); hi=F[pp].max(axis=1)
        LB=np.zeros((nt,nt),dtype=np.float32)
        for fidx in range(F.shape[1]):
            LB+=np.maximum(0,np.maximum(lo[:,None,fidx]-hi[None,:,fidx],lo[None,:,fidx]-hi[:,None,fidx])).astype(np.float32)
    else:
        lo=deg[pp].min(axis=1); hi=deg[pp].max(axis=1)
        LB=np.maximum(0,np.maximum(lo[:,None]-hi[None,:],lo[None,:]-hi[:,None])).astype(np.float32)
        for k in range(2):
            slo=sm[pp,k].min(axis=1); shi=sm[pp,k].max(axis=1); nmax=nk[pp,k].max(axis=1)
            gap=np.maximum(0,np.maximum(slo[:,None]-shi[None,:],slo[None,:]-shi[:,None]))
            LB+=(gap/np.maximum(nmax[:,None],nmax[None,:])).astype(np.float32)
    thr=np.maximum(tmax[:,None],tmax[None,:])
    surv=(LB<=thr); iu=np.triu_indices(nt)
    return surv[iu].mean()
def run(name,perm):
    perm=perm.astype(np.int32); perm.tofile('perm.bin')
    subprocess.check_call(['./band',str(n),'hist100k.bin','perm.bin',str(W),'20',str(variant),'tau.bin'])
    tau=np.fromfile('tau.bin',np.float64)
    out=[f"{name}: tau0 median {np.median(tau):.2f}"]
    for T in (128,64): out.append(f"T={T}: {tile_stats(perm,tau,T)*100:.1f}%")
    # with the true tau approximated by 0.8*tau0 (tightening after the first pruned phase)
    for T in (128,64): out.append(f"T={T},0.8tau: {tile_stats(perm,0.8*tau,T)*100:.1f}%")
    print('  '.join(out),flush=True)
def kdorder(idx,leaf,Fs):
    if len(idx)<=leaf: return [idx]
    ext=Fs[idx].max(axis=0)-Fs[idx].min(axis=0)
    d=int(np.argmax(ext))
    o=idx[np.argsort(Fs[idx,d],kind='stable')]
    half=(len(o)//2+leaf-1)//leaf*leaf
    if half>=len(o): half=len(o)//2
    return kdorder(o[:half],leaf,Fs)+kdorder(o[half:],leaf,Fs)
sys.setrecursionlimit(10000)
run('sort deg, mu1',np.lexsort((F[:,1],F[:,0])))
run('kd (raw features) leaf 64',np.concatenate(kdorder(np.arange(n),64,F)))
Fl=F.copy(); Fl[:,0]=np.log(F[:,0])*8; 
run('kd (log deg) leaf 64',np.concatenate(kdorder(np.arange(n),64,Fl)))
