"""Survival of (row tile of TR nodes) x (column group of TC nodes) blocks under the exact lower bound, for a node order: what a k-nearest
sweep that gives every ROW TILE its own list of surviving 16-node column granules could skip.  No symmetry in that form (each row tile
meets the columns its own rows need), so it must stay well below 50 % to beat the symmetric sweep.  usage: prune_rect_sim.py <variant 0=A 1=B>"""
import numpy as np, subprocess, sys
n=100000;K=3
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B).astype(np.float64); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
rp=np.fromfile('ba100k.bin',np.int64,n+3)[2:]; deg=np.diff(rp).astype(np.float64)
nk=H.sum(axis=2).astype(np.float64); sm=(H*sup[None,None,:]).sum(axis=2); mean=sm/nk
variant=int(sys.argv[1])
F=np.column_stack([deg,mean[:,0],mean[:,1],mean[:,2]]) if variant==0 else np.column_stack([deg,mean[:,0],mean[:,1]])
def kdorder(idx,leaf,Fs):
    if len(idx)<=leaf: return [idx]
    ext=Fs[idx].max(axis=0)-Fs[idx].min(axis=0)
    d=int(np.argmax(ext)); o=idx[np.argsort(Fs[idx,d],kind='stable')]
    half=(len(o)//2+leaf-1)//leaf*leaf
    if half>=len(o): half=len(o)//2
    return kdorder(o[:half],leaf,Fs)+kdorder(o[half:],leaf,Fs)
sys.setrecursionlimit(10000)
def stats(perm,tau,TR,TC):
    def groups(T):
        nt=(n+T-1)//T; pad=nt*T-n
        return np.concatenate([perm,np.full(pad,perm[-1],dtype=perm.dtype)]).reshape(nt,T)
    R=groups(TR); C=groups(TC)
    tmax=tau[R].max(axis=1)
    surv=0; tot=R.shape[0]*C.shape[0]
    if variant==0:
        rlo=F[R].min(axis=1); rhi=F[R].max(axis=1); clo=F[C].min(axis=1); chi=F[C].max(axis=1)
        for a in range(0,R.shape[0],64):
            LB=np.zeros((min(64,R.shape[0]-a),C.shape[0]),dtype=np.float32)
            for fi in range(F.shape[1]):
                LB+=np.maximum(0,np.maximum(rlo[a:a+64,None,fi]-chi[None,:,fi],clo[None,:,fi]-rhi[a:a+64,None,fi])).astype(np.float32)
            surv+=(LB<=tmax[a:a+64,None]).sum()
    else:
        rlo=deg[R].min(axis=1); rhi=deg[R].max(axis=1); clo=deg[C].min(axis=1); chi=deg[C].max(axis=1)
        for a in range(0,R.shape[0],64):
            LB=np.maximum(0,np.maximum(rlo[a:a+64,None]-chi[None,:],clo[None,:]-rhi[a:a+64,None])).astype(np.float32)
            for k in range(2):
                rs_lo=sm[R,k].min(axis=1); rs_hi=sm[R,k].max(axis=1); cs_lo=sm[C,k].min(axis=1); cs_hi=sm[C,k].max(axis=1)
                rn=nk[R,k].max(axis=1); cn=nk[C,k].max(axis=1)
                gap=np.maximum(0,np.maximum(rs_lo[a:a+64,None]-cs_hi[None,:],cs_lo[None,:]-rs_hi[a:a+64,None]))
                LB+=(gap/np.maximum(rn[a:a+64,None],cn[None,:])).astype(np.float32)
            surv+=(LB<=tmax[a:a+64,None]).sum()
    return surv/tot
Fl=F.copy(); Fl[:,0]=np.log(F[:,0])*8
for name,perm in (('kd (log deg) leaf 16',np.concatenate(kdorder(np.arange(n),16,Fl))),('kd (log deg) leaf 64',np.concatenate(kdorder(np.arange(n),64,Fl)))):
    perm=perm.astype(np.int32); perm.tofile('perm.bin')
    subprocess.check_call(['./band',str(n),'hist100k.bin','perm.bin','128','20',str(variant),'tau.bin'])
    tau=np.fromfile('tau.bin',np.float64)
    out=[f"{name}: tau0 median {np.median(tau):.2f}"]
    for TR,TC in ((128,128),(128,64),(128,16),(64,16),(32,16)):
        for sc in (1.0,0.8):
            out.append(f"{TR}x{TC} tau*{sc}: {stats(perm,sc*tau,TR,TC)*100:.1f}%")
    print('\n   '.join(out),flush=True)
