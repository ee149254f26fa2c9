import numpy as np, time
n=100000;K=3
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B).astype(np.float64); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
g=np.fromfile('ba100k.bin',np.int64,2); rp=np.fromfile('ba100k.bin',np.int64,n+3)[2:]; deg=np.diff(rp).astype(np.float64)
nk=H.sum(axis=2).astype(np.float64)            # [n][3]
sm=(H*sup[None,None,:]).sum(axis=2)            # sum of member degrees
mean=sm/nk
dl=np.diff(sup)                                # Delta_b
# A operands: G[k] = CDF*Delta  [n][B-1]
def GA(k):
    c=np.cumsum(H[:,k,:],axis=1,dtype=np.float64)[:,:-1]/nk[:,k:k+1]
    return (c*dl[None,:]).astype(np.float64)
# B operands: U = [n*s0, T(b)*Delta]
def UB(k):
    c=np.cumsum(H[:,k,:],axis=1,dtype=np.int64)
    T=(nk[:,k:k+1]-c)[:,:-1]
    return np.concatenate([nk[:,k:k+1]*sup[0],T*dl[None,:]],axis=1)
t=time.time()
GAs=[GA(k) for k in range(3)]; UBs=[UB(k) for k in range(2)]
print('ops',time.time()-t)
rng=np.random.default_rng(1)
rows=rng.choice(n,200,replace=False)
kk=20
resA=[];resB=[]
for i in rows:
    DA=np.abs(deg-deg[i])
    for k in range(3): DA+=np.abs(GAs[k]-GAs[k][i]).sum(axis=1)
    LBA=np.abs(deg-deg[i])+np.abs(mean-mean[i]).sum(axis=1)
    DB=np.abs(deg-deg[i])
    LBB=np.abs(deg-deg[i])
    for k in range(2):
        S=np.abs(UBs[k]-UBs[k][i]).sum(axis=1)
        mx=np.maximum(nk[:,k],nk[i,k])
        DB+=S/mx; LBB+=np.abs(sm[:,k]-sm[i,k])/mx
    assert (LBA<=DA+1e-9).all() and (LBB<=DB+1e-9).all()
    DA[i]=np.inf;DB[i]=np.inf
    tA=np.partition(DA,kk-1)[kk-1]; tB=np.partition(DB,kk-1)[kk-1]
    resA.append((deg[i],tA,(LBA<=tA).sum()-1,(DA<=tA).sum()))
    resB.append((deg[i],tB,(LBB<=tB).sum()-1,(DB<=tB).sum()))
resA=np.array(resA);resB=np.array(resB)
for nm,r in (('A',resA),('B',resB)):
    print(nm,'tau median',np.median(r[:,1]),'mean',r[:,1].mean(),'max',r[:,1].max())
    print(nm,'survivors per row (LB<=tau): median',np.median(r[:,2]),'mean',r[:,2].mean(),'p90',np.percentile(r[:,2],90),'max',r[:,2].max(), ' ties<=tau',r[:,3].mean())
    o=np.argsort(r[:,0]); 
    for q in (0,50,100,150,190,199): print('   deg',r[o[q],0],'tau',r[o[q],1],'surv',r[o[q],2])
np.save('rows.npy',rows)
