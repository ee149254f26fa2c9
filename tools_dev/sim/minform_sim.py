import numpy as np
n=100000;K=3
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
T=128; nt=(n+T-1)//T
iu=np.triu_indices(nt,1)
bin_words=(B+1)//2
tail=[H[:,r,:].astype(np.int64)[:,::-1].cumsum(axis=1)[:,::-1] for r in (0,1)]
cands=[0,2,4,6,8,10,12,16,20,24,30,36,40,44,48,52,56,60,64,70,80,90,100,120]
def cost(ring,perm_g,form,gsz):
    best=None
    for fb in cands:
        above=tail[ring][:,fb]; low=(fb+1)//2
        m=np.zeros(nt*T,np.int64); m[:n]=above
        if perm_g is not None: m=m.reshape(-1,gsz)[perm_g].ravel()
        tm=m.reshape(nt,T).max(axis=1); a=tm[iu[0]]; b=tm[iu[1]]
        if form=='max': c=np.minimum(bin_words, low+(np.maximum(a,b)+1)//2).mean()
        else: c=np.minimum(bin_words, low+(np.minimum(a,b)+1)//2+1).mean()
        if best is None or c<best[0]: best=(c,fb)
    return best
for gsz in (16,64,128):
  for form in ('max','min'):
    res=[]
    for fb2 in (40,48,56,64,70,80):
        above=tail[1][:,fb2]; m=np.zeros(nt*T,np.int64); m[:n]=above
        perm_g=np.argsort(m.reshape(-1,gsz).max(axis=1),kind='stable')
        c2=cost(1,perm_g,form,gsz); c1=cost(0,perm_g,form,gsz)
        res.append((c1[0]+c2[0],fb2,c1,c2))
    r=min(res)
    print(f"granule {gsz:3d} sorted by ring-2 key (bin {r[1]}), {form}-form: ring1 {r[2][0]:.1f} (fb {r[2][1]}) + ring2 {r[3][0]:.1f} (fb {r[3][1]}) = {r[0]:.1f} words")
