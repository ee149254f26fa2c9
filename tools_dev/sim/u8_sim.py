import numpy as np
n=100000;K=3      # reads hist100k.bin (see README.md)
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
print("bins",B,"support head",sup[:12],"tail",sup[-5:])
T=128; nt=(n+T-1)//T
iu=np.triu_indices(nt,1)
bin_words=(B+1)//2
# tail[ring][i,b] = members of ring with bin >= b
tail=[H[:,r,:].astype(np.int64)[:,::-1].cumsum(axis=1)[:,::-1] for r in (1,2)] if False else [H[:,r,:].astype(np.int64)[:,::-1].cumsum(axis=1)[:,::-1] for r in (0,1)]
print("ring sizes: ring a mean/max", tail[0][:,0].mean(), tail[0][:,0].max(), " ring b", tail[1][:,0].mean(), tail[1][:,0].max())
gsz=16
def tile_max(v, perm_g):
    m=np.zeros(nt*T,np.int64); m[:n]=v
    m=m.reshape(-1,gsz)[perm_g].ravel()
    return m.reshape(nt,T).max(axis=1)
def first_below(ring, lim):
    # per node: smallest bin b with tail[b] < lim
    t=tail[ring]
    return (t>=lim).sum(axis=1)       # tail is non-increasing in b
cands=[0,2,4,6,8,10,12,16,20,24,30,36,40,44,48,52,56,60,64,70,80,90,100,120]
def cost(ring, perm_g, u8):
    best=None
    b8n=first_below(ring,256)
    b8t=tile_max(b8n,perm_g)
    for fb in cands:
        above=tail[ring][:,fb]
        tm=tile_max(above,perm_g); a=tm[iu[0]]; b=tm[iu[1]]
        rmin=np.minimum(a,b)
        if not u8:
            c=np.minimum(bin_words, (fb+1)//2 + (rmin+1)//2 + 1)
        else:
            b8p=np.minimum(np.maximum(b8t[iu[0]],b8t[iu[1]]), fb)
            binw=(b8p+1)//2 + (fb-b8p+3)//4
            # ranks: values above sup[fb] ; >= 256 over it
            lim=sup[fb]+256 if fb<B else 1<<30
            hb=np.searchsorted(sup, lim)       # first bin with support >= lim
            r8n=tail[ring][:,hb] if hb<B else np.zeros(n,np.int64)
            r8t=tile_max(r8n,perm_g); r8p=np.maximum(r8t[iu[0]],r8t[iu[1]])
            r16=np.minimum(r8p,rmin)
            rankw=(r16+1)//2 + (rmin-r16+3)//4
            c=binw+rankw+1
        cm=c.mean()
        if best is None or cm<best[0]: best=(cm,fb)
    return best
for key_fb in (40,48,56,64,70,80):
    above=tail[1][:,key_fb]; m=np.zeros(nt*T,np.int64); m[:n]=above
    perm_g=np.argsort(m.reshape(-1,gsz).max(axis=1),kind='stable')
    for u8 in (False,True):
        c1=cost(0,perm_g,u8); c2=cost(1,perm_g,u8)
        print(f"key bin {key_fb}: u8={u8}: ring1 {c1[0]:.1f} (fb {c1[1]}) + ring2 {c2[0]:.1f} (fb {c2[1]}) = {c1[0]+c2[0]:.1f} SAD instr per pair")
