import numpy as np, itertools
n=100000;K=3      # reads hist100k.bin (see README.md)
f=open('hist100k.bin','rb'); B=np.fromfile(f,np.int32,1)[0]; sup=np.fromfile(f,np.int32,B); H=np.fromfile(f,np.uint16).reshape(n,K,B); f.close()
T=128; nt=(n+T-1)//T
iu=np.triu_indices(nt,0)
bin_words=(B+1)//2
tail=[H[:,r,:].astype(np.int64)[:,::-1].cumsum(axis=1)[:,::-1] for r in (0,1)]
gsz=16
def tile_max(v, perm_g):
    m=np.zeros(nt*T,np.int64); m[:n]=v
    m=m.reshape(-1,gsz)[perm_g].ravel()
    return m.reshape(nt,T).max(axis=1)
key_fb=56
above=tail[1][:,key_fb]; m=np.zeros(nt*T,np.int64); m[:n]=above
perm_g=np.argsort(m.reshape(-1,gsz).max(axis=1),kind='stable')
ring=1
for fb in (48,56):
    tm=tile_max(tail[ring][:,fb],perm_g); a=tm[iu[0]]; b=tm[iu[1]]; rmin=np.minimum(a,b)
    c16=np.minimum(bin_words,(fb+1)//2+(rmin+1)//2+1)
    lim=sup[fb]+256; hb=np.searchsorted(sup,lim)
    big=tail[ring][:,hb]
    paths={}
    for b8 in (0,4,8,12,16,20,24,28,32,40,48,56):
      if b8>fb: continue
      okb=tail[ring][:,min(b8,B-1)]<256
      for r16 in (0,2,4,8,12,16,24,32,48):
        okn=okb&(big<=r16)
        okt=tile_max((~okn).astype(np.int64),perm_g)==0
        ok=okt[iu[0]]&okt[iu[1]]
        fixed=b8//2 + (fb-b8+3)//4 + r16//2
        c8=fixed+np.maximum(rmin-r16,0)//4+1+1
        paths[(b8,r16)]=np.where(ok,c8,1<<20)
    single=min(((np.minimum(c16,v).mean(),k) for k,v in paths.items()))
    print(f"fb {fb}: u16 {c16.mean():.1f}; best single u8 path {single}")
    best2=None
    keys=list(paths)
    for k1,k2 in itertools.combinations(keys,2):
        c=np.minimum(np.minimum(c16,paths[k1]),paths[k2]).mean()
        if best2 is None or c<best2[0]: best2=(c,k1,k2)
    print("   best two u8 paths",best2)
