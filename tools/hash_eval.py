import numpy as np, itertools, sys
M32=np.uint64(0xFFFFFFFF)
def u32(x): return (x & M32).astype(np.uint64)
def mad24(a,b,c): return u32((a & np.uint64(0xFFFFFF))*(np.uint64(b) & np.uint64(0xFFFFFF)) + c)
def old(x,seed):
    h=u32(x*np.uint64(0x9E3779B1)+np.uint64(seed)); h^=h>>np.uint64(16); h=u32(h*np.uint64(0x21f0aaad)); h^=h>>np.uint64(15); h=u32(h*np.uint64(0x735a2d97)); h^=h>>np.uint64(15); return h
def mk(K,S,final=16, pre='add'):
    def f(x,seed):
        h=u32(x+np.uint64(seed)) if pre=='add' else (x ^ np.uint64(seed))
        for k,s in zip(K,S):
            h=mad24(h,k,h>>np.uint64(s))
        if final: h^=h>>np.uint64(final)
        return h
    return f
def avalanche(f, n=200000, seed=12345, seq=True):
    rng=np.random.default_rng(1)
    x=(np.arange(n,dtype=np.uint64)+np.uint64(rng.integers(0,1<<26))) if seq else rng.integers(0,1<<32,n,dtype=np.uint64)
    h0=f(x,seed)
    worst=0; mat=np.zeros((32,32))
    for i in range(28):     # input bits that occur (index < 2^28)
        h1=f(x ^ np.uint64(1<<i),seed)
        d=h0^h1
        for o in range(32):
            mat[i,o]=((d>>np.uint64(o))&np.uint64(1)).mean()
    dev=np.abs(mat[:28]-0.5)
    return dev.max(), dev.mean()
def seed_avalanche(f,n=100000):
    x=np.arange(n,dtype=np.uint64)
    rng=np.random.default_rng(2)
    s=int(rng.integers(0,1<<32))
    h0=f(x,s); worst=0
    for i in range(32):
        h1=f(x,s^(1<<i)); d=h0^h1
        for o in range(32):
            worst=max(worst,abs(((d>>np.uint64(o))&np.uint64(1)).mean()-0.5))
    return worst
def keep_stats(f,seed=777,n=1<<22,thr=6553):
    x=np.arange(n,dtype=np.uint64)
    h=f(x,seed)
    lo=(h&np.uint64(0xFFFF))>=thr; hi=(h>>np.uint64(16))>=thr
    k=np.empty(2*n,bool); k[0::2]=lo; k[1::2]=hi
    rate=k.mean()
    # serial correlations at several lags
    kk=k.astype(np.float64)-rate
    cors={}
    for lag in (1,2,3,4,16,64,164,165,328,1024,26896):
        cors[lag]=float((kk[:-lag]*kk[lag:]).mean()/(rate*(1-rate)))
    # chi-square of 16-bit halves into 256 bins (top byte)
    tb=np.bincount(((h>>np.uint64(24))&np.uint64(0xFF)).astype(np.int64),minlength=256); e=n/256
    chi_hi=((tb-e)**2/e).sum()
    tb=np.bincount(((h>>np.uint64(8))&np.uint64(0xFF)).astype(np.int64),minlength=256)
    chi_lo=((tb-e)**2/e).sum()
    return rate, max(abs(v) for v in cors.values()), chi_hi, chi_lo
def cross_seed(f,n=1<<20,thr=6553):
    x=np.arange(n,dtype=np.uint64)
    a=(f(x,1000)&np.uint64(0xFFFF))>=thr; w=0
    for s in (1001,1002,1000+(1<<8),1000+(1<<16),1000+(1<<24),0x9E3779B9):
        b=(f(x,s)&np.uint64(0xFFFF))>=thr
        ra,rb=a.mean(),b.mean()
        c=((a-ra)*(b-rb)).mean()/np.sqrt(ra*(1-ra)*rb*(1-rb)); w=max(w,abs(c))
    return w
cands={'old':old}
import random
random.seed(5)
Ks=[0x9E3779,0x85EBCA,0xC2B2AE,0x27D4EB,0x165667,0xB5297A,0x68E31D,0x1B873D]
for K,S in [((0x9E3779,0x85EBCB,0xC2B2AF),(15,13,16)), ((0x9E3779,0x85EBCB),(15,13)), ((0xB5297B,0x68E31D,0x1B873D),(16,12,15)), ((0x9E3779,0x85EBCB,0xC2B2AF),(12,12,12)),((0x9E3779,0x85EBCB,0xC2B2AF,0x27D4EB),(15,13,12,16))]:
    cands['mad24 K=%s S=%s'%([hex(k) for k in K],S)]=mk(K,S)
for name,f in cands.items():
    a=avalanche(f); sa=seed_avalanche(f); ks=keep_stats(f); cs=cross_seed(f)
    print('%-60s aval max %.3f mean %.4f | seed aval %.3f | keep %.5f maxcorr %.4f chi %.0f %.0f | xseed %.4f'%(name,a[0],a[1],sa,ks[0],ks[1],ks[2],ks[3],cs))
print('---- xorshift + mad24(h,K,h) rounds')
def mk2(K,S,final=16):
    def f(x,seed):
        h=u32(x+np.uint64(seed))
        for k,s in zip(K,S):
            if s: h^=h>>np.uint64(s)
            h=mad24(h,k,h)
        if final: h^=h>>np.uint64(final)
        return h
    return f
c2={}
for K,S,fin in [((0x9E3779,0x85EBCB,0xC2B2AF),(16,16,16),16), ((0x9E3779,0x85EBCB),(16,16),16), ((0x9E3779,0x85EBCB,0xC2B2AF,0x27D4EB),(16,16,16,16),16),
                ((0x9E3779,0x85EBCB,0xC2B2AF),(0,16,16),16), ((0x9E3779,0x85EBCB,0xC2B2AF),(0,15,13),16), ((0x9E3779,0x85EBCB,0xC2B2AF),(0,16,16),15),
                ((0x3779B1,0xf0aaad,0x5a2d97),(0,16,16),16), ((0x9E3779,0x85EBCB,0xC2B2AF,0x27D4EB),(0,16,16,16),16)]:
    c2['xs+mad24 K=%s S=%s fin=%d'%([hex(k) for k in K],S,fin)]=mk2(K,S,fin)
for name,f in c2.items():
    a=avalanche(f); sa=seed_avalanche(f); ks=keep_stats(f); cs=cross_seed(f)
    print('%-70s aval max %.3f mean %.4f | seed aval %.3f | keep %.5f maxcorr %.4f chi %.0f %.0f | xseed %.4f'%(name,a[0],a[1],sa,ks[0],ks[1],ks[2],ks[3],cs))
