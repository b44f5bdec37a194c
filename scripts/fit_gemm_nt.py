import json, math
shapes=[]
for l in open('profiles/r02/gemm_sweep_r02b.txt'):
    if not l.startswith('JSON '): continue
    d=json.loads(l[5:])
    if d['mode']=='tn': continue
    t={}
    for us,c in d['results']:
        if 'tile' in c and c['tile'] in (0,1,5): t[(c['tile'],c['bk'],c.get('splits',1))]=us
    shapes.append((d,t))
DIM={0:(128,128),1:(128,96),5:(64,128)}
def quant(b):
    pc=b/256; r=math.ceil(b/256); return r/pc
def pad(n,b): return (n+b-1)//b*b
def choose(d,h,o_lo,o_sl,bkK,bk1K,split_c):
    M,N,K,G=d['M'],d['N'],d['K'],d['G']
    best=None
    for i,(tile,(bm,bn)) in enumerate(DIM.items()):
        tiles=((M+bm-1)//bm+(G//2 if G>1 else 0))*((N+bn-1)//bn)
        waste=pad(N,bn)/N
        pc=tiles/256
        occ=o_lo if pc<1 else (1+o_sl*(2-pc) if pc<2 else 1)
        cost=quant(tiles)*h[i]*waste*occ
        if best is None or cost<best[0]-1e-9: best=(cost,tile,tiles)
    tile=best[1]
    bk=32 if (K>=bkK or (tile==1 and K>=bk1K)) else 16
    s=1
    kt=K//bk
    if best[2]<256 and kt>=24:
        bc=None
        for cs in range(1,5):
            if kt/cs<6: break
            c=quant(best[2]*cs)*(1+split_c*(cs-1))
            if bc is None or c<bc-1e-9: bc=c; s=cs
    return tile,bk,s
def regret(p,verbose=False):
    tot=base=0
    for d,t in shapes:
        tile,bk,s=choose(d,*p)
        key=(tile,bk,s)
        if key not in t:
            # nearest s measured
            cands=[k for k in t if k[0]==tile and k[1]==bk]
            key=min(cands,key=lambda k:abs(k[2]-s))
        tb=min(t.values()); tc=t[key]
        tot+=d['count']*tc; base+=d['count']*tb
        if verbose: print(f"{d['mode']} {d['M']}x{d['N']}x{d['K']} g{d['G']} e{d['epi']}: chose {key} {tc:.1f} best {tb:.1f} ({tc/tb:.3f}) {min(t,key=t.get)}")
    return tot,base
cur=((1.0,1.02,1.03),1.2,0.2,1024,768,0.03)
print('current',regret(cur,True))
best=None
for h1 in (0.98,1.0,1.02,1.04):
  for h5 in (0.98,1.0,1.02,1.03,1.05):
    for o_lo in (1.0,1.1,1.2,1.3):
      for o_sl in (0.0,0.05,0.1,0.2):
        for bkK in (768,1024,1536,4096):
          for bk1K in (384,768,4096):
            for sc in (0.0,0.03,0.06):
              p=((1.0,h1,h5),o_lo,o_sl,bkK,bk1K,sc)
              r=regret(p)[0]
              if best is None or r<best[0]: best=(r,p)
print(best)
print(regret(best[1],True))

def choose2(d,h,o_lo,o_sl,bkK,sp):
    M,N,K,G=d['M'],d['N'],d['K'],d['G']
    best=None
    for i,(tile,(bm,bn)) in enumerate(DIM.items()):
        tiles=((M+bm-1)//bm+(G//2 if G>1 else 0))*((N+bn-1)//bn)
        waste=pad(N,bn)/N
        bk=32 if K>=bkK else 16
        kt=K//bk
        for s in range(1,5):
            if s>1 and (tiles>=256 or kt<24 or kt/s<6): break
            pc=tiles*s/256
            occ=o_lo if pc<1 else (1+o_sl*(2-pc) if pc<2 else 1)
            cost=quant(tiles*s)*h[i]*waste*occ*(1+sp*(s-1))
            if best is None or cost<best[0]-1e-9: best=(cost,tile,bk,s)
    return best[1],best[2],best[3]
def regret2(p,verbose=False):
    tot=base=0
    for d,t in shapes:
        tile,bk,s=choose2(d,*p)
        key=(tile,bk,s)
        if key not in t:
            cands=[k for k in t if k[0]==tile and k[1]==bk]
            key=min(cands,key=lambda k:abs(k[2]-s))
        tb=min(t.values()); tc=t[key]
        tot+=d['count']*tc; base+=d['count']*tb
        if verbose and tc/tb>1.01: print(f"{d['mode']} {d['M']}x{d['N']}x{d['K']} g{d['G']} e{d['epi']}: chose {key} {tc:.1f} best {tb:.1f} ({tc/tb:.3f}) {min(t,key=t.get)}")
    return tot,base
best=None
for h1 in (1.0,1.02,1.04,1.06):
  for h5 in (1.0,1.02,1.03,1.05):
    for o_lo in (1.0,1.1,1.2,1.3):
      for o_sl in (0.0,0.05,0.1,0.2):
        for bkK in (768,1024,1536):
            for sp in (0.0,0.03,0.06,0.1):
              p=((1.0,h1,h5),o_lo,o_sl,bkK,sp)
              r=regret2(p)[0]
              if best is None or r<best[0]: best=(r,p)
print('joint',best)
print(regret2(best[1],True))
