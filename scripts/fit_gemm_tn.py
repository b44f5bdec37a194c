import json, itertools, math
shapes=[]
for l in open('profiles/r02/gemm_sweep_r02b.txt'):
    if not l.startswith('JSON '): continue
    d=json.loads(l[5:])
    if d['mode']!='tn': continue
    t={}
    for us,c in d['results']:
        if c.get('bk')!=16: continue
        if c.get('separate') and c['splits']!=256: t[(c['tile'],c['splits'])]=us
        if c.get('splits')==1 and not c.get('separate'): t[(c['tile'],1)]=us
    shapes.append((d,t))
DIM={0:(128,128),1:(128,96),2:(96,128)}
def quant(b):
    pc=b/256; r=math.ceil(b/256); return r/pc
def pad(n,b): return (n+b-1)//b*b
def choose(d,t,h,penc,oa,og,opc,gpc):
    M,N,K,G=d['M'],d['N'],d['K'],d['G']
    rows=K//G; pen=penc*G/K
    best=None
    for tile,(bm,bn) in DIM.items():
        tiles=((M+bm-1)//bm)*((N+bn-1)//bn)*G
        waste=pad(M,bm)*pad(N,bn)/(M*N)
        for s in sorted(set(s for (tt,s) in t if tt==tile)):
            if s>1 and rows/s<64: break
            if s>192: break
            blocks=tiles*s; pc=blocks/256
            occ=1+oa*max(0,opc-pc)
            if G>1 and pc<gpc: occ+=og*(gpc-pc)
            cost=quant(blocks)*occ*h[tile]*waste+pen*(s if s>1 else 0)
            if best is None or cost<best[0]-1e-9: best=(cost,tile,s)
    return best[1],best[2]
def regret(params,verbose=False):
    tot=0;base=0
    for d,t in shapes:
        tile,s=choose(d,t,*params)
        tb=min(t.values()); tc=t[(tile,s)]
        tot+=d['count']*tc; base+=d['count']*tb
        if verbose: print(f"tn {d['M']}x{d['N']}x{d['K']} g{d['G']}: chose tile {tile} s{s} {tc:.1f} best {tb:.1f} ({tc/tb:.3f})")
    return tot,base
cur=((1.0,1.13,1.13),100,0.15,0.05,3.0,4.5)
print('current',regret(cur,True))
best=None
for h0 in (1.0,1.03,1.05,1.08):
  for h1 in (0.97,1.0,1.03,1.06,1.1):
    for penc in (40,60,80,100,130,160):
      for oa in (0.0,0.05,0.1,0.15,0.25):
        for og in (0.0,0.03,0.05,0.1):
          for opc in (2.0,3.0):
            p=((h0,h1,h1),penc,oa,og,opc,4.5)
            r=regret(p)[0]
            if best is None or r<best[0]: best=(r,p)
print(best)
print(regret(best[1],True))
