import numpy as np, sys
z=np.load(sys.argv[1])
names=['nt','nn','tn']
F=None
rows=[]
for k in sorted(x for x in z.files if x.startswith('s')):
    t=z[k].astype(np.int64); m=[int(v) for v in z['m'+k[1:]]]
    mode,M,N,K,G,epi,cnt,ev=m
    hw=t[:,5]; cu=((hw>>32)&0xf)*256+((hw>>8)&0xff)
    # clock: per CU slope
    rs=[]
    for c in np.unique(cu):
        w=t[cu==c]
        if len(w)<2: continue
        d_rt=w[:,7].max()-w[:,7].min(); i=np.argmax(w[:,7]); j=np.argmin(w[:,7])
        if d_rt>1000: rs.append((w[i,0]-w[j,0])/d_rt)
    f=np.median(rs)*100 if rs else 2090.0   # cycles per us
    rt0=t[:,7].min()
    start=(t[:,7]-rt0)/100.0  # us
    def T(i):
        return start+(t[:,i]-t[:,0])/f
    has_loop=t[:,2]>0
    endt=np.where(t[:,4]>0,T(4),np.where(t[:,3]>0,T(3),T(0)))
    span=endt.max()
    flops=2.0*M*N*K
    ideal_cyc=flops/65536.0
    ideal_us=ideal_cyc/f
    # per-CU: union time with >=1 WG in loop, and WG-in-loop count shares
    sh=np.zeros(5); 
    l0,l1=T(1),T(2)
    for c in np.unique(cu):
        idx=np.where((cu==c)&has_loop)[0]
        ev_=[]
        for i in idx:
            ev_.append((l0[i],1)); ev_.append((l1[i],-1))
        ev_.sort()
        cur=0; last=0.0
        for tt,d in ev_:
            sh[min(cur,4)]+=tt-last; last=tt; cur+=d
        sh[0]+=span-last
    sh/=sh.sum()
    pro=np.median((t[has_loop,1]-t[has_loop,0]))/f
    loop=np.median((t[has_loop,2]-t[has_loop,1]))/f
    d=t[:,4]>0
    epi_=np.median((t[d,4]-t[d,3]))/f if d.any() else 0
    fix=np.median((t[has_loop,3]-t[has_loop,2]))/f
    first_loop=l0[has_loop].min(); last_loop_end=l1[has_loop].max()
    print(f'{names[mode]} {M}x{N}x{K} g{G} e{epi} x{cnt} | WGs {len(t)} | clk {f:.0f} MHz | span {span:.1f} us (event {ev}) | ideal@clk {ideal_us:.1f} eff {ideal_us/span:.2f} | pro {pro:.1f} loop {loop:.1f} fix {fix:.1f} epi {epi_:.1f} | inloop0/1/2/3+ {sh[0]:.2f} {sh[1]:.2f} {sh[2]:.2f} {sh[3]+sh[4]:.2f} | first loop at {first_loop:.1f}, last loop end {last_loop_end:.1f}')
    rows.append((cnt,span,ideal_us,f))
tot=sum(c*s for c,s,i,f in rows); toti=sum(c*i for c,s,i,f in rows)
print('step total span', tot/1e3,'ms ideal@clk', toti/1e3, 'ms; mean clk', np.mean([f for *_,f in rows]))
