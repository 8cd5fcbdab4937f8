import numpy as np, sys
for fn in sys.argv[1:]:
    a=np.fromfile(fn,dtype=np.uint64).reshape(-1,8).astype(np.int64)
    t0=a[:,0].min()
    st=(a[:,:7]-t0)*0.01  # us (100 MHz)
    qn=a[:,7]
    print(fn, "tiles",len(a),"qn mean",qn.mean(),"nonzero",(qn>0).sum())
    names=["start","tile+compact","phase1","phase2","barrier","phase3","end"]
    for i,n in enumerate(names):
        col=st[:,i]; v=col[a[:,i]>0]
        if len(v): print(f"  {n:14s} abs: min {v.min():7.2f} med {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f}")
    m=(qn>0)&(a[:,5]>0)
    d=np.diff(st[m][:,:7],axis=1)
    print("  deltas (WGs with work) median:", np.round(np.median(d,axis=0),2), " lifetime med", np.round(np.median(st[m][:,6]-st[m][:,0]),2))
    e=~(qn>0)
    if e.any(): print("  empty WG lifetime med", np.round(np.median(st[e][:,6]-st[e][:,0]),2))
