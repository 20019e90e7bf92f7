import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import common, oracle_lib as O, motion_planning_amd as mp
for w,n in ((common.config2(n=20000),20000),(common.config3(n=4000),4000),(common.config5(n=3000),3000)):
    ref=O.eval_batch(w.model,w.boxes,w.params,w.soa[:,:n],w.layout,threads=8)
    eng=mp.Engine(w.model,w.boxes,w.params); full=eng.eval_batch(w.soa[:,:n],w.layout); eng.close()
    for k in ("fc_t_min","fc_rp_min","margin"):
        a,b=full[k],ref[k]; ok=np.isfinite(a)&np.isfinite(b)
        print(w.name,k,"max abs",np.max(np.abs(a[ok]-b[ok])),"max rel",np.max(np.abs(a[ok]-b[ok])/np.maximum(np.abs(b[ok]),1e-9)))
