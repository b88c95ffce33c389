import sys; sys.path.insert(0,'.')
import numpy as np, torch, ibs_amd, bench
from oracle import c_oracle as co
dev=torch.device('cuda',0); ctx=ibs_amd.Context(0); EPS=2.220446049250313e-16
for N,n in ((2561,96),(4097,64),(16385,8)):
    h,g,c,f=bench.c5_family(dev,'rough',n,N,seed=31+N); nA=bench.norm_a(h,g,c,f).cpu().numpy()
    r=ctx.solve_gcf(h,g,c,f,want_info=True); gn,cn,fn=g.cpu().numpy(),c.cpu().numpy(),f.cpu().numpy()
    lam=r['lam'].cpu().numpy(); lam_c=co.lam_batch(h,gn,cn,fn)
    torch.cuda.synchronize(); import time; t0=time.perf_counter(); ctx.solve_gcf(h,g,c,f); torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(N,'lam err/N eps',(np.abs(lam-lam_c)/nA).max()/(N*EPS), 'flags',int(((r['info']>>16)!=0).sum()), 'passes mean %.2f max %d'%(float((r['info']&0xffff).double().mean()), int((r['info']&0xffff).max())), '%.3f ms for %d systems'%(dt*1e3,n))
    gam_c,lam_s,_=co.solve_gcf_batch(h,gn,cn,fn)
    d=np.abs(r['gam'].cpu().numpy()-gam_c)/np.maximum(1.0,np.abs(gam_c))
    for gapf in (1e-4,1e-5,1e-6,1e-7):
        ok=ctx.sturm_count(h,gn,cn,fn,lam-gapf*nA)==1
        print('  gap>',gapf,'systems',int(ok.sum()),'max rel dgam',d[ok].max() if ok.any() else None)
    print('  all',d.max(),'|gam| range',np.abs(gam_c).min(),np.abs(gam_c).max(),'nA',nA.mean())
