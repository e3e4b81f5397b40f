"""debug: how often neighbouring pixels hold bitwise-identical planes before / after each raster sweep (C3 pair)."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np
import crossscalepatchmatch_amd as cs
from crossscalepatchmatch_amd import synth
cfg,l,r,gl,gr = synth.make_config("C3")
ctx = cs.StereoContext(0); ctx.set_images(l,r); ctx.build_cost_grd(cfg["max_dis"],35,5,0.3)
kw=dict(seed=12345, schedule=0)
ctx.pm_init(**kw)
for it in range(3):
    P,_ = ctx.get_planes(0)
    eqx = np.all(P[:,1:]==P[:,:-1],axis=-1).mean(); eqy=np.all(P[1:]==P[:-1],axis=-1).mean()
    print("before spatial",it,"x-neighbour identical %.3f y %.3f"%(eqx,eqy))
    ctx.pm_spatial(it,**kw)
    P,_ = ctx.get_planes(0); PR,_=ctx.get_planes(1)
    print("  after spatial: x-identical %.3f"%np.all(P[:,1:]==P[:,:-1],axis=-1).mean())
    ctx.pm_view(it,**kw); ctx.pm_refine(it,**kw)
