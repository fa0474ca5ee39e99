import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, util, gpuutil as G
from oracle import oracle as O
from swiftvideo_amd import compute as sv, chipvideo as cv
cv.set_switch("CHV_BGRA_PATH","stream")
ctx = sv.makeComputeContext(forType="GPU")
cw,ch=320,180
specs=[("img_nv12_bgra",304,176,dict(rect=(40,20,200,120),opacity=o)) for o in (0.9,)]
exp=util.alloc_image("bgra",cw,ch); O.run_kernel("img_clear_bgra",exp)
layers=[]
for i,(k,sw,sh,kw) in enumerate(specs):
    u=util.make_uniforms((cw,ch),in_size=(sw,sh),**kw)
    src=util.alloc_image("nv12",sw,sh,seed=180+i)
    O.run_kernel(k,exp,src,u,threads=4)
    layers.append((sv.defaultComputeKernelFromString(k),G.to_gpu(ctx,"nv12",sw,sh,src),u,0))
gd=G.to_gpu(ctx,"bgra",cw,ch,util.alloc_image("bgra",cw,ch,seed=5))
h,name,keep=G.make_batch(ctx,[(gd,True,layers)]); print(name)
G.run_batch(ctx,h)
got=G.from_gpu(ctx,gd,"bgra",cw,ch)[0]; e=exp[0]
d=np.any(got!=e,axis=2)
rows=np.where(d.any(axis=1))[0]; cols=np.where(d.any(axis=0))[0]
print("rows",rows[:40], len(rows)); print("cols",cols[:40], len(cols))
print("per-row counts", d.sum(axis=1)[18:60])
