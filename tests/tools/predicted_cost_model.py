#!/usr/bin/env python3
"""Can a frame with a NEW camera be dealt by cost too?  The repeated view deals a region's pixels to waves by the evaluation
counts of the frame before (LABNOTES.md §3.9).  For a new camera the only exact knowledge about a frame comes from the frame
itself, so: render every 4th pixel of every 4th row first (1/16 of the pixels; in a 64x16 region exactly one wave), predict
the cost of the other 15/16 from the samples around them (nearest / bilinear / maximum of the four), deal those by the
prediction.  CPU model on the oracle's per-pixel step counts (test infrastructure), evaluations a frame executes as in
sorted_region_model.py.  Result on scene4 at 4K: no gain (-0.3 ... -0.7 % against rectangles; dealing by the true cost: +15 %):
what makes a pixel expensive here is finer than four pixels (silhouettes and penumbra edges).
Usage: python tests/tools/predicted_cost_model.py > profiles/r4_predicted_cost_model.json"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tests')
import oracle_lib as O
from loltracer_amd import scene as S
def wave_cost(march, hit, sh):
    return march.max(axis=1) + 4 * hit.any(axis=1) + sh.max(axis=1).sum(axis=1)
w,h=3840,2160
sc=S.Scene.parse_file(ROOT+'/tests/golden/scenes/scene4.lol')
rw,rh=64,16
res=dict(now=0,true=0,pred_nearest=0,pred_mean=0,pred_max=0, need=0, px=0, pass1=0)
for strip in range(0,h//64,6):
    y0=strip*64
    _,_,st=O.render_rows(sc,w,h,y0,y0+64,256,want_steps=True)
    st=st[y0:y0+64].astype(np.int64)
    hit=st[...,2]!=0; dark=st[...,3]; march=st[...,0]
    sh=[]
    for li in range(4):
        s=st[...,8+li].copy(); s[((dark>>li)&1)==1]=0; s[~hit]=0; sh.append(s)
    sh=np.stack(sh,-1)
    total=march+4*hit+sh.sum(-1)
    res['need']+=int(total.sum()); res['px']+=64*w
    # predictions from samples at (4i,4j)
    samp=total[0::4,0::4]            # [16, w/4]
    near=np.repeat(np.repeat(samp,4,0),4,1)
    # mean of 4 surrounding samples (clamped)
    sp=np.pad(samp,((0,1),(0,1)),mode='edge')
    s00=sp[:-1,:-1]; s01=sp[:-1,1:]; s10=sp[1:,:-1]; s11=sp[1:,1:]
    def up(x): return np.repeat(np.repeat(x,4,0),4,1)
    fy=(np.arange(64)%4/4.0)[:,None]; fx=(np.arange(w)%4/4.0)[None,:]
    mean=(up(s00)*(1-fy)*(1-fx)+up(s01)*(1-fy)*fx+up(s10)*fy*(1-fx)+up(s11)*fy*fx)
    mx=np.maximum(np.maximum(up(s00),up(s01)),np.maximum(up(s10),up(s11)))
    def cut(x):
        tail=x.shape[2:]
        return x.reshape(64//rh,rh,w//rw,rw,*tail).swapaxes(1,2).reshape(-1,rh*rw,*tail)
    m,hh,s4=cut(march),cut(hit),cut(sh)
    n_reg=m.shape[0]; nw=16
    def rect(x):
        tail=x.shape[2:]
        return x.reshape(n_reg,rh//4,4,rw//16,16,*tail).swapaxes(2,3).reshape(n_reg*nw,64,*tail)
    res['now']+=int(wave_cost(rect(m),rect(hh),rect(s4)).sum())
    def dealt(keyimg, two_pass):
        k=cut(keyimg).astype(np.float64)
        if two_pass:
            # sample mask
            yy,xx=np.mgrid[0:64,0:w]
            ms=cut(((yy%4==0)&(xx%4==0)))
            k=np.where(ms,-1.0,k)          # samples sort first -> wave 0 of region = the 64 samples
        idx=np.argsort(k,axis=1,kind='stable')
        g=lambda x: np.take_along_axis(x, idx if x.ndim==2 else idx[...,None],axis=1).reshape(n_reg*nw,64,*x.shape[2:])
        return int(wave_cost(g(m),g(hh),g(s4)).sum())
    res['true']+=dealt(total,False)
    res['pred_nearest']+=dealt(near,True)
    res['pred_mean']+=dealt(mean,True)
    res['pred_max']+=dealt(mx,True)
out = dict(scene="scene4.lol", size="3840x2160", region="64x16", sampled_pixels=res['px'],
           lane_efficiency_rectangles=res['need'] / (res['now'] * 64))
for k in ('true', 'pred_nearest', 'pred_mean', 'pred_max'):
    out[k] = dict(gain_over_rectangles=res['now'] / res[k] - 1, lane_efficiency=res['need'] / (res[k] * 64))
print(json.dumps(out, indent=1))
