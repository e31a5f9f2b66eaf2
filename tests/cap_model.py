#!/usr/bin/env python3
"""Why trace_cap = 5 (search_kernel.cuh, discrete mode): a cost model of "several traces per simulation step" driven by the C oracle's
own trace record of BASELINE config B (which trace of which tree creates a non-terminal node, i.e. needs the network).  Per step a
workgroup pays one network phase + barriers (c_step) and its slowest wave's tree work: a first A -> B round (c_first: it may finish
a leaf and expand) plus one cheaper round (c_more) for every further trace its busiest tree runs; a tree stops at a trace that needs
an evaluation or at the cap.  With the stamped build's costs (k cycles: 4.4 / 6.1 / 4.0) the model gives, per trace, 10.5k at cap 1,
8.2k at 2, 7.7k at 3, 7.5k at 4, 7.4k at 5, 7.5k at 6, 8.5k at 8 -- the shape that was measured on the GPU (0.484 / 0.41 / 0.388 /
0.375 / 0.371 / 0.371 / 0.43 ms).  "Catch-up" caps for trees that lag behind a target pace come out worse (8.6k+): they lengthen the
steps of everybody.  CPU only (uses the oracle: test infrastructure):  python tests/cap_model.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
from alphazero_gym_amd import _capi  # noqa: E402
B=4096
e=O.OracleEngine(env_id=0,mode=0,n_trees=B,n_sims=100,c_uct=1.5,gamma=1.0,num_actions=2,seed=34)
e.set_weights(_capi.make_desc(4,[128,128],2,'relu'),O.make_weights(34,4,[128,128],2))
e.trace_enable(); e.search(e.synthetic_roots()); leaf,m=e.trace_get(); d=e.dump_tree()
l=leaf&0xffff
fl=d['node_flags']; term=np.take_along_axis(fl,l.astype(np.int64),1)&2
new=np.zeros_like(l,bool)
for t in range(B):
    _,idx=np.unique(l[t],return_index=True); new[t,idx]=True
need=new&(term==0)     # trace i creates a non-terminal node -> needs eval before trace i+1's ... (its backup)
NS=100
def simulate(policy, c_first=6.1, c_more=4.0, c_step=4.4):
    total=0.0; steps_all=[]
    for w in range(B//16):
        nd=need[w*16:(w+1)*16]
        pos=np.zeros(16,int)   # next trace index
        step=0; cost=0.0
        while (pos<NS).any():
            step+=1
            iters=np.zeros(16,int)
            for t in range(16):
                k=0
                cap=policy(pos[t], step)
                while pos[t]<NS:
                    i=pos[t]; pos[t]+=1; k+=1
                    if nd[t,i] or k>=cap: break
                iters[t]=k
            # wave-level: 4 waves of 4 trees: step time = max over waves of sum over iterations cost; iteration j costs c_first if j==0 else c_more, if any tree of the wave active
            wt=[]
            for wv in range(4):
                mx=iters[wv*4:(wv+1)*4].max()
                wt.append((c_first if mx>0 else 0)+max(mx-1,0)*c_more)
            cost+=c_step+max(wt)
        total+=cost; steps_all.append(step)
    return total/(B//16)/100.0, np.mean(steps_all)
for cap in (1,2,3,4,5,6,8,12):
    c,s=simulate(lambda p,st,cap=cap: cap)
    print('cap',cap,'cost/trace %.2fk'%c,'steps %.1f'%s)
# adaptive: target rate
for T in (25,30,35):
    for lo in (1,2):
        c,s=simulate(lambda p,st,T=T,lo=lo: max(lo, int(np.ceil((st*NS/T)-p))) )
        print('adaptive T',T,'lo',lo,'cost/trace %.2fk'%c,'steps %.1f'%s)
