import sys, os
ROOT="/root/repo"; sys.path.insert(0, ROOT+"/peps-torch_amd"); sys.path.insert(0, ROOT)
import numpy as np, torch, config as cfg
cfg.global_args.device="cuda:0"
import _native
from ipeps.ipeps_c4v import read_ipeps_c4v
from ipeps.ipeps import IPEPS
from ctm.generic.env import ENV, init_env
from ctm.generic import ctmrg
eng=_native.engine()
s4=read_ipeps_c4v(ROOT+"/tests/golden/test-input/RVB_1x1.in")
A=s4.site().cuda()
res={}
for blk in (64,32):
    eng.set_option("lz_block", blk)
    st=IPEPS({(x,y):A.clone() for x in range(2) for y in range(2)})
    env=ENV(80, st); init_env(st, env)
    lz0=eng.stat("lz_hits")
    for sw in range(4):
        for d in cfg.ctm_args.ctm_move_sequence:
            for _ in range(2): ctmrg.ctm_MOVE(d, st, env)
    print("block",blk,"krylov solves",eng.stat("lz_hits")-lz0, "fallbacks", eng.stat("si_fallbacks"), "async fb", eng.stat("lz_async_fallbacks"))
    res[blk]={k:(v/v[0]).cpu().numpy() for k,v in env.get_spectra().items()}
print("max spectra diff", max(np.abs(res[64][k]-res[32][k]).max() for k in res[64]))
print(list(res[32].values())[0][:12])
