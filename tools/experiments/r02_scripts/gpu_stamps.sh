#!/bin/bash
# diagnostic build with in-kernel stamps: print average cycles per phase per wave of k_factor
make -C hydra_pspec_amd/csrc clean > /dev/null; make -C hydra_pspec_amd/csrc -j8 HPX_STAMP=1 "$@" > gpurun_out/stamp_build.log 2>&1 || { tail -5 gpurun_out/stamp_build.log; exit 1; }
python3 - <<'PY'
import ctypes as C, numpy as np, sys, json, subprocess
sys.path.insert(0, '.')
import torch
from hydra_pspec_amd import hpx, pspec, synthetic
import os
N,T,M,nbl=512,32,12,int(os.environ.get("NBL","1024"))
d=synthetic.make_baselines(N,T,M,nbl=nbl,dense=False)
gb=pspec.GibbsBatch(d["vis"],d["flags"],d["fgmodes"],d["ninv_diag"],d["ps_prior"],3,seed=1)
ps0=np.broadcast_to(d["ps0"],(nbl,N)).copy()
gb.run(3,ps0=ps0)
L=C.CDLL(str(hpx._LIB_PATH))
n=nbl*4*8
buf=(C.c_longlong*n)()
assert L.hpx_debug_stamps(buf,n)==0
a=np.frombuffer(buf,dtype=np.int64).reshape(nbl,4,8).astype(float)
names=["diag last32+combine","next-diag partial","potf2 loop","final scale","tile init(gen)","k-loop","X mult+store","end barrier"]
tot=a.sum(axis=2)
print("cycles per wave (mean over %d WGs), total %.0f"%(nbl,tot.mean()))
for i,nm in enumerate(names):
    print("  %-16s %10.0f  %5.1f%%   per-wave-id: %s"%(nm,a[:,:,i].mean(),100*a[:,:,i].mean()/tot.mean(), np.round(a[:,:,i].mean(axis=0)).astype(int)))
PY
