#!/bin/bash
export PYTHONPATH=$PWD
out=$PWD/gpurun_out
mkdir -p $out
timeout 1200 python tools/chip_trsv_probe.py u60 p3d41 u80 q400 u100 > $out/r06f_trsv_probe.log 2>&1; echo "probe rc=$?"
cat $out/r06f_trsv_probe.log | tail -60
python - <<'P'
import ctypes as C, sys
sys.path.insert(0,'.')
from deeppreconditioning_amd import _lib as L
import torch
torch.cuda.set_device(0)
offs=(C.c_int32*7)(-10000,-100,-1,0,1,100,10000)
for depth,wt in ((2,0),(4,0),(2,1)):
    for _ in range(2):
        g,u,l=C.c_double(),C.c_double(),C.c_int()
        L.check(L.lib().dpcg_debug_l2_gather(124992, 200 if not wt else 40, offs, depth, wt, None, C.byref(g), C.byref(u), C.byref(l)))
    print("l2 gather probe depth",depth,"written_through",wt,":",round(g.value,1),"GB/s",round(u.value,3),"us per pass; groups on one XCD:",l.value)
P
