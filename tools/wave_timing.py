#!/usr/bin/env python3
"""Phase timing of the fused wave kernel (a library built with -DMGP_WAVE_TIMING=1: tools/mkvariant.sh wtiming
mgp_fused_wave_inst_f64.hip -DMGP_WAVE_TIMING=1 (or _f32); built-in instantiations only: MUYGPYS_HIP_JIT=0).  Share of
a wave's life per phase on a BASELINE config.

    MUYGPYS_HIP_JIT=0 MUYGPYS_HIP_LIB=variants/lib_wtiming.so python tools/wave_timing.py [config]
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from muygpys_amd import _lib
from muygpys_amd.fused import PackedTable, pack_table

cid = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = dict(bench.CONFIGS[cid])
if cid == 4:
    cfg["batch"] = 2_000_000
dev = torch.device("cuda", 0)
w = bench.build_workload(cfg, dev, 0, False)
use_packed = PackedTable.supported(w["d"], w["R"], w["k"], w["td"])
if use_packed:
    pack_table(w["X"], w["y"])
step = bench.make_step(cfg, w, "fused", "auto", use_packed)
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 8)()
for it in range(3):
    step()
    torch.cuda.synchronize()
    getattr(lib, "mgp_debug_wave_timing_" + cfg["dtype"])(out, 1)
print(_lib.last_kernel())
names = ["tile wait + barrier", "centre rows, norms", "Gram distances", "cov + exchange + read-back (+ next gather issue)",
         "elimination", "outputs", "-", "-"]
tot = sum(out)
for nm, v in zip(names, out):
    print(f"{nm:48s} {v / tot * 100:6.1f} %   {v / w['b']:10.1f} ticks per neighbourhood")
