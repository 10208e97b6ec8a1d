"""Single-band timings (400 MP u16 band resident in HBM): u8 and u16 output of a few strategies."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import sarpro_amd as S
from sarpro_amd import synth
from sarpro_amd.types import AutoscaleStrategy as St, BitDepth as Bd
rows=cols=20000; pitch=20032
ctx=S.Context(0, timing=True); q=synth.q_tables()
band=torch.empty((rows,pitch),dtype=torch.int16,device="cuda")
out=torch.empty((rows,pitch),dtype=torch.uint8,device="cuda")
ctx.dev_synth_scene_u16(synth.SEED_SCENE_A,0,q,rows,cols,0,rows,band.data_ptr(),pitch)
for st in (St.Robust, St.Standard, St.Clahe):
    for i in range(4):
        torch.cuda.synchronize(); t=time.perf_counter()
        ctx.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, st, Bd.U8, out.data_ptr(), pitch)
        dt=(time.perf_counter()-t)*1e3
    print(st.name, "u8", round(dt,3), {k: round(v,3) for k,v in ctx.last_kernel_times()})
out16=torch.empty((rows,pitch),dtype=torch.int16,device="cuda")
for st in (St.Robust, St.Standard):
    for i in range(4):
        torch.cuda.synchronize(); t=time.perf_counter()
        ctx.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, st, Bd.U16, out16.data_ptr(), pitch)
        dt=(time.perf_counter()-t)*1e3
    print(st.name, "u16", round(dt,3), {k: round(v,3) for k,v in ctx.last_kernel_times()})
