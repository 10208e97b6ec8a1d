"""Per-kernel HIP-event times and host segments of the headline pass (400 MP dual-pol CLAHE + synRGB)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import sarpro_amd as S
from sarpro_amd import synth
rows=cols=20000; pitch=20032
ctx=S.Context(0, timing=True); q=synth.q_tables()
band=[torch.empty((rows,pitch),dtype=torch.int16,device="cuda") for _ in range(2)]
rgb=torch.empty((rows,pitch*3),dtype=torch.uint8,device="cuda")
for b in range(2): ctx.dev_synth_scene_u16(synth.SEED_SCENE_A,b,q,rows,cols,0,rows,band[b].data_ptr(),pitch)
def run(n=6):
    acc={}
    for i in range(n):
        t=time.perf_counter()
        ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(),band[1].data_ptr(),rows,cols,pitch,4,0,rgb.data_ptr(),pitch)
        dt=(time.perf_counter()-t)*1e3
        if i>=2:
            for k,v in ctx.last_kernel_times(): acc.setdefault(k,[]).append(v)
            acc.setdefault("TOTAL",[]).append(dt)
    return {k: round(float(np.mean(v)),3) for k,v in acc.items()}
r = run()
for k, v in r.items():
    print(f"{k:45s} {v:8.3f} ms")
