"""A few headline passes with nothing else in the process: the target of rocprofv3 runs
(rocprofv3 --kernel-trace --stats | --pmc ... -- python3 tools/profile_one.py <passes> <strategy>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import synth
rows=cols=20000; pitch=20032
ctx=S.Context(0); q=synth.q_tables()
band=[torch.empty((rows,pitch),dtype=torch.int16,device="cuda") for _ in range(2)]
rgb=torch.empty((rows,pitch*3),dtype=torch.uint8,device="cuda")
for b in range(2): ctx.dev_synth_scene_u16(synth.SEED_SCENE_A,b,q,rows,cols,0,rows,band[b].data_ptr(),pitch)
n=int(sys.argv[1]) if len(sys.argv)>1 else 3
strat=int(sys.argv[2]) if len(sys.argv)>2 else 4
for i in range(n):
    ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(),band[1].data_ptr(),rows,cols,pitch,strat,0,rgb.data_ptr(),pitch)
