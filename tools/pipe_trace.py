#!/usr/bin/env python3
"""One resident batch (sarpro_hip_batch_dualpol_synrgb_u16_dev) of a few 400 MP scenes with nothing else in the process: the target of
`rocprofv3 --kernel-trace` for the lane-overlap timeline (tools/trace_overlap.py reads the trace).  usage: pipe_trace.py [scenes] [lanes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows = cols = 20000; pitch = 20032
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
q0 = synth.q_tables()
with S.Context(0) as c:
    scenes, outs = [], []
    for name, off, flags, qkw, _ in synth.BENCH_SCENES[:3]:
        q = synth.q_tables(**qkw) if qkw else q0
        d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + off, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch, flags)
        scenes.append(d)
        outs.append(torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    batch = [(scenes[i % 3][0].data_ptr(), scenes[i % 3][1].data_ptr(), outs[i % 3].data_ptr()) for i in range(n)]
    for _ in range(3):  # (the first batches make the lanes' plans and workspaces; the last one is the one to read)
        rep, st, routes = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=lanes)
    print(rep, routes)
