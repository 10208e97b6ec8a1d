"""Batch driver throughput (BASELINE config 5 style): N in-memory dual-pol scenes -> 1024^2 padded synRGB,
with 1, 2 and 3 workers on one GPU (a device listed k times = k worker threads/contexts on it)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sarpro_amd as S
from sarpro_amd import synth

side, n = 10000, 8
ctx = S.Context(0); q = synth.q_tables()
pitch = (side + 63) // 64 * 64
scenes = []
for k in range(n):
    hb = []
    for b in range(2):
        d = torch.empty((side, pitch), dtype=torch.int16, device="cuda")
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + k, b, q, side, side, 0, side, d.data_ptr(), pitch)
        h = torch.empty((side, side), dtype=torch.int16, pin_memory=True)
        h.copy_(d[:, :side])
        hb.append(h.numpy().view(np.uint16))
    scenes.append(tuple(hb))
ctx.close()
for devs in ([0], [0, 0], [0, 0, 0], [0, 0, 0, 0]):
    S.batch_dualpol_synrgb_resized(devs, scenes[:len(devs)], S.AutoscaleStrategy.Clahe, 1024, True)  # warm-up
    t = time.perf_counter()
    outs, rep, st, rc = S.batch_dualpol_synrgb_resized(devs, scenes, S.AutoscaleStrategy.Clahe, 1024, True)
    dt = time.perf_counter() - t
    assert rc == 0 and rep.processed == n
    print(f"workers on GPU0: {len(devs)}  {dt*1e3/n:7.1f} ms/scene  {n*side*side/dt/1e6:8.0f} Mpix/s  ({n*2*side*side*2/dt/1e9:.1f} GB/s H2D)")
