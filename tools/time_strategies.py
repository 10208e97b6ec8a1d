"""Per-strategy step time of the HBM-resident dual-pol pass (400 MP), and the f32 flavour of one band."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sarpro_amd as S
from sarpro_amd import synth
rows = cols = 20000; pitch = 20032
ctx = S.Context(0, timing=True); q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
rgb = torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda")
for b in range(2):
    ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
for st in S.AutoscaleStrategy:
    for i in range(4):
        t = time.perf_counter()
        ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, st, 0, rgb.data_ptr(), pitch)
        dt = (time.perf_counter() - t) * 1e3
    print(f"dual-pol {st.name:10s} {dt:7.2f} ms  ", {k: round(v, 3) for k, v in ctx.last_kernel_times()})
# f32 flavour: one band as float (what GDAL hands over), CLAHE U16 and Robust U8
f = torch.empty((rows, pitch), dtype=torch.float32, device="cuda")
f.copy_((band[0].to(torch.int32) & 0xFFFF).to(torch.float32))
out16 = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
for st, bd in ((S.AutoscaleStrategy.Clahe, S.BitDepth.U16), (S.AutoscaleStrategy.Clahe, S.BitDepth.U8), (S.AutoscaleStrategy.Robust, S.BitDepth.U8)):
    for i in range(3):
        t = time.perf_counter()
        ctx.dev_autoscale_band_f32(f.data_ptr(), rows, cols, pitch, st, bd, out16.data_ptr(), pitch * (2 if bd == S.BitDepth.U8 else 1))
        dt = (time.perf_counter() - t) * 1e3
    print(f"f32 band {st.name:8s} {bd.name:4s} {dt:7.2f} ms  ", {k: round(v, 3) for k, v in ctx.last_kernel_times()})
