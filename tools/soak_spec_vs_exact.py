"""Soak of the speculative CLAHE kernels: N full-size synthetic scenes (different seeds), each processed three times --
the fused CLAHE -> RGB pass (the product path: speculative f32 blend with exact fallback, predicted and verified floor), the
apply + compose route with the same speculative blend (SARPRO_HIP_NO_FUSED_RGB=1), and every pixel through the exact f64
blend (SARPRO_HIP_NO_SPEC=1) -- and the RGB rasters compared byte for byte on the device.
usage: python tools/soak_spec_vs_exact.py [n_scenes] [rows] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sw
import torch
import sarpro_amd as S
from sarpro_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
pitch = (cols + 63) // 64 * 64
ctx = S.Context(0); q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
rgb = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in range(3)]
SW = ("SARPRO_HIP_NO_SPEC", "SARPRO_HIP_NO_FUSED_RGB")
bad = 0
t0 = time.time()
for k in range(n):
    for b in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + 1000 + k, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    for which, env in ((0, None), (1, "SARPRO_HIP_NO_SPEC"), (2, "SARPRO_HIP_NO_FUSED_RGB")):
        for k2 in SW: sw.pop(k2)
        if env: sw.set(env, "1")
        ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, S.AutoscaleStrategy.Clahe,
                                   S.SyntheticRgbMode.Default, rgb[which].data_ptr(), pitch)
    for k2 in SW: sw.pop(k2)
    diff = int((rgb[0].view(rows, pitch, 3)[:, :cols] != rgb[1].view(rows, pitch, 3)[:, :cols]).sum().item())
    diff += int((rgb[2].view(rows, pitch, 3)[:, :cols] != rgb[1].view(rows, pitch, 3)[:, :cols]).sum().item())
    bad += diff != 0
    print(f"scene {k}: {'equal' if diff == 0 else f'{diff} bytes differ'}", flush=True)
print(f"{n} scenes of {rows}x{cols}: {bad} with differences, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
