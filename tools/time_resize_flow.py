#!/usr/bin/env python3
"""BASELINE config 2 as a device-resident flow (400 MP dual-pol -> Robust u8 x2 -> Lanczos3 to 2048^2 -> pad -> synRGB): wall
time per scene and every kernel of the composite call, with the horizontal pass reading the DN raster through the autoscale table (default), with the u8 level raster and the
register-resident resize kernels (SARPRO_HIP_NO_RESIZE_LUT=1), and with the generic kernels (SARPRO_HIP_RESIZE_GENERIC=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sw
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, synth, resize_output_dims
rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
for env in ("", "nolut", "1"):
    if env == "nolut":
        sw.set("SARPRO_HIP_NO_RESIZE_LUT", "1")
    if env == "1":
        sw.set("SARPRO_HIP_RESIZE_GENERIC", "1")
    with S.Context(0, timing=True) as c:
        band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for b in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
        fc, fr = resize_output_dims(cols, rows, target, True)
        rgb = torch.empty((fr * fc * 3,), dtype=torch.uint8, device="cuda")
        dts = []
        for it in range(5):
            t = time.perf_counter()
            c.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, target, True, rgb.data_ptr())
            dts.append((time.perf_counter() - t) * 1e3)
        kt = {}
        for k, v in c.last_kernel_times():
            kt[k] = round(kt.get(k, 0) + v, 4)
        print({"": "DN through the table in the horizontal pass", "nolut": "level raster, register-resident passes", "1": "level raster, generic passes"}[env], "ms/scene", round(sorted(dts)[2], 3), kt, flush=True)
