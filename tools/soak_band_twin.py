#!/usr/bin/env python3
"""Stress of the band twin's thread handshake (context.h BandWorker): tens of thousands of back-to-back dual-pol f32 calls on small
rasters (the second band on the helper thread every time), calls separated by pauses longer than the spin phase (the worker
sleeps on its condition variable in between), and contexts created and destroyed around a few calls each.  Every RGB must equal
the first one; the script must end.  usage: python tools/soak_band_twin.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import f32data
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows, cols, pitch = 200, 264, 320
b = []
for k in range(2):
    t = torch.zeros((rows, pitch), dtype=torch.float32, device="cuda")
    t[:, :cols] = torch.from_numpy(f32data.resampled_scene(rows, cols, k)).cuda()
    b.append(t)
rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
t0 = time.time()
with S.Context(0) as c:
    c.dev_dualpol_synrgb_f32(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, St.Default, Mode.Default, rgb.data_ptr(), pitch)
    ref = {s: None for s in (St.Default, St.Tamed, St.Clahe)}
    for s in ref:
        c.dev_dualpol_synrgb_f32(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, s, Mode.Default, rgb.data_ptr(), pitch)
        ref[s] = rgb.clone()
    bad = 0
    for i in range(n):
        s = (St.Default, St.Tamed, St.Clahe)[i % 3]
        c.dev_dualpol_synrgb_f32(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, s, Mode.Default, rgb.data_ptr(), pitch)
        if i % 97 == 0:
            bad += int(not torch.equal(rgb, ref[s]))
        if i % 1000 == 999:
            time.sleep(0.01)  # longer than the worker's spin: it goes to sleep and must wake up
    print(f"{n} back-to-back calls, {time.time() - t0:.1f} s, mismatches {bad}", flush=True)
t0 = time.time()
for i in range(150):
    with S.Context(0) as c:
        for s in (St.Default, St.Clahe):
            c.dev_dualpol_synrgb_f32(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, s, Mode.Default, rgb.data_ptr(), pitch)
            bad += int(not torch.equal(rgb, ref[s]))
print(f"150 contexts created, used and destroyed, {time.time() - t0:.1f} s, mismatches {bad}")
