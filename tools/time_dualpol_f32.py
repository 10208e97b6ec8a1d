#!/usr/bin/env python3
"""The reference's default flow at its usual size (`--size N` resamples on read: the hot path sees two non-integer f32 bands of
N x N): per-band autoscale -> (Tamed re-autoscale) -> synRGB, bands resident in HBM, one synchronous call; ms per call and the
kernels of the context's timing table.  usage: python tools/time_dualpol_f32.py [side ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import f32data

for side in [int(x) for x in (sys.argv[1:] or ["1024", "2048", "4096", "8192"])]:
    rows = cols = side
    pitch = (cols + 63) // 64 * 64
    b = [torch.zeros((rows, pitch), dtype=torch.float32, device="cuda") for _ in range(2)]
    for k in range(2):
        b[k][:, :cols] = torch.from_numpy(f32data.resampled_scene(rows, cols, band=k)).cuda()
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for timing in (False, True):
        with S.Context(0, timing=timing) as c:
            for strategy in (St.Default, St.Tamed, St.Standard, St.Clahe):
                for plain in (False, True):
                    def call():
                        c.dev_dualpol_synrgb_f32(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch,
                                                 plain_pipeline=plain)
                    call(); call()
                    n = 20
                    t = time.perf_counter()
                    for _ in range(n):
                        call()
                    ms = (time.perf_counter() - t) / n * 1e3
                    if not timing:
                        print(f"{side:5d}^2 {strategy.name:9s} plain={int(plain)}  {ms:7.3f} ms per call", flush=True)
                    elif not plain:
                        k = {}
                        for n_, v in c.last_kernel_times():
                            if not n_.startswith("host:"):
                                k[n_] = round(k.get(n_, 0.0) + v, 4)
                        print(f"{side:5d}^2 {strategy.name:9s} kernels {k}", flush=True)
