#!/usr/bin/env python3
"""Soak of the f32 flavour at sizes where the oracle is too slow to sit in a loop: random rasters of 600..3600 px a side
(f32 bands and log-ratio / normalised-difference of u16 bands), every strategy and depth; the default route's raster must
equal the raster of each cross-check switch: the 4096-bin sweep instead of the zone route, the f64 blend for every sample,
copy / fill commands instead of the mailbox, four samples per lane, the term-by-term level expression.
usage: python tools/soak_f32_switches.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sw
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op

SWITCHES = [{"SARPRO_HIP_F32_ZONES": "0"}, {"SARPRO_HIP_NO_SPEC": "1"}, {"SARPRO_HIP_NO_MAILBOX": "1"}, {"SARPRO_HIP_F32_NO_VEC8": "1"},
            {"SARPRO_HIP_F32_LEVEL_GENERAL": "1"}, {"SARPRO_HIP_F32_DIRECT": "1"}]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = runs = 0
t0 = time.time()
sw.set("SARPRO_HIP_F32_DIRECT", "0")
with S.Context(0) as c:
    for seed in range(first, first + n):
        g = torch.Generator(device="cuda"); g.manual_seed(9100 + seed)
        rng = np.random.default_rng(9100 + seed)
        rows, cols = int(rng.integers(600, 3600)), int(rng.integers(600, 3600))
        pitch = (cols + 7) // 8 * 8 if seed % 3 else (cols + 3) // 4 * 4
        kind = seed % 4
        strategy = list(St)[seed % len(St)]
        bd = Bd.U8 if (seed // 7) % 2 else Bd.U16
        if kind < 2:   # f32 band: log-normal, a share of invalid samples
            x = torch.exp(torch.randn((rows, pitch), generator=g, device="cuda") * float(rng.uniform(0.3, 2.5)) + float(rng.uniform(-3, 5)))
            if kind == 1:
                x = torch.where(torch.rand((rows, pitch), generator=g, device="cuda") < 0.2, torch.zeros_like(x), x)
            call = lambda o: c.dev_autoscale_band_f32(x.data_ptr(), rows, cols, pitch, strategy, bd, o.data_ptr(), pitch, want_stats=False)
        else:          # pol-op of two u16 bands (speckle-like)
            a = (torch.rand((rows, pitch), generator=g, device="cuda").log().neg() * float(rng.uniform(50, 900))).clamp(0, 65535).to(torch.int32)
            b = (torch.rand((rows, pitch), generator=g, device="cuda").log().neg() * float(rng.uniform(50, 900))).clamp(0, 65535).to(torch.int32)
            a16 = a.to(torch.int16) if False else (a - (a >= 32768).int() * 65536).to(torch.int16)
            b16 = (b - (b >= 32768).int() * 65536).to(torch.int16)
            op = Op.LogRatio if kind == 2 else Op.NDiff
            call = lambda o: c.dev_polop_autoscale_band(op, a16.data_ptr(), b16.data_ptr(), True, rows, cols, pitch, strategy, bd, o.data_ptr(), pitch, want_stats=False)
        dt = torch.uint8 if bd == Bd.U8 else torch.int16
        outs = []
        for env in [{}] + SWITCHES:
            saved = {k: os.environ.get(k) for k in env}
            sw.update(env)
            o = torch.zeros((rows, pitch), dtype=dt, device="cuda")
            torch.cuda.synchronize()
            call(o)
            outs.append(o[:, :cols].clone())
            for k, v in saved.items():
                if v is None: sw.pop(k)
                else: sw.set(k, v)
            runs += 1
        if int(torch.unique(outs[0][::7, ::5]).numel()) < 8:
            print(f"SUSPICIOUS seed {seed}: the default raster holds fewer than 8 distinct levels", flush=True); bad += 1
        for i in range(1, len(outs)):
            if not torch.equal(outs[0], outs[i]):
                bad += 1
                print(f"MISMATCH seed {seed} {rows}x{cols} pitch {pitch} kind {kind} {strategy.name} {bd.name} switch {SWITCHES[i - 1]}: {int((outs[0] != outs[i]).sum())} samples", flush=True)
        if (seed - first) % 200 == 199:
            print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s, {runs} runs, mismatches {bad}", flush=True)
print(f"cases {n}: {runs} runs, mismatches {bad}")
