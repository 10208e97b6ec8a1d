#!/usr/bin/env python3
"""CLAHE with u16 output of one 400 MP band (BASELINE config 3(i)): per-kernel times, for A/B runs of library builds.
usage: python tools/time_clahe_u16.py [lib ...]   ('-' = the default build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, %r)
import torch
import sarpro_amd as S
from sarpro_amd import synth
rows = cols = 20000; pitch = 20032
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    band = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
    out = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
    c.dev_synth_scene_u16(synth.SEED_SCENE_A, 0, q, rows, cols, 0, rows, band.data_ptr(), pitch)
    acc = {}
    for it in range(8):
        torch.cuda.synchronize()
        c.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, S.AutoscaleStrategy.Clahe, S.BitDepth.U16, out.data_ptr(), pitch)
        if it >= 2:
            for n, ms in c.last_kernel_times():
                acc.setdefault(n, []).append(ms)
    print(json.dumps({n: round(sorted(v)[len(v) // 2], 4) for n, v in acc.items()}))
''' % ROOT
for lib in sys.argv[1:] or ["-"]:
    env = dict(os.environ)
    if lib != "-":
        env["SARPRO_HIP_LIB"] = os.path.join(ROOT, "sarpro_amd", lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(lib, line[-1] if line else ("FAILED: " + out.stderr[-400:]), flush=True)
