#!/usr/bin/env python3
"""Per-kernel times of the headline chain on each of bench.py's scenes (synth.BENCH_SCENES), 400 MP unless SIDE is set."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
q0 = synth.q_tables()
with S.Context(0, timing=True) as c:
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    first = True
    for name, off, flags, qkw, what in synth.BENCH_SCENES:
        q = synth.q_tables(**qkw) if qkw else q0
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + off, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch, flags)
        torch.cuda.synchronize()
        acc = {}
        if first:  # (the first calls of a process read 3-5 % slow whatever the scene: not charged to scene A)
            for _ in range(10):
                c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            c.last_kernel_times(); first = False
        for it in range(5):
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            if it >= 2:
                for n, ms in c.last_kernel_times():
                    acc.setdefault(n, []).append(ms)
        print(name, c.spec_report()["outcome"], json.dumps({n: round(sorted(v)[len(v) // 2], 4) for n, v in acc.items()}), flush=True)
