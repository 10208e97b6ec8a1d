#!/usr/bin/env python3
"""Timeline of one f32-flavour call (config 3(ii): log-ratio pol-op of two u16 bands -> CLAHE u16) at 400 MP, for
rocprofv3 --kernel-trace: python tools/trace_f32.py [rows cols]; tools/trace_gaps.py prints the kernel timeline with the gaps."""
import sys, time
sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth

rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (20000, 20000)
pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
with S.Context(0, timing=False) as c:
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for b in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    out = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    for strategy, bd in ((St.Clahe, Bd.U16), (St.Clahe, Bd.U8), (St.Robust, Bd.U16)):
        ts = []
        for i in range(4):
            t = time.perf_counter()
            c.dev_polop_autoscale_band(Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, rows, cols, pitch, strategy, bd, out.data_ptr(), pitch,
                                       want_stats=False)
            ts.append((time.perf_counter() - t) * 1e3)
        print(strategy.name, bd.name, "wall ms per call:", " ".join(f"{x:.3f}" for x in ts), flush=True)
