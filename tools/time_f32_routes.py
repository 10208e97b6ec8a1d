#!/usr/bin/env python3
"""400 MP, u16 DN bands resident in HBM: the log-ratio pol-op followed by the autoscale of its f32 raster, four ways --
unfused (k_polop_f32 + f32 flavour), fused, fused without the zone route, fused with the 65535-entry level table."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sw
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth

rows = cols = 20000; pitch = 20032
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for b in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    out = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
    f = []
    for b in band:
        x = b.to(torch.float32); x[x < 0] += 65536.0; f.append(x.contiguous())
    ratio = torch.empty((rows, pitch), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def timed(fn, n=4):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    for strategy, bd in ((St.Clahe, Bd.U16), (St.Clahe, Bd.U8), (St.Robust, Bd.U16), (St.Robust, Bd.U8), (St.Standard, Bd.U16)):
        def unfused():
            c.dev_polop_f32(Op.LogRatio, f[0].data_ptr(), f[1].data_ptr(), rows * pitch, ratio.data_ptr())
            c.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, pitch, strategy, bd, out.data_ptr(), pitch, want_stats=False)
        def fused():
            c.dev_polop_autoscale_band(Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, rows, cols, pitch, strategy, bd, out.data_ptr(), pitch, want_stats=False)
        res = {}
        for name, env, fn in (("unfused (polop_f32 + f32 flavour, 4096-bin sweep, level table)", {"SARPRO_HIP_F32_ZONES": "0", "SARPRO_HIP_F32_LEVEL_TABLE": "1"}, unfused),
                              ("fused, 4096-bin sweep, level table", {"SARPRO_HIP_F32_ZONES": "0", "SARPRO_HIP_F32_LEVEL_TABLE": "1"}, fused),
                              ("fused, zone route, level table", {"SARPRO_HIP_F32_LEVEL_TABLE": "1"}, fused),
                              ("fused, zone route, queued levels (default)", {}, fused)):
            for k in ("SARPRO_HIP_F32_ZONES", "SARPRO_HIP_F32_LEVEL_TABLE"):
                sw.pop(k)
            sw.update(env)
            ms = timed(fn)
            kern = {k: round(v, 3) for k, v in c.last_kernel_times() if not k.startswith("host:")}
            print(f"{strategy.name:9s} {bd.name:4s} {name:66s} {ms:6.3f} ms  {kern}", flush=True)
