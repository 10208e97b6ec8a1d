import os, sys, time
sys.path.insert(0, "/root/repo")
import sw
import torch, numpy as np
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, synth
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    for side in ([int(x) for x in sys.argv[1:]] or [2048, 3000, 4096, 8192]):
        rows = cols = side; pitch = (cols + 63) // 64 * 64
        d = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, 0, q, rows, cols, 0, rows, d.data_ptr(), pitch)
        f = d.to(torch.float32); f[f < 0] += 65536.0; f = f.contiguous()
        out = torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for strategy in (St.Standard, St.Clahe):
            res = []
            for env in ("0", "force"):
                sw.set("SARPRO_HIP_F32_ZONES", env)
                fn = lambda: c.dev_autoscale_band_f32(f.data_ptr(), rows, cols, pitch, strategy, Bd.U8, out.data_ptr(), pitch, want_stats=False)
                fn(); fn(); torch.cuda.synchronize()
                t = time.perf_counter()
                for _ in range(10): fn()
                torch.cuda.synchronize()
                res.append((time.perf_counter() - t) / 10 * 1e3)
            print(f"{side}^2 {strategy.name}: sweep route {res[0]:.3f} ms, zone route {res[1]:.3f} ms", flush=True)
