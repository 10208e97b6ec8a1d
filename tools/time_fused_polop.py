import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, synth
rows = cols = 20000; pitch = 20032
q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
out = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
res = {}
for timing in (False, True):
    with S.Context(0, timing=timing) as c:
        if not timing:
            for b in range(2): c.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
        f = lambda: c.dev_polop_autoscale_band(Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, rows, cols, pitch, St.Clahe, Bd.U16, out.data_ptr(), pitch, want_stats=False)
        for _ in range(2): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(6): f()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 6 * 1e3
        if timing: res["kernels"] = {k: round(v, 4) for k, v in c.last_kernel_times() if not k.startswith("host:")}
        else: res["plain_ms"] = round(ms, 3)
print(json.dumps(res))
