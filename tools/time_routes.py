#!/usr/bin/env python3
"""What a scene costs on each outcome of the speculative route: scene A (400 MP unless SIDE is set) accepted, with the predicted floor
forced one level off (retried: a second fused pass), two levels off (refuted twice: the exact kernels) and with the retry switched off
(round 5's behaviour), one call per scene on one stream; per-kernel event times of the last call beside the wall-clock mean."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
with S.Context(0, timing=True, async_dev=True) as c:
    if rows * cols < (32 << 20):
        c.set_attr("SAMPLED_HIST_MIN_PX", 0)
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    def run(n):
        for _ in range(n):
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch, want_stats=False)
        c.synchronize()
    run(10)
    for force in (None, "mispredict", "mispredict2", "mispredict,noretry", None):
        c.set_attr("SPEC_FORCE", force)
        run(3); c.last_kernel_times()
        c.time_only("none")  # (no event pairs inside the timed loop)
        torch.cuda.synchronize(); t = time.perf_counter(); run(20); ms = (time.perf_counter() - t) / 20 * 1e3
        c.time_only(None)
        run(1)
        kt = {}
        for n, v in c.last_kernel_times():
            if not n.startswith("host:"):
                kt[n] = round(kt.get(n, 0.0) + v, 4)
        print(json.dumps({"force": force, "outcome": c.spec_report()["outcome"], "ms_per_scene": round(ms, 4), "kernels_ms": kt}), flush=True)
