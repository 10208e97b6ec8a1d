#!/usr/bin/env python3
"""Per-pixel cost of the headline chain's sweeps (histograms, sample pass, fused CLAHE -> RGB pass; or apply + compose) at several scene sizes, scaled to 400 MP: scenes whose level rasters
fit the 256-MiB Infinity Cache show what the compose pass would cost if its inputs did not come from HBM."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
os.environ.setdefault("SARPRO_HIP_SAMPLED_HIST_MIN_PX", "0")
q = synth.q_tables()
for side in [int(x) for x in (sys.argv[1:] or ["20000", "10000", "7000", "5000", "3500"])]:
    rows = cols = side; pitch = (cols + 63) // 64 * 64
    with S.Context(0, timing=True) as c:
        d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        acc = {}
        for it in range(10):
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            if it >= 2:
                for n, ms in c.last_kernel_times():
                    acc.setdefault(n, []).append(ms)
        med = {n: sorted(v)[len(v) // 2] for n, v in acc.items()}
        sc = 4e8 / (rows * cols)
        print(side, {k: round(med[k] * sc, 4) for k in ("dn_hist_u16", "clahe_sample", "clahe_rgb_fused", "clahe_apply_u8_spec", "compose_u8") if k in med}, "ms scaled to 400 MP; raw total", round(sum(v for k, v in med.items() if not k.startswith("host")), 4), flush=True)
    del d, rgb
