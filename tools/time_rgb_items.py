#!/usr/bin/env python3
"""The fused CLAHE -> RGB pass by item geometry: one context per configuration ("RGB_ITEM_ROWS:RGB_TAIL_ROWS:RGB_TAIL_ITEM_ROWS", - = default),
the configurations interleaved round after round on the same scene, median / min of the pass's event time.  CONFIGS, ROUNDS, SIDE."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
rounds = int(os.environ.get("ROUNDS", "15"))
cfgs = os.environ.get("CONFIGS", "-:-:-,256:0:96,256:1024:96,384:1250:96,512:2500:128,512:1250:96,640:2500:128").split(",")
q = synth.q_tables()
ctxs = []
for cfg in cfgs:
    c = S.Context(0, timing=True)
    for name, v in zip(("RGB_ITEM_ROWS", "RGB_TAIL_ROWS", "RGB_TAIL_ITEM_ROWS"), cfg.split(":")):
        if v != "-":
            c.set_attr(name, int(v))
    ctxs.append(c)
d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
for k in range(2):
    ctxs[0].dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
ref = None
acc = {cfg: [] for cfg in cfgs}
tot = {cfg: [] for cfg in cfgs}
for r in range(rounds + 2):
    for cfg, c in zip(cfgs, ctxs):
        c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        kt = dict(c.last_kernel_times())
        if r == 0:
            if ref is None:
                ref = rgb.clone()
            else:
                assert torch.equal(ref, rgb), cfg
        if r >= 2:
            acc[cfg].append(kt["clahe_rgb_fused"]); tot[cfg].append(kt["host:chain(enqueue+final sync)"])
for cfg in cfgs:
    v = np.array(acc[cfg]); t = np.array(tot[cfg])
    print(json.dumps({"config": cfg, "fused_ms_median": round(float(np.median(v)), 4), "fused_ms_min": round(float(v.min()), 4), "fused_ms_p25": round(float(np.percentile(v, 25)), 4),
                      "chain_ms_median": round(float(np.median(t)), 4)}), flush=True)
