#!/usr/bin/env python3
"""Small-scene throughput of the f32 dual-pol flow -- the reference's default size (api/mod.rs:374-449 feeds the core bands of a few
megapixels): N scenes of SIDE x SIDE f32 bands -> CLAHE / Tamed / Robust -> resize + pad to SIDE -> synRGB,
  * resident: sarpro_hip_batch_dualpol_synrgb_resized_f32_dev over 1, 2, 4, 8 lanes, and one call per scene on one context (the baseline),
  * host to host: sarpro_hip_batch_dualpol_synrgb_resized_f32 with 1, 2, 4, 8 workers on one device.
Scenes per second, best of three batches after a warm-up batch.  usage: batch_rate_f32.py [side] [scenes]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, resize_output_dims
from f32data import resampled_scene

side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pitch = (side + 63) // 64 * 64
distinct = 4
host = [(resampled_scene(side, side, 0, seed=10 + i), resampled_scene(side, side, 1, seed=20 + i)) for i in range(distinct)]
fc, fr = resize_output_dims(side, side, side, True)
dev = []
for a, b in host:
    pair = []
    for x in (a, b):
        t = torch.zeros((side, pitch), dtype=torch.float32, device="cuda")
        t[:, :side] = torch.from_numpy(x).cuda()
        pair.append(t)
    dev.append(pair)
outs = [torch.zeros((fr * fc * 3,), dtype=torch.uint8, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
batch = [(dev[i % distinct][0].data_ptr(), dev[i % distinct][1].data_ptr(), outs[i].data_ptr()) for i in range(n)]


def best(fn, reps=3):
    fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    return min(ts)


for strategy, plain in ((St.Clahe, True), (St.Tamed, False), (St.Robust, True)):
    with S.Context(0) as c:
        def single():
            for b1, b2, o in batch:
                c.dev_dualpol_synrgb_resized_f32(b1, b2, side, side, pitch, strategy, side, True, o, plain_pipeline=plain)
        t1 = best(single)
        rec = {"side": side, "scenes": n, "strategy": strategy.name, "plain_pipeline": plain, "resident_one_call_per_scene_scenes_per_s": round(n / t1, 1), "resident_lanes": {}}
        ref = outs[0].clone()
        for lanes in (1, 2, 4, 8):
            t = best(lambda: c.dev_batch_dualpol_synrgb_resized_f32(batch, side, side, pitch, strategy, side, True, plain_pipeline=plain, lanes=lanes))
            rec["resident_lanes"][str(lanes)] = round(n / t, 1)
            assert torch.equal(outs[0], ref)
        rec["resident_speedup_4_lanes"] = round(rec["resident_lanes"]["4"] / rec["resident_one_call_per_scene_scenes_per_s"], 2)
    hs = [host[i % distinct] for i in range(n)]
    rec["host_to_host_workers"] = {}
    for w in (1, 2, 4, 8):
        t = best(lambda: S.batch_dualpol_synrgb_resized_f32([0], hs, strategy, side, True, plain_pipeline=plain, workers_per_device=w), reps=2)
        rec["host_to_host_workers"][str(w)] = round(n / t, 1)
    print(json.dumps(rec), flush=True)
