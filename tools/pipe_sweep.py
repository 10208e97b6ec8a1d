#!/usr/bin/env python3
"""Resident batch (sarpro_hip_batch_dualpol_synrgb_u16_dev) over bench.py's nine scenes: ms per scene by lanes / order of the fused
passes / grids of the two sweeps, each configuration's rasters compared with those of one call per scene on one stream.
SIDE (default 20000), CONFIGS ("lanes:order:rgb_grid:piece_grid,..." ; 0 = the default grid) select the sweep."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth

rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
reps = int(os.environ.get("REPS", "3"))
q0 = synth.q_tables()
scenes = []
with S.Context(0) as c:
    for name, off, flags, qkw, what in synth.BENCH_SCENES:
        q = synth.q_tables(**qkw) if qkw else q0
        d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + off, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch, flags)
        scenes.append(d)
torch.cuda.synchronize()
K = len(scenes)
ref = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in range(K)]
out = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in range(K)]
min_px = {"SAMPLED_HIST_MIN_PX": 0} if rows * cols < (32 << 20) else {}


def view(t):
    return t.view(rows, pitch, 3)[:, :cols]


# reference: one stream, one call per scene, enqueued back to back (bench.py's loop of rounds 3-4)
with S.Context(0, async_dev=True) as c:
    for k, v in min_px.items():
        c.set_attr(k, v)
    routes = []
    for i in range(K):
        c.dev_dualpol_synrgb_u16(scenes[i][0].data_ptr(), scenes[i][1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, ref[i].data_ptr(), pitch, want_stats=False)
        c.synchronize()
        routes.append(c.spec_report()["outcome"])
    runs = []
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(3 * K):
            c.dev_dualpol_synrgb_u16(scenes[i % K][0].data_ptr(), scenes[i % K][1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, out[i % K].data_ptr(), pitch, want_stats=False)
        c.synchronize(); runs.append((time.perf_counter() - t) / (3 * K) * 1e3)
    print(json.dumps({"config": "one stream, one call per scene", "ms_per_scene": round(sorted(runs)[len(runs) // 2], 4), "routes": routes}), flush=True)

cfgs = os.environ.get("CONFIGS", "1:1:0:0,2:1:0:0,2:0:0:0,3:1:0:0,3:0:0:0,2:1:224:32,2:1:208:48,2:1:192:64,2:1:176:80,3:1:208:48")
for cfg in cfgs.split(","):
    lanes, order, rgrid, pgrid = (int(x) for x in cfg.split(":"))
    for t in out:
        t.zero_()
    with S.Context(0, timing=bool(os.environ.get("TIMING"))) as c:
        if os.environ.get("TIMING") == "2":
            c.time_only("clahe_rgb_fused")
        for k, v in min_px.items():
            c.set_attr(k, v)
        c.set_attr("PIPE_ORDER", order)
        if rgrid:
            c.set_attr("RGB_GRID", rgrid)
        if pgrid:
            c.set_attr("PIECE_GRID", pgrid)
        batch = [(scenes[i][0].data_ptr(), scenes[i][1].data_ptr(), out[i].data_ptr()) for i in range(K)]
        rep, st, rt = c.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=lanes)
        same = all(bool(torch.equal(view(ref[i]), view(out[i]))) for i in range(K))
        runs = []
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter()
            nb = int(os.environ.get("NBATCH", str(3 * K)))
            c.dev_batch_dualpol_synrgb_u16((batch * 3)[:nb], rows, cols, pitch, St.Clahe, Mode.Default, pitch, lanes=lanes)
            runs.append((time.perf_counter() - t) / nb * 1e3)
            c.last_kernel_times()
        print(json.dumps({"config": {"lanes": lanes, "order": order, "rgb_grid": rgrid, "piece_grid": pgrid}, "ms_per_scene": round(sorted(runs)[len(runs) // 2], 4),
                          "runs": [round(x, 4) for x in runs], "rasters_equal_one_stream": same, "report": rep, "routes": rt}), flush=True)
