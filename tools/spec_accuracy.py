#!/usr/bin/env python3
"""How good is the speculative CLAHE chain's floor prediction?  For several 400 MP scenes and sample strides: the sample's
estimate of the two cumulative counts against the exact ones the compose pass counts, the verdict, and the distance of the
target from the nearest boundary (what the estimate's error has to stay under).  usage: spec_accuracy.py [nseeds] [side]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sw
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows = cols = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
sw.set("SARPRO_HIP_SAMPLED_HIST_MIN_PX", "0")
with S.Context(0, timing=True) as c:
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    for seed in range(nseeds):
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + seed, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        for stride in (9, 17, 33, 65):
            sw.set("SARPRO_HIP_SAMPLE_STRIDE", str(stride))
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            r = c.spec_report()
            err = [r["est_lt"][i] - r["n_lt"][i] for i in range(2)] if r["spec_ok"] else None
            room = min(r["target"] - r["n_lt"][0], r["n_lt"][1] - r["target"]) if r["spec_ok"] and r["verdict"] == 0 else None
            print(json.dumps({"seed": seed, "stride": stride, "spec_ok": r["spec_ok"], "verdict": r["verdict"], "floor": r["floor_pred"],
                              "est_minus_exact": err, "room_to_boundary": room, "level_population": r["n_lt"][1] - r["n_lt"][0]}), flush=True)
