#!/usr/bin/env python3
"""How good is the fused pass's floor prediction?  For several scenes and sample strides: the stratified estimate of the
two verified counts against their exact values, in pixels and in units of the margin to the target."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth

side = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rows = cols = side
pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
out = []
with S.Context(0, timing=True, fused_clahe=True) as c:
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    for seed_off in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + seed_off, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        torch.cuda.synchronize()
        for stride in (16, 32, 64, 128):
            os.environ["SARPRO_HIP_FUSED_SAMPLE"] = str(stride)
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            r = c.fused_report()
            t = dict(c.last_kernel_times())
            target = round(2 * r["total_px"] * 0.05)
            out.append({"seed": seed_off, "stride": stride, "floor_pred": r["floor_pred"], "verdict": r["verdict"], "spec_ok": r["spec_ok"],
                        "err_px": [round(r["cum_est"][i] - r["n_lt"][i]) for i in range(2)],
                        "margin_px": [r["n_lt"][0] - target, r["n_lt"][1] - target],
                        "sample_ms": round(t.get("fused_sample", 0), 4), "predict_ms": round(t.get("fused_predict", 0), 4), "queued": r["queued"][1], "overflowed": r["overflowed"][1]})
            print(json.dumps(out[-1]), flush=True)
