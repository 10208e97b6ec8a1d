#!/usr/bin/env python3
"""Fused CLAHE pass on the headline scene: diagnostics (sarpro_hip_ctx_fused_report), per-kernel times, and the forced
routes compared with the default one."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth

rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
    out = {}
    ref = None
    for force in (None, "nospec", "mispredict", "twolevel", "tinyqueue"):
        if force:
            os.environ["SARPRO_HIP_FUSED_FORCE"] = force
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for it in range(2):
            c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        img = rgb.view(rows, pitch, 3)[:, :cols]
        if ref is None:
            ref = img.clone()
        nd = int((img != ref).any(dim=2).sum().item())
        first = None
        if nd:
            idx = (img != ref).any(dim=2).nonzero()[:5].tolist()
            first = [(r, cc, img[r, cc].tolist(), ref[r, cc].tolist()) for r, cc in idx]
        out[str(force)] = {"report": c.fused_report(), "kernels_ms": {n: round(ms, 4) for n, ms in c.last_kernel_times() if n.startswith(("fused", "clahe"))},
                           "pixels_differing_from_default": nd, "first": first}
    print(json.dumps(out, indent=1))
