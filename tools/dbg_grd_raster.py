#!/usr/bin/env python3
"""One raster of tools/soak_grd_like.py (by index) through the routes and the oracle: which one is off, and what the speculation saw.
usage: dbg_grd_raster.py <i> [seed0]"""
import importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
i = int(sys.argv[1]); seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sys.argv = [sys.argv[0], "0", str(seed0)]
spec = importlib.util.spec_from_file_location("soak", os.path.join(ROOT, "tools", "soak_grd_like.py"))
src = open(os.path.join(ROOT, "tools", "soak_grd_like.py")).read().split("tally, forms")[0]
ns = {"__name__": "soak", "__file__": os.path.join(ROOT, "tools", "soak_grd_like.py")}
exec(compile(src, "soak_grd_like.py", "exec"), ns)
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode
import oracle
rng = np.random.default_rng(seed0)
for k in range(i + 1):
    rows = int(rng.integers(6000, 7200)); cols = int(rng.integers(6000, 7200))
pitch = (cols + 63) // 64 * 64
bands, what = ns["make_scene"](rows, cols, pitch, seed0 * 100003 + i)
print(rows, cols, what)
host = [b[:, :cols].cpu().numpy().view(np.uint16) for b in bands]
rc, ref, r1, r2 = oracle.dualpol_synrgb(host[0].astype(np.float32), host[1].astype(np.float32), int(St.Clahe))
assert rc == 0
print("oracle levels: band0 min", int(r1.min()), "band1 min", int(r2.min()))
with S.Context(0) as c:
    for attrs in ({}, {"NO_SPEC": 1}, {"SPEC_FORCE": "noretry"}, {"NO_SPEC_RESCALE": 1}, {"NO_FUSED_RGB": 1}, {"NO_SAMPLED_HIST": 1}):
        for k, v in attrs.items():
            c.set_attr(k, v)
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        c.dev_dualpol_synrgb_u16(bands[0].data_ptr(), bands[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch, want_stats=False)
        got = rgb.view(rows, pitch, 3)[:, :cols].cpu().numpy()
        try:
            rep = c.spec_report()
        except Exception as e:
            rep = {"error": str(e)}
        cr = c.chain_report()
        bad = int((got != ref).any(axis=2).sum())
        print(json.dumps({"attrs": attrs, "pixels_off_oracle": bad, "outcome": rep.get("outcome"), "spec_ok": rep.get("spec_ok"), "floor_first": rep.get("floor_first"), "floor_pred": rep.get("floor_pred"),
                          "n_lt": rep.get("n_lt"), "target": rep.get("target"), "est_lt": rep.get("est_lt"), "min_pred": rep.get("min_pred"), "n_below_min": rep.get("n_below_min"),
                          "fwc": int(cr["floor_with_cushion"]), "identity": [int(x) for x in cr["identity"]], "resc_head": [[int(x) for x in cr["rescale"][b][:6]] for b in range(2)],
                          "hist_head": [[int(x) for x in cr["level_hist"][b][:10]] for b in range(2)]}), flush=True)
        for k in attrs:
            c.set_attr(k, None)
lv = np.concatenate([r1.ravel(), r2.ravel()])
h = np.bincount(lv, minlength=256)
cum = np.cumsum(h); target = round(lv.size * 0.05)
print("oracle floor:", int(np.argmax(cum >= target)), "target", target, "cum[:12]", cum[:12].tolist())
