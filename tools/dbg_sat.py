import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows, cols = 6000, 6016
pitch = cols
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for b in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A + 9, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    out = []
    for attr in (None, "NO_FUSED_RGB"):
        if attr: c.set_attr(attr, 1)
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        print([n for n, _ in c.last_kernel_times()], c.spec_report()["outcome"])
        out.append(rgb.view(rows, pitch, 3).cpu().numpy())
a, b = out
d = (a != b).any(axis=2)
print("differing px", d.sum())
rr, cc = np.nonzero(d)
print("rows hist (per 375):", np.bincount(rr // 375, minlength=16))
print("cols hist (per 376):", np.bincount(cc // 376, minlength=16))
for k in range(min(10, len(rr))):
    r, cl = rr[k], cc[k]
    dn = [int(band[x][r, cl].item()) & 0xFFFF for x in range(2)]
    print(r, cl, dn, a[r, cl], b[r, cl])
cc_, rb_ = S.host_clahe_saturated_levels(rows, cols)
print("classes", np.bincount(cc_), np.bincount(rb_, minlength=8))
