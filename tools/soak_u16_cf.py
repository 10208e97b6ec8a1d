"""The conflict-free exact CLAHE kernel with u16 levels out (kernels.hip 4a) against the kernel of rounds 1-3 (NO_U16_CF=1) on random
shapes, scenes and item heights: python tools/soak_u16_cf.py [n_cases]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rnd = random.Random(20261003)
bad = 0; checked_oracle = 0
for case in range(n):
    rows, cols = rnd.randint(16, 2600), rnd.randint(64, 2600)
    if not oracle.clahe_shape_ok(rows, cols):
        continue
    item_rows = rnd.choice([None, 16, 48, 160, 512, 1024])
    if item_rows: os.environ["SARPRO_HIP_U16_ITEM_ROWS"] = str(item_rows)
    else: os.environ.pop("SARPRO_HIP_U16_ITEM_ROWS", None)
    flags = rnd.choice([0, synth.NO_WEDGE, synth.NO_BRIGHT, synth.NO_WEDGE | synth.NO_BRIGHT])
    sig = rnd.choice([None, (3.0, 2.0), (40.0, 25.0), (420.0, 260.0), (3000.0, 2000.0)])
    q = synth.q_tables(sigma=sig) if sig else synth.q_tables()
    pitch = (cols + 63) // 64 * 64 + rnd.choice([0, 64])
    with S.Context(0) as c:
        band = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
        c.dev_synth_scene_u16(1000 + case, rnd.randint(0, 1), q, rows, cols, 0, rows, band.data_ptr(), pitch, flags)
        outs = []
        for off in (None, 1):
            if off: c.set_attr("NO_U16_CF", 1)
            o = torch.full((rows, pitch), -1, dtype=torch.int16, device="cuda")
            c.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, o.data_ptr(), pitch)
            torch.cuda.synchronize(); outs.append(o)
        same = bool(torch.equal(outs[0][:, :cols], outs[1][:, :cols])) and bool((outs[0][:, cols:] == -1).all())
        if same and rows * cols <= 600_000:  # the oracle on the small ones
            rc, ref = oracle.pipeline(band[:, :cols].cpu().numpy().view(np.uint16).astype(np.float32), int(Bd.U16), int(St.Clahe))
            same = rc == 0 and np.array_equal(outs[0][:, :cols].cpu().numpy().view(np.uint16), ref)
            checked_oracle += 1
        if not same:
            bad += 1
            print(f"case {case}: {rows}x{cols} pitch {pitch} items {item_rows} flags {flags} sigma {sig}: DIFFERENT", flush=True)
print(f"{n} cases ({checked_oracle} also against the oracle): {bad} with differences")
sys.exit(1 if bad else 0)
