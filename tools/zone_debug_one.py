import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SARPRO_HIP_F32_ZONES_DEBUG"] = "1"
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, synth
side = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    rows = cols = side; pitch = (cols + 63) // 64 * 64
    d = torch.zeros((rows, pitch), dtype=torch.int16, device="cuda")
    c.dev_synth_scene_u16(synth.SEED_SCENE_A, 0, q, rows, cols, 0, rows, d.data_ptr(), pitch)
    f = d.to(torch.float32); f[f < 0] += 65536.0; f = f.contiguous()
    out = torch.zeros((rows, pitch), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for strategy in (St.Clahe, St.Standard):
        c.dev_autoscale_band_f32(f.data_ptr(), rows, cols, pitch, strategy, Bd.U8, out.data_ptr(), pitch, want_stats=False)
        print(strategy.name, [n for n, _ in c.last_kernel_times()], flush=True)
