"""Which earlier call on ANOTHER context slows the dual-pol 2048^2 f32 call of a fresh context (bench.py's secondary sequence)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SyntheticRgbMode as Mode, synth, resize_output_dims
side = 2048
rows = cols = 20000; pitch = 20032
fb = [torch.rand((side, side), dtype=torch.float32, device="cuda") * 900.0 + 1.0 for _ in range(2)]
rgb = torch.empty((side, side * 3), dtype=torch.uint8, device="cuda")
o1 = torch.empty((side, side), dtype=torch.uint8, device="cuda")
def timed(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def measure(tag):
    c2 = S.Context(0)
    ms = timed(lambda: c2.dev_dualpol_synrgb_f32(fb[0].data_ptr(), fb[1].data_ptr(), side, side, side, St.Default, Mode.Default, rgb.data_ptr(), side))
    c2.close(); print(f"{tag}: {ms:.4f} ms", flush=True)
measure("start")
cp = S.Context(0); q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
for b in range(2): cp.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
torch.cuda.synchronize(); measure("after the scene generator on cp")
fc, fr = resize_output_dims(cols, rows, 2048, True)
small = torch.empty((fr * fc * 3,), dtype=torch.uint8, device="cuda")
timed(lambda: cp.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, 2048, True, small.data_ptr()), n=3, warm=1)
measure("after the resized flow on cp")
timed(lambda: cp.dev_autoscale_band_f32(fb[0].data_ptr(), side, side, side, St.Standard, Bd.U8, o1.data_ptr(), side, want_stats=False), n=20, warm=3)
measure("after single-band f32 on cp")
big = torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda")
timed(lambda: cp.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, Mode.Default, big.data_ptr(), pitch), n=5, warm=2)
measure("after dual-pol u16 Robust on cp")
cp.close(); measure("cp closed")
