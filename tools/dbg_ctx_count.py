"""The dual-pol 2048^2 f32 call of a fresh context after various things happened in the process (allocations, other contexts, their
use): the 0.2-ms figure moves by 10 % with what ran just before it, not with the number of contexts or streams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
side = 2048
fb = [torch.rand((side, side), dtype=torch.float32, device="cuda") * 900.0 + 1.0 for _ in range(2)]
rgb = torch.empty((side, side * 3), dtype=torch.uint8, device="cuda")
def timed(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
def measure(tag):
    c2 = S.Context(0)
    ms = timed(lambda: c2.dev_dualpol_synrgb_f32(fb[0].data_ptr(), fb[1].data_ptr(), side, side, side, St.Default, Mode.Default, rgb.data_ptr(), side))
    c2.close(); print(f"{tag}: {ms:.4f} ms", flush=True)
measure("start")
big = [torch.empty((20000, 20032), dtype=torch.int16, device="cuda") for _ in range(2)]
measure("1.6 GB of tensors allocated (untouched)")
big[0].zero_(); big[1].zero_(); torch.cuda.synchronize(); measure("... and written by torch")
a = S.Context(0); measure("one idle context")
q = synth.q_tables()
small = torch.empty((2048, 2048), dtype=torch.int16, device="cuda")
a.dev_synth_scene_u16(1, 0, q, 2048, 2048, 0, 2048, small.data_ptr(), 2048); torch.cuda.synchronize(); measure("it generated a 4 MP scene")
a.dev_synth_scene_u16(1, 0, q, 20000, 20000, 0, 20000, big[0].data_ptr(), 20032); torch.cuda.synchronize(); measure("it generated a 400 MP band")
b = S.Context(0, timing=True); measure("+ one idle timing context")
a.close(); measure("first context closed"); b.close(); measure("both closed")
del big; torch.cuda.empty_cache(); measure("tensors freed")
