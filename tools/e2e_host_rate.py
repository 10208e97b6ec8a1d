"""PCIe-inclusive rate of the host entry point (host u16 bands in, RGB out), pinned vs pageable buffers.
Never the bench `value` (that is HBM-resident); reported in NOTEBOOK.md section 6 (and the state table of DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sarpro_amd as S
from sarpro_amd import synth

rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ctx = S.Context(0)
q = synth.q_tables()
pitch = (cols + 63) // 64 * 64
dev = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
for b in range(2):
    ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, dev[b].data_ptr(), pitch)
for pinned in (True, False):
    host = [torch.empty((rows, cols), dtype=torch.int16, pin_memory=pinned) for _ in range(2)]
    for b in range(2):
        host[b].copy_(dev[b][:, :cols])
    rgb = torch.empty((rows, cols, 3), dtype=torch.uint8, pin_memory=pinned)
    b1, b2 = (h.numpy().view(np.uint16) for h in host)
    out = rgb.numpy()
    import ctypes as C
    from sarpro_amd._lib import lib
    def run():
        rc = lib.sarpro_hip_dualpol_synrgb_u16(ctx._h, b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), rows, cols, 4, 0,
                                               out.ctypes.data_as(C.c_void_p), None, None, None)
        assert rc == 0
    run()
    t = time.perf_counter(); n = 3
    for _ in range(n): run()
    dt = (time.perf_counter() - t) / n
    print(f"{'pinned' if pinned else 'pageable'} host buffers: {dt*1e3:.1f} ms per scene = {rows*cols/dt/1e6:.0f} Mpix/s "
          f"({(2*rows*cols*2 + rows*cols*3)/dt/1e9:.1f} GB/s over PCIe)")

# BASELINE config 2: 400 MP dual-pol -> Robust autoscale -> Lanczos3 to 2048^2 -> pad -> synRGB (host bands in, small RGB out)
host = [torch.empty((rows, cols), dtype=torch.int16, pin_memory=True) for _ in range(2)]
for b in range(2):
    host[b].copy_(dev[b][:, :cols])
b1, b2 = (h.numpy().view(np.uint16) for h in host)
for strat, name in ((1, "Robust"), (4, "Clahe")):
    ctx.dualpol_synrgb_resized(b1, b2, S.AutoscaleStrategy(strat), 2048, True)
    t = time.perf_counter(); n = 3
    for _ in range(n):
        rgb2, m = ctx.dualpol_synrgb_resized(b1, b2, S.AutoscaleStrategy(strat), 2048, True)
    dt = (time.perf_counter() - t) / n
    print(f"config-2 flow ({name}, -> {m.final_cols}x{m.final_rows} padded synRGB): {dt*1e3:.1f} ms per scene = {rows*cols/dt/1e6:.0f} Mpix/s; kernels:",
          {k: round(v, 3) for k, v in ctx.last_kernel_times()})
