"""Device-resident timings of the BASELINE configurations other than the headline (bench.py times that one):
config 1 (2048^2 f32, Standard, U8), config 2's hot path (400 MP dual-pol Robust u8 -> synRGB), config 3
(400 MP CLAHE U16 per band; log-ratio pol-op -> f32 -> CLAHE U16).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import synth
from sarpro_amd.types import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, SyntheticRgbMode as Mode

rows = cols = int(os.environ.get("SARPRO_CFG_SIZE", "20000"))
pitch = (cols + 31) // 32 * 32
ctx = S.Context(0, timing=True)
q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
for b in range(2):
    ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
torch.cuda.synchronize()


def timed(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


out = {}
# config 2 hot path: Robust, u8, both bands -> default synRGB at full resolution
rgb = torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda")
out["cfg2_robust_dualpol_synrgb_ms"] = timed(lambda: ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, Mode.Default, rgb.data_ptr(), pitch))
del rgb
# config 3 (i): CLAHE, U16 output, per band
o16 = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
out["cfg3_clahe_u16_per_band_ms"] = timed(lambda: ctx.dev_autoscale_band_u16(band[0].data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, o16.data_ptr(), pitch))
out["cfg3_clahe_u16_kernels"] = {k: round(v, 3) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}
# config 3 (ii): log-ratio pol-op on the f32 bands, then CLAHE U16 of the f32 result
f = [b[:, :cols].to(torch.float32).contiguous() for b in band]  # uint16 bit pattern viewed as int16: fix the sign
for x in f:
    x[x < 0] += 65536.0
ratio = torch.empty((rows, cols), dtype=torch.float32, device="cuda")
out["cfg3_logratio_polop_ms"] = timed(lambda: ctx.dev_polop_f32(Op.LogRatio, f[0].data_ptr(), f[1].data_ptr(), rows * cols, ratio.data_ptr()))
del f
out["cfg3_ratio_f32_clahe_u16_with_stats_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Clahe, Bd.U16, o16.data_ptr(), pitch), n=3, warm=1)
out["cfg3_ratio_f32_clahe_u16_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Clahe, Bd.U16, o16.data_ptr(), pitch, want_stats=False), n=3, warm=1)
out["cfg3_ratio_f32_kernels"] = {k: round(v, 3) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}
o8f = torch.empty((rows, pitch), dtype=torch.uint8, device="cuda")
out["f32_robust_u8_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Robust, Bd.U8, o8f.data_ptr(), pitch, want_stats=False), n=3, warm=1)
out["f32_robust_u8_kernels"] = {k: round(v, 3) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}
out["f32_standard_u8_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Standard, Bd.U8, o8f.data_ptr(), pitch, want_stats=False), n=3, warm=1)
out["f32_standard_u8_kernels"] = {k: round(v, 3) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}
out["f32_robust_u16_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Robust, Bd.U16, o16.data_ptr(), pitch, want_stats=False), n=3, warm=1)
out["f32_robust_u16_kernels"] = {k: round(v, 3) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}
del o8f
del ratio, o16
# config 1: 2048 x 2048 f32, Standard, U8
n1 = 2048
x = band[0][:n1, :n1].to(torch.float32).contiguous()
x[x < 0] += 65536.0
o8 = torch.empty((n1, n1), dtype=torch.uint8, device="cuda")
out["cfg1_2048_f32_standard_u8_ms"] = timed(lambda: ctx.dev_autoscale_band_f32(x.data_ptr(), n1, n1, n1, St.Standard, Bd.U8, o8.data_ptr(), n1), n=20)
out = {k: (round(v, 3) if isinstance(v, float) else v) for k, v in out.items()}
out["size"] = [rows, cols]
print(json.dumps(out))
