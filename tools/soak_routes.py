"""Full-size cross-check of the fast routes against their slow twins over several scenes (different seeds):
every strategy, dual-pol -> RGB, device chain + fused pass vs SARPRO_HIP_NO_CHAIN=1 (host-orchestrated phases) and
vs SARPRO_HIP_NO_FUSED=1 (table pass + compose), rasters compared byte for byte on the device.
usage: python tools/soak_routes.py [n_scenes] [rows] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sw
import torch
import sarpro_amd as S
from sarpro_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
pitch = (cols + 63) // 64 * 64
ctx = S.Context(0); q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
rgb = [torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda") for _ in range(2)]
SWITCHES = ("SARPRO_HIP_NO_CHAIN", "SARPRO_HIP_NO_FUSED", "SARPRO_HIP_NO_LINEAR_HIST", "SARPRO_HIP_FULL_LEVEL_HIST", "SARPRO_HIP_NO_FUSED_RGB", "SARPRO_HIP_NO_SAMPLED_HIST")
bad = 0; t0 = time.time()
for k in range(n):
    for b in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + 2000 + k, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    for strategy in S.AutoscaleStrategy:
        for name in SWITCHES: sw.pop(name)
        ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, strategy, S.SyntheticRgbMode.Default, rgb[0].data_ptr(), pitch)
        for name in SWITCHES:
            sw.set(name, "1")
            rgb[1].zero_()
            ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, strategy, S.SyntheticRgbMode.Default, rgb[1].data_ptr(), pitch)
            sw.pop(name)
            diff = int((rgb[0].view(rows, pitch, 3)[:, :cols] != rgb[1].view(rows, pitch, 3)[:, :cols]).sum().item())
            if diff:
                bad += 1
                print(f"scene {k} {strategy.name} {name}: {diff} bytes differ", flush=True)
    print(f"scene {k}: done, {time.time() - t0:.0f} s", flush=True)
print(f"{n} scenes x {len(list(S.AutoscaleStrategy))} strategies x {len(SWITCHES)} switches at {rows}x{cols}: {bad} differences")
sys.exit(1 if bad else 0)
