#!/usr/bin/env python3
"""Soak of the CLAHE chain's routes on rasters the synthetic scenes do not cover: N random rasters (random shape, random kind: natural
scene with / without its no-data wedge, DNs coarsened to a few values, a handful of values mixed by a spatial gradient, a crop lifted
off zero, mostly-invalid with valid patches, a band swapped for a constant or for zeros), each through
  * the product's default route (the fused CLAHE -> RGB pass, identity or predicted rescale),
  * apply + compose (NO_FUSED_RGB), the partial histogram of every row (NO_SAMPLED_HIST), a floor forced wrong (SPEC_FORCE=mispredict:
    the gated fallback), no predicted rescale (NO_SPEC_RESCALE),
  * every pixel through the exact f64 blend (NO_SPEC) -- the route the oracle tests pin --
and the RGB rasters compared byte for byte on the device.  Round 5's lost-LDS-adds bug was of the kind this
finds: a fallback route that is only wrong on rasters without level 0, and only from a few MP up.
usage: python tools/soak_random_rasters.py [n_rasters] [max_megapixels] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
max_mp = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
q = synth.q_tables()
KINDS = ["natural", "no_wedge", "coarse", "few_gradient", "lifted_crop", "patches", "const_band", "zero_band", "bright"]
ROUTES = [None, "NO_FUSED_RGB", "NO_SAMPLED_HIST", "SPEC_FORCE", "NO_SPEC_RESCALE", "NO_SPEC"]


def make(ctx, kind, rows, cols, pitch, seed):
    band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    flags = 1 if kind in ("no_wedge", "lifted_crop") else 0  # synth: bit 0 = no no-data wedge
    for b in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + 5000 + seed, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch, flags)
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    for b in range(2):
        t = band[b].to(torch.int32) & 0xFFFF
        if kind == "coarse":
            sh = int(rng.integers(6, 10))
            t = ((t >> sh) << sh) + int(rng.integers(1, 200))
        elif kind == "few_gradient":
            m = int(rng.integers(2, 9))
            vals = torch.tensor(np.sort(rng.integers(20, 6000, m)).astype(np.int32), device="cuda")
            w = torch.linspace(0, 1, cols, device="cuda")[None, :] * (m - 1)
            idx = (w + torch.rand((rows, cols), device="cuda", generator=g) * 1.5 - 0.75).round().clamp_(0, m - 1).long()
            t = torch.zeros((rows, pitch), dtype=torch.int32, device="cuda")
            t[:, :cols] = vals[idx]
        elif kind == "lifted_crop":
            t = t + int(rng.integers(1, 60))
        elif kind == "patches":
            keep = torch.zeros((rows, pitch), dtype=torch.bool, device="cuda")
            for _ in range(int(rng.integers(1, 5))):
                r0, c0 = int(rng.integers(0, rows - 32)), int(rng.integers(0, cols - 32))
                keep[r0:r0 + int(rng.integers(32, max(33, rows // 2))), c0:c0 + int(rng.integers(32, max(33, cols // 2)))] = True
            t = torch.where(keep, t, torch.zeros_like(t))
        elif kind == "const_band" and b == int(seed) % 2:
            t = torch.full_like(t, int(rng.integers(1, 3000)))
        elif kind == "zero_band" and b == int(seed) % 2:
            t = torch.zeros_like(t)
        elif kind == "bright":
            t = t * int(rng.integers(2, 12)) + int(rng.integers(0, 500))
        t = t.clamp_(0, 65535)
        band[b].copy_(torch.where(t >= 32768, t - 65536, t).to(torch.int16))
    torch.cuda.synchronize()
    return band


bad = 0
t0 = time.time()
with S.Context(0) as c:
    c.set_attr("SAMPLED_HIST_MIN_PX", 0)
    for k in range(n):
        kind = KINDS[k % len(KINDS)]
        mp = float(np.exp(rng.uniform(np.log(0.2), np.log(max_mp))))
        aspect = float(np.exp(rng.uniform(-1.0, 1.0)))
        rows = max(264, int((mp * 1e6 * aspect) ** 0.5)); cols = max(264, int(mp * 1e6 / rows))
        pitch = (cols + 63) // 64 * 64
        band = make(c, kind, rows, cols, pitch, k)
        out, reps = [], []
        for route in ROUTES:
            if route:
                c.set_attr(route, "mispredict" if route == "SPEC_FORCE" else 1)
            rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
            c.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
            try:
                r = c.spec_report(); reps.append((r["spec_ok"], r["outcome"]))
            except Exception:
                reps.append(None)
            if route:
                c.set_attr(route, None)
            out.append(rgb.view(rows, pitch, 3)[:, :cols])
        diffs = [int((o != out[-1]).any(dim=2).sum().item()) for o in out[:-1]]
        bad += any(diffs)
        print(f"raster {k} {kind} {rows}x{cols}: default {reps[0]}  px differing from the exact blend's raster by route {dict(zip([r or 'default' for r in ROUTES[:-1]], diffs))}", flush=True)
        del band, out
print(f"{n} rasters up to {max_mp} MP: {bad} with differences, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
