#!/usr/bin/env python3
"""Soak of the f32 zone route and the queued u16 levels (f32_path.cpp) over random rasters: shapes 64..420, log-normal /
quantised / tie-heavy / mostly-invalid distributions, every strategy and depth, f32 bands and pol-ops of u16 bands; the raster
must be the oracle's whether the route answered or stepped aside.  usage: python tools/soak_zones.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["SARPRO_HIP_F32_DIRECT"] = "0"  # rasters this small take the direct route otherwise: the zone route is what is soaked here
import numpy as np
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = answered = aside = 0
t0 = time.time()
with S.Context(0, timing=True) as c:
    for seed in range(first, first + n):
        rng = np.random.default_rng(77000 + seed)
        rows, cols = int(rng.integers(64, 420)), int(rng.integers(64, 420))
        kind = seed % 5
        if kind == 0:
            x = np.exp(rng.normal(rng.uniform(-3, 6), rng.uniform(0.2, 3.0), (rows, cols))).astype(np.float32)
        elif kind == 1:   # quantised: many ties, thresholds land ON sample values
            x = (np.round(np.exp(rng.normal(1.0, 1.0, (rows, cols))) * 8) / 8).astype(np.float32)
        elif kind == 2:   # mostly invalid
            x = np.where(rng.random((rows, cols)) < 0.9, 0.0, rng.gamma(2.0, 5.0, (rows, cols))).astype(np.float32)
        elif kind == 3:   # two populations far apart (wide span: bins wider than the sample's buckets)
            x = np.where(rng.random((rows, cols)) < 0.5, rng.uniform(1e-4, 2e-4, (rows, cols)), rng.uniform(1e3, 1e4, (rows, cols))).astype(np.float32)
        else:             # pol-op of two u16 bands
            x = None
        if x is not None:
            x[rng.random((rows, cols)) < 0.03] = 0.0
            x[rng.random((rows, cols)) < 0.01] = np.nan
        for strategy in St:
            for bd in Bd:
                if x is not None:
                    rc, ref = oracle.pipeline(x, int(bd), int(strategy))
                    got = c.process_scalar_data_pipeline(x, bd, strategy)
                else:
                    a = rng.integers(0, 4000, (rows, cols)).astype(np.uint16); b = rng.integers(0, 900, (rows, cols)).astype(np.uint16)
                    op = Op(int(rng.integers(0, 5)))
                    rc, ref = oracle.pipeline(oracle.polop(int(op), a.astype(np.float32), b.astype(np.float32)), int(bd), int(strategy))
                    got = c.polop_autoscale_band(op, a, b, bd, strategy)
                out = got[0] if bd == Bd.U8 else got[1]
                names = [k for k, _ in c.last_kernel_times()]
                answered += "f32_zone_count" in names and "f32_hist4096" not in names
                aside += "f32_hist4096" in names
                if rc != 0 or not np.array_equal(out, ref):
                    bad += 1
                    print("MISMATCH seed", seed, "kind", kind, rows, cols, strategy.name, bd.name, flush=True)
        if (seed - first) % 10 == 9:
            print(f"{seed - first + 1} cases, {time.time() - t0:.0f} s, answered {answered}, stepped aside {aside}, mismatches {bad}", flush=True)
print(f"cases {n}: zone route answered {answered} calls, stepped aside (or not applicable) {aside}, mismatches {bad}")
sys.exit(1 if bad else 0)
