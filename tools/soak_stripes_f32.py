#!/usr/bin/env python3
"""Soak of the f32 row-stripe protocol (sarpro_hip_stripe_*_f32): random shapes, random splits (incl. empty and one-row
stripes), f32 bands and on-the-fly pol-ops, every strategy and depth; the concatenated stripes must be the oracle's raster.
usage: python tools/soak_stripes_f32.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op
from test_gpu_stripes_f32 import run_striped

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = calls = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(88000 + seed)
    rows, cols = int(rng.integers(40, 300)), int(rng.integers(40, 300))
    k = int(rng.integers(1, 6))
    cuts = sorted(int(c) for c in rng.integers(0, rows + 1, k - 1))
    edges = [0] + cuts + [rows]
    splits = [(edges[i], edges[i + 1] - edges[i]) for i in range(k)]
    pitch = (cols + 3) // 4 * 4 if seed % 3 else cols + 1  # vector and scalar kernels
    polop = seed % 2 == 1
    if polop:
        a = rng.integers(0, 5000, (rows, cols)).astype(np.uint16); b = rng.integers(0, 1200, (rows, cols)).astype(np.uint16)
        op = Op(int(rng.integers(0, 5)))
        x = oracle.polop(int(op), a.astype(np.float32), b.astype(np.float32))
        bands = [a, b] if seed % 4 == 1 else [a.astype(np.float32), b.astype(np.float32)]
    else:
        x = np.exp(rng.normal(rng.uniform(-2, 4), rng.uniform(0.3, 2.5), (rows, cols))).astype(np.float32)
        x[rng.random((rows, cols)) < 0.05] = 0.0
        bands, op = [x], None
    for strategy in St:
        if strategy == St.Clahe and oracle.pipeline(x, 0, int(strategy))[0] != 0:
            continue
        for bd in Bd:
            rc, ref = oracle.pipeline(x, int(bd), int(strategy))
            if rc != 0:
                continue
            got, _ = run_striped(bands, op, rows, cols, strategy, bd, splits, pitch)
            calls += 1
            if not np.array_equal(got, ref):
                # Adaptive reads mean / std, whose f64 sums follow the partition: a flipped discrete test shows as a different window
                bad += 1
                print("MISMATCH seed", seed, rows, cols, splits, strategy.name, bd.name, "polop" if polop else "band", int((got != ref).sum()), "px", flush=True)
    if (seed - first) % 10 == 9:
        print(f"{seed - first + 1} cases, {calls} striped scenes, {time.time() - t0:.0f} s, mismatches {bad}", flush=True)
print(f"cases {n}: {calls} striped scenes, mismatches {bad}")
sys.exit(1 if bad else 0)
