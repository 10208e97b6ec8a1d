"""The N > 1 paths that a one-GPU box can execute, kept warm for the day a multi-GPU node runs them:
the batch driver with FOUR workers (the same device listed four times: four threads, four contexts, scenes dealt dynamically,
streamed from files) with and without the NUMA binding, and `bench.py --gpus 2` on a node with one GPU: it must fail from the
child, with the child's message and a non-zero exit code -- no hang, no re-exec of a process that has touched the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("no_numa", [False, True])
def test_batch_of_16_streamed_scenes_on_four_workers(tmp_path, no_numa, monkeypatch):
    if no_numa:
        monkeypatch.setenv("SARPRO_HIP_BATCH_NO_NUMA", "1")
    else:
        monkeypatch.delenv("SARPRO_HIP_BATCH_NO_NUMA", raising=False)
    shapes = [(300 + 7 * i, 420 + 13 * (i % 5)) for i in range(16)]
    scenes, keep, want = [], [], []
    for i, (rows, cols) in enumerate(shapes):
        b = [synth.scene_u16(rows, cols, k, seed=synth.SEED_SCENE_A + i) for k in (0, 1)]
        readers = []
        for k in (0, 1):
            p = str(tmp_path / f"s{i}_b{k}.tif")
            w = S.TiffWriter(p, cols, rows, 1, 16)
            w.write_rows(0, b[k])
            w.finish()
            readers.append(S.TiffReader(p))
        pair = S.TiffPair(*readers)
        keep.append(pair)
        scenes.append((pair.reader(), rows, cols))
        u8 = [oracle.resize_image_data_with_meta(oracle.pipeline(x.astype(np.float32), 0, int(St.Default))[1], 128, True)[0] for x in b]
        want.append(oracle.synrgb(0, int(St.Default), u8[0], u8[1]))
    got, rep, st, rc = S.batch_dualpol_synrgb_resized([0, 0, 0, 0], scenes, St.Default, 128, True)
    assert rc == 0 and rep.processed == 16 and rep.errors == 0 and rep.skipped == 0 and st == [0] * 16
    for i, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(a, b), i


def test_bench_with_more_ranks_than_gpus_fails_from_the_child_without_hanging():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--mode", "stripe", "--steps", "1", "--warmup", "1",
                        "--rows", "2000", "--cols", "2048", "--no-secondary", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode != 0
    assert f"needs GPU {n - 1}" in p.stderr, p.stderr[-1500:]       # the child's own message reaches the caller
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]  # and no record is printed for a run that did not happen
