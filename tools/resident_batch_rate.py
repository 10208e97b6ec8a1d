"""Throughput of HBM-resident scenes processed by 1, 2, 3 contexts (threads) at once on one GPU: kernels of different scenes
overlap (the HBM-bound compose of one under the LDS-bound apply of another)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import synth
from sarpro_amd.types import AutoscaleStrategy as St, SyntheticRgbMode as Mode

rows = cols = 20000
pitch = 20032
q = synth.q_tables()
strategy = {s.name.lower(): s for s in St}[(sys.argv[1] if len(sys.argv) > 1 else "clahe").lower()]
NMAX = 3
ctxs = [S.Context(0) for _ in range(NMAX)]
bands, rgbs = [], []
for i in range(NMAX):
    b = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        ctxs[i].dev_synth_scene_u16(synth.SEED_SCENE_A + i, k, q, rows, cols, 0, rows, b[k].data_ptr(), pitch)
    bands.append(b)
    rgbs.append(torch.empty((rows, pitch * 3), dtype=torch.uint8, device="cuda"))
torch.cuda.synchronize()


def worker(i, n):
    for _ in range(n):
        ctxs[i].dev_dualpol_synrgb_u16(bands[i][0].data_ptr(), bands[i][1].data_ptr(), rows, cols, pitch, strategy, Mode.Default,
                                       rgbs[i].data_ptr(), pitch)


for nw in (1, 2, 3):
    for rep in range(2):
        ths = [threading.Thread(target=worker, args=(i, 3 if rep == 0 else 20)) for i in range(nw)]
        t = time.perf_counter()
        [x.start() for x in ths]
        [x.join() for x in ths]
        dt = time.perf_counter() - t
    scenes = nw * 20
    print(f"{strategy.name}: {nw} scene(s) in flight: {dt / scenes * 1e3:.3f} ms per scene = {rows * cols * scenes / dt / 1e9:.1f} Gpix/s")
