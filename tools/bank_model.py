"""LDS bank model of the fused pass's compose lookups on the bench scenes (CPU only: the oracle's level rasters of a 2048^2 version of
each scene): a byte read is served in two groups of 32 lanes over 32 four-byte banks, identical dwords broadcast, every further distinct
dword on a bank adds a cycle.  Prints the mean cycles per group for R2[level1], G2[level2] and B2[level1][level2] with 256- and with
260-byte rows (the quantised-VH scene: 11.4 -> 3.3; the others 3.5 either way).  usage: python tools/bank_model.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from sarpro_amd import synth
def levels(scene_idx, rows=2048, cols=2048):
    name, off, flags, qkw, what = synth.BENCH_SCENES[scene_idx]
    q = synth.q_tables(**qkw) if qkw else synth.q_tables()
    out=[]
    for b in range(2):
        dn = synth.scene_u16(rows, cols, b, seed=synth.SEED_SCENE_A+off, q=q, flags=flags)
        rc, lv = oracle.pipeline(dn.astype(np.float32), 0, 4)
        out.append(lv)
    return name, out
def conflict(addr_dword):  # addr_dword: int array [..., 32] of dword addresses per lane group; cycles = max over banks of #distinct dwords on that bank
    n = addr_dword.shape[0]
    tot = 0
    for g in range(n):
        a = np.unique(addr_dword[g])
        tot += np.bincount(a % 32, minlength=32).max()
    return tot / n
for idx in (0,1,5,7):
    name,(l1,l2)=levels(idx)
    rows,cols=l1.shape
    # lane l of a wave handles pixels 8l..8l+7 of a 512-px strip; instruction j reads pixel j of each lane; byte reads: groups of 32 lanes
    strips = cols//512
    res={'R2':[], 'G2':[], 'B2_256':[], 'B2_260':[]}
    rng=np.random.default_rng(0)
    for r in rng.integers(0, rows, 200):
        s = int(rng.integers(0, strips)); j = int(rng.integers(0, 8)); half = int(rng.integers(0,2))
        px = s*512 + (np.arange(32)+32*half)*8 + j
        v1 = l1[r, px].astype(np.int64); v2 = l2[r, px].astype(np.int64)
        res['R2'].append(v1//4); res['G2'].append(64+v2//4); res['B2_256'].append((v1*256+v2)//4); res['B2_260'].append((v1*260+v2)//4)
    print(name, {k: round(conflict(np.array(v)),2) for k,v in res.items()}, 'distinct levels VV/VH:', len(np.unique(l1)), len(np.unique(l2)))
