#!/usr/bin/env python3
"""A/B timing of library builds inside ONE process-per-build on the same box: per-kernel times of the headline chain.
usage: python tools/time_variants.py SPEC ...   SPEC = lib[:ENV=VAL[,ENV=VAL...]], lib relative to sarpro_amd/ ('-' = the default build)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    acc = {}
    N = int(os.environ.get('N', '8'))
    for it in range(N + 2):
        c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        if it >= 2:
            for n, ms in c.last_kernel_times():
                acc.setdefault(n, []).append(ms)
    print(json.dumps({n: round(sorted(v)[len(v) // 2], 4) for n, v in acc.items()} | ({'fused_min': round(min(acc['clahe_rgb_fused']), 4), 'fused_mean': round(sum(acc['clahe_rgb_fused']) / len(acc['clahe_rgb_fused']), 4)} if 'clahe_rgb_fused' in acc else {})))
''' % ROOT
for spec in sys.argv[1:]:
    lib, _, envs = spec.partition(":")
    env = dict(os.environ)
    for kv in filter(None, envs.split(",")):
        k, _, v = kv.partition("=")
        env[k] = v
    if lib != "-":
        env["SARPRO_HIP_LIB"] = os.path.join(ROOT, "sarpro_amd", lib)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    print(spec, line[-1] if line else ("FAILED: " + out.stderr[-400:]), flush=True)
