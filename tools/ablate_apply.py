"""Scratch: time the CLAHE apply kernel of the headline pass under SARPRO_HIP_* environment variants (each in a child process)."""
import os, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
variants = [v.split(",") if v else [] for v in sys.argv[1:]] or [[]]
for var in variants:
    env = dict(os.environ)
    for kv in var:
        k, v = kv.split("=")
        env[k] = v
    out = subprocess.run([sys.executable, os.path.join(HERE, "time_kernels.py")], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("clahe_apply")]
    tot = [l for l in out.splitlines() if l.startswith("TOTAL")]
    print(",".join(var) or "default", "|", " ".join(line[0].split()) if line else out[-300:], "|", " ".join(tot[0].split()) if tot else "")
