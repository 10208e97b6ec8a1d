#!/usr/bin/env python3
"""A/B timing of library builds on the config-2 resident flow (resize kernels): python tools/time_resize_variants.py lib ... ('-' = default)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    env = dict(os.environ)
    if spec != "-":
        env["SARPRO_HIP_LIB"] = os.path.join(ROOT, "sarpro_amd", spec)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "time_resize_flow.py")], env=env, capture_output=True, text=True)
    lines = [l for l in out.stdout.splitlines() if "resize_h" in l]
    print(spec, " | ".join(l.split("ms/scene")[1][:200] for l in lines[:2]) if lines else out.stderr[-300:], flush=True)
