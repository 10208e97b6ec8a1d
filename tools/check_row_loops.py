#!/usr/bin/env python3
"""Guard of the fused CLAHE -> RGB pass's row loops (kernels.hip 6a): the pass keeps the next row of both bands in
flight and waits for it with a COUNTED s_waitcnt at the loop's bottom.  Two things have silently turned that wait into a wait for the
row's own stores, 5 % of the pass each time: a store on one path of a divergent branch inside the loop (the compiler must then wait
with the smaller path's count), and a register spilled across the loop (a scratch reload waits with vmcnt(0)).  This tool compiles
kernels.hip to assembly, finds the innermost loops of k_clahe_rgb_fused / k_clahe_rgb_fused_rescaled that hold the pass's two 128-bit
row stores, and FAILS when such a loop holds a scratch access, an s_waitcnt vmcnt(0), or a global / buffer store other than its three
(two chunks + the edge bytes) on its straight-line path.

    python tools/check_row_loops.py            (compiles sarpro_amd/csrc/kernels.hip)
    python tools/check_row_loops.py --asm f.s  (checks a listing as it is)
Exit status 1 with the offending lines; 0 and a one-line summary otherwise.  __graft_entry__.build() runs it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "--offload-device-only", "-S"]
KERNELS = ("17k_clahe_rgb_fusedENS", "26k_clahe_rgb_fused_rescaledENS")


def loops_of(body):
    """innermost (label, backward branch) spans of a function body"""
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    spans = []
    for i, l in enumerate(body):
        m = re.match(r"\s*s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            spans.append((labels[m.group(1)], i))
    return [s for s in spans if not any(o != s and s[0] <= o[0] and o[1] <= s[1] for o in spans)]


def check(asm):
    bad, seen = [], 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n", asm, re.M):
        if not any(k in m.group(1) for k in KERNELS):
            continue
        body = asm[m.end():asm.find(".Lfunc_end", m.end())].split("\n")
        for lo, hi in loops_of(body):
            span = body[lo:hi + 1]
            if sum(1 for l in span if re.match(r"\s*buffer_store_dwordx4 .*s\d+ offen", l)) < 2:
                continue
            seen += 1
            for i, l in enumerate(span):
                t = l.strip()
                if t.startswith("scratch_") or re.match(r"s_waitcnt vmcnt\(0\)", t):
                    bad.append(f"{m.group(1)[-48:]} loop at +{lo}: line +{lo + i}: {t}")
            stores = [l.strip().split()[0] for l in span if re.match(r"\s*(buffer_store|global_store)", l)]
            if sorted(stores) != ["buffer_store_byte", "buffer_store_dwordx4", "buffer_store_dwordx4"]:
                bad.append(f"{m.group(1)[-48:]} loop at +{lo}: stores on the straight-line path: {stores}")
    return bad, seen


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--asm":
        asm = open(sys.argv[2]).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.run([HIPCC] + FLAGS + [os.path.join(ROOT, "sarpro_amd", "csrc", "kernels.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
            asm = open(out).read()
    bad, seen = check(asm)
    if seen < 8:
        bad.append(f"expected the 2 x 4 row loops of the fused pass, found {seen}: the tool no longer recognises them")
    if bad:
        print("check_row_loops: FAILED")
        for b in bad:
            print("  " + b)
        return 1
    print(f"check_row_loops: {seen} row loops of the fused pass, none with a scratch access, vmcnt(0) or a store off the counted path")
    return 0


if __name__ == "__main__":
    sys.exit(main())
