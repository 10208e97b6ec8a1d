#!/usr/bin/env python3
"""Lane-overlap timeline from a `rocprofv3 --kernel-trace --output-format csv` run of tools/pipe_trace.py: the kernels of the LAST
batch, per queue (= lane stream), with start / end in microseconds from the batch's first kernel, and -- for the two sweeps (the DN
histogram pass and the fused CLAHE -> RGB pass) -- how much of each launch ran while a sweep of ANOTHER lane was running.
usage: trace_overlap.py <dir with *_kernel_trace.csv> [kernels of the last batch = scenes * per-scene count, default: last 90]"""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?")), short(r["Kernel_Name"])) for r in rows]
ev.sort()
nscenes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
hists = [e for e in ev if e[3] == "k_dn_hist_pieces"]
t_first = min(e[0] for e in hists[-nscenes:])  # the last batch = everything from the earliest of its histogram passes on
start_i = next(i for i, e in enumerate(ev) if e[0] >= t_first)
batch = ev[start_i:]
t0 = batch[0][0]
queues = sorted({e[2] for e in batch})
print(f"# {os.path.basename(f)}: {len(batch)} kernels of the last batch on {len(queues)} queues (lanes); times in us from the batch's first kernel")
sweeps = [e for e in batch if e[3] in ("k_dn_hist_pieces", "k_clahe_rgb_fused")]
for e in batch:
    if e[1] - e[0] < int(os.environ.get("MIN_US", "15")) * 1000 and e[3] not in ("k_chain_predict",):
        continue  # (only kernels of 15 us and more, and the prediction: the rest are listed in the count below)
    ov = 0
    if e in sweeps:
        for o in sweeps:
            if o[2] != e[2]:
                ov += max(0, min(e[1], o[1]) - max(e[0], o[0]))
    print(f"lane {queues.index(e[2])}  {e[3]:24s} {(e[0] - t0) / 1e3:9.1f} -> {(e[1] - t0) / 1e3:9.1f}  ({(e[1] - e[0]) / 1e3:7.1f} us)" + (f"  beside another lane's sweep: {ov / 1e3:6.1f} us" if e in sweeps else ""))
span = (batch[-1][1] - t0) / 1e3
busy = sum(e[1] - e[0] for e in sweeps) / 1e3
small = [e for e in batch if e not in sweeps]
hidden = 0
for e in small:
    for o in sweeps:
        if o[2] != e[2]:
            hidden += max(0, min(e[1], o[1]) - max(e[0], o[0]))
print(f"# batch span {span:.1f} us for {nscenes} scenes = {span / nscenes:.1f} us per scene; the sweeps' own durations add up to {busy:.1f} us; "
      f"{len(small)} short kernels, {sum(e[1] - e[0] for e in small) / 1e3:.1f} us in total, {hidden / 1e3:.1f} us of it beside another lane's sweep")
