"""File-to-file rate of the streaming entry point: two 400 MP u16 band TIFFs (one strip per row, like S1 GRD
measurement rasters) in tmpfs -> sarpro_hip_dualpol_synrgb_stream_u16 with the C strip-TIFF reader / sink of
tiff_io.cpp -> RGB TIFF in tmpfs.  Prints one JSON line (wall time, reader / sink share, Mpix/s)."""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sarpro_amd as S
from sarpro_amd import synth
from sarpro_amd.types import AutoscaleStrategy as St, SyntheticRgbMode as Mode

rows = cols = int(os.environ.get("SARPRO_E2E_SIZE", "20000"))
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
paths = [os.path.join(d, f"b{k}.tif") for k in (0, 1)]
t0 = time.perf_counter()
for k, p in enumerate(paths):  # synthetic scene, written in row blocks so the generator never holds 800 MB twice
    w = S.TiffWriter(p, cols, rows, 1, 16)
    for r0 in range(0, rows, 2000):
        w.write_rows(r0, synth.scene_u16(rows, cols, k, row0=r0, rows_local=min(2000, rows - r0)))
    w.finish()
gen_s = time.perf_counter() - t0
ctx = S.Context(0, timing=True)
res = []
for it in range(3):
    ra, rb = S.TiffReader(paths[0]), S.TiffReader(paths[1])
    out = os.path.join(d, "rgb.tif")
    w = S.TiffWriter(out, cols, rows, 3, 8)
    pair = S.TiffPair(ra, rb)
    t = time.perf_counter()
    ctx.dualpol_synrgb_stream(pair.reader(), rows, cols, St.Clahe, Mode.Default, w.sink())
    w.finish()
    dt = time.perf_counter() - t
    times = {}
    for n, v in ctx.last_kernel_times():
        times[n] = times.get(n, 0.0) + v
    res.append((dt, times.get("host:reader", 0.0), times.get("host:sink", 0.0), times.get("host:chain(enqueue+final sync)", 0.0)))
    ra.close(); rb.close()
best = min(res)
print(json.dumps({"rows": rows, "cols": cols, "wall_s": round(best[0], 4), "reader_ms": round(best[1], 1), "sink_ms": round(best[2], 1),
                  "chain_ms": round(best[3], 2), "Mpix_per_s": round(rows * cols / best[0] / 1e6, 1), "all_runs_s": [round(r[0], 3) for r in res],
                  "scene_generation_s": round(gen_s, 1)}))
for p in paths + [os.path.join(d, "rgb.tif")]:
    os.remove(p)
os.rmdir(d)
