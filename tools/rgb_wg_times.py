#!/usr/bin/env python3
"""When each persistent workgroup of the fused CLAHE -> RGB pass starts and ends (instrumented build: tools/build_variant.sh wgtimes
"-DSARPRO_RGB_WG_TIMES"; SARPRO_HIP_LIB=lib_wgtimes.so): the tail of the pass = the time between the mean and the last workgroup's end."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
from sarpro_amd._lib import lib
rows = cols = int(os.environ.get("SIDE", "20000")); pitch = (cols + 63) // 64 * 64
q = synth.q_tables()
with S.Context(0, timing=True) as c:
    for k, v in [x.split("=") for x in os.environ.get("ATTRS", "").split(",") if x]:
        c.set_attr(k, int(v))
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for k in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    which = os.environ.get("KERNEL", "clahe_rgb_fused")  # or dn_hist_u16 (the piece histogram: build with -DSARPRO_PIECE_WG_TIMES, file piece_kernels.hip), or clahe_apply_u16 (config 3's exact kernel)
    for it in range(4):
        if which == "clahe_apply_u16":  # per workgroup: duration against rows in extrapolating cells / rows of items with a straddling lane / rows / items / table builds
            c.dev_autoscale_band_u16(d[0].data_ptr(), rows, cols, pitch, St.Clahe, S.BitDepth.U16, d[1].data_ptr(), pitch)
            t = np.zeros((1024, 8), np.uint64)
            assert lib.sarpro_hip_debug_rgb_wg_times(t.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
            kt = dict(c.last_kernel_times())
            t = t[:256].astype(np.float64); t0 = t[:, 0].min()
            st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0
            X = np.stack([t[:, 4], t[:, 2], t[:, 3], t[:, 5], t[:, 6]], axis=1)  # rows, edge rows, straddle rows, items, builds
            coef, *_ = np.linalg.lstsq(X, en - st, rcond=None)
            res = (en - st) - X @ coef
            print(json.dumps({"kernel_ms": round(kt.get("clahe_apply_u16", 0), 4), "last_end_us": round(float(en.max()), 1), "mean_end_us": round(float(en.mean()), 1), "first_end_us": round(float(en.min()), 1),
                              "us_per": dict(zip(["row", "edge_row_extra", "straddle_row_extra", "item", "table_build"], [round(float(x), 4) for x in coef])),
                              "residual_us_rms": round(float(np.sqrt((res ** 2).mean())), 1), "means": [round(float(x), 1) for x in X.mean(axis=0)]}), flush=True)
            continue
        c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        which = os.environ.get("KERNEL", "clahe_rgb_fused")  # or dn_hist_u16 (the piece histogram: build with -DSARPRO_PIECE_WG_TIMES, file piece_kernels.hip)
        fn = lib.sarpro_hip_debug_rgb_wg_times if which == "clahe_rgb_fused" else lib.sarpro_hip_debug_piece_wg_times
        t = np.zeros((1024, 8 if which == "clahe_rgb_fused" else 2), np.uint64)
        assert fn(t.ctypes.data_as(C.POINTER(C.c_ulonglong))) == 0
        kt = dict(c.last_kernel_times())
        t = t[:256].astype(np.int64)
        if which == "clahe_rgb_fused":  # thread 0's view of its workgroup: time at the item barrier, in the prologues, in the rows (100 MHz ticks -> us)
            print(json.dumps({"items_per_wg_mean": round(float(t[:, 5].mean()), 2), "wait_us_mean": round(float(t[:, 2].mean()) / 100, 1), "prologue_us_mean": round(float(t[:, 3].mean()) / 100, 1),
                              "rows_us_mean": round(float(t[:, 4].mean()) / 100, 1), "prologue_us_per_item": round(float(t[:, 3].sum()) / max(float(t[:, 5].sum()), 1) / 100, 2),
                              "wait_us_per_item": round(float(t[:, 2].sum()) / max(float(t[:, 5].sum()), 1) / 100, 2)}), flush=True)
        t0 = t[:, 0].min()
        st, en = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0  # us
        busy = en - st
        print(json.dumps({"kernel_ms": round(kt.get(which, 0), 4), "last_end_us": round(float(en.max()), 1), "mean_end_us": round(float(en.mean()), 1),
                          "first_end_us": round(float(en.min()), 1), "p10_end_us": round(float(np.percentile(en, 10)), 1), "p90_end_us": round(float(np.percentile(en, 90)), 1),
                          "latest_start_us": round(float(st.max()), 1), "mean_busy_us": round(float(busy.mean()), 1), "max_busy_us": round(float(busy.max()), 1)}), flush=True)
