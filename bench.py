#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mpix/s calibrate+CLAHE+synRGB on a 400 MP dual-pol scene.

One "step" = one pass of the hot path (save.rs:317-367 at native resolution: per-band dB +
CLAHE autoscale to u8, then suppressed synthetic-RGB composition) over one synthetic
20000 x 20000 dual-pol u16 scene that is already resident in HBM.  Output: the interleaved
RGB raster in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode batch|stripe]

N > 1 runs one rank per GPU under torch.distributed.run.  Started WITHOUT that launcher (`python bench.py --gpus N`) the
script starts it itself, as a child process and before anything touches a GPU, and relays rank 0's JSON line and the
child's exit code.  Modes:
  batch  (default, weak scaling): every rank processes its own scene, no collective
         (BASELINE.json config 5 style sharding; pixel data never leaves a GPU)
  stripe (strong scaling): ONE scene split into row stripes; the three small histogram
         reductions of the path are RCCL all-reduces (BASELINE.json config 4)

Prints ONE JSON line on rank 0 (see the task contract), including
  roofline     -- dominant kernel, algorithmic bytes / HIP-event time vs 8 TB/s HBM peak
  cpu_baseline -- the CPU oracle (single thread, = the reference's behaviour) on a bounded
                  sample of the same workload, timed on this box's host cores (N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=["batch", "stripe"], default="batch")
    ap.add_argument("--rows", type=int, default=20000)
    ap.add_argument("--cols", type=int, default=20000)
    ap.add_argument("--strategy", default="clahe")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="one host round trip per scene instead of stream-ordered enqueue")
    ap.add_argument("--cpu-sample", type=int, default=7000, help="side of the square CPU-baseline sample scene (median of 3 runs)")
    ap.add_argument("--no-cpu-full", action="store_true", help="skip the single full-size (metric configuration) run of the CPU baseline (~30 s)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary records (N = 1: end-to-end over PCIe, BASELINE configs 1-3, resize; "
                                                               "N > 1: row-stripe mode over RCCL and the PCIe-inclusive leg on all ranks)")
    ap.add_argument("--secondary-only", action="store_true", help=argparse.SUPPRESS)  # the child process that measures the N = 1 secondary records
    ap.add_argument("--traffic-only", action="store_true", help=argparse.SUPPRESS)    # the child process that measures roofline.traffic under rocprofv3
    ap.add_argument("--no-traffic", action="store_true", help="roofline.traffic from the committed PMC passes (profiles/) instead of two rocprofv3 --pmc passes "
                                                               "started by this run after the timed region (~1 min)")
    ap.add_argument("--pitch-align", type=int, default=64, help="row pitch of the resident rasters, rounded up to this many elements")
    ap.add_argument("--lanes", type=int, default=3, help="internal lanes of the resident batch entry point the timed steps go through (sarpro_hip_batch_dualpol_synrgb_u16_dev: "
                                                         "scene i + 1's histogram chain is enqueued beside scene i's fused pass); 0 = one call per scene on one stream, the loop of rounds 3-4")
    ap.add_argument("--scenes", type=int, default=0, help="how many of sarpro_amd.synth.BENCH_SCENES the timed steps cycle over (0 = all of them; 1 = scene A only, "
                                                          "the workload of rounds 1-3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.secondary_only:
        sys.exit(secondary_child(args))
    if args.traffic_only:
        sys.exit(traffic_child(args))
    # N = 1: the secondary records are measured by a CHILD process, started here -- before this process touches a GPU runtime --
    # and parked on its stdin until the headline is done (see secondary_start)
    sec_child = secondary_start(args) if (args.gpus == 1 and not args.no_secondary) else None
    # roofline.traffic is MEASURED by this run: a second child, parked like the first, runs the headline scene under
    # rocprofv3 --pmc (FETCH_SIZE and WRITE_SIZE in separate passes) once the timed region is over
    traffic_proc = traffic_start(args) if (args.gpus == 1 and not args.no_traffic and args.strategy.lower() == "clahe"
                                           and (args.rows, args.cols) == (20000, 20000)) else None

    import torch
    import torch.distributed as dist

    import sarpro_amd
    from sarpro_amd import AutoscaleStrategy, SyntheticRgbMode, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, this node has {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    strategy = {s.name.lower(): s for s in AutoscaleStrategy}[args.strategy.lower()]
    rows, cols = args.rows, args.cols
    pitch = (cols + args.pitch_align - 1) // args.pitch_align * args.pitch_align
    # scenes are enqueued back to back on the library's stream (SARPRO_HIP_CTX_ASYNC_DEV); --sync-steps: one host round trip per scene
    use_async = not args.sync_steps and not (args.mode == "stripe" and world > 1)
    ctx = sarpro_amd.Context(local_rank, timing=True, async_dev=use_async)
    q = synth.q_tables()

    if args.mode == "stripe" and world > 1:
        r0s, nrs = sarpro_amd.host_stripe_plan(rows, world)
        row0, rows_local = r0s[rank], nrs[rank]
        seed = synth.SEED_SCENE_A
    else:
        row0, rows_local = 0, rows
        seed = synth.SEED_SCENE_A + 16 * rank  # batch: every rank its own scenes

    # The timed steps cycle over K resident scenes that differ in DISTRIBUTION (class maps, sigma sets, with and without invalid
    # pixels, amplitude windows beyond the fused pass's LDS pool, a constant band): the speculative route of the CLAHE chain can
    # miss on them, and a miss costs what it costs inside `value`.  Scene A (rounds 1-3) is step 0.  A row stripe is one scene.
    striped = args.mode == "stripe" and world > 1
    scene_defs = synth.BENCH_SCENES[:1] if striped else synth.BENCH_SCENES[:(args.scenes or len(synth.BENCH_SCENES))]
    scenes = []
    for name, off, flags, qkw, what in scene_defs:
        qs = synth.q_tables(**qkw) if qkw else q
        bands = [torch.empty((max(rows_local, 1), pitch), dtype=torch.int16, device=dev) for _ in range(2)]
        for b in range(2):
            ctx.dev_synth_scene_u16(seed + off, b, qs, rows, cols, row0, rows_local, bands[b].data_ptr(), pitch, flags)
        scenes.append(bands)
    K = len(scenes)
    band = scenes[0]
    rgb = torch.empty((max(rows_local, 1), pitch * 3), dtype=torch.uint8, device=dev)
    # The timed steps of a whole-scene run go through the library's RESIDENT BATCH entry point (csrc/pipeline.cpp: one context, `lanes`
    # internal streams + workspace sets; the batch loop of api/mod.rs:484-533 for rasters that are already in HBM): two or three
    # scenes are in flight, so each lane's scenes get an RGB raster of their own (scene i -> rgbs[i mod lanes]; the scenes of one
    # lane run in stream order).  --lanes 0: one call per scene on one stream, as rounds 3-4 timed it (`ms_per_step_scene_a` and
    # `ms_per_step_one_stream` keep measuring that loop, so the rounds stay comparable).
    pipelined = (not striped) and args.lanes > 0 and use_async
    rgbs = [rgb] + [torch.empty_like(rgb) for _ in range(max(args.lanes, 1) - 1)] if pipelined else [rgb]

    def batch_steps(n, first=0):
        """n steps (scenes first, first + 1, ... of the cycle) in ONE call of the resident batch entry point; returns the scenes' routes"""
        batch = [(scenes[(first + i) % K][0].data_ptr(), scenes[(first + i) % K][1].data_ptr(), rgbs[i % len(rgbs)].data_ptr()) for i in range(n)]
        rep, st, routes = ctx.dev_batch_dualpol_synrgb_u16(batch, rows, cols, pitch, strategy, SyntheticRgbMode.Default, pitch, lanes=len(rgbs))
        assert rep["processed"] == n, (rep, st)
        return routes

    if args.mode == "stripe" and world > 1:
        # library-owned RCCL communicator (xGMI): rank 0 makes the id, everyone joins
        uid = [sarpro_amd.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])

    def step(i=0):
        b = scenes[i % K]
        if striped:
            # one call: device-resident chain with its RCCL all-reduces enqueued on the library's stream
            ctx.stripe_run_u16(b[0].data_ptr(), b[1].data_ptr(), rows, cols, row0, rows_local, pitch, strategy,
                               SyntheticRgbMode.Default, rgb.data_ptr(), pitch)
        else:
            ctx.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, strategy,
                                       SyntheticRgbMode.Default, rgb.data_ptr(), pitch, want_stats=False)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Warm-up steps are timed kernel by kernel to find the dominant kernel.  In the timed region only THAT kernel is
    # bracketed by HIP events (on the library's stream): an event pair idles the stream ~10 us between two kernels, a
    # dozen of them per step would be measurement overhead inside `value`.  The per-kernel breakdown comes from
    # EXTRA_STEPS fully instrumented steps AFTER the timed region.
    EXTRA_STEPS = 3
    # Before anything is timed every scene runs once, synchronously: which route it takes (sarpro_hip_ctx_spec_report: accepted /
    # refuted / unproven / pool_overflow -- a property of the scene, the chain is deterministic) and what one synchronous call costs.
    # Rank 0 keeps scene A's RGB raster on the host: the CPU baseline below compares it with the oracle's, pixel by pixel.
    outcomes, scene_sync_ms, rgb_a_host = [], [], None
    step(0)  # (the first call of a shape builds its plan and sizes the workspaces: not what "one synchronous call" is meant to show)
    ctx.synchronize()
    for i in range(K):
        step(i)  # (first touch of this scene's rasters: the second call is the one that is timed -- a cold call read 1.19-1.46 ms where the cycle shows 1.05)
        ctx.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step(i)
        ctx.synchronize()
        scene_sync_ms.append((time.perf_counter() - t0) * 1e3)
        try:
            outcomes.append(ctx.spec_report()["outcome"] if (strategy == AutoscaleStrategy.Clahe and not striped) else "n/a")
        except Exception:
            outcomes.append("n/a")  # (no speculative chain ran: a scene below the route's size threshold)
        if i == 0 and rank == 0 and world == 1 and not args.no_cpu_baseline and not args.no_cpu_full and not striped:
            rgb_a_host = rgb.view(max(rows_local, 1), pitch, 3)[:, :cols].cpu().numpy()
    wtimes: dict[str, float] = {}
    for w in range(args.warmup):
        step(w)
        ctx.synchronize()
        for name, ms in ctx.last_kernel_times():
            if not name.startswith("host:"):
                wtimes[name] = wtimes.get(name, 0.0) + ms
    dom_warm = max(wtimes, key=wtimes.get) if wtimes else None
    if dom_warm is None:  # --warmup 0: no measurement to choose by; the kernel that dominates this strategy's chain in every profile
        dom_warm = "clahe_rgb_fused" if strategy == AutoscaleStrategy.Clahe else "lut_compose_u16"
    ctx.last_kernel_times()  # (drained: with --warmup 0 the classification pass's event pairs would otherwise count as timed launches)
    if dom_warm:
        ctx.time_only(dom_warm)
    dtimes: list[float] = []  # the dominant kernel's launches inside the timed region
    routes_timed = None
    warm_pipelined = 0
    if pipelined:
        # The lanes' contexts, plans and workspaces are made by their first scenes, and the first ~30 scenes of a process through the
        # lanes run 4-5 % slower than the ones after them whatever the scenes are (tools/pipe_sweep.py prints every repetition:
        # 1.03, 0.989, 0.985, 0.983 ms per scene for four batches in a row; the one-stream loop shows the same drift): beside the W
        # warm-up steps above, untimed batches of the same K-step shape run here until 40 scenes have gone through, and the
        # record says so (`config.warmup_through_lanes`).  The timed region is exactly K steps.
        while warm_pipelined < max(40, args.warmup):
            batch_steps(args.steps)
            warm_pipelined += args.steps
        ctx.last_kernel_times()
    barrier()
    t0 = time.perf_counter()
    if pipelined:
        routes_timed = batch_steps(args.steps)  # EXACTLY args.steps scenes, one call; it returns when every raster is complete
    else:
        for i in range(args.steps):
            step(i)
            if not use_async:
                dtimes += [ms for name, ms in ctx.last_kernel_times() if name == dom_warm]  # HIP events on the library's stream
    barrier()
    elapsed = time.perf_counter() - t0
    if use_async:  # the event pairs of all the enqueued scenes (pipelined: of every lane, on the lane's stream), read after the timed region
        dtimes += [ms for name, ms in ctx.last_kernel_times() if name == dom_warm]
        if pipelined:  # (reported lane after lane: back into step order, scene i ran on lane i mod lanes)
            L = len(rgbs)
            per_lane = [(args.steps - l + L - 1) // L for l in range(L)]
            if len(dtimes) == args.steps:
                starts = np.cumsum([0] + per_lane[:-1])
                dtimes = [dtimes[int(starts[i % L]) + i // L] for i in range(args.steps)]
    # the fused pass returns at once on a scene whose speculation never started (unproven) or whose windows exceed its pool:
    # such launches moved no bytes and are left out of the roofline average (one launch of the dominant kernel per step)
    if K > 1 and len(dtimes) == args.steps and dom_warm == "clahe_rgb_fused":
        rts = routes_timed if routes_timed is not None else [outcomes[i % K] for i in range(args.steps)]
        dtimes = [ms for i, ms in enumerate(dtimes) if rts[i] in ("accepted", "retried", "refuted")]

    subset_dom_ms: list[float] = []  # the dominant kernel's event times of the LAST timed_subset run (one stream: nothing beside the kernel)
    subset_dom_scene: list[int] = []  # ... and the scene of each

    def timed_subset(idx):  # the same enqueue pattern over a subset of the scenes (beside `value`, after the timed region)
        idx = idx or [0]  # (every rank takes part in the barriers, whatever its scenes did)
        runs = []
        for _ in range(3):  # median of three: K steps are ~25 ms of wall clock, one host hiccup in them is 5-10 % (`value` takes no such liberty)
            barrier()
            t = time.perf_counter()
            for i in range(args.steps):
                step(idx[i % len(idx)])
            barrier()
            runs.append((time.perf_counter() - t) / args.steps * 1e3)
            kt = [ms for name, ms in ctx.last_kernel_times() if name == dom_warm]  # (drained: nothing of these steps reaches the per-kernel table below)
            if len(kt) == args.steps:
                keep = [i for i in range(len(kt)) if outcomes[idx[i % len(idx)]] in ("accepted", "retried", "refuted", "n/a")]
                subset_dom_ms[:] = [kt[i] for i in keep]
                subset_dom_scene[:] = [idx[i % len(idx)] for i in keep]
        return sorted(runs)[1]
    ms_accepted = timed_subset([i for i in range(K) if outcomes[i] in ("accepted", "n/a")]) if K > 1 else None
    ms_scene_a = timed_subset([0]) if (K > 1 or pipelined) else None
    ms_one_stream = timed_subset(list(range(K))) if pipelined else None  # the whole cycle, one call per scene on one stream (the headline loop of rounds 3-4)
    # Under the lanes a kernel's event pair brackets more than the kernel when another lane's sweep shares the chip (round 5's free
    # run: its workgroups are dispatched as the other sweep releases compute units).  `roofline.frac` / `achieved` / `ms_per_launch`
    # are the figures of the TIMED REGION, whatever they bracket; the same kernel on the one-stream cycle measured right after it
    # (same scenes, same process, nothing beside the kernel) is reported beside them as `frac_one_stream` / `ms_per_launch_one_stream`,
    # and the slowest scene of that cycle as `frac_worst_scene`.
    one_stream_ms = list(subset_dom_ms) if (pipelined and subset_dom_ms) else None
    per_scene_ms = {}
    for sc_i, ms in zip(subset_dom_scene, subset_dom_ms):
        per_scene_ms.setdefault(sc_i, []).append(ms)
    ctx.time_only(None)
    ktimes: dict[str, list[float]] = {}
    for _ in range(EXTRA_STEPS):  # (scene A: the per-kernel breakdown of the route that is taken when the speculation holds)
        step(0)
        ctx.synchronize()
        for name, ms in ctx.last_kernel_times():
            ktimes.setdefault(name, []).append(ms)
    barrier()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        scenes_per_step = world if not (args.mode == "stripe" and world > 1) else 1
        px = rows * cols * scenes_per_step * args.steps
        value = px / elapsed / 1e6
        # dominant kernel = largest total event time; one launch covers both bands of the local rows
        per_launch = {k: float(np.mean(v)) for k, v in ktimes.items()}
        launches = {k: len(v) / EXTRA_STEPS for k, v in ktimes.items()}
        total_ms = {k: per_launch[k] * launches[k] for k in per_launch}
        kern = {k: v for k, v in total_ms.items() if not k.startswith("host:")}  # host:* entries are wall-clock segments
        dom = dom_warm if dom_warm in kern else (max(kern, key=kern.get) if kern else None)
        if dom == dom_warm and dtimes:  # the roofline figure is the one measured inside the timed region
            per_launch[dom] = float(np.mean(dtimes))
            launches[dom] = len(dtimes) / args.steps
        # algorithmic bytes per pixel PER LAUNCH (both bands), DESIGN.md section 4
        alg_bpp = {"dn_hist_u16": 4.0, "clahe_rgb_fused": 7.0, "clahe_apply_u8_spec": 6.0, "clahe_apply_u16": 6.0, "compose_u8": 5.0, "lut_apply_u16": 3.0, "lut_compose_u16": 7.0}
        roofline = None
        if dom:
            local_px = rows_local * cols
            # alg_bpp covers both bands; a kernel launched once per band (the staggered two-stream chain) moves half per launch
            bytes_launch = alg_bpp.get(dom, 0.0) * local_px / max(launches[dom], 1.0)
            achieved = bytes_launch / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
            tr = pmc_traffic_gb(dom, rows_local * cols / max(launches[dom], 1.0))
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        # scalars only at this level (a reader that flattens the record keeps them); the dicts follow under *_detail
                        "traffic": tr.get("value") if isinstance(tr, dict) else None,
                        "traffic_unit": "GB per launch (HBM bytes by the PMC counters: FETCH_SIZE x 2 + WRITE_SIZE, KiB)",
                        "algorithmic_gb": round(bytes_launch / 1e9, 3),
                        "traffic_ratio": round(tr["value"] / (bytes_launch / 1e9), 3) if isinstance(tr, dict) and tr.get("value") and bytes_launch else None,
                        "traffic_detail": tr,
                        "launches_per_step": launches[dom],
                        "ms_per_launch": round(per_launch[dom], 4),
                        "frac_in_timed_region": round(achieved / HBM_PEAK_GBS, 4) if (dom == dom_warm and dtimes) else None,
                        "timed_in": "timed region (HIP events on this kernel only, on the stream it is launched on)" if (dom == dom_warm and dtimes) else "extra steps",
                        "kernels_ms_per_step": {k: round(total_ms[k], 4) for k in sorted(total_ms)},
                        "kernels_ms_per_step_from": f"{EXTRA_STEPS} fully instrumented steps after the timed region"}
            if one_stream_ms:
                m1 = float(np.mean(one_stream_ms))
                roofline["ms_per_launch_one_stream"] = round(m1, 4)
                roofline["frac_one_stream"] = round(alg_bpp.get(dom, 0.0) * local_px / (m1 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if m1 > 0 else None
            if per_scene_ms:
                worst_i = max(per_scene_ms, key=lambda i: float(np.mean(per_scene_ms[i])))
                mw = float(np.mean(per_scene_ms[worst_i]))
                roofline["frac_worst_scene"] = round(alg_bpp.get(dom, 0.0) * local_px / (mw * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if mw > 0 else None
                roofline["worst_scene"] = scene_defs[worst_i][0]
                roofline["ms_per_launch_by_scene"] = {scene_defs[i][0]: round(float(np.mean(v)), 4) for i, v in sorted(per_scene_ms.items())}
        out = {
            "metric": "Mpix/s calibrate+CLAHE+synRGB, 400MP dual-pol scene; % HBM roofline",
            "value": round(value, 1), "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if (args.mode == "stripe" and world > 1) else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{rows}x{cols} dual-pol u16 in HBM -> dB+{strategy.name} u8 x2 -> synRGB u8, native res (save.rs:317-367)",
                       "workload_detail": f"dual-pol u16 {rows}x{cols} scene resident in HBM -> dB + {strategy.name} autoscale u8 x2 "
                                          f"-> synRGB ({'suppressed' if strategy.name in ('Clahe', 'Tamed') else 'default'} variant) interleaved u8, native resolution (save.rs:317-367); "
                                          + (f"the steps cycle over {K} resident scenes that differ in distribution ({', '.join(d[0] for d in scene_defs)}), scene A = the scene of rounds 1-3 first"
                                             if K > 1 else "one scene (A) in every step"),
                       "scenes_in_cycle": K,
                       "mode": args.mode if world > 1 else "single", "rows": rows, "cols": cols,
                       "scenes_per_step": scenes_per_step,
                       "scenes": [{"name": d[0], "what": d[4], "route": outcomes[i], "ms_one_synchronous_call": round(scene_sync_ms[i], 3)} for i, d in enumerate(scene_defs)],
                       "warmup_through_lanes": warm_pipelined,
                       "enqueue": (f"resident batch entry point (sarpro_hip_batch_dualpol_synrgb_u16_dev), {len(rgbs)} internal lanes, ONE call for the {args.steps} timed steps, "
                                   "synchronous (returns when every raster is complete)") if pipelined
                                  else "stream-ordered, one synchronisation after the K steps" if use_async else "one host round trip per step"},
            "roofline": roofline,
        }
        if strategy == AutoscaleStrategy.Clahe and not striped:
            # what the speculative route (sampled level histogram -> proven identity + predicted floor -> fused CLAHE -> RGB pass
            # that verifies the floor) did over the TIMED steps: accepted = its RGB stood; refuted = the floor was mispredicted,
            # unproven = level 0 or 255 not proven in both bands, pool_overflow = DN windows beyond the pass's LDS pool: in those
            # three the exact apply -> finish -> compose kernels produced the raster (inside `value`)
            per_step = routes_timed if routes_timed is not None else [outcomes[i % K] for i in range(args.steps)]  # pipelined: what the batch reported per scene
            out["spec"] = {k: per_step.count(k) for k in ("accepted", "retried", "refuted", "unproven", "pool_overflow")}
            out["ms_per_step_accepted_scenes"] = round(ms_accepted, 3) if ms_accepted is not None else out["ms_per_step"]
            out["ms_per_step_scene_a"] = round(ms_scene_a, 3) if ms_scene_a is not None else out["ms_per_step"]
            if ms_one_stream is not None:
                out["ms_per_step_one_stream"] = round(ms_one_stream, 3)
                out["ms_per_step_note"] = ("ms_per_step / value: the resident batch entry point (lanes overlap the scenes' chains); ms_per_step_one_stream: the same nine-scene cycle, "
                                           "one call per scene enqueued on ONE stream (the timed loop of rounds 3-4); ms_per_step_scene_a: scene A only, one stream (rounds 1-4)")
    else:
        out = None
    # ---- everything below is beside the headline: it runs after the timed region, with the headline's rasters freed, and a
    # failure in it costs its own record, never the line
    # one rank, one whole scene A (batch mode): the N = 1 reference of the stripe leg, which runs scene A
    single_ms = ms_scene_a if ms_scene_a is not None else elapsed / args.steps * 1e3
    if rank == 0:  # the measured headline, on stderr, BEFORE anything beside it runs: a hang or a kill further down still leaves it in the log
        print("bench.py headline (the full record follows on stdout): " + json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step")}),
              file=sys.stderr, flush=True)
    del band, rgb, rgbs, scenes
    torch.cuda.empty_cache()
    if world > 1 and not args.no_secondary:
        sec = {}
        try:
            sec = multi_rank_records(torch, dist, dev, rank, world, rows, cols, strategy, single_ms if args.mode == "batch" else None)
        except Exception as e:
            sec = {"error": f"{type(e).__name__}: {e}"}
        if rank == 0:
            out["secondary"] = sec
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(ctx, q, args.cpu_sample, int(strategy), torch, dev, rows if not args.no_cpu_full else 0, cols, rgb_a_host)
            except Exception as e:
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
    ctx.close()
    parity_failed = False
    if rank == 0:
        if traffic_proc is not None and out.get("roofline"):
            live = traffic_collect(traffic_proc, out["roofline"]["kernel"])
            if live and "value" in live:
                out["roofline"]["traffic"] = live["value"]
                out["roofline"]["traffic_detail"] = live
                if out["roofline"].get("algorithmic_gb"):
                    out["roofline"]["traffic_ratio"] = round(live["value"] / out["roofline"]["algorithmic_gb"], 3)
            elif live:
                out["roofline"]["traffic_live_error"] = live.get("error")
        if sec_child is not None:
            out["secondary"] = secondary_collect(sec_child)
            fr = out["secondary"].get("forced_routes") if isinstance(out["secondary"], dict) else None
            if isinstance(fr, dict) and "error" not in fr:  # scalars at the top level: what a refuted or unproven scene costs beside an accepted one
                for label in ("retried", "refuted", "unproven"):
                    out[f"ms_per_step_{label}"] = fr.get(label)
                out["retried_over_accepted"] = fr.get("retried_over_accepted")
                out["refuted_over_accepted"] = fr.get("refuted_over_accepted")
                out["ms_per_step_forced_note"] = ("scene A on one stream with SPEC_FORCE = mispredict (retried: a second fused pass), mispredict2 (refuted twice: exact kernels), "
                                                  "nospec (unproven: exact kernels only), measured by the secondary child; compare with secondary.forced_routes.accepted")
        print(json.dumps(out), flush=True)
        parity_failed = isinstance(out.get("cpu_baseline"), dict) and out["cpu_baseline"].get("gpu_equals_oracle_full_size") is False
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:  # the line is out (a reader sees the flag); the exit code says the same
        sys.exit("bench.py: the GPU's RGB raster of scene A differs from the CPU oracle's at full size")


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD (never exec: this process may
    not replace itself once a GPU runtime is loaded, and nothing here has touched one yet), one rank per GPU on 127.0.0.1,
    relay its output and return its exit code.  A node with fewer than N GPUs fails in the child, with its message."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    launcher = os.environ.get("SARPRO_BENCH_LAUNCHER")  # tests substitute a stub for torch.distributed.run
    if launcher:
        cmd = [sys.executable, launcher] + cmd[3:]
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def traffic_start(args):
    """A child that has loaded no GPU runtime, parked on its stdin: on "go" it runs tools/profile_one.py (three headline scenes, nothing
    else in the process) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (the two do not fit one
    pass; no other trace domain) and prints the per-kernel means."""
    import subprocess
    try:
        return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--traffic-only"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)
    except Exception as e:
        return e


def traffic_child(args):
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if sys.stdin.readline().strip() != "go":
        return 0
    out = {}
    try:
        if not shutil.which("rocprofv3"):
            raise RuntimeError("rocprofv3 is not on PATH")
        env = dict(os.environ, TMPDIR="/tmp")
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = tempfile.mkdtemp(prefix="sarpro_pmc_", dir="/tmp")
            try:
                subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
                                sys.executable, os.path.join(ROOT, "tools", "profile_one.py"), "3", "4"],
                               cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
                files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
                acc, n = {}, {}
                for r in csv.DictReader(open(files[0])):
                    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0].split("::")[-1]
                    acc[k] = acc.get(k, 0.0) + float(r["Counter_Value"])
                    n[k] = n.get(k, 0) + 1
                out[counter] = {k: acc[k] / n[k] for k in acc}
            finally:
                shutil.rmtree(d, ignore_errors=True)
    except Exception as e:
        out = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    return 0


def traffic_collect(child, kernel):
    """-> roofline.traffic of `kernel` (bench's name for it) from the child's two PMC passes, with the gfx950 corrections of
    MI355X_MICROARCH.md (HBM section): the counters are KiB; FETCH_SIZE reports half the bytes of a 16-B-per-lane streaming read (x 2),
    WRITE_SIZE is exact for 16-B-per-lane streaming stores."""
    names = {"clahe_rgb_fused": "k_clahe_rgb_fused", "clahe_apply_u8_spec": "k_clahe_apply_u8_spec", "dn_hist_u16": "k_dn_hist_pieces",
             "compose_u8": "k_compose_u8", "lut_compose_u16": "k_lut_compose_u16"}
    if isinstance(child, Exception):
        return {"error": f"{type(child).__name__}: {child}"}
    try:
        stdout, stderr = child.communicate("go\n", timeout=600)
        lines = [l for l in stdout.splitlines() if l.startswith("{")]
        raw = json.loads(lines[-1]) if lines else {"error": f"no output (exit {child.returncode}): {stderr[-200:]}"}
        if "error" in raw:
            return raw
        k = names.get(kernel)
        if not k or k not in raw.get("FETCH_SIZE", {}) or k not in raw.get("WRITE_SIZE", {}):
            return {"error": f"no counters for {kernel}"}
        rd, wr = raw["FETCH_SIZE"][k] * 1024 * 2 / 1e9, raw["WRITE_SIZE"][k] * 1024 / 1e9
        return {"value": round(rd + wr, 3), "read": round(rd, 3), "write": round(wr, 3), "unit": "GB",
                "source": "measured by this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE over three headline "
                          "scenes (scene A) after the timed region, same box and build; KiB counters, FETCH_SIZE x 2 (gfx950 reports half of a 16-B-per-lane "
                          "streaming read), WRITE_SIZE as is"}
    except Exception as e:
        try:
            child.kill()
        except Exception:
            pass
        return {"error": f"{type(e).__name__}: {e}"}


def secondary_start(args):
    """N = 1: the secondary records are measured by a CHILD process (`bench.py --secondary-only`): they allocate pinned host
    memory and several full-size rasters and run many passes, and an out-of-memory kill or a GPU fault there must not take
    the headline line with it.  The child is started before this process has loaded a GPU runtime (nothing is ever exec'ed
    from a process that has), imports its modules, and waits on its stdin for "go" -- sent after the timed region, when the
    headline's rasters are freed -- so it cannot disturb the headline either."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--secondary-only", "--rows", str(args.rows), "--cols", str(args.cols)]
    try:
        return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    except Exception as e:
        return e


def secondary_collect(child):
    if isinstance(child, Exception):
        return {"error": f"{type(child).__name__}: {child}"}
    try:
        stdout, stderr = child.communicate("go\n", timeout=1200)
        lines = [l for l in stdout.splitlines() if l.startswith("{")]
        if child.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"error": f"secondary child exited with {child.returncode}: {stderr[-300:]}"}
    except Exception as e:
        try:
            child.kill()
        except Exception:
            pass
        return {"error": f"{type(e).__name__}: {e}"}


def secondary_child(args):
    import torch
    import sarpro_amd  # noqa: F401  (imported while parked: the first import on a fresh box takes a minute)
    if sys.stdin.readline().strip() != "go":  # the parent went away (or failed) before the headline was done
        return 0
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    try:
        out = secondary_records(torch, dev, args.rows, args.cols)
    except Exception as e:
        out = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    return 0


def multi_rank_records(torch, dist, dev, rank, world, rows, cols, strategy, single_ms):
    """N > 1, in the same processes as the headline (BASELINE.json configs 4 and 5 as the driver's one command sees them):
      ranks_seen     -- an all-reduce of ones over torch.distributed's RCCL group: the record proves N ranks took part
      stripe         -- ONE 400 MP scene as N row stripes, sarpro_hip_stripe_run_u16 over the library's own communicator: three
                        u64 all-reduces inside the chain (DN histograms, CLAHE tile histograms, level histograms)
      e2e_all_ranks  -- the PCIe-inclusive leg on every rank at once: pinned host bands -> H2D -> chain -> D2H of the RGB raster"""
    import ctypes as C
    import sarpro_amd
    from sarpro_amd import SyntheticRgbMode, synth
    from sarpro_amd._lib import lib
    out = {}
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(ones)
    out["ranks_seen"] = int(ones.item())
    q = synth.q_tables()
    pitch = (cols + 63) // 64 * 64

    def max_over_ranks(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def barrier():
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()

    def all_ranks_ok(ok):
        """A rank that failed its own set-up (context, allocations) must not leave its peers inside a collective: every rank
        learns here whether ALL of them are ready, and the leg is skipped by all of them together otherwise."""
        t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # (i) row stripes of one scene
    ctx = band = rgb = None
    err = None
    try:  # the fallible per-rank set-up, no collective in it
        ctx = sarpro_amd.Context(dev.index, timing=True)
        r0s, nrs = sarpro_amd.host_stripe_plan(rows, world)
        row0, rl = r0s[rank], nrs[rank]
        band = [torch.empty((max(rl, 1), pitch), dtype=torch.int16, device=dev) for _ in range(2)]
        rgb = torch.empty((max(rl, 1), pitch * 3), dtype=torch.uint8, device=dev)
        for b in range(2):
            ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, row0, rl, band[b].data_ptr(), pitch)
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    try:
        if not all_ranks_ok(err is None):
            raise RuntimeError(err or "another rank failed its set-up: the leg is skipped on every rank")
        uid = [sarpro_amd.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ctx.comm_init(world, rank, uid[0])

        def stripe():
            ctx.stripe_run_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, row0, rl, pitch, strategy, SyntheticRgbMode.Default, rgb.data_ptr(), pitch)
        for _ in range(3):
            stripe()
        n = 10
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            stripe()
        barrier()
        ms = max_over_ranks((time.perf_counter() - t0) / n * 1e3)
        kt = dict(ctx.last_kernel_times())  # the last scene's event pairs on this rank
        ar = {k: round(max_over_ranks(v) * 1e3, 1) for k, v in sorted(kt.items()) if k.startswith("allreduce_")}
        out["stripe"] = {"what": f"one {rows}x{cols} dual-pol scene as {world} row stripes, sarpro_hip_stripe_run_u16 over the library's RCCL communicator (BASELINE config 4)",
                         "ms_per_scene": round(ms, 3), "value": round(rows * cols / ms / 1e3, 1), "unit": "Mpix/s", "scaling": "strong",
                         "speedup_vs_n1": round(single_ms / ms, 2) if single_ms else None, "n1_ms_per_scene": round(single_ms, 3) if single_ms else None,
                         "allreduce_us_max_over_ranks": ar, "kernels_ms_rank0": {k: round(v, 4) for k, v in kt.items() if not k.startswith("host:")}}
    except Exception as e:
        out["stripe"] = {"error": f"{type(e).__name__}: {e}"}
    del band, rgb
    if ctx is not None:
        ctx.close()
    torch.cuda.empty_cache()
    # (ii) PCIe-inclusive leg, all ranks at once
    ctx = None
    err = None
    try:  # set-up without collectives, then one agreement
        ctx = sarpro_amd.Context(dev.index)
        dband = torch.empty((rows, pitch), dtype=torch.int16, device=dev)
        host = []
        for b in range(2):
            ctx.dev_synth_scene_u16(synth.SEED_SCENE_A + rank, b, q, rows, cols, 0, rows, dband.data_ptr(), pitch)
            h = torch.empty((rows, cols), dtype=torch.int16, pin_memory=True)
            h.copy_(dband[:, :cols])
            host.append(h)
        del dband
        rgb_h = torch.empty((rows, cols, 3), dtype=torch.uint8, pin_memory=True)
        b1, b2 = (h.numpy().view(np.uint16) for h in host)
        o = rgb_h.numpy()
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    try:
        if not all_ranks_ok(err is None):
            raise RuntimeError(err or "another rank failed its set-up: the leg is skipped on every rank")

        def e2e():
            rc = lib.sarpro_hip_dualpol_synrgb_u16(ctx._h, b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), rows, cols, int(strategy), 0,
                                                   o.ctypes.data_as(C.c_void_p), None, None, None)
            assert rc == 0
        e2e()
        n = 3
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            e2e()
        barrier()
        ms = max_over_ranks((time.perf_counter() - t0) / n * 1e3)
        px = rows * cols
        out["e2e_all_ranks"] = {"what": f"every rank at once: pinned host u16 bands -> H2D -> calibrate + {strategy.name} + synRGB -> D2H of the RGB raster (sarpro_hip_dualpol_synrgb_u16), one scene per rank (BASELINE config 5's transfer pattern)",
                                "ms_per_scene_max_over_ranks": round(ms, 2), "value": round(world * px / ms / 1e3, 1), "unit": "Mpix/s aggregate",
                                "pcie_gb_s_aggregate": round(world * (2 * px * 2 + px * 3) / ms / 1e6, 1), "pcie_gb_s_per_rank": round((2 * px * 2 + px * 3) / ms / 1e6, 1)}
        del host, rgb_h
    except Exception as e:
        out["e2e_all_ranks"] = {"error": f"{type(e).__name__}: {e}"}
    if ctx is not None:
        ctx.close()
    return out


def pmc_traffic_gb(kernel, local_px):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/r2_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 corrections applied), scaled
    to this run's pixel count.  None when the kernel has no committed measurement."""
    names = {"clahe_rgb_fused": "k_clahe_rgb_fused", "clahe_apply_u8_spec": "k_clahe_apply_u8_spec", "clahe_apply_u16": "k_clahe_apply_u8_spec", "dn_hist_u16": "k_dn_hist_pieces",
             "compose_u8": "k_compose_u8", "lut_apply_u16": "k_lut_apply_u16", "lut_compose_u16": "k_lut_compose_u16"}
    try:
        for fn in ("r4_traffic.json", "r3_traffic.json", "r2_traffic.json"):
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                t = json.load(f)
            if names[kernel] in t:
                gb = t[names[kernel]]["total"] * (local_px / 4.0e8)
                return {"value": round(gb, 3), "unit": "GB", "source": f"profiles/{fn} (rocprofv3 PMC, separate passes, an earlier run of this command's kernels: a committed measurement, not one of this run)"}
        return None
    except Exception:
        return None


def cpu_baseline(ctx, q, side, strategy, torch, dev, full_rows=0, full_cols=0, gpu_rgb=None):
    """Time the CPU oracle on the same synthetic workload: (i) ONE thread -- the reference's behaviour, its hot path has no
    threads -- median of 3 runs on a side x side sample, and ONE run on the metric's own configuration
    (full_rows x full_cols, ~30 s) beside it, so that "the per-pixel cost does not depend on the scene size" is shown, not
    stated; (ii) next to each, labelled as NOT the reference's behaviour, a row-parallel variant of the same loops on all host
    cores (oracle/sarpro_oracle_mt.c, OpenMP).  `value` is the full-size single-thread figure when it was measured."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    from sarpro_amd import synth

    def scene(rows, cols):
        pitch = (cols + 63) // 64 * 64
        bands = []
        for b in range(2):
            t = torch.empty((rows, pitch), dtype=torch.int16, device=dev)
            ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, t.data_ptr(), pitch)
            bands.append(t[:, :cols].contiguous().cpu().numpy().view(np.uint16).astype(np.float32))
            del t
        return bands

    def timed(fn, n):
        runs, last = [], None
        for _ in range(n):
            t0 = time.perf_counter()
            last = fn()
            runs.append(time.perf_counter() - t0)
        return runs, last

    oracle.lib()
    bands = scene(side, side)
    runs, (rc, rgb, _, _) = timed(lambda: oracle.dualpol_synrgb(bands[0], bands[1], strategy), 3)
    assert rc == 0
    dt = sorted(runs)[1]
    out = {"value": round(side * side / dt / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
           "sample": f"{side}x{side} dual-pol scene (same generator), whole path, median of 3 runs ({', '.join(f'{x:.1f}' for x in runs)} s) on 1 of {os.cpu_count()} host threads",
           "context": "the reference's README quotes ~40 s per 400 MP dual-band native synRGB scene incl. I/O on an M4 Pro (README.md:63); not measured here"}
    mt = None
    if strategy == 4:  # the row-parallel variant restates the CLAHE path only
        try:
            mt = oracle.lib_mt()
            nthr = int(mt.sarpro_oracle_mt_threads())
            runs2, (rc2, rgb2) = timed(lambda: oracle.dualpol_clahe_synrgb_mt(bands[0], bands[1]), 3)
            assert rc2 == 0
            dt2 = sorted(runs2)[1]
            out["row_parallel"] = {"value": round(side * side / dt2 / 1e6, 1), "unit": "Mpix/s", "cores": nthr,
                                   "note": f"same loops split over rows with OpenMP on all {nthr} host threads, median of 3 runs "
                                           f"({', '.join(f'{x:.2f}' for x in runs2)} s); NOT the reference's behaviour (its hot path is single-threaded)",
                                   "raster_equals_single_thread": bool(np.array_equal(rgb2, rgb))}
        except Exception as e:  # the single-thread figure is the baseline; this one is informative
            out["row_parallel"] = {"error": str(e)}
            mt = None
    del bands, rgb
    if full_rows and full_cols and (full_rows, full_cols) != (side, side):
        try:  # the metric's configuration itself, once
            fb = scene(full_rows, full_cols)
            px = full_rows * full_cols
            (t1,), (rc, rgbf, _, _) = timed(lambda: oracle.dualpol_synrgb(fb[0], fb[1], strategy), 1)
            assert rc == 0
            out["sample_value"] = out["value"]
            out["value"] = round(px / t1 / 1e6, 2)
            out["full_size"] = {"value": out["value"], "unit": "Mpix/s", "cores": 1, "seconds": round(t1, 1),
                                "what": f"{full_rows}x{full_cols} dual-pol scene (the metric's configuration), whole path, one run on 1 host thread"}
            out["sample"] = f"value: one {full_rows}x{full_cols} run ({t1:.1f} s); sample_value: " + out["sample"]
            if gpu_rgb is not None:  # the oracle as the checker: EVERY pixel of the GPU's raster of the same scene (scene A, fused route)
                out["gpu_equals_oracle_full_size"] = bool(gpu_rgb.shape == rgbf.shape and np.array_equal(gpu_rgb, rgbf))
                if not out["gpu_equals_oracle_full_size"]:
                    out["gpu_vs_oracle_differing_bytes"] = int((gpu_rgb != rgbf).sum()) if gpu_rgb.shape == rgbf.shape else -1
            if mt is not None:
                (t2,), (rc2, rgb2) = timed(lambda: oracle.dualpol_clahe_synrgb_mt(fb[0], fb[1]), 1)
                out["row_parallel"]["full_size"] = {"value": round(px / t2 / 1e6, 1), "unit": "Mpix/s", "seconds": round(t2, 2),
                                                    "raster_equals_single_thread": bool(rc2 == 0 and np.array_equal(rgb2, rgbf))}
        except Exception as e:
            out["full_size"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def secondary_records(torch, dev, rows, cols):
    """Records beside the headline (never `value`): the PCIe-inclusive end-to-end leg (SURVEY 8d timing protocol 2) and the
    other BASELINE.json configurations, device-resident, at full size.  Each is a handful of calls after the timed region.  The ms
    figures are synchronous calls on a context without the timing table; the per-kernel tables come from an instrumented twin."""
    import sarpro_amd
    from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, PolarizationOperation as Op, SyntheticRgbMode as Mode, synth
    import ctypes as C
    from sarpro_amd._lib import lib

    out = {}
    q = synth.q_tables()
    pitch = (cols + 63) // 64 * 64

    def timed(fn, n=5, warm=2):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3

    # (first: a 0.2-ms call reads 0.21-0.23 ms in a quiet process and 0.25-0.27 right after the 400 MP records below -- tools/dbg_ctx_count.py:
    # neither the number of live contexts nor of streams explains it, what ran just before does)
    # The reference's DEFAULT flow at its usual size (`--size 2048` resamples on read: the raster core sees two non-integer f32 bands):
    # per-band autoscale -> synRGB, device-resident, one synchronous call.  A context without the timing table: the second band runs on
    # the context's twin (own stream) from its helper thread.
    try:
        side = 2048
        fb = [torch.rand((side, side), dtype=torch.float32, device=dev) * 900.0 + 1.0 for _ in range(2)]
        rgb1 = torch.empty((side, side * 3), dtype=torch.uint8, device=dev)
        rec = {"what": "dual-pol 2048x2048 f32 bands resident in HBM -> per-band autoscale -> synRGB (one synchronous call)", "unit": "ms per call"}
        c2 = sarpro_amd.Context(dev.index)
        try:
            for name, st in (("default", St.Default), ("tamed", St.Tamed), ("clahe", St.Clahe)):
                rec[name] = round(timed(lambda: c2.dev_dualpol_synrgb_f32(fb[0].data_ptr(), fb[1].data_ptr(), side, side, side, st, Mode.Default,
                                                                         rgb1.data_ptr(), side), n=20, warm=3), 4)
        finally:
            c2.close()
        rec["value"] = round(side * side / rec["default"] / 1e3, 1)
        rec["value_unit"] = "Mpix/s (default strategy)"
        out["config1_dualpol_f32"] = rec
        del fb, rgb1
    except Exception as e:
        out["config1_dualpol_f32"] = {"error": f"{type(e).__name__}: {e}"}
    # The same flow as a BATCH (api/mod.rs:474-536 loops over a directory's scenes): 64 scenes of 2048 x 2048 and of 1024 x 1024 f32 bands
    # -> CLAHE (plain pipeline, api/mod.rs:404-437) -> resize + pad to the same side -> synRGB, resident: one call per scene on one
    # context against sarpro_hip_batch_dualpol_synrgb_resized_f32_dev (the context's lanes, a host thread each: the f32 chain's host turns
    # overlap), and host to host through the batch driver (pageable numpy bands in, RGB out: PCIe and the staging copies bound it).
    try:
        rec = {"what": "64 dual-pol f32 scenes -> CLAHE -> resize + pad -> synRGB (the reference's default flow, api/mod.rs:404-437, 474-536)", "unit": "scenes per second"}
        for side in (2048, 1024):
            n, distinct = 64, 4
            fb = [[torch.rand((side, side), dtype=torch.float32, device=dev) * 900.0 + 1.0 for _ in range(2)] for _ in range(distinct)]
            outs = [torch.empty((side * side * 3,), dtype=torch.uint8, device=dev) for _ in range(n)]
            batch = [(fb[i % distinct][0].data_ptr(), fb[i % distinct][1].data_ptr(), outs[i].data_ptr()) for i in range(n)]
            c3 = sarpro_amd.Context(dev.index)
            try:
                def single():
                    for b1, b2, o in batch:
                        c3.dev_dualpol_synrgb_resized_f32(b1, b2, side, side, side, St.Clahe, side, True, o, plain_pipeline=True)
                t1 = timed(single, n=3, warm=1)
                r = {"resident_one_call_per_scene": round(n / t1 * 1e3, 1)}
                for lanes in (1, 4, 8):
                    tl = timed(lambda: c3.dev_batch_dualpol_synrgb_resized_f32(batch, side, side, side, St.Clahe, side, True, plain_pipeline=True, lanes=lanes), n=3, warm=1)
                    r[f"resident_batch_{lanes}_lanes"] = round(n / tl * 1e3, 1)
                r["resident_batch_over_one_call_per_scene"] = round(max(r["resident_batch_4_lanes"], r["resident_batch_8_lanes"]) / r["resident_one_call_per_scene"], 2)  # (the better of 4 / 8 lanes)
            finally:
                c3.close()
            hs = [(fb[i % distinct][0].cpu().numpy(), fb[i % distinct][1].cpu().numpy()) for i in range(distinct)]
            hs = [hs[i % distinct] for i in range(n)]
            for w in (1, 2):
                t0 = time.perf_counter()
                sarpro_amd.batch_dualpol_synrgb_resized_f32([dev.index], hs, St.Clahe, side, True, plain_pipeline=True, workers_per_device=w)
                r[f"host_to_host_{w}_workers"] = round(n / (time.perf_counter() - t0), 1)
            rec[f"{side}x{side}"] = r
            del fb, outs, hs
        out["small_scene_batch_f32"] = rec
    except Exception as e:
        out["small_scene_batch_f32"] = {"error": f"{type(e).__name__}: {e}"}
    ctx = sarpro_amd.Context(dev.index, timing=True)  # per-kernel tables (an event pair costs the stream ~10 us per kernel)
    cp = sarpro_amd.Context(dev.index)                # the ms figures: what a caller sees
    band = [torch.empty((rows, pitch), dtype=torch.int16, device=dev) for _ in range(2)]
    for b in range(2):
        ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
    torch.cuda.synchronize()

    def kernels():
        return {k: round(v, 4) for k, v in ctx.last_kernel_times() if not k.startswith("host:")}

    px = rows * cols
    # (0) What the speculative route costs when it does NOT hold -- scene A, one call per scene enqueued on one stream, in THIS child process
    # (the headline process launches the fused pass on accepted scenes only: its rocprofv3 kernel statistics stay those of the timed
    # kernel): the floor predicted one level off (SPEC_FORCE = mispredict: refuted, the second fused pass with the floor the counts point
    # to stands -- `retried`), two levels off (mispredict2: both passes refuted, the exact apply -> finish -> compose kernels), no proof
    # at all (nospec: `unproven`, the exact kernels alone).  None of the nine scenes of the cycle takes these routes by itself; on 200
    # GRD-like rasters of 36-52 MP (profiles/r6/soak_grd_like.txt) 189 were accepted, 11 took the second pass (6 floors one level off, 5
    # lowest levels the sample missed), none went to the exact kernels.
    try:
        rec = {"what": "scene A, 400 MP, CLAHE + synRGB, one call per scene enqueued on one stream; SPEC_FORCE = mispredict / mispredict2 / nospec", "unit": "ms per scene"}
        cf = sarpro_amd.Context(dev.index, async_dev=True)
        rgbf = torch.empty((rows, pitch * 3), dtype=torch.uint8, device=dev)
        try:
            one = lambda: cf.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgbf.data_ptr(), pitch, want_stats=False)
            for label, force in (("accepted", None), ("retried", "mispredict"), ("refuted", "mispredict2"), ("unproven", "nospec")):
                cf.set_attr("SPEC_FORCE", force)
                one(); cf.synchronize()
                got = cf.spec_report()["outcome"]
                rec[label] = round(timed(one, n=20, warm=3), 3) if got == label else None
            cf.set_attr("SPEC_FORCE", None)
            if rec.get("accepted"):
                for label in ("retried", "refuted", "unproven"):
                    if rec.get(label):
                        rec[f"{label}_over_accepted"] = round(rec[label] / rec["accepted"], 3)
        finally:
            cf.close()
        del rgbf
        out["forced_routes"] = rec
    except Exception as e:
        out["forced_routes"] = {"error": f"{type(e).__name__}: {e}"}
    # (1) end to end: pinned host u16 bands -> H2D -> chain -> D2H of the RGB raster, through the host entry point
    try:
        host = [torch.empty((rows, cols), dtype=torch.int16, pin_memory=True) for _ in range(2)]
        for b in range(2):
            host[b].copy_(band[b][:, :cols])
        rgb_h = torch.empty((rows, cols, 3), dtype=torch.uint8, pin_memory=True)
        b1, b2 = (h.numpy().view(np.uint16) for h in host)
        o = rgb_h.numpy()

        def e2e():
            rc = lib.sarpro_hip_dualpol_synrgb_u16(ctx._h, b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), rows, cols, int(St.Clahe), 0,
                                                   o.ctypes.data_as(C.c_void_p), None, None, None)
            assert rc == 0
        ms = timed(e2e, n=3, warm=1)
        out["e2e"] = {"what": "pinned host u16 bands -> H2D -> calibrate + CLAHE + synRGB -> D2H of the RGB raster (sarpro_hip_dualpol_synrgb_u16)",
                      "ms_per_scene": round(ms, 2), "value": round(px / ms / 1e3, 1), "unit": "Mpix/s",
                      "pcie_gb_s": round((2 * px * 2 + px * 3) / ms / 1e6, 1), "north_star_target_mpix_s": 500}
        # two scenes in flight through PCIe: a second context + host thread on the same device, so that scene i's D2H (RGB, 1.2 GB)
        # crosses the link beside scene i + 1's H2D (bands, 1.6 GB) -- the link is full duplex, one synchronous call per scene uses one
        # direction at a time.  What listing a device twice in sarpro_hip_batch_* does.
        try:
            import threading
            c2 = sarpro_amd.Context(dev.index)
            rgb_h2 = torch.empty((rows, cols, 3), dtype=torch.uint8, pin_memory=True)
            o2 = rgb_h2.numpy()

            def e2e_on(cx, oo, n):
                for _ in range(n):
                    rc = lib.sarpro_hip_dualpol_synrgb_u16(cx._h, b1.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), rows, cols, int(St.Clahe), 0,
                                                           oo.ctypes.data_as(C.c_void_p), None, None, None)
                    assert rc == 0
            e2e_on(c2, o2, 1)  # (plan, workspaces)
            n2 = 3
            ths = [threading.Thread(target=e2e_on, args=(cx, oo, n2)) for cx, oo in ((ctx, o), (c2, o2))]
            torch.cuda.synchronize()
            t = time.perf_counter()
            [x.start() for x in ths]
            [x.join() for x in ths]
            ms2 = (time.perf_counter() - t) / (2 * n2) * 1e3
            out["e2e_two_in_flight"] = {"what": "the same end-to-end call from two host threads on two contexts of this GPU (scene i's D2H beside scene i + 1's H2D)",
                                        "ms_per_scene": round(ms2, 2), "value": round(px / ms2 / 1e3, 1), "unit": "Mpix/s",
                                        "pcie_gb_s_both_directions": round((2 * px * 2 + px * 3) / ms2 / 1e6, 1), "rasters_equal": bool(np.array_equal(o, o2))}
            c2.close()
            del rgb_h2
        except Exception as e:
            out["e2e_two_in_flight"] = {"error": f"{type(e).__name__}: {e}"}
        # BASELINE config 2 as a flow: host bands -> Robust -> Lanczos3 to 2048^2 -> pad -> synRGB (small RGB back)
        ms = timed(lambda: ctx.dualpol_synrgb_resized(b1, b2, St.Robust, 2048, True), n=3, warm=1)
        out["config2_flow"] = {"what": "host u16 bands -> Robust autoscale x2 -> Lanczos3 to 2048^2 -> pad -> default synRGB -> 2048x2048x3 back",
                               "ms_per_scene": round(ms, 2), "value": round(px / ms / 1e3, 1), "unit": "Mpix/s"}
        del host, rgb_h
    except Exception as e:
        out["e2e"] = {"error": str(e)}
    # BASELINE config 2 as a flow, device-resident (no PCIe): Robust x2 -> Lanczos3 to 2048^2 -> pad -> default synRGB; carries the
    # resize kernels' own times (resize.rs:32-89: horizontal pass over the u8 level raster, then vertical pass)
    try:
        from sarpro_amd import resize_output_dims
        fc, fr = resize_output_dims(cols, rows, 2048, True)
        rgb_small = torch.empty((fr * fc * 3,), dtype=torch.uint8, device=dev)
        ms = timed(lambda: cp.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, 2048, True, rgb_small.data_ptr()), n=3, warm=1)
        timed(lambda: ctx.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, 2048, True, rgb_small.data_ptr()), n=1, warm=1)
        kt = {}
        for k, v in ctx.last_kernel_times():
            if not k.startswith("host:"):
                kt[k] = round(kt.get(k, 0.0) + v, 4)
        out["config2_flow_resident"] = {"what": "dual-pol u16 resident in HBM -> Robust autoscale x2 (u8 level rasters) -> Lanczos3 to 2048^2 -> pad -> default synRGB, RGB stays on the device",
                                        "ms_per_scene": round(ms, 3), "value": round(px / ms / 1e3, 1), "unit": "Mpix/s", "kernels_ms_summed_over_both_bands": kt}
        del rgb_small
    except Exception as e:
        out["config2_flow_resident"] = {"error": f"{type(e).__name__}: {e}"}
    # BASELINE config 1 (the reference's own CPU-runnable case): single band 2048 x 2048, f32 samples, Standard, u8 -- latency of one call
    try:
        side = 2048
        f1 = torch.rand((side, side), dtype=torch.float32, device=dev) * 900.0 + 1.0
        o1 = torch.empty((side, side), dtype=torch.uint8, device=dev)
        ms = timed(lambda: cp.dev_autoscale_band_f32(f1.data_ptr(), side, side, side, St.Standard, Bd.U8, o1.data_ptr(), side, want_stats=False), n=20, warm=3)
        timed(lambda: ctx.dev_autoscale_band_f32(f1.data_ptr(), side, side, side, St.Standard, Bd.U8, o1.data_ptr(), side, want_stats=False), n=1, warm=2)
        out["config1"] = {"what": "single band 2048x2048 f32 resident in HBM -> Standard autoscale -> u8 (one synchronous call)", "ms_per_call": round(ms, 4),
                          "value": round(side * side / ms / 1e3, 1), "unit": "Mpix/s", "kernels_ms": kernels()}
        del f1, o1
    except Exception as e:
        out["config1"] = {"error": f"{type(e).__name__}: {e}"}
    # (2) BASELINE config 2, hot path, device-resident: Robust x2 -> default synRGB at full resolution
    rgb = torch.empty((rows, pitch * 3), dtype=torch.uint8, device=dev)
    ms = timed(lambda: cp.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, Mode.Default, rgb.data_ptr(), pitch))
    timed(lambda: ctx.dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, Mode.Default, rgb.data_ptr(), pitch), n=1, warm=1)
    out["config2"] = {"what": "dual-pol u16 resident in HBM -> Robust autoscale x2 -> default synRGB, native resolution", "ms_per_scene": round(ms, 3),
                      "value": round(px / ms / 1e3, 1), "unit": "Mpix/s", "kernels_ms": kernels()}
    del rgb
    # (3) BASELINE config 3: CLAHE with u16 output per band; the log-ratio pol-op; CLAHE u16 of the f32 ratio band
    o16 = torch.empty((rows, pitch), dtype=torch.int16, device=dev)
    ms_band = timed(lambda: ctx.dev_autoscale_band_u16(band[0].data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, o16.data_ptr(), pitch))
    k_band = kernels()
    f = [b[:, :cols].to(torch.float32).contiguous() for b in band]
    for x in f:
        x[x < 0] += 65536.0  # the u16 bit pattern was held as int16
    ratio = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    ms_op = timed(lambda: ctx.dev_polop_f32(Op.LogRatio, f[0].data_ptr(), f[1].data_ptr(), px, ratio.data_ptr()))
    del f
    ms_f32 = timed(lambda: ctx.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Clahe, Bd.U16, o16.data_ptr(), pitch, want_stats=False), n=3, warm=1)
    k_f32 = kernels()
    # (ii) fused: the pol-op is computed from the u16 DN inside every pass of the f32 flavour, no f32 raster exists
    ms_fused = timed(lambda: ctx.dev_polop_autoscale_band(Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, rows, cols, pitch, St.Clahe, Bd.U16,
                                                          o16.data_ptr(), pitch, want_stats=False), n=3, warm=1)
    k_fused = kernels()
    # The ms figures of this record come from a context WITHOUT the timing table -- what a caller sees: an event pair costs the stream
    # ~10 us per kernel, 0.03-0.08 ms per call of these chains -- ; the per-kernel tables, and the same calls' ms with the events in
    # (`with_timing_events`), from the instrumented context above.
    plain = {}
    try:
        c3 = cp
        try:
            plain["clahe_u16_per_band_ms"] = round(timed(lambda: c3.dev_autoscale_band_u16(band[0].data_ptr(), rows, cols, pitch, St.Clahe, Bd.U16, o16.data_ptr(), pitch)), 3)
            plain["ratio_f32_clahe_u16_ms"] = round(timed(lambda: c3.dev_autoscale_band_f32(ratio.data_ptr(), rows, cols, cols, St.Clahe, Bd.U16, o16.data_ptr(), pitch, want_stats=False), n=3, warm=1), 3)
            for name, st in (("clahe", St.Clahe), ("robust", St.Robust), ("standard", St.Standard)):
                plain[f"ratio_fused_polop_{name}_u16_ms"] = round(timed(lambda: c3.dev_polop_autoscale_band(
                    Op.LogRatio, band[0].data_ptr(), band[1].data_ptr(), True, rows, cols, pitch, st, Bd.U16, o16.data_ptr(), pitch, want_stats=False), n=5, warm=2), 3)
        finally:
            pass
    except Exception as e:
        plain = {"error": f"{type(e).__name__}: {e}"}
    instrumented = {"clahe_u16_per_band_ms": round(ms_band, 3), "ratio_f32_clahe_u16_ms": round(ms_f32, 3), "ratio_fused_polop_clahe_u16_ms": round(ms_fused, 3)}
    v_band = plain.get("clahe_u16_per_band_ms", instrumented["clahe_u16_per_band_ms"])
    v_f32 = plain.get("ratio_f32_clahe_u16_ms", instrumented["ratio_f32_clahe_u16_ms"])
    v_fused = plain.get("ratio_fused_polop_clahe_u16_ms", instrumented["ratio_fused_polop_clahe_u16_ms"])
    out["config3"] = {"what": "CLAHE u16 per band (i); log-ratio pol-op -> f32 band -> CLAHE u16 (ii); all resident in HBM; ms per synchronous call on a context "
                              "without the timing table, kernel tables from an instrumented context",
                      "clahe_u16_per_band_ms": v_band, "clahe_u16_kernels_ms": k_band,
                      "logratio_polop_ms": round(ms_op, 3), "ratio_f32_clahe_u16_ms": v_f32, "ratio_f32_kernels_ms": k_f32,
                      "ratio_fused_polop_clahe_u16_ms": v_fused, "ratio_fused_kernels_ms": k_fused,
                      "ratio_fused_polop_robust_u16_ms": plain.get("ratio_fused_polop_robust_u16_ms"), "ratio_fused_polop_standard_u16_ms": plain.get("ratio_fused_polop_standard_u16_ms"),
                      "with_timing_events": instrumented, "plain_context_error": plain.get("error"),
                      "scene_ms": round(2 * v_band + v_fused, 3), "scene_unfused_ms": round(2 * v_band + ms_op + v_f32, 3),
                      "value": round(px / (2 * v_band + v_fused) / 1e3, 1), "unit": "Mpix/s"}
    ctx.close()
    cp.close()
    del o16, ratio
    # (3b) what ONE rank of an 8-rank row-stripe run executes (BASELINE config 4), timed alone on this GPU.  First the real thing: eight
    # contexts + host threads joined by the in-process communicator run the single-call stripe chain on the eight 2500-row stripes of
    # scene A (the accepted fused route, as tests/test_gpu_full_size_oracle.py checks against the oracle), with COMM_RECORD on rank 3:
    # it keeps the RESULT of each of its six all-reduces.  Then rank 3 runs the same call ALONE with COMM_REPLAY: every all-reduce is
    # answered from the recorded sums (one device copy), so the rank executes exactly the chain it executed among eight -- the scene's
    # histograms, proof, prediction, verdict -- without seven neighbours sharing the GPU.  What is missing is the wire (five small
    # xGMI all-reduces, section 7 of DESIGN.md prices them); no scaling curve can be measured on a one-GPU box.
    try:
        import threading
        nr = 8
        r0s, nrs = sarpro_amd.host_stripe_plan(rows, nr)
        group = sarpro_amd.LocalGroup(nr)
        cr = [sarpro_amd.Context(dev.index, timing=(k == 3)) for k in range(nr)]
        try:
            for k, c_ in enumerate(cr):
                c_.comm_init_local(group, k)
            rgb_s = torch.empty((rows, pitch * 3), dtype=torch.uint8, device=dev)
            errs = []

            def rank_call(k):
                try:
                    r0, nrw = int(r0s[k]), int(nrs[k])
                    cr[k].stripe_run_u16(band[0].data_ptr() + r0 * pitch * 2, band[1].data_ptr() + r0 * pitch * 2, rows, cols, r0, nrw, pitch, St.Clahe, Mode.Default,
                                         rgb_s.data_ptr() + r0 * pitch * 3, pitch)
                except Exception as e:
                    errs.append(f"rank {k}: {e}")

            def all_ranks():
                ths = [threading.Thread(target=rank_call, args=(k,)) for k in range(nr)]
                [x.start() for x in ths]
                [x.join() for x in ths]
            torch.cuda.synchronize()
            all_ranks()                      # plans, workspaces
            cr[3].set_attr("COMM_RECORD", 1)
            all_ranks()                      # the recorded run
            if errs:
                raise RuntimeError("; ".join(errs))
            route8 = cr[3].spec_report()["outcome"]
            cr[3].reset_attr("COMM_RECORD")
            for c_ in cr:
                c_.comm_destroy()
            cs = cr[3]
            cs.set_attr("COMM_REPLAY", 1)
            r0, nrw = int(r0s[3]), int(nrs[3])

            def stripe_call():
                cs.stripe_run_u16(band[0].data_ptr() + r0 * pitch * 2, band[1].data_ptr() + r0 * pitch * 2, rows, cols, r0, nrw, pitch, St.Clahe, Mode.Default,
                                  rgb_s.data_ptr() + r0 * pitch * 3, pitch)
            ms = timed(stripe_call, n=10, warm=3)
            kt = {}
            for k, v in cs.last_kernel_times():
                if not k.startswith("host:"):
                    kt[k] = round(kt.get(k, 0.0) + v, 4)
            sweeps = sum(v for k, v in kt.items() if k in ("dn_hist_u16", "clahe_rgb_fused", "clahe_sample"))
            out["stripe_rank_model"] = {"what": f"one rank of 8, alone on the GPU: rows [{r0}, {r0 + nrw}) of the {rows}x{cols} scene through sarpro_hip_stripe_run_u16, its six all-reduces "
                                                "answered from the sums recorded in a real 8-rank run of the same scene (COMM_RECORD / COMM_REPLAY, in-process communicator): the chain a "
                                                "rank of an 8-GPU stripe run executes, minus the wire",
                                        "route_in_the_8_rank_run": route8, "route_replayed": cs.spec_report()["outcome"],
                                        "ms_per_call_with_timing_events": round(ms, 4), "kernels_ms": kt,
                                        "sweeps_ms": round(sweeps, 4), "serial_term_ms": round(sum(kt.values()) - sweeps, 4),
                                        "note": "sweeps_ms scales with 1/N; serial_term_ms (statistics, tables, CDFs, prediction, verdict, the replayed all-reduces' copies, gated launches) does "
                                                "not; an event pair per kernel adds ~10 us each to ms_per_call; no multi-GPU node was available to measure the curve itself"}
            del rgb_s
        finally:
            for c_ in cr:
                c_.close()
            group.close()
    except Exception as e:
        out["stripe_rank_model"] = {"error": f"{type(e).__name__}: {e}"}
    # (3c) the same for the RESIZED product (config 2 / 5's flow over row stripes: each rank's PCIe link carries 1/8 of the scene, the
    # product is 2048^2): eight ranks through sarpro_hip_stripe_run_resized_u16, the assembled raster compared with the one-piece flow's,
    # then rank 3 alone with its all-reduces (level chain x 2 bands, geometry, halo, floor histogram) replayed.
    try:
        import threading
        nr, target = 8, 2048
        r0s, nrs = sarpro_amd.host_stripe_plan(rows, nr)
        fc, fr = sarpro_amd.resize_output_dims(cols, rows, target, True)
        one = torch.zeros((fr * fc * 3,), dtype=torch.uint8, device=dev)
        with sarpro_amd.Context(dev.index) as c1:
            c1.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, target, True, one.data_ptr())
        group = sarpro_amd.LocalGroup(nr)
        cr = [sarpro_amd.Context(dev.index, timing=(k == 3)) for k in range(nr)]
        try:
            for k, c_ in enumerate(cr):
                c_.comm_init_local(group, k)
            want = [sarpro_amd.host_stripe_resized_rows(rows, cols, int(r0s[k]), int(nrs[k]), target, True) for k in range(nr)]
            sl = [torch.zeros((max(w[1], 1) * fc * 3,), dtype=torch.uint8, device=dev) for w in want]
            errs, res = [], [None] * nr

            def rank_call(k, c_=None):
                try:
                    r0, nrw = int(r0s[k]), int(nrs[k])
                    res[k] = (c_ or cr[k]).stripe_run_resized_u16(band[0].data_ptr() + r0 * pitch * 2, band[1].data_ptr() + r0 * pitch * 2, rows, cols, r0, nrw, pitch,
                                                                  St.Clahe, Mode.Default, target, True, sl[k].data_ptr())
                except Exception as e:
                    errs.append(f"rank {k}: {e}")

            def all_ranks():
                ths = [threading.Thread(target=rank_call, args=(k,)) for k in range(nr)]
                [x.start() for x in ths]
                [x.join() for x in ths]
            torch.cuda.synchronize()
            all_ranks()
            cr[3].set_attr("COMM_RECORD", 1)
            all_ranks()
            if errs:
                raise RuntimeError("; ".join(errs))
            got = torch.cat([t[: res[k][1] * fc * 3] for k, t in enumerate(sl)])
            equal = bool(torch.equal(got, one))
            cr[3].reset_attr("COMM_RECORD")
            for c_ in cr:
                c_.comm_destroy()
            cs = cr[3]
            cs.set_attr("COMM_REPLAY", 1)
            ms = timed(lambda: rank_call(3, cs), n=10, warm=3)
            if errs:
                raise RuntimeError("; ".join(errs))
            kt = {}
            for k, v in cs.last_kernel_times():
                if not k.startswith("host:"):
                    kt[k] = round(kt.get(k, 0.0) + v, 4)
            out["stripe_resized_rank_model"] = {"what": f"one rank of 8, alone on the GPU: rows [{int(r0s[3])}, {int(r0s[3]) + int(nrs[3])}) of the {rows}x{cols} scene through "
                                                        f"sarpro_hip_stripe_run_resized_u16 (CLAHE u8 x 2 -> Lanczos3 to {target}^2 -> pad -> suppressed synRGB), its all-reduces answered from the sums "
                                                        "recorded in a real 8-rank run (COMM_RECORD / COMM_REPLAY); the rank returns its rows of the product",
                                                "assembled_raster_of_the_8_rank_run_equals_the_one_piece_flow": equal,
                                                "rows_of_the_product_per_rank": [w[1] for w in want], "ms_per_call_with_timing_events": round(ms, 4), "kernels_ms": kt,
                                                }
            del sl, got
        finally:
            for c_ in cr:
                c_.close()
            group.close()
        del one
    except Exception as e:
        out["stripe_resized_rank_model"] = {"error": f"{type(e).__name__}: {e}"}
    # (4) several scenes in flight on the one GPU: one context (own stream, own workspaces) and one host thread per scene, as the batch
    # driver runs when a device is listed more than once -- the short dependent kernels of one scene's chain run beside another
    # scene's sweeps.  Beside the headline, which keeps ONE context and stream.
    try:
        import threading
        nmax = 3
        ctxs = [sarpro_amd.Context(dev.index) for _ in range(nmax)]
        rgbs = [torch.empty((rows, pitch * 3), dtype=torch.uint8, device=dev) for _ in range(nmax)]
        torch.cuda.synchronize()

        def worker(i, n):
            for _ in range(n):
                ctxs[i].dev_dualpol_synrgb_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgbs[i].data_ptr(), pitch)
        rec = {}
        for nw in range(1, nmax + 1):
            for n in (3, 20):  # (warm, timed)
                ths = [threading.Thread(target=worker, args=(i, n)) for i in range(nw)]
                t = time.perf_counter()
                [x.start() for x in ths]
                [x.join() for x in ths]
                dt = time.perf_counter() - t
            rec[str(nw)] = {"ms_per_scene": round(dt / (nw * 20) * 1e3, 3), "value": round(px * nw * 20 / dt / 1e6, 1)}
        same = all(bool(torch.equal(rgbs[0].view(rows, pitch, 3)[:, :cols], r.view(rows, pitch, 3)[:, :cols])) for r in rgbs[1:])  # (the pad columns are never written)
        out["scenes_in_flight"] = {"what": "scene A (CLAHE -> synRGB, resident in HBM) by 1, 2, 3 contexts at once on this GPU, one synchronous call per scene and thread; "
                                           "20 scenes per context", "unit": "Mpix/s", "by_contexts": rec, "rasters_equal": same}
        for c in ctxs:
            c.close()
    except Exception as e:
        out["scenes_in_flight"] = {"error": f"{type(e).__name__}: {e}"}
    return out


if __name__ == "__main__":
    main()
