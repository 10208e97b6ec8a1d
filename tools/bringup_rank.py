#!/usr/bin/env python3
"""One rank of the multi-GPU bring-up (tools/bringup_8gpu.sh), launched by torch.distributed.run with one rank per GPU.
  leg = comm    : sarpro_hip_comm_* self-test -- unique id from rank 0, every rank joins over RCCL, all-reduce of a u64 buffer whose
                  sum is known in closed form (rank r contributes r + 1 + i), twice (a second call on a warm communicator);
  leg = stripes : one 403 x 520 dual-pol scene as row stripes (the speculative fused route forced on: SAMPLED_HIST_MIN_PX = 0), every
                  strategy of the u16 chain, the assembled raster against the CPU oracle on rank 0 -- the test of
                  tests/test_gpu_multirank_local.py over REAL ranks on real devices instead of the in-process communicator.
A failing rank prints its sarpro_hip_last_error and exits non-zero; torch.distributed.run then tears the others down.
`--backend gloo --device-less` runs the protocol's host side only (no GPU: the CPU suite checks that the script still parses, joins
and reduces what it should)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("leg", choices=["comm", "stripes"])
ap.add_argument("--device-less", action="store_true")
args = ap.parse_args()
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
dist.init_process_group("gloo")  # the side channel (unique id, gathered rasters); the data path's collectives are the library's own


def fail(msg, ctx=None):
    err = ""
    if ctx is not None:
        from sarpro_amd._lib import lib
        err = (lib.sarpro_hip_last_error(ctx._h) or b"").decode()
    print(f"[bringup rank {rank}/{world}] FAILED: {msg} {err}", file=sys.stderr, flush=True)
    sys.exit(1)


if args.device_less:  # the protocol without devices: what every rank would contribute and what the sum must be
    n = 4096
    mine = torch.arange(n, dtype=torch.int64) + (rank + 1)
    dist.all_reduce(mine)
    want = world * torch.arange(n, dtype=torch.int64) + world * (world + 1) // 2
    assert torch.equal(mine, want)
    if rank == 0:
        print(f"bringup {args.leg}: device-less protocol check passed on {world} ranks", flush=True)
    dist.destroy_process_group()
    sys.exit(0)

import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
if local >= torch.cuda.device_count():
    fail(f"needs GPU {local}, this node has {torch.cuda.device_count()}")
torch.cuda.set_device(local)
with S.Context(local) as c:
    uid = [S.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    try:
        c.comm_init(world, rank, uid[0])
    except Exception as e:
        fail(f"comm_init: {e}", c)
    if args.leg == "comm":
        n = 65536 * 2  # the size of the path's largest all-reduce (the DN histograms of two bands)
        for rep in range(2):
            buf = (torch.arange(n, dtype=torch.int64, device="cuda") + (rank + 1)).contiguous()
            torch.cuda.synchronize()
            try:
                c.comm_allreduce_sum_u64(buf.data_ptr(), n)
            except Exception as e:
                fail(f"comm_allreduce_sum_u64 (call {rep}): {e}", c)
            want = world * torch.arange(n, dtype=torch.int64, device="cuda") + world * (world + 1) // 2
            if not torch.equal(buf, want):
                fail(f"all-reduce sum wrong at {int((buf != want).nonzero()[0])} (call {rep})", c)
        print(f"[bringup rank {rank}] comm self-test passed", flush=True)
    else:
        import oracle
        rows, cols, pitch = 403, 520, 576
        b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
        r0s, nrs = S.host_stripe_plan(rows, world)
        r0, nr = r0s[rank], nrs[rank]
        c.set_attr("SAMPLED_HIST_MIN_PX", 0); c.set_attr("SAMPLE_STRIDE", 5)
        for strategy in (St.Clahe, St.Robust, St.Tamed, St.Standard):
            d = []
            for x in b:
                t = torch.zeros((max(nr, 1), pitch), dtype=torch.int16, device="cuda")
                if nr:
                    t[:nr, :cols] = torch.from_numpy(x[r0:r0 + nr].view(np.int16)).cuda()
                d.append(t)
            rgb = torch.zeros((max(nr, 1), pitch * 3), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            try:
                c.stripe_run_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, r0, nr, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch)
            except Exception as e:
                fail(f"stripe_run_u16 {strategy.name}: {e}", c)
            mine = rgb.cpu().numpy().reshape(-1, pitch, 3)[:nr, :cols]
            parts = [None] * world
            dist.all_gather_object(parts, mine)
            if rank == 0:
                rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), int(strategy))
                got = np.concatenate(parts, axis=0)
                if rc != 0 or not np.array_equal(got, ref):
                    fail(f"stripes {strategy.name}: {int((got != ref).any(axis=2).sum())} pixels differ from the oracle")
                print(f"[bringup] {world} row stripes, {strategy.name}: == oracle", flush=True)
    c.comm_destroy()
dist.barrier()
dist.destroy_process_group()
