#!/usr/bin/env python3
"""Attempt: two processes, BOTH on GPU 0, joined by the library's RCCL communicator (the only way to execute a world-size-2
RCCL all-reduce on a 1-GPU box, if RCCL accepts two ranks on one device).  Each rank runs sarpro_hip_stripe_run_u16 /
_run_f32 on its half of a scene; rank 0 compares the concatenated stripes with the one-piece result.
launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/two_rank_one_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SyntheticRgbMode as Mode, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")                 # CPU side channel for the unique id and the result
torch.cuda.set_device(0)
rows, cols, pitch = 2400, 1984, 1984
q = synth.q_tables()
r0s, nrs = S.host_stripe_plan(rows, world)
row0, nr = r0s[rank], nrs[rank]
with S.Context(0) as c:
    uid = [S.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    c.comm_init(world, rank, uid[0])
    band = [torch.zeros((nr, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    for b in range(2):
        c.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, row0, nr, band[b].data_ptr(), pitch)
    out = {}
    for strategy in (St.Clahe, St.Robust):
        rgb = torch.zeros((nr, pitch * 3), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.stripe_run_u16(band[0].data_ptr(), band[1].data_ptr(), rows, cols, row0, nr, pitch, strategy, Mode.Default, rgb.data_ptr(), pitch)
        out[("u16", strategy.name)] = rgb.cpu().numpy().reshape(nr, pitch, 3)[:, :cols]
        f = band[0].to(torch.float32); f[f < 0] += 65536.0; f = f.contiguous()
        o = torch.zeros((nr, pitch), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        c.stripe_run_f32(f.data_ptr(), rows, cols, row0, nr, pitch, strategy, Bd.U8, o.data_ptr(), pitch)
        out[("f32", strategy.name)] = o.cpu().numpy()[:, :cols]
    gathered = [None] * world
    dist.all_gather_object(gathered, out)
    c.comm_destroy()
if rank == 0:
    ok = True
    with S.Context(0) as c1:
        b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
        for strategy in (St.Clahe, St.Robust):
            ref = c1.dualpol_synrgb(b[0], b[1], strategy)
            got = np.concatenate([g[("u16", strategy.name)] for g in gathered], axis=0)
            e1 = np.array_equal(got, ref)
            ref8 = c1.process_scalar_data_pipeline(b[0].astype(np.float32), Bd.U8, strategy)[0]
            got8 = np.concatenate([g[("f32", strategy.name)] for g in gathered], axis=0)
            e2 = np.array_equal(got8, ref8)
            print(f"{strategy.name}: 2-rank RCCL stripes == one piece: u16 dual-pol {e1}, f32 band {e2}", flush=True)
            ok &= e1 and e2
    print("TWO-RANK RCCL ON ONE GPU:", "PASS" if ok else "FAIL", flush=True)
dist.barrier()
dist.destroy_process_group()
