"""world_size-2 gloo test (CPU): the row-stripe reduction protocol.  Each rank holds a row stripe,
produces its local integer histograms (numpy stand-ins for the kernels, tests/emul.py), merges them
with torch.distributed all-reduce(sum) exactly where the GPU path all-reduces over RCCL, then runs
the product's host half.  The N-rank RGB must equal the 1-rank RGB and the oracle, bit for bit."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, strategy, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import emul
    import sarpro_amd as S
    from sarpro_amd import AutoscaleStrategy as St, synth

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = 211, 160
    r0s, nrs = S.host_stripe_plan(rows, world)
    r0, nr = r0s[rank], nrs[rank]
    b = [synth.scene_u16(rows, cols, k, row0=r0, rows_local=nr) for k in (0, 1)]

    def allreduce(a):
        t = torch.from_numpy(a.astype(np.int64))  # sums of counts: exact in int64
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy().astype(np.uint64)

    rgb, _, _ = emul.dualpol_synrgb(b[0], b[1], St(strategy), rows_total=rows, row0=r0, reduce=allreduce)
    np.save(os.path.join(out_dir, f"rgb_{rank}.npy"), rgb)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strategy", [4, 5, 1])  # Clahe, Tamed, Robust
def test_two_rank_stripes_equal_single_rank_and_oracle(tmp_path, strategy):
    import torch.multiprocessing as mp

    import oracle
    from sarpro_amd import synth

    port = 29500 + (os.getpid() % 2000) + strategy
    mp.spawn(_worker, args=(2, port, strategy, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"rgb_{r}.npy") for r in range(2)], axis=0)
    rows, cols = 211, 160
    b = [synth.scene_u16(rows, cols, k).astype(np.float32) for k in (0, 1)]
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0], b[1], strategy)
    assert rc == 0 and np.array_equal(got, ref)


# ----------------------------------------------------------------------------- f32 flavour (sarpro_hip_stripe_*_f32)
def _worker_f32(rank, world, port, strategy, bit_depth, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    import emul
    import f32data
    import sarpro_amd as S
    from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd

    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, cols = 211, 160
    r0s, nrs = S.host_stripe_plan(rows, world)
    r0, nr = r0s[rank], nrs[rank]
    x = f32data.ratio_scene(rows, cols)[r0:r0 + nr]

    def allreduce(a):
        t = torch.from_numpy(a.astype(np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy().astype(np.uint64).reshape(a.shape)

    def gather(p):
        # the partials travel the way sarpro_hip_stripe_run_f32 ships them: an all-reduce(sum) of a u64 buffer that is zero
        # outside the rank's own 4-word slot
        buf = np.zeros((world, 4), np.uint64)
        buf[rank] = np.frombuffer(bytes(p), np.uint64)
        t = torch.from_numpy(buf.view(np.int64))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [S.F32Partial.from_buffer_copy(t.numpy()[r].tobytes()) for r in range(world)]

    out, st = emul.f32_pipeline(x, Bd(bit_depth), St(strategy), rows_total=rows, row0=r0, reduce=allreduce, gather=gather)
    np.save(os.path.join(out_dir, f"out_{rank}.npy"), out)
    np.save(os.path.join(out_dir, f"st_{rank}.npy"), np.array([st.valid_count, st.min_db, st.max_db, st.low_clip, st.high_clip]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("strategy,bit_depth", [(4, 0), (4, 1), (1, 0), (0, 1)])  # Clahe u8 / u16, Robust u8, Standard u16
def test_two_rank_f32_stripes_equal_oracle(tmp_path, strategy, bit_depth):
    import torch.multiprocessing as mp

    import f32data
    import oracle

    port = 31500 + (os.getpid() % 2000) + strategy * 2 + bit_depth
    mp.spawn(_worker_f32, args=(2, port, strategy, bit_depth, str(tmp_path)), nprocs=2, join=True)
    got = np.concatenate([np.load(tmp_path / f"out_{r}.npy") for r in range(2)], axis=0)
    rc, ref, so = oracle.pipeline(f32data.ratio_scene(211, 160), bit_depth, strategy, want_stats=True)
    assert rc == 0 and np.array_equal(got, ref)
    for r in range(2):
        st = np.load(tmp_path / f"st_{r}.npy")
        assert list(st) == [so.valid_count, so.min_db, so.max_db, so.low_clip, so.high_clip]
