"""bench.py's N > 1 records (secondary.stripe, secondary.e2e_all_ranks, ranks_seen) driven on the CPU: two ranks under
torch.distributed.run over gloo, the GPU context replaced by a stand-in that records the calls.  What is checked is the code path
the driver's one multi-GPU command takes after the headline timing: rendezvous of the library communicator id, the stripe plan of
each rank, max-over-ranks timing, and that a leg that cannot run (no GPU here: the PCIe leg calls the real library) costs its own
record only."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import json, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import sarpro_amd
from sarpro_amd import AutoscaleStrategy as St
import bench

calls = []
class FakeCtx:
    def __init__(self, device=0, timing=False, async_dev=False):
        self._h = None
    def comm_init(self, n, r, uid):
        assert len(uid) == 128
        calls.append(("comm_init", n, r))
    def dev_synth_scene_u16(self, seed, band, q, rows, cols, row0, rows_local, ptr, pitch):
        calls.append(("synth", row0, rows_local))
    def stripe_run_u16(self, b1, b2, rows, cols, row0, rows_local, pitch, strategy, mode, rgb, rgb_pitch):
        calls.append(("stripe", row0, rows_local))
    def last_kernel_times(self):
        return [("dn_hist_u16", 0.1), ("allreduce_dn_hist", 0.02), ("allreduce_tile_hists", 0.01), ("allreduce_level_hist", 0.005), ("host:x", 1.0)]
    def close(self):
        pass
sarpro_amd.Context = FakeCtx
sarpro_amd.comm_unique_id = lambda: bytes(range(128))
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda: None
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
out = bench.multi_rank_records(torch, dist, torch.device("cpu"), rank, world, 1000, 640, St.Clahe, 1.5)
stripes = [c for c in calls if c[0] == "stripe"]
assert len(stripes) == 13 and all(c[1:] == stripes[0][1:] for c in stripes), stripes
allr = [None] * world
dist.all_gather_object(allr, stripes[0][1:])
if rank == 0:
    out["_stripes"] = allr
    print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()
''' % ROOT


def test_multi_rank_records_over_gloo(tmp_path):
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER)
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), str(drv)], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks_seen"] == 2
    st = out["stripe"]
    assert "error" not in st, st
    assert st["scaling"] == "strong" and st["ms_per_scene"] > 0 and st["speedup_vs_n1"] is not None and st["n1_ms_per_scene"] == 1.5
    assert set(st["allreduce_us_max_over_ranks"]) == {"allreduce_dn_hist", "allreduce_tile_hists", "allreduce_level_hist"}
    assert st["allreduce_us_max_over_ranks"]["allreduce_dn_hist"] == 20.0
    # the two ranks' stripes tile the scene: whole CLAHE tile rows, in rank order
    (r0a, na), (r0b, nb) = out["_stripes"]
    assert r0a == 0 and r0b == na and na + nb == 1000
    # no GPU here: the PCIe leg fails inside its own record
    assert "e2e_all_ranks" in out and ("error" in out["e2e_all_ranks"] or out["e2e_all_ranks"]["ms_per_scene_max_over_ranks"] > 0)


def test_headline_line_survives_a_failing_secondary(tmp_path):
    """The N = 1 secondary records come from a child process parked on its stdin; whatever it does, the parent's merge returns a record."""
    sys.path.insert(0, ROOT)
    import bench
    child = subprocess.Popen([sys.executable, "-c", "import sys; sys.stdin.readline(); sys.exit(3)"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True)
    assert "error" in bench.secondary_collect(child)
    child = subprocess.Popen([sys.executable, "-c", "import sys, json; assert sys.stdin.readline().strip() == 'go'; print(json.dumps({'e2e': 1}))"],
                             stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert bench.secondary_collect(child) == {"e2e": 1}
    assert "error" in bench.secondary_collect(OSError("spawn failed"))
