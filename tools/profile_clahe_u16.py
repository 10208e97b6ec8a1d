"""A few CLAHE u16-output calls of one 400 MP band (BASELINE config 3(i)) with nothing else in the process: the target of the
rocprofv3 --pmc passes on k_clahe_apply_u16 (tools/pmc_apply.sh with PROFILE_SCRIPT set)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import synth
rows = cols = 20000; pitch = 20032
ctx = S.Context(0); q = synth.q_tables()
band = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
out = torch.empty((rows, pitch), dtype=torch.int16, device="cuda")
ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, 0, q, rows, cols, 0, rows, band.data_ptr(), pitch)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    ctx.dev_autoscale_band_u16(band.data_ptr(), rows, cols, pitch, S.AutoscaleStrategy.Clahe, S.BitDepth.U16, out.data_ptr(), pitch)
