"""BASELINE config 2 as a device-resident flow, a few scenes with nothing else in the process: the target of rocprofv3 runs on the
resize kernels (rocprofv3 --kernel-trace --stats | --pmc ... -- python3 tools/profile_resize.py <scenes>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, synth, resize_output_dims
rows = cols = 20000; pitch = 20032
ctx = S.Context(0); q = synth.q_tables()
band = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
for b in range(2):
    ctx.dev_synth_scene_u16(synth.SEED_SCENE_A, b, q, rows, cols, 0, rows, band[b].data_ptr(), pitch)
fc, fr = resize_output_dims(cols, rows, 2048, True)
rgb = torch.empty((fr * fc * 3,), dtype=torch.uint8, device="cuda")
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ctx.dev_dualpol_synrgb_resized(band[0].data_ptr(), band[1].data_ptr(), rows, cols, pitch, St.Robust, 2048, True, rgb.data_ptr())
