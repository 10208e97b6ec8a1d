import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["SARPRO_HIP_F32_ZONES"] = "force"; os.environ["SARPRO_HIP_F32_ZONES_DEBUG"] = "1"
import numpy as np, f32data, oracle, sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd
x = f32data.ratio_scene(403, 520)
with S.Context(0, timing=True) as c:
    for st in (St.Standard, St.Clahe):
        out = c.process_scalar_data_pipeline(x, Bd.U8, st)[0]
        rc, ref = oracle.pipeline(x, 0, int(st))
        print(st.name, np.array_equal(out, ref), [n for n, _ in c.last_kernel_times()], flush=True)
