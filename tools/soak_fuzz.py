"""Extended run of the randomised parity tests (tests/test_gpu_fuzz.py) over seed ranges far beyond the suite's.
usage: python tools/soak_fuzz.py [first_offset] [count]   -- seeds offset .. offset+count-1 of every fuzz test"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import sarpro_amd as S
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = S.Context(0)
t0 = time.time(); bad = 0
for name in ("test_random_shapes_single_band", "test_random_shapes_dualpol", "test_random_f32_bands", "test_random_medium_shapes_clahe"):
    fn = getattr(F, name)
    fn = getattr(fn, "__wrapped__", fn)
    n = count if "medium" not in name else max(count // 8, 1)
    for seed in range(first, first + n):
        try:
            fn(ctx, seed)
        except AssertionError as e:
            bad += 1
            print("MISMATCH", name, seed, str(e)[:200], flush=True)
    print(f"{name}: seeds {first}..{first + n - 1} done, {time.time() - t0:.0f} s", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
