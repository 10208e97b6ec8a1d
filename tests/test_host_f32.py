"""CPU suite, f32 flavour: the host-built threshold tables (+ numpy binary search = the kernel)
reproduce the oracle bit for bit, and glibc log10/pow are monotone where the tables rely on it."""
import numpy as np
import pytest

import emul
import f32data
import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd

CASES = [("ratio", lambda: f32data.ratio_scene(150, 210)),
         ("resampled", lambda: f32data.resampled_scene(128, 96)),
         ("nasty", lambda: f32data.nasty_scene(90, 131))]


@pytest.mark.parametrize("name,make", CASES)
@pytest.mark.parametrize("strategy", list(St))
@pytest.mark.parametrize("bit_depth", list(Bd))
def test_threshold_tables_reproduce_oracle(name, make, strategy, bit_depth):
    x = make()
    got, st = emul.f32_pipeline(x, bit_depth, strategy)
    rc, ref, so = oracle.pipeline(x, int(bit_depth), int(strategy), want_stats=True)
    assert rc == 0
    assert np.array_equal(got, ref), f"{(got != ref).sum()} px differ"
    for k in ("valid_count", "min_db", "max_db", "median_db", "p01", "p25", "p75", "p99", "low_clip", "high_clip", "gamma"):
        assert getattr(st, k) == getattr(so, k), k


def test_valid_threshold_is_the_first_float_above_minus_50_db():
    t = np.float32(S.host_f32_valid_threshold())
    below = np.nextafter(t, np.float32(0))
    db, mask = oracle.db_mask(np.array([[below, t]], np.float32))
    assert list(mask.ravel()) == [0, 1]


def test_thresholds_are_sorted_and_tight():
    x = f32data.ratio_scene(100, 100)
    db, mask = oracle.db_mask(x)
    so = oracle.stats(db, mask)
    thr = S.host_f32_bin4096_thresholds(so.min_db, so.max_db)
    fin = thr[1:][np.isfinite(thr[1:])]
    assert np.all(np.diff(fin) >= 0)
    # tight: the float just below thr[k] has a smaller index than thr[k]
    span = so.max_db - so.min_db
    def idx(v):
        d = 10.0 * np.log10(np.maximum(v.astype(np.float64), 1e-10))
        t = np.clip((d - so.min_db) * (1.0 / span), 0.0, 1.0)
        return np.minimum((t * 4096.0).astype(np.int64), 4095)
    ks = np.arange(1, 4096)[np.isfinite(thr[1:])]
    at = idx(thr[ks])
    below = idx(np.nextafter(thr[ks], np.float32(0)))
    assert np.all(at >= ks) and np.all(below < ks)


def test_glibc_log10_is_monotone_on_sampled_floats():
    # the threshold construction assumes weak monotonicity of the reference's dB formula
    rng = np.random.default_rng(1)
    bits = np.sort(rng.integers(0x00800000, 0x7F7FFFFF, 2_000_000, dtype=np.int64)).astype(np.uint32)
    v = bits.view(np.float32)
    db, _ = oracle.db_mask(v.reshape(1, -1))
    assert np.all(np.diff(db.ravel()) >= 0)
    # and on runs of consecutive floats
    for start in (0x3F800000, 0x42C80000, 0x3A000000):
        run = (np.arange(200000, dtype=np.uint32) + np.uint32(start)).view(np.float32)
        d, _ = oracle.db_mask(run.reshape(1, -1))
        assert np.all(np.diff(d.ravel()) >= 0)
