"""CPU-side sanitizer runs (AddressSanitizer + UBSan, g++): the TIFF reader parses foreign files, so it is fuzzed
under the sanitizers; the host logic (tables, statistics, thresholds) is exercised the same way.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sarpro_amd", "csrc")
FLAGS = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
         "-D_FILE_OFFSET_BITS=64", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_tiff_reader_survives_truncated_and_corrupted_files_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_tiff_fuzz")
    subprocess.check_call(["g++", *FLAGS, os.path.join(ROOT, "tests", "native", "asan_tiff_fuzz.cpp"),
                           os.path.join(CSRC, "tiff_io.cpp"), "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "fuzz done" in p.stdout


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_logic_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_host_logic")
    subprocess.check_call(["g++", *FLAGS, "-ffp-contract=off", os.path.join(ROOT, "tests", "native", "asan_host_logic.cpp"),
                           os.path.join(CSRC, "host_logic.cpp"), "-lpthread", "-o", exe])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "host logic done" in p.stdout
