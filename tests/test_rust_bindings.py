"""The Rust crates ship as source (no rustc in the image): check them against the header the way a compiler + linker would.

* rust/sarpro-hip-sys/src/lib.rs declares every function of include/sarpro_hip.h with the same name, arity and types
  (the header is parsed here with a parser of this file's own, not the generator's);
* the generated file is up to date;
* every `sys::` item the safe crate (rust/sarpro-hip) uses exists, and every call passes the declared number of arguments;
* the safe crate's enums have the reference's variant names in the reference's order (types.rs:8-14,115-123,170-182 as
  restated by the header's discriminants and sarpro_amd.types)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sarpro_hip.h")
SYS_RS = os.path.join(ROOT, "rust", "sarpro-hip-sys", "src", "lib.rs")
SAFE_RS = os.path.join(ROOT, "rust", "sarpro-hip", "src", "lib.rs")

C2RUST = {"int": "c_int", "unsigned": "c_uint", "size_t": "usize", "uint64_t": "u64", "int64_t": "i64", "uint32_t": "u32", "uint16_t": "u16",
          "uint8_t": "u8", "int32_t": "i32", "float": "f32", "double": "f64", "char": "c_char", "void": "c_void"}


def header_functions():
    """{name: (ret, [ctype, ...])} with each ctype normalised to 'const T **'-style text"""
    src = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    src = re.sub(r"typedef\s+(?:struct|enum)\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    src = re.sub(r"typedef[^;]*;", " ", src)
    out = {}
    for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?\w+(?:\s*\*+|\s+))\s*(sarpro_hip_\w+)\s*\(([^()]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        for a in [x.strip() for x in args.split(",")]:
            if a in ("", "void"):
                continue
            arr = "[" in a
            a = re.sub(r"\[.*?\]", "", a)
            toks = re.findall(r"\w+|\*", a)
            toks = toks[:-1]  # the parameter's name
            if arr:
                toks.append("*")
            params.append(" ".join(toks))
        out[name] = (" ".join(re.findall(r"\w+|\*", ret)), params)
    return out


def to_rust(ctype):
    toks = ctype.split()
    stars = toks.count("*")
    const = "const" in toks
    base = [t for t in toks if t not in ("*", "const")]
    base = " ".join(base)
    r = C2RUST.get(base, base)
    if stars == 0:
        return r
    # 'const T *' -> *const T; 'T **' -> *mut *mut T; 'const T **' -> *mut *const T
    inner = ("*const " if const else "*mut ") + r
    for _ in range(stars - 1):
        inner = "*mut " + inner
    return inner


def rust_externs():
    src = open(SYS_RS).read()
    block = src[src.index('extern "C" {'):]
    out = {}
    for m in re.finditer(r"pub fn (sarpro_hip_\w+)\((.*?)\)( -> ([^;]+))?;", block):
        params = [p.split(":", 1)[1].strip() for p in m.group(2).split(", ") if p.strip()]
        out[m.group(1)] = (m.group(4).strip() if m.group(4) else "void", params)
    return out


def test_every_header_function_is_declared_with_the_same_name_arity_and_types():
    h, r = header_functions(), rust_externs()
    assert len(h) >= 80
    assert sorted(h) == sorted(r)
    for name, (ret, params) in h.items():
        rret, rparams = r[name]
        assert len(params) == len(rparams), name
        assert [to_rust(p) for p in params] == rparams, (name, [to_rust(p) for p in params], rparams)
        assert (to_rust(ret) if ret != "void" else "void") == rret, (name, ret, rret)


def test_header_parser_agrees_with_the_python_binding():
    from sarpro_amd import _lib
    assert sorted(header_functions()) == sorted(_lib.SYMBOLS)


def test_generated_file_is_up_to_date():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"]).returncode == 0


def split_top_level(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_safe_crate_calls_match_the_declarations():
    src = re.sub(r"//[^\n]*", "", open(SAFE_RS).read())
    sys_src = open(SYS_RS).read()
    externs = rust_externs()
    sys_items = set(re.findall(r"pub (?:const|struct|type|fn) (\w+)", sys_src))
    used = set(re.findall(r"\bsys::(\w+)", src))
    assert used and not (used - sys_items), used - sys_items
    calls = 0
    for m in re.finditer(r"\bsys::(sarpro_hip_\w+)\s*\(", src):
        name = m.group(1)
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = split_top_level(src[m.end():i - 1])
        assert len(args) == len(externs[name][1]), (name, len(args), len(externs[name][1]))
        calls += 1
    assert calls >= 40
    # function-pointer uses (Stripe::phase) name declared functions too
    for name in re.findall(r"self\.phase\(sys::(\w+)\)", src):
        assert externs[name][1] == ["*mut sarpro_hip_stripe", "*mut *mut u64", "*mut usize"]


def test_safe_crate_enums_keep_the_reference_order():
    from sarpro_amd import AutoscaleStrategy, BitDepth, PolarizationOperation, SyntheticRgbMode
    src = open(SAFE_RS).read()
    hdr = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    cases = {"AutoscaleStrategy": (AutoscaleStrategy, "SARPRO_STRATEGY_"), "BitDepth": (BitDepth, "SARPRO_BITDEPTH_"),
             "PolarizationOperation": (PolarizationOperation, "SARPRO_OP_"), "SyntheticRgbMode": (SyntheticRgbMode, "SARPRO_SYNRGB_")}
    for name, (py_enum, prefix) in cases.items():
        m = re.search(r"#\[repr\(i32\)\]\s*#\[derive\([^\]]*\)\]\s*pub enum %s \{([^}]*)\}" % name, src)
        assert m, name
        variants = [(k, int(v)) for k, v in re.findall(r"(\w+)\s*=\s*(\d+)", m.group(1))]
        assert variants == [(e.name, int(e)) for e in py_enum], name          # names and order of types.rs
        cvals = [(k, int(v)) for k, v in re.findall(r"(%s\w+)\s*=\s*(\d+)" % prefix, hdr)]
        assert [v for _, v in cvals] == [v for _, v in variants], name         # the C ABI's discriminants
        assert [k[len(prefix):].replace("_", "").lower() for k, _ in cvals] == [k.lower() for k, _ in variants], name


def test_safe_crate_keeps_the_reference_signatures():
    """pipeline.rs:42-46, autoscale.rs:710-714, synthetic_rgb.rs:182-187, api/mod.rs:803-857: argument order and result shapes"""
    src = " ".join(open(SAFE_RS).read().split())
    assert ("pub fn process_scalar_data_pipeline(processed: &Array2<f32>, bit_depth: BitDepth, strategy: AutoscaleStrategy) "
            "-> (DbImage, Vec<bool>, Vec<u8>, Option<Vec<u16>>)") in src
    assert "impl std::ops::Deref for DbImage { type Target = Array2<f64>;" in src
    assert "pub fn process_scalar_data_inplace(processed: &Array2<f32>) -> (Array2<f64>, Vec<bool>)" in src
    assert "pub fn autoscale_db_image_tamed_synrgb_u8(db: &DbImage, valid_mask: &[bool], is_copol: bool) -> Vec<u8>" in src
    assert ("pub fn create_synthetic_rgb_by_mode_and_strategy(mode: SyntheticRgbMode, strategy: AutoscaleStrategy, band1_data: &[u8], "
            "band2_data: &[u8]) -> Vec<u8>") in src
    assert re.search(r"pub fn save_image<[^(]*>\(processed: &Array2<f32>, output: &Path, format: OutputFormat, bit_depth: BitDepth, "
                     r"target_size: Option<usize>, metadata: Option<&M>, pad: bool, autoscale: AutoscaleStrategy, operation: ProcessingOperation, ", src)
    assert re.search(r"pub fn save_multiband_image<[^(]*>\(processed1: &Array2<f32>, processed2: &Array2<f32>, output: &Path, format: OutputFormat, "
                     r"bit_depth: BitDepth, target_size: Option<usize>, metadata: Option<&M>, pad: bool, autoscale: AutoscaleStrategy, "
                     r"operation: ProcessingOperation, ", src)
    for f in ("sum_arrays", "difference_arrays", "ratio_arrays", "normalized_diff_arrays", "log_ratio_arrays"):
        assert f"pub fn {f}(a: &Array2<f32>, b: &Array2<f32>) -> Array2<f32>" in src
    for field in ("width: usize", "height: usize", "bit_depth: BitDepth", "format: OutputFormat", "gray: Option<Vec<u8>>", "gray16: Option<Vec<u16>>",
                  "rgb: Option<Vec<u8>>", "gray_band2: Option<Vec<u8>>", "gray16_band2: Option<Vec<u16>>"):
        assert f"pub {field}" in src
    assert "pub struct BatchReport { pub processed: usize, pub skipped: usize, pub errors: usize }" in src
