#!/usr/bin/env python3
"""Writes rust/sarpro-hip-sys/src/lib.rs from include/sarpro_hip.h: every struct, constant, callback type and function of
the header as a Rust `extern "C"` item (the image has no rustc, so the crate ships as source; tests/test_rust_bindings.py
parses the result back and checks it against the header with a parser of its own).

    python tools/gen_rust_sys.py            # rewrites the file
    python tools/gen_rust_sys.py --check    # exit 1 when the file is stale
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sarpro_hip.h")
OUT = os.path.join(ROOT, "rust", "sarpro-hip-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "size_t": "usize", "uint64_t": "u64", "uint32_t": "u32",
           "uint16_t": "u16", "uint8_t": "u8", "int32_t": "i32", "int64_t": "i64", "float": "f32", "double": "f64", "char": "c_char",
           "void": "c_void", "unsigned long long": "u64"}
KEYWORDS = {"in": "input", "type": "kind", "ref": "reference", "box": "bbox", "fn": "func", "mod": "module", "use": "used"}


def strip_comments(src):
    return re.sub(r"/\*.*?\*/", " ", src, flags=re.S)


def rust_type(ctype, known):
    """C type (without the declarator's name, pointers included as '*') -> Rust type."""
    t = " ".join(ctype.replace("*", " * ").split())
    toks = t.split(" ")
    # peel pointers from the right: "const T * const *" etc.
    def parse(toks):
        if toks and toks[-1] == "const" and "*" in toks:
            toks = toks[:-1]  # const applied to the pointer itself: irrelevant for the ABI
        if toks and toks[-1] == "*":
            inner = toks[:-1]
            const = False
            base = [x for x in inner if x != "*"]
            # pointer to const iff the pointee (right before this '*') is const-qualified
            if "*" in inner:
                # pointee is itself a pointer: const only if 'const' directly precedes this '*'
                const = inner[-1] == "const"
                return ("*const " if const else "*mut ") + parse(inner[:-1] if const else inner)
            const = "const" in base
            return ("*const " if const else "*mut ") + parse([x for x in inner if x != "const"])
        name = " ".join(x for x in toks if x != "const")
        if name in SCALARS:
            return SCALARS[name]
        if name in known:
            return name
        raise ValueError(f"unknown C type {ctype!r}")
    return parse(toks)


def split_decl(decl):
    """'const uint16_t *band1' / 'uint8_t uid_out[128]' -> (ctype, name, array_len or None)"""
    decl = decl.strip()
    m = re.match(r"^(.*?)(\w+)\s*(\[\s*(\w*)\s*\])?$", decl, flags=re.S)
    ctype, name, arr = m.group(1).strip(), m.group(2), m.group(3)
    return ctype, name, (m.group(4) if arr else None)


def parse_header(src):
    src = strip_comments(src)
    consts = re.findall(r"^#define\s+(SARPRO_HIP_\w+)\s+\(?(-?\d+)(u?)\)?\s*$", src, flags=re.M)
    enums = []
    for body, name in re.findall(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        enums.append((name, [(k, int(v)) for k, v in re.findall(r"(\w+)\s*=\s*(-?\d+)", body)]))
    opaque = re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)
    callbacks = []
    for ret, name, args in re.findall(r"typedef\s+(\w+)\s*\(\s*\*\s*(\w+)\s*\)\s*\((.*?)\)\s*;", src, flags=re.S):
        callbacks.append((name, ret, [split_decl(a) for a in args.split(",")]))
    structs = []
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for stmt in body.split(";"):
            stmt = stmt.strip()
            if not stmt:
                continue
            first, *rest = [x.strip() for x in stmt.split(",")]
            ctype, fname, arr = split_decl(first)
            base = ctype.replace("*", "").strip()
            fields.append((ctype, fname, arr))
            for r in rest:  # 'a, *b, c[4]' share the base type
                stars = "*" * r.count("*")
                _, n2, arr2 = split_decl(r.replace("*", ""))
                fields.append((base + " " + stars if stars else base, n2, arr2))
        structs.append((name, fields))
    body = re.sub(r"typedef\s+(struct|enum)\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"typedef[^;{]*;", " ", body)
    body = re.sub(r"^\s*#.*$", " ", body, flags=re.M)
    body = re.sub(r'extern\s+"C"\s*\{', " ", body)
    funcs = []
    for ret, name, args in re.findall(r"([\w\s\*]+?)\b(sarpro_hip_\w+)\s*\(([^;{}()]*)\)\s*;", body, flags=re.S):
        ret = " ".join(ret.split())
        params = [] if args.strip() in ("", "void") else [split_decl(a) for a in args.split(",")]
        funcs.append((name, ret, params))
    return consts, enums, opaque, callbacks, structs, funcs


def param_type(ctype, arr, known):
    t = rust_type(ctype, known)
    if arr is not None:  # array parameter decays to a pointer
        return ("*const " if "const" in ctype.split() else "*mut ") + t
    return t


def ident(n):
    return KEYWORDS.get(n, n)


def generate():
    consts, enums, opaque, callbacks, structs, funcs = parse_header(open(HEADER).read())
    known = set(opaque) | {s for s, _ in structs} | {c for c, _, _ in callbacks}
    o = []
    o.append("//! Raw bindings of `include/sarpro_hip.h` (libsarpro_hip.so: the MI355X raster core of sarpro).")
    o.append("//! GENERATED by tools/gen_rust_sys.py from the header -- do not edit; the safe surface with sarpro's own")
    o.append("//! signatures (`process_scalar_data_pipeline`, `save_multiband_image`, ...) lives in the `sarpro-hip` crate.")
    o.append("#![allow(non_camel_case_types, clippy::too_many_arguments)]")
    o.append("")
    o.append("use std::os::raw::{c_char, c_int, c_uint, c_void};")
    o.append("")
    for name, val, u in consts:
        o.append(f"pub const {name}: {'c_uint' if u else 'c_int'} = {val};")
    o.append("")
    for name, items in enums:
        o.append(f"// {name}")
        for k, v in items:
            o.append(f"pub const {k}: c_int = {v};")
    o.append("")
    for name in opaque:
        o.append("#[repr(C)]")
        o.append(f"pub struct {name} {{ _private: [u8; 0] }}")
    o.append("")
    for name, ret, args in callbacks:
        a = ", ".join(f"{ident(n)}: {param_type(t, arr, known)}" for t, n, arr in args)
        o.append(f"pub type {name} = Option<unsafe extern \"C\" fn({a}) -> {rust_type(ret, known)}>;")
    o.append("")
    for name, fields in structs:
        o.append("#[repr(C)]")
        o.append("#[derive(Debug, Clone, Copy)]")
        o.append(f"pub struct {name} {{")
        for t, n, arr in fields:
            rt = rust_type(t, known)
            o.append(f"    pub {ident(n)}: {f'[{rt}; {arr}]' if arr is not None else rt},")
        o.append("}")
    o.append("")
    o.append('extern "C" {')
    for name, ret, params in funcs:
        a = ", ".join(f"{ident(n)}: {param_type(t, arr, known)}" for t, n, arr in params)
        r = "" if ret == "void" else f" -> {rust_type(ret, known)}"
        o.append(f"    pub fn {name}({a}){r};")
    o.append("}")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == text else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
