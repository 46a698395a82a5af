#!/usr/bin/env python3
"""Generate shim/perceive-core/src/ffi.rs — the `extern "C"` block of the Rust shim — from
include/perceive_hip.h, so that the two cannot drift (tests/test_rust_shim.py re-parses both with its
own parsers and compares symbol sets, argument counts and integer widths).

    python tools/gen_rust_ffi.py            # rewrites the file
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "perceive_hip.h")
OUT = os.path.join(ROOT, "shim", "perceive-core", "src", "ffi.rs")

SCALARS = {
    "pcv_status": "c_int", "int": "c_int", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64", "uint64_t": "u64",
    "size_t": "usize", "float": "f32", "double": "f64", "uint8_t": "u8", "char": "c_char", "void": "c_void",
    "pcv_ctx": "pcv_ctx", "pcv_searcher": "pcv_searcher", "pcv_model": "pcv_model", "pcv_tokenizer": "pcv_tokenizer",
    "pcv_comm": "pcv_comm", "pcv_hit": "pcv_hit", "pcv_scan_stats": "pcv_scan_stats", "pcv_model_desc": "pcv_model_desc",
    "pcv_encode_stats": "pcv_encode_stats",
}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(ctype):
    """`const int64_t*` -> `*const i64`, `pcv_ctx**` -> `*mut *mut pcv_ctx`, `const char* const*` -> `*const *const c_char`."""
    t = ctype.strip()
    arr = re.search(r"\[\d*\]$", t)
    if arr:
        t = t[: arr.start()].strip() + "*"
    toks = re.findall(r"const|\*|[A-Za-z_][A-Za-z0-9_]*", t)
    base = [x for x in toks if x not in ("const", "*")][0]
    # walk the declarator left to right: each '*' wraps what is to its left; `const` binds to what precedes it
    # (or to the base when it leads)
    out = SCALARS[base]
    const = toks[0] == "const" or (len(toks) > 1 and toks[1] == "const" and toks[0] == base)
    i = toks.index(base) + 1
    if i < len(toks) and toks[i] == "const":
        const = True
        i += 1
    while i < len(toks):
        assert toks[i] == "*", ctype
        nxt_const = i + 1 < len(toks) and toks[i + 1] == "const"
        out = ("*const " if const else "*mut ") + out
        const = nxt_const
        i += 2 if nxt_const else 1
    return out


def split_param(p):
    p = p.strip()
    m = re.match(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[\d*\])?$", p)
    ctype, name, arr = m.group(1), m.group(2), m.group(3) or ""
    return ctype.strip() + arr, name


def functions(text):
    text = strip_comments(text)
    text = re.sub(r"typedef struct \w+ \{.*?\} \w+;", "", text, flags=re.S)
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(pcv_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, params = m.group(1).strip(), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("", "void") else [split_param(p) for p in params.split(",")]
        yield ret, name, plist


def structs(text):
    text = strip_comments(text)
    for m in re.finditer(r"typedef struct (\w+) \{(.*?)\} \w+;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            ctype, names = decl.split(None, 1) if not decl.startswith("const") else decl.rsplit(None, 1)
            for n in names.split(","):
                fields.append((ctype, n.strip()))
        yield m.group(1), fields


def callbacks(text):
    """`typedef int (*pcv_x)(void* user, ...);` -> (ret, name, [(ctype, pname)])"""
    text = strip_comments(text)
    for m in re.finditer(r"typedef\s+([\w \*]+?)\(\s*\*\s*(pcv_\w+)\s*\)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        params = " ".join(m.group(3).split())
        yield m.group(1).strip(), m.group(2), [split_param(p) for p in params.split(",")]


def enums(text):
    text_nc = strip_comments(text)
    for m in re.finditer(r"enum\s*\{(.*?)\};", text_nc, flags=re.S):
        for item in m.group(1).split(","):
            item = item.strip()
            if item:
                name, value = [x.strip() for x in item.split("=")]
                yield name, int(value)


KEYWORDS = {"type", "ref", "in", "match", "move", "async"}


def main():
    text = open(HEADER).read()
    lines = [
        "//! `extern \"C\"` binding of libperceive_hip.so — GENERATED from include/perceive_hip.h by",
        "//! tools/gen_rust_ffi.py; do not edit.  One declaration per symbol of the header, same order.",
        "//! NOT COMPILED in the build environment of this repository (it has no Rust toolchain):",
        "//! tests/test_rust_shim.py checks it mechanically against the header instead (symbol set, argument",
        "//! counts, integer widths, struct fields).",
        "#![allow(non_camel_case_types, dead_code)]",
        "use std::os::raw::{c_char, c_int, c_void};",
        "",
    ]
    for h in ("pcv_ctx", "pcv_searcher", "pcv_model", "pcv_tokenizer", "pcv_comm"):
        lines += [f"/// opaque handle", "#[repr(C)]", f"pub struct {h} {{", "    _private: [u8; 0],", "}"]
    lines.append("")
    for name, fields in structs(text):
        lines += ["#[repr(C)]", "#[derive(Debug, Clone, Copy, Default)]", f"pub struct {name} {{"]
        for ctype, fname in fields:
            lines.append(f"    pub {fname}: {SCALARS[ctype]},")
        lines += ["}", ""]
    for name, value in enums(text):
        lines.append(f"pub const {name}: c_int = {value};")
    for m in re.finditer(r"#define\s+(PCV_\w+)\s+INT64_MIN\b", strip_comments(text)):
        lines.append(f"pub const {m.group(1)}: i64 = i64::MIN;")
    lines.append("")
    for ret, name, params in callbacks(text):
        SCALARS[name] = name
        args = ", ".join(f"{n}: {rust_type(t)}" for t, n in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"pub type {name} = Option<unsafe extern \"C\" fn({args}){r}>;")
    lines += ["", "#[link(name = \"perceive_hip\")]", "extern \"C\" {"]
    for ret, name, params in functions(text):
        args = ", ".join(f"{(n + '_') if n in KEYWORDS else n}: {rust_type(t)}" for t, n in params)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        lines.append(f"    pub fn {name}({args}){r};")
    lines += ["}", ""]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write("\n".join(lines))
    print(f"wrote {OUT}: {sum(1 for _ in functions(text))} functions")


if __name__ == "__main__":
    main()
