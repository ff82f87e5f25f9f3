#!/usr/bin/env python3
"""Parses C function prototypes out of header text into a canonical form, so that the prototypes of
include/tfhe_hip_backend.h can be compared — return type, parameter types AND their order — with the
prototypes the reference's Rust FFI binds (backends/tfhe-cuda-backend/cuda/include/**/*.h,
backends/tfhe-cuda-common/cuda/include/device.h).  Also the source of the generated Rust `extern "C"` block
of backends/tfhe-hip-backend/src/bindings.rs (tools/gen_rust_bindings.py)."""
import re

_KEYWORDS = {"const", "void", "bool", "int", "unsigned", "char", "enum", "struct", "float", "double", "long"}


def _strip(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#[^\n]*", " ", text, flags=re.M)
    return text


def canonical_type(tokens):
    """['void', 'const', '*'] / ['const', 'void', '*'] -> 'const void *'; enum/struct keywords dropped;
    `T *const *` keeps its inner const."""
    toks = [t for t in tokens if t not in ("enum", "struct")]
    # split into base (up to the first '*') and pointer suffix
    if "*" in toks:
        i = toks.index("*")
        base, suffix = toks[:i], toks[i:]
    else:
        base, suffix = toks, []
    const = "const" in base
    base = [t for t in base if t != "const"]
    out = (["const"] if const else []) + base + suffix
    return " ".join(out)


def parse_prototypes(text):
    """-> {name: (ret_type, [param_type, ...])} for every top-level `ret name(params);`"""
    text = _strip(text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b([A-Za-z_]\w*)\s*\(([^;{}()]*)\)\s*;", text):
        ret, name, params = m.group(1), m.group(2), m.group(3)
        ret_toks = re.findall(r"[A-Za-z_]\w*|\*", ret)
        if not ret_toks or ret_toks[0] in ("typedef", "return", "else") or name in _KEYWORDS:
            continue
        ret_toks = [t for t in ret_toks if t not in ("extern", "static", "inline")]
        plist = []
        params = params.strip()
        if params and params != "void":
            for p in params.split(","):
                toks = re.findall(r"[A-Za-z_]\w*|\*", p)
                # drop the parameter name: the last identifier, unless the parameter is unnamed
                idents = [t for t in toks if t != "*"]
                if len(idents) >= 2 and toks[-1] != "*" and toks[-1] not in _KEYWORDS:
                    toks = toks[:-1]
                plist.append(canonical_type(toks))
        protos[name] = (canonical_type(ret_toks), plist)
    return protos


_RUST = {"void": "()", "bool": "bool", "int": "ffi::c_int", "uint32_t": "u32", "uint64_t": "u64",
         "int8_t": "i8", "float": "f32", "double": "f64", "char": "ffi::c_char"}


def rust_type(ctype):
    toks = ctype.split()
    const = toks[0] == "const"
    if const:
        toks = toks[1:]
    base, stars = toks[0], toks[1:]
    inner_const = "const" in stars
    stars = [t for t in stars if t == "*"]
    r = _RUST.get(base, base)   # enums / FFI structs keep their C name (bindgen does the same)
    if not stars:
        return r
    if base == "void":
        r = "ffi::c_void"
    # `T *const *` : outer pointer to const pointer
    for i, _ in enumerate(stars):
        is_const = const if i == 0 else inner_const
        r = ("*const " if is_const else "*mut ") + r
    return r
