#!/usr/bin/env python3
"""Generates tests/golden/reference_prototypes.json: for every function include/tfhe_hip_backend.h declares
under a reference name (everything not prefixed hip_), the prototype the REFERENCE declares for it — return
type and parameter types in order, canonicalised by tools/c_prototypes.py — with the header it came from.
Sources: backends/tfhe-cuda-backend/cuda/include/**/*.h (what build.rs:78-137 feeds to bindgen) and
backends/tfhe-cuda-common/cuda/include/device.h (cuda_bind.rs:5-150).
Run in the build container (the GPU box has no /root/reference; tests read the committed JSON and, where the
reference tree is present, re-derive it and compare):  python tests/golden/make_prototypes.py"""
import glob
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, os.pardir, os.pardir))
sys.path.insert(0, ROOT)
from tools.c_prototypes import parse_prototypes  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(HERE, "reference_prototypes.json")


def reference_prototypes():
    headers = sorted(glob.glob(os.path.join(REF, "backends/tfhe-cuda-backend/cuda/include/**/*.h"), recursive=True))
    headers.append(os.path.join(REF, "backends/tfhe-cuda-common/cuda/include/device.h"))
    found = {}
    for h in headers:
        for name, (ret, params) in parse_prototypes(open(h).read()).items():
            found.setdefault(name, {"ret": ret, "params": params, "header": os.path.relpath(h, REF)})
    return found


def rust_declarations(text):
    """{name: 'pub fn name(args) -> ret;' with whitespace collapsed} for every extern fn of a bindings file"""
    out = {}
    for m in re.finditer(r"pub fn ([A-Za-z_]\w*)\s*\((.*?)\)\s*(->\s*[^;]+)?;", text, flags=re.S):
        args = re.sub(r"\s+", " ", m.group(2)).strip().rstrip(",").strip()
        ret = re.sub(r"\s+", " ", m.group(3) or "").strip()
        out[m.group(1)] = f"pub fn {m.group(1)}({args})" + (f" {ret}" if ret else "") + ";"
    return out


def reference_rust():
    found = {}
    for f in ("backends/tfhe-cuda-backend/src/bindings.rs", "backends/tfhe-cuda-common/src/cuda_bind.rs"):
        for name, decl in rust_declarations(open(os.path.join(REF, f)).read()).items():
            found.setdefault(name, {"rust": decl, "rust_file": f})
    return found


def in_scope_names():
    ours = parse_prototypes(open(os.path.join(ROOT, "include", "tfhe_hip_backend.h")).read())
    return sorted(n for n in ours if not n.startswith("hip_"))


def build():
    ref = reference_prototypes()
    missing = [n for n in in_scope_names() if n not in ref]
    if missing:
        raise SystemExit(f"declared under a reference name but absent from the reference headers: {missing}")
    rust = reference_rust()
    missing = [n for n in in_scope_names() if n not in rust]
    if missing:
        raise SystemExit(f"no Rust binding of the reference for: {missing}")
    return {n: dict(ref[n], **rust[n]) for n in in_scope_names()}


if __name__ == "__main__":
    data = build()
    json.dump({"_about": "reference prototypes of the in-scope entry points; see make_prototypes.py",
               "prototypes": data}, open(OUT, "w"), indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(data)} prototypes")
