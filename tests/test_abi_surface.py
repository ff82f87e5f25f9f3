"""CPU-only checks of the drop-in boundary: the product shared object must export every symbol
that include/tfhe_hip_backend.h declares (no compute calls here — there is no GPU), the ctypes
binding must cover exactly that set, and the product package must not depend on the oracle."""
import os
import re
import subprocess

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.pardir))
HEADER = os.path.join(ROOT, "include", "tfhe_hip_backend.h")
LIB = os.path.join(ROOT, "tfhe_rs_amd", "lib", "libtfhe_hip_backend.so")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b((?:cuda|scratch_cuda|cleanup_cuda|has_support_to_cuda|hip)_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_reference_entry_points():
    syms = declared_symbols()
    for must in ("cuda_create_stream_ffi", "cuda_malloc_async", "cuda_drop",
                 "cuda_convert_lwe_programmable_bootstrap_key_64_async",
                 "scratch_cuda_programmable_bootstrap_64_async", "cuda_programmable_bootstrap_64_async",
                 "cleanup_cuda_programmable_bootstrap_64", "cuda_multi_bit_programmable_bootstrap_64_async",
                 "cuda_keyswitch_lwe_ciphertext_vector_64_64_async", "cuda_keyswitch_gemm_64_64_async",
                 "cuda_glwe_sample_extract_64_async", "cuda_centered_modulus_switch_64_async"):
        assert must in syms


def test_shared_object_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB], text=True)
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_ctypes_binding_matches_header():
    import tfhe_rs_amd  # noqa: F401
    from tfhe_rs_amd import ffi
    assert sorted(ffi.SIGNATURES) == declared_symbols()
    # loading the product library resolves every symbol (dlopen only, nothing is launched)
    if os.path.exists(LIB):
        lib = ffi.Library(LIB)
        assert b"gfx950" in lib.hip_backend_version()


def test_product_package_does_not_touch_the_oracle():
    """No file of the product package imports, includes, links or opens anything of oracle/ or
    tests/ (comments may mention the word)."""
    pkg = os.path.join(ROOT, "tfhe_rs_amd")
    bad = re.compile(r"tfhe_oracle|libtfhe_oracle|orc_[a-z]|oracle/|from tests|import tests|tests\.oracle|"
                     r"#include\s*[\"<][^\">]*oracle|_emu\.so")
    for dirpath, dirs, files in os.walk(pkg):
        dirs[:] = [d for d in dirs if d not in ("build", "__pycache__")]
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                m = bad.search(text)
                assert not m, f"{os.path.join(dirpath, f)}: {m.group(0)}"


def test_missing_library_fails_loudly(tmp_path):
    import tfhe_rs_amd  # noqa: F401
    from tfhe_rs_amd import ffi
    with pytest.raises(ImportError):
        ffi.Library(str(tmp_path / "nope.so"))


# ------------------------------------------------------------------ prototypes, types and order
def _protos():
    import sys
    sys.path.insert(0, ROOT)
    from tools.c_prototypes import parse_prototypes
    return parse_prototypes(open(HEADER).read())


def _golden():
    import json
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_prototypes.json")))["prototypes"]


def test_every_reference_named_prototype_equals_the_reference_header():
    """Return type, every parameter type and their order, for EVERY function the header declares under a
    reference name (50 of them), against the prototypes of the reference's own headers (snapshot:
    tests/golden/reference_prototypes.json, made by tests/golden/make_prototypes.py from
    backends/tfhe-cuda-backend/cuda/include/**/*.h and tfhe-cuda-common/cuda/include/device.h)."""
    ours, ref = _protos(), _golden()
    in_scope = sorted(n for n in ours if not n.startswith("hip_"))
    assert in_scope == sorted(ref), "the snapshot does not cover the header: re-run tests/golden/make_prototypes.py"
    assert len(in_scope) >= 50
    for name in in_scope:
        ret, params = ours[name]
        assert ret == ref[name]["ret"], f"{name}: return type {ret} != {ref[name]['ret']} ({ref[name]['header']})"
        assert params == ref[name]["params"], f"{name}: parameters differ from {ref[name]['header']}"


@pytest.mark.skipif(not os.path.isdir("/root/reference/backends"), reason="reference tree absent")
def test_prototype_snapshot_is_what_the_reference_tree_says():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_prototypes
    assert make_prototypes.build() == _golden()


def test_rust_crate_bindings_are_generated_from_the_header_and_equal_the_reference_bindings():
    """backends/tfhe-hip-backend/src/{bindings,cuda_bind}.rs are the generator's output for the current header,
    and every declaration under a reference name is, token for token, the declaration bindgen produced for the
    reference (backends/tfhe-cuda-backend/src/bindings.rs, tfhe-cuda-common/src/cuda_bind.rs; snapshot)."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from tools import gen_rust_bindings as g
    from make_prototypes import rust_declarations
    crate = os.path.join(ROOT, "backends", "tfhe-hip-backend", "src")
    ours = {}
    for which, fname in (("backend", "bindings.rs"), ("runtime", "cuda_bind.rs")):
        text = open(os.path.join(crate, fname)).read()
        assert text == g.generate(which), f"{fname} is stale: run tools/gen_rust_bindings.py"
        ours.update(rust_declarations(text))
    ref = _golden()
    assert sorted(n for n in ours if not n.startswith("hip_")) == sorted(ref)
    for name, entry in ref.items():
        assert ours[name] == entry["rust"], f"{name}:\n  ours {ours[name]}\n  ref  {entry['rust']}"
    for f in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/ffi_types.rs"):
        assert os.path.exists(os.path.join(ROOT, "backends", "tfhe-hip-backend", f))


def test_split_key_ntt_engine_accepts_exactly_the_sets_its_limb_products_stay_exact_for():
    """hip_programmable_bootstrap_ntt64_split_supported is a pure host predicate: N = 2048, k = 1, one level, base_log 22
    or 23 — the sets the throughput kernel's LIMBS mode is instantiated for, with limb products
    (k+1) l N 2^(base_log-1) 2^15 <= 2^49 that the f64 transforms reproduce to well within 1/4."""
    import tfhe_rs_amd  # noqa: F401
    from tfhe_rs_amd import ffi
    if not os.path.exists(LIB):
        pytest.skip("product library not built")
    ok = ffi.Library(LIB).hip_programmable_bootstrap_ntt64_split_supported
    assert ok(1, 2048, 1, 23) and ok(1, 2048, 1, 22)     # PARAM_MESSAGE_2_CARRY_2 and the GPU multi-bit sets' base
    assert not ok(1, 2048, 1, 24) and not ok(1, 2048, 2, 15) and not ok(2, 1024, 1, 23) and not ok(1, 4096, 1, 22)
