"""Import shim: the package directory is `tfhe-rs_amd/` (named after the reference repo);
Python cannot import a hyphenated name, so `import tfhe_rs_amd` loads it from there."""
import os as _os

_real = _os.path.normpath(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), _os.pardir, "tfhe-rs_amd"))
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
