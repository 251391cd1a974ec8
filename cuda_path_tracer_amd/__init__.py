"""Import shim: the package directory is named ``cuda-path-tracer_amd`` (not a Python
identifier), so ``import cuda_path_tracer_amd`` resolves to it through this module."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cuda-path-tracer_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
