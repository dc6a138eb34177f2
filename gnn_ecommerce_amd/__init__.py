"""Import alias: ``gnn_ecommerce_amd`` -> the package directory ``gnn-ecommerce_amd/``.

The product directory carries the project's hyphenated name, which Python cannot import; this
shim points the package's search path at it and runs its ``__init__``.  No code lives here.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gnn-ecommerce_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
