"""Importable alias of the `cooperative-search_amd/` package directory (a hyphen cannot be imported).

`import cooperative_search_amd` executes cooperative-search_amd/__init__.py with this package's __path__
pointing at that directory, so `cooperative_search_amd.env` etc. resolve to the files that live there.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "cooperative-search_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
