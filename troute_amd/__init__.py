"""Import alias for the package that lives in ``t-route_amd/``.

The project directory is named after the upstream repository (``t-route``),
which is not a valid Python identifier; this three-line package makes it
importable as ``troute_amd`` without moving or duplicating any code:
submodules resolve through ``__path__`` into ``t-route_amd/``.
"""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "t-route_amd"))

from ._version import __version__  # noqa: E402,F401
