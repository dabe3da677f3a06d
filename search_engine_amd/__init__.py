"""Importable alias of the ``search-engine_amd`` package directory (a hyphen cannot appear in a
module name): submodules resolve inside that directory."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "search-engine_amd"))
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
