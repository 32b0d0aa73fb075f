"""Import shim: the package directory is ``kobato-eyes_amd/`` (hyphen, as the project is named),
which Python cannot import by that name.  This module IS the package ``kobato_eyes_amd``: it
points ``__path__`` at that directory and runs its ``__init__``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "kobato-eyes_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__, "r", encoding="utf-8") as _fh:
    exec(compile(_fh.read(), __file__, "exec"))
del _os, _fh
