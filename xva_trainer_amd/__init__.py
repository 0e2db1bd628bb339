"""Import shim: the product package lives in ``xva-trainer_amd/`` (the directory name the
build contract fixes).  A hyphen is not importable, so this module re-points its package
path at that directory and executes its ``__init__``; ``import xva_trainer_amd.<sub>``
therefore resolves to ``xva-trainer_amd/<sub>``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "xva-trainer_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
