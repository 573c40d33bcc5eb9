"""Import shim: the product package lives in the directory ``very-large-scale-face-recognition_amd/``
(not a legal Python identifier); ``import vlsfr_amd`` binds that directory as a regular package."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "very-large-scale-face-recognition_amd")
_spec = _u.spec_from_file_location("vlsfr_amd", _os.path.join(_dir, "__init__.py"),
                                   submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["vlsfr_amd"] = _mod
_spec.loader.exec_module(_mod)
