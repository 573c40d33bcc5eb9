"""Drop-in for the reference's top-level `lru` module."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from vlsfr_amd.lru import LRU  # noqa: E402,F401
