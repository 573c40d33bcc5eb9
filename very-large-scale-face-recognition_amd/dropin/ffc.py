"""Drop-in for the reference's top-level `ffc` module."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
from vlsfr_amd.ffc import FFC  # noqa: E402,F401
