"""Drop-in for the reference's `util/lmdb_loader.py`."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))))
from vlsfr_amd.data import MultiLMDBDataset, PairLMDBDataset  # noqa: E402,F401
