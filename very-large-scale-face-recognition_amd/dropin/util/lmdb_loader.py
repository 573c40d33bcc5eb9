"""Drop-in for the reference's `util/lmdb_loader.py`."""
from vlsfr_amd.data import MultiLMDBDataset, PairLMDBDataset  # noqa: F401
