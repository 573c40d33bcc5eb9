"""Drop-in for the reference's `util` package (util/__init__.py re-exports the loader classes)."""
from vlsfr_amd.data import MultiLMDBDataset, PairLMDBDataset  # noqa: F401
