"""MI355X-native FFC face-embedding training hot path (drop-in for the reference's ffc.py / lru.py /
model/ / optim/ / main.py surface).  Import as ``vlsfr_amd`` (root shim ``vlsfr_amd.py``), or put
``very-large-scale-face-recognition_amd/dropin`` first on ``sys.path`` to keep the reference's own
top-level module names (``from ffc import FFC`` ...).  See INTEGRATION.md."""
__version__ = "0.1.0"
