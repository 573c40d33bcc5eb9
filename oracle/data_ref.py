"""ORACLE — test infrastructure only (never imported by the product package).

numpy restatement of the reference loader's per-image transform (util/lmdb_loader.py:109-127 and :206-233), given the
decoded uint8 image cv2.imdecode(..., -1) returns ([H, W, 3] BGR or [H, W] grey) and the flip decision.  Parity
unpinned by the reference's own tests (it has none, and its loader does not import here: `from data import Datum`,
cv2, lmdb are missing — SURVEY F9); the arithmetic is three lines of numpy restated verbatim."""
import numpy as np


def loader_transform_ref(img, flip):
    img = np.asarray(img)
    if img.ndim == 3 and img.shape[2] == 1:
        img = img[:, :, 0]
    if flip:
        img = img[:, ::-1]                                                       # cv2.flip(img, 1)  :109-110
    if img.ndim == 2:                                                            # :111-117
        buf = np.zeros((3, img.shape[0], img.shape[1]), dtype=np.uint8)
        buf[0] = img
        buf[1] = img
        buf[2] = img
        return (buf - 127.5).astype(np.float32) * 0.0078125
    return (img.transpose((2, 0, 1)).astype(np.float32) - 127.5) * 0.0078125    # :124
