"""Data path of the training loop — the reference's `util/lmdb_loader.py` surface on the MI355X path.

The reference reads JPEG bytes from LMDB environments (`data/creat_lmdb.py:45-70` writes them, key
`'%s_%d_%d' % (db_name, label, j)`, one `<key> <label>` line per image in `<db_name>_kv.txt`) and turns each image into
a float32 CHW tensor on the host (`util/lmdb_loader.py:101-132`, `:191-237`).  `lmdb` / `cv2` / the `Datum` protobuf
are not importable here, so:

  * `FaceStore` is a self-contained key -> image-bytes store with the same key / kv-file conventions (raw uint8 HWC
    BGR records, or JPEG records decoded on the host with PIL — cv2.imdecode's channel order is restored);
  * `MultiLMDBDataset` / `PairLMDBDataset` keep the reference's constructor arguments, kv parsing, multi-database
    label offsetting, `random()` flip draws and `sample(keys, 2)` pair choice — but a sample is the DECODED uint8
    image plus its flip flag; the arithmetic of the transform runs on the GPU for the whole batch
    (`vlsfr_faces_normalize`, csrc/norm.hip) in `DeviceBatcher` / `device_collate`, which yield exactly the batch
    tensors the reference's DataLoader yields: float32 [B, 3, H, W] = (v - 127.5) * 0.0078125, labels int64.
There is no host implementation of the transform in this package (oracle/data_ref.py restates it for the tests).
"""
import ctypes
import io
import os
from random import random, sample

import numpy as np
import torch

from . import _lib


class FaceStore(object):
    """Directory `<path>/` with `<name>.bin` (records back to back) and `<name>.idx` (text: key offset nbytes fmt H W C).
    Stands in for the LMDB environment + Datum of the reference (util/lmdb_loader.py:63-70, 105-108)."""

    def __init__(self, path, name=None, readonly=True):
        self.path = path
        if name is None:
            idx = [f for f in os.listdir(path) if f.endswith(".idx")]
            if len(idx) != 1:
                raise ValueError("FaceStore: %s holds %d stores, name one" % (path, len(idx)))
            name = idx[0][:-4]
        self.name = name
        self._bin = os.path.join(path, name + ".bin")
        self._idx = os.path.join(path, name + ".idx")
        self.index = {}
        self._map = None
        self._w = None
        if readonly:
            with open(self._idx) as f:
                for line in f:
                    it = line.split()
                    if it:
                        self.index[it[0]] = (int(it[1]), int(it[2]), it[3], int(it[4]), int(it[5]), int(it[6]))
        else:
            os.makedirs(path, exist_ok=True)
            self._w = (open(self._bin, "wb"), open(self._idx, "w"))
            self._off = 0

    # ---- writer (data/creat_lmdb.py:60-66) --------------------------------------------------------
    def put(self, key, img_hwc_bgr, fmt="raw"):
        img = np.ascontiguousarray(img_hwc_bgr, dtype=np.uint8)
        if img.ndim == 2:
            img = img[:, :, None]
        H, W, C = img.shape
        if fmt == "raw":
            data = img.tobytes()
        elif fmt == "jpeg":
            from PIL import Image
            buf = io.BytesIO()
            Image.fromarray(img[:, :, ::-1] if C == 3 else img[:, :, 0]).save(buf, format="JPEG", quality=95)
            data = buf.getvalue()
        else:
            raise ValueError("FaceStore.put: fmt must be 'raw' or 'jpeg'")
        self._w[0].write(data)
        self._w[1].write("%s %d %d %s %d %d %d\n" % (key, self._off, len(data), fmt, H, W, C))
        self._off += len(data)

    def close(self):
        if self._w is not None:
            for f in self._w:
                f.close()
            self._w = None
        self._map = None

    # ---- reader -------------------------------------------------------------------------------------
    def get(self, key):
        """Decoded uint8 image [H, W, C] (C = 3 in cv2's BGR order, or 1) — what cv2.imdecode(..., -1) returns."""
        if self._map is None:
            self._map = np.memmap(self._bin, dtype=np.uint8, mode="r")
        off, n, fmt, H, W, C = self.index[key]
        rec = self._map[off:off + n]
        if fmt == "raw":
            return np.array(rec).reshape(H, W, C)
        from PIL import Image
        im = np.asarray(Image.open(io.BytesIO(rec.tobytes())))
        return im[:, :, ::-1].copy() if im.ndim == 3 else im[:, :, None]


def _as_list(v):
    return list(v) if isinstance(v, (list, tuple)) else [v]


class MultiLMDBDataset(object):
    """util/lmdb_loader.py:12-132.  `source_lmdbs` are FaceStore directories, `source_files` their kv files.
    `__getitem__` -> (uint8 image [H, W, C], flip flag, label, -1); `device_collate` / `DeviceBatcher` make the
    reference's (float32 CHW image, label, -1) batches out of it on the GPU."""

    def __init__(self, source_lmdbs, source_files, feat_lmdbs=None, feat_files=None, transforms=None, return_feats=False):
        source_lmdbs, source_files = _as_list(source_lmdbs), _as_list(source_files)
        assert len(source_files) == len(source_lmdbs)
        assert len(source_lmdbs) > 0
        self.source_lmdbs = source_lmdbs
        self.train_list = []
        max_label = 0
        last_label = 0
        for db_id, file_path in enumerate(source_files):                        # :32-43
            with open(file_path, 'r') as fin:
                for line in fin:
                    l = line.rstrip().lstrip()
                    if len(l) > 0:
                        items = l.split(' ')
                        self.train_list.append([items[0], db_id, int(items[1]) + last_label])
                        max_label = max(max_label, int(items[1]) + last_label)
            if max_label != last_label:
                max_label += 1
                last_label = max_label
        self.num_class = last_label
        self.transform = transforms
        self.return_feats = return_feats
        self.stores = None
        if transforms is not None:
            assert isinstance(transforms, (list, tuple)) and len(transforms) == len(source_lmdbs)

    def __len__(self):
        return len(self.train_list)

    def open_lmdb(self):
        self.stores = [FaceStore(p) for p in self.source_lmdbs]

    def close(self):
        if self.stores is not None:
            for s in self.stores:
                s.close()
            self.stores = None

    def __getitem__(self, index):
        if self.stores is None:
            self.open_lmdb()
        key, db_id, label = self.train_list[index][:3]
        img = self.stores[db_id].get(key)
        flip = 1 if random() < 0.5 else 0                                        # :109
        return torch.from_numpy(np.ascontiguousarray(img)), flip, label, -1


class PairLMDBDataset(object):
    """util/lmdb_loader.py:134-237: one item per identity, two of its images (the same one twice when it has only one).
    `__getitem__` -> (uint8 img1, flip1, uint8 img2, flip2, label)."""

    def __init__(self, source_lmdbs, source_files, exclude_id_set=None):
        source_lmdbs, source_files = _as_list(source_lmdbs), _as_list(source_files)
        assert len(source_files) == len(source_lmdbs)
        assert len(source_lmdbs) > 0
        self.source_lmdbs = source_lmdbs
        self.stores = None
        max_label = 0
        last_label = 0
        self.label2files = {}
        self.label_set = []
        for db_id, file_path in enumerate(source_files):                        # :156-169
            with open(file_path, 'r') as fin:
                for line in fin:
                    l = line.strip()
                    if len(l) > 0:
                        items = l.split(' ')
                        the_label = int(items[1]) + last_label
                        if the_label not in self.label2files:
                            self.label2files[the_label] = [db_id, []]
                            self.label_set.append(the_label)
                        self.label2files[the_label][1].append(items[0])
                        max_label = max(max_label, the_label)
            max_label += 1
            last_label = max_label

    def __len__(self):
        return len(self.label_set)

    def open_lmdb(self):
        self.stores = [FaceStore(p) for p in self.source_lmdbs]

    def close(self):
        if self.stores is not None:
            for s in self.stores:
                s.close()
            self.stores = None

    def __getitem__(self, index):
        if self.stores is None:
            self.open_lmdb()
        label = self.label_set[index]
        db_id, keys = self.label2files[label]
        if len(keys) >= 2:                                                       # :195-198
            key1, key2 = sample(keys, 2)
        else:
            key1, key2 = keys[0], keys[0]
        img1 = self.stores[db_id].get(key1)
        f1 = 1 if random() < 0.5 else 0                                          # :206-208
        img2 = self.stores[db_id].get(key2)
        f2 = 1 if random() < 0.5 else 0                                          # :221-223
        return (torch.from_numpy(np.ascontiguousarray(img1)), f1, torch.from_numpy(np.ascontiguousarray(img2)), f2, label)


def faces_to_device(raw_list, flips, device):
    """Stacks decoded uint8 images [H, W, C] (one shape per call), ships them with the flip flags in one pinned
    transfer each and runs the loader transform on the GPU -> float32 [B, 3, H, W] (vlsfr_faces_normalize)."""
    if torch.device(device).type != "cuda":
        raise _lib.VlsfrError("faces_to_device: the loader transform runs on the GPU; there is no host path")
    rows = [r if torch.is_tensor(r) else torch.from_numpy(np.ascontiguousarray(r)) for r in raw_list]
    if len(set(int(r.shape[2]) for r in rows)) > 1:
        # a batch that mixes grey [H, W, 1] and colour [H, W, 3] records (the reference converts sample by sample,
        # util/lmdb_loader.py:111-127 / :209-233): grey records take their three equal planes here, which is what the
        # kernel's C = 1 path writes, so the result is the per-sample conversion's bit for bit
        rows = [r.expand(r.shape[0], r.shape[1], 3) if r.shape[2] == 1 else r for r in rows]
    raw = torch.stack(rows)
    B, H, W, C = raw.shape
    raw_d = raw.pin_memory().to(device, non_blocking=True)
    flip_d = torch.as_tensor(np.asarray(flips, dtype=np.uint8)).pin_memory().to(device, non_blocking=True)
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
    fn = _lib.lib().vlsfr_faces_normalize
    fn.restype = ctypes.c_int
    _lib.check(fn(ctypes.c_void_p(raw_d.data_ptr()), ctypes.c_void_p(flip_d.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                  ctypes.c_int32(B), ctypes.c_int32(H), ctypes.c_int32(W), ctypes.c_int32(C),
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "vlsfr_faces_normalize")
    out._vlsfr_keep = (raw_d, flip_d)      # inputs stay alive until the kernel has run
    return out


def device_collate(device):
    """collate_fn for torch.utils.data.DataLoader over the two datasets: the batch the reference's loop receives
    (main.py:35,43: `images, labels, _` / `images1, images2, id_indexes`), images already on `device`."""
    def collate(items):
        if len(items[0]) == 4:                                                   # MultiLMDBDataset
            imgs, flips, labels, _ = zip(*items)
            return faces_to_device(imgs, flips, device), torch.tensor(labels, dtype=torch.int64), torch.full((len(items),), -1)
        i1, f1, i2, f2, labels = zip(*items)
        both = faces_to_device(list(i1) + list(i2), list(f1) + list(f2), device)
        return both[:len(items)], both[len(items):], torch.tensor(labels, dtype=torch.int64)
    return collate


class DeviceBatcher(object):
    """The two loaders of the reference's loop (main.py:102-111: RandomSampler over the instance dataset, an id
    iterator that is re-armed when exhausted, main.py:42-46) without worker processes: batches of decoded images go to
    the GPU as uint8 and are normalised / flipped there.  Yields what SyntheticFaces yields:
    (instance images [B], instance labels, id images 1 [B/2], id images 2 [B/2], ids)."""

    def __init__(self, inst_dataset, id_dataset, batch_size, device, n_batches=None, seed=0):
        self.inst, self.ids, self.B, self.dev = inst_dataset, id_dataset, batch_size, device
        self.n = n_batches if n_batches is not None else len(inst_dataset) // batch_size
        self.rng = np.random.default_rng(seed)
        self._id_order = []

    def __len__(self):
        return self.n

    def _next_ids(self, k):
        """k distinct identities: the next k of a random permutation; when fewer than k are left the id iterator is
        re-armed (main.py:42-46) — a partial last batch is dropped rather than mixed with the next epoch's ids."""
        if k > len(self.ids):
            raise ValueError("the id dataset holds %d identities, a batch needs %d" % (len(self.ids), k))
        if len(self._id_order) < k:
            self._id_order = self.rng.permutation(len(self.ids)).tolist()
        out, self._id_order = self._id_order[:k], self._id_order[k:]
        return out

    def __iter__(self):
        col = device_collate(self.dev)
        h = self.B // 2
        for _ in range(self.n):
            idx = self.rng.integers(0, len(self.inst), size=self.B)               # RandomSampler(replacement draws per epoch)
            images, labels, _ = col([self.inst[int(i)] for i in idx])
            img1, img2, ids = col([self.ids[i] for i in self._next_ids(h)])
            yield images, labels, img1, img2, ids


def make_synthetic_store(path, db_name, n_ids, imgs_per_id, hw=112, seed=0, fmt="raw", grey_every=0):
    """Writes a FaceStore + kv file with the conventions of data/creat_lmdb.py:45-70 (key '%s_%d_%d', kv line
    '<key> <label>') from seeded uniform-uint8 images (every `grey_every`-th image single-channel, to exercise the
    grey branch of the loader).  Returns (store directory, kv file path)."""
    rng = np.random.default_rng(seed)
    st = FaceStore(path, db_name, readonly=False)
    kv_path = os.path.join(path, "%s_kv.txt" % db_name)
    k = 0
    with open(kv_path, "w") as kv:
        for label in range(n_ids):
            n_img = imgs_per_id(label) if callable(imgs_per_id) else imgs_per_id
            for j in range(n_img):
                grey = grey_every and (k % grey_every == grey_every - 1)
                img = rng.integers(0, 256, size=(hw, hw, 1 if grey else 3), dtype=np.uint8)
                key = '%s_%d_%d' % (db_name, label, j)
                st.put(key, img, fmt)
                kv.write('%s %d\n' % (key, label))
                k += 1
    st.close()
    return path, kv_path
