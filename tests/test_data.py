"""Data path: kv parsing / multi-database label offsets / pair sampling of the reference loader's two
datasets over the self-contained store (CPU), and the device-side transform against the oracle's restatement of
util/lmdb_loader.py:109-127 (GPU, bit-exact: (v - 127.5) * 0.0078125 is exact in fp32)."""
import os
import random

import numpy as np
import pytest
import torch

from oracle.data_ref import loader_transform_ref


def _two_stores(tmp_path):
    from vlsfr_amd.data import make_synthetic_store
    a = make_synthetic_store(str(tmp_path / "a"), "dba", 4, lambda l: 1 if l == 2 else 3, hw=16, seed=1, grey_every=5)
    b = make_synthetic_store(str(tmp_path / "b"), "dbb", 3, 2, hw=16, seed=2, fmt="jpeg")
    return a, b


def test_kv_parsing_label_offsets_and_pairs(tmp_path):
    from vlsfr_amd.data import FaceStore, MultiLMDBDataset, PairLMDBDataset
    (pa, kva), (pb, kvb) = _two_stores(tmp_path)
    assert open(kva).readline() == "dba_0_0 0\n"                                # data/creat_lmdb.py:65-66
    multi = MultiLMDBDataset([pa, pb], [kva, kvb])
    # store a has labels 0..3 -> the second store's labels start at 4 (util/lmdb_loader.py:32-43)
    assert len(multi) == (3 + 3 + 1 + 3) + 3 * 2 and multi.num_class == 7
    assert [t[2] for t in multi.train_list if t[1] == 1] == [4, 4, 5, 5, 6, 6]
    assert multi.train_list[0] == ["dba_0_0", 0, 0]
    pair = PairLMDBDataset([pa, pb], [kva, kvb])
    assert len(pair) == 7 and pair.label_set == list(range(7))
    assert pair.label2files[2] == [0, ["dba_2_0"]] and pair.label2files[5][0] == 1
    random.seed(3)
    img1, f1, img2, f2, label = pair[2]                                         # one image: used twice (:197-198)
    assert label == 2 and torch.equal(img1, img2)
    img1, f1, img2, f2, label = pair[0]
    assert label == 0 and not torch.equal(img1, img2)                           # sample(keys, 2): two different images
    # flips follow python's random(), seeded by the loop (main.py:24)
    random.seed(11)
    want = [1 if random.random() < 0.5 else 0 for _ in range(6)]
    random.seed(11)
    got = [multi[i][1] for i in range(6)]
    assert got == want
    # raw records come back bit for bit; JPEG records decode to the stored size in BGR order
    st = FaceStore(pa)
    rng = np.random.default_rng(1)
    assert np.array_equal(st.get("dba_0_0"), rng.integers(0, 256, size=(16, 16, 3), dtype=np.uint8))
    assert FaceStore(pb).get("dbb_1_1").shape == (16, 16, 3)
    assert multi[4][0].shape == (16, 16, 1)                                      # the grey record (every 5th image)


@pytest.mark.gpu
def test_device_transform_matches_reference_arithmetic(tmp_path):
    from vlsfr_amd.data import DeviceBatcher, MultiLMDBDataset, PairLMDBDataset, device_collate, faces_to_device
    rng = np.random.default_rng(0)
    for shape in ((112, 112, 3), (112, 112, 1), (17, 23, 3)):
        imgs = [rng.integers(0, 256, size=shape, dtype=np.uint8) for _ in range(5)]
        flips = [0, 1, 1, 0, 1]
        got = faces_to_device(imgs, flips, "cuda").cpu().numpy()
        want = np.stack([loader_transform_ref(im, f) for im, f in zip(imgs, flips)])
        assert got.dtype == np.float32 and np.array_equal(got, want)
    (pa, kva), (pb, kvb) = _two_stores(tmp_path)
    multi, pair = MultiLMDBDataset(pa, kva), PairLMDBDataset(pa, kva)
    random.seed(5)
    items = [multi[i] for i in (0, 1, 2)]
    images, labels, feats = device_collate("cuda")(items)
    assert images.shape == (3, 3, 16, 16) and labels.tolist() == [0, 0, 0] and feats.tolist() == [-1, -1, -1]
    for k, it in enumerate(items):
        assert np.array_equal(images[k].cpu().numpy(), loader_transform_ref(it[0].numpy(), it[1]))
    random.seed(7)
    b = DeviceBatcher(MultiLMDBDataset(pb, kvb), PairLMDBDataset(pb, kvb), batch_size=4, device="cuda", n_batches=3)
    n = 0
    for inst, inst_label, i1, i2, ids in b:                                      # the tuple main.py's loop consumes
        assert inst.shape == (4, 3, 16, 16) and i1.shape == (2, 3, 16, 16) and i2.shape == (2, 3, 16, 16)
        assert inst.is_cuda and inst.dtype == torch.float32 and float(inst.abs().max()) <= 1.0
        assert inst_label.shape == (4,) and ids.shape == (2,) and len(set(ids.tolist())) == 2
        n += 1
    assert n == 3
    # batches that MIX grey [H, W, 1] and colour [H, W, 3] records (store a: every 5th image is grey) — the reference
    # converts sample by sample (util/lmdb_loader.py:111-127), so such batches are ordinary input
    random.seed(9)
    items = [multi[i] for i in (2, 3, 4, 5, 9)]                                  # records 4 and 9 are grey
    assert sorted(set(int(it[0].shape[2]) for it in items)) == [1, 3]
    images, labels, _ = device_collate("cuda")(items)
    for k, it in enumerate(items):
        assert np.array_equal(images[k].cpu().numpy(), loader_transform_ref(it[0].numpy(), it[1]))
    random.seed(13)
    b = DeviceBatcher(multi, pair, batch_size=8, device="cuda", n_batches=4, seed=1)
    seen = 0
    for inst, inst_label, i1, i2, ids in b:
        assert inst.shape == (8, 3, 16, 16) and i1.shape == (4, 3, 16, 16) and bool(torch.isfinite(inst).all())
        seen += 1
    assert seen == 4
