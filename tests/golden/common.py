"""Deterministic synthetic inputs shared by the golden-vector generator (make_golden.py, runs in the
build container where /root/reference exists) and by the tests (run anywhere).  Everything is drawn
from numpy's PCG64 so the same seed gives the same bytes on every machine."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.dirname(os.path.abspath(__file__))


def fill_state(sd, seed):
    """Overwrite every floating tensor of a backbone state dict with seeded values that are more
    discriminating than the constructors' constants (BN weight 1 / bias 0 / PReLU 0.25)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name in sorted(sd.keys()):
        v = sd[name]
        if not v.is_floating_point():
            out[name] = v.clone()
            continue
        shape = tuple(v.shape)
        if name.endswith("running_mean"):
            a = 0.05 * rng.standard_normal(shape)
        elif name.endswith("running_var"):
            a = 1.0 + 0.1 * rng.random(shape)
        elif name == "features.weight":
            a = np.ones(shape)                      # frozen at 1 in the reference
        elif name.endswith("prelu.weight") or (v.dim() == 1 and name.split(".")[-2] in ("2", "5")):
            a = 0.25 + 0.05 * rng.standard_normal(shape)
        elif v.dim() == 1 and name.endswith("weight"):
            a = 1.0 + 0.1 * rng.standard_normal(shape)      # BN gamma
        elif v.dim() == 1:
            a = 0.1 * rng.standard_normal(shape)            # BN beta / fc bias
        else:
            std = 1.0 / np.sqrt(float(np.prod(shape[1:])))  # conv / linear weights: fan-in scaling
            a = std * rng.standard_normal(shape)
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(shape).clone()
    return out


def unit_rows(rng, *shape):
    a = rng.standard_normal(shape).astype(np.float32)
    a /= np.linalg.norm(a, axis=-1, keepdims=True)
    return a


def synth_images_u8(rng, n, hw=112):
    return rng.integers(0, 256, size=(n, 3, hw, hw), dtype=np.uint8)


def images_from_u8(u8):
    """Loader contract (reference util/lmdb_loader.py:127): (v - 127.5) * 0.0078125, float32 CHW."""
    return torch.from_numpy((u8.astype(np.float32) - 127.5) * 0.0078125)


def head_case(seed, Q, D, B, T, n_id):
    """Embeddings and labels for T steps of (rollback pass, commit pass).  Label structure follows
    main.py:53-60: the first half of x/y labels are shared identities, the second half are instance
    labels; a few deliberate repeats inside a batch exercise duplicate (row, slot) writes."""
    rng = np.random.default_rng(seed)
    P = unit_rows(rng, T, 2, B, D)       # [:,0] = probe(x), [:,1] = probe(y)
    G = unit_rows(rng, T, 2, B, D)       # [:,0] = gallery(y), [:,1] = gallery(x)
    XL = np.zeros((T, B), dtype=np.int64)
    YL = np.zeros((T, B), dtype=np.int64)
    h = B // 2
    for t in range(T):
        ids = rng.choice(n_id, size=h, replace=False)
        XL[t, :h] = ids
        YL[t, :h] = ids
        XL[t, h:] = rng.integers(0, n_id, size=B - h)
        YL[t, h:] = rng.integers(0, n_id, size=B - h)
        if t % 2 == 1 and B >= 6:        # the same identity three times in one gallery batch
            YL[t, h] = YL[t, 0]
            YL[t, h + 1] = YL[t, 0]
            XL[t, h] = XL[t, 1]
    queue0 = unit_rows(rng, 2, Q, D)
    return dict(P=P, G=G, XL=XL, YL=YL, queue0=queue0)


def step_inputs(seed, Q, D, B, n_id=1000):
    """Inputs of the full-step golden cases (images as uint8, main.py:53-60 label structure)."""
    rng = np.random.default_rng(seed + 5000)
    queue0 = unit_rows(rng, 2, Q, D)
    xu8 = synth_images_u8(rng, B)
    yu8 = synth_images_u8(rng, B)
    ids = rng.choice(n_id, size=B // 2, replace=False)
    xl = np.concatenate([ids, rng.integers(0, n_id, size=B - B // 2)]).astype(np.int64)
    yl = np.concatenate([ids, rng.integers(0, n_id, size=B - B // 2)]).astype(np.int64)
    warm = unit_rows(rng, 2, B, D)
    return dict(queue0=queue0, xu8=xu8, yu8=yu8, xl=xl, yl=yl, warm=warm)
