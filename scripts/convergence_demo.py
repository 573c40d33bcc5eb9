"""End-to-end sanity beyond one-step parity: the drop-in training driver (vlsfr_amd.main, the reference's loop) on a synthetic
face store with learnable structure — every identity is a random low-frequency template, every image of it the template plus
pixel noise — for a few hundred iterations, with the bf16 and with the fp8 class matmul.  The loss must fall.
usage: python scripts/convergence_demo.py [iters] [net_type]     (writes nothing but its log to stdout)"""
import os, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlsfr_amd  # noqa
from vlsfr_amd.data import FaceStore
from vlsfr_amd.main import parse_args, train

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
net = sys.argv[2] if len(sys.argv) > 2 else "ir18"
n_ids, per_id = 1500, 4
tmp = tempfile.mkdtemp(prefix="vlsfr_demo_")
rng = np.random.default_rng(0)
st = FaceStore(tmp, "demo", readonly=False)
kv_path = os.path.join(tmp, "demo_kv.txt")
with open(kv_path, "w") as kv:
    for label in range(n_ids):
        t = rng.integers(40, 216, size=(14, 14, 3)).astype(np.float32)
        t = np.kron(t, np.ones((8, 8, 1), dtype=np.float32))                      # 112 x 112 low-frequency template
        for j in range(per_id):
            img = np.clip(t + rng.normal(0, 25, size=t.shape), 0, 255).astype(np.uint8)
            key = "demo_%d_%d" % (label, j)
            st.put(key, img, "raw")
            kv.write("%s %d\n" % (key, label))
st.close()
for dtype in ("bf16", "fp8"):
    losses = []
    def log(msg):
        if " loss " in msg:
            losses.append(float(msg.split(" loss ")[1].split()[0]))
    conf = parse_args(["--net_type", net, "--feat_dim", "512", "--queue_size", "2048", "--num_class", str(n_ids), "--batch_size", "64",
                       "--print_freq", "20", "--iters_per_epoch", str(iters), "--saved_dir", "", "--data_store", tmp, "--data_kv", kv_path,
                       "--head_dtype", dtype, "--loss_type", "Arc"])
    torch.manual_seed(0)
    train(conf, log=log)
    head = " ".join("%.2f" % v for v in losses)
    print("%s class matmul, %s, %d iterations, loss every 20 iterations: %s" % (dtype, net, iters, head), flush=True)
    k = max(1, len(losses) // 5)
    print("   first fifth mean %.3f -> last fifth mean %.3f" % (float(np.mean(losses[:k])), float(np.mean(losses[-k:]))), flush=True)
