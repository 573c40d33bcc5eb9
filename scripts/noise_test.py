import os, sys, ctypes
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from tests.test_step_gpu import build_ffc, G
from vlsfr_amd import _lib
z = np.load(os.path.join(G, "step_mobile.npz"))
def run(conc, kb=64):
    _lib.lib().vlsfr_set_option(b"bn_block_kb", ctypes.c_int32(kb))
    m, x, y, xl, yl = build_ffc(z, "mobile")
    m.concurrent_streams = conc; m.probe_net.concurrent_backward = conc
    loss = m(x, y, xl, yl); torch.cuda.synchronize()
    return float(loss.detach())
print("serial kb64", run(False), run(False))
print("serial kb32", run(False, 32), "kb16", run(False, 16), "kb128", run(False, 128))
print("concurrent ", run(True), run(True), run(True))
