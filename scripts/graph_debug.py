"""Diagnostic: per-step loss / gradient norm of a small MobileFaceNet FFC with the executor calls replayed from HIP graphs
(argv[1]: none | both) — how the memset-node problem was found: with hipMemsetAsync in the captured passes the gradient norm
jumped by 3 - 4 orders of magnitude every few steps (LR=0 keeps the weights fixed so every step should look alike)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vlsfr_amd.ffc import FFC
from vlsfr_amd.optim.fused import FusedSGD
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
torch.manual_seed(0)
B, Q, D = 32, 1000, 128
m = FFC("mobile", D, Q, 32.0, "Arc", 0.5, 0.99).cuda()
m.probe_net.use_graphs = m.gallery_net.use_graphs = mode != "none"
LR = float(os.environ.get("LR", "0.05"))
opt = FusedSGD([p for p in m.parameters() if p.requires_grad], LR, momentum=0.9 if LR else 0.0, weight_decay=1e-4 if LR else 0.0, nesterov=bool(LR))
g = torch.Generator(device="cuda").manual_seed(1)
SYNC = os.environ.get("SYNC") == "1"
for step in range(int(os.environ.get("STEPS", "8"))):
    x = torch.randn(B, 3, 112, 112, device="cuda", generator=g); y = torch.randn(B, 3, 112, 112, device="cuda", generator=g)
    xl = torch.randint(0, Q, (B,)); yl = xl.clone()
    opt.zero_grad()
    loss = m(x, y, xl, yl)
    if SYNC: torch.cuda.synchronize()
    loss.backward()
    gn = float(torch.sqrt(sum((p.grad.float() ** 2).sum() for p in m.probe_net.parameters() if p.grad is not None)))
    opt.step()
    print("%s step %d loss %.4f grad norm %.4f" % (mode, step, float(loss), gn), flush=True)
