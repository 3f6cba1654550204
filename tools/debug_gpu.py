import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from probabilistic_domain_adaptation_amd import _lib as L
from oracle import punet_oracle as O
dev = torch.device("cuda:0")
sys.path.insert(0, "tests")
import test_gpu_model as T
gd = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
name, mode = "uniform32_literal", "weight"
fx, cfg, Pnp, lay, eng, flat, (x, x1, x2, y), (B, C, H, W, s0, s1, s2) = T._setup(gd, name, dev)
nlast = max(int(k[3:]) for k in fx.files if k.startswith("y_n"))
y_soft = torch.from_numpy(fx[f"y_n{nlast}"]); consm = torch.from_numpy(fx[f"z_n{nlast}_{mode}"]).float()
eps_s = T._eps(1, B, 6, s2 + 100)[0]
st = eng.elbo_forward(flat, eng.to_nhwc(x2), y_soft.to(dev), consm.to(dev), eps_s.to(dev), "bce", 1.0, 1.0, True)
gbuf = torch.zeros(lay.numel, device=dev)
eng.elbo_backward(flat, gbuf, st, False)
torch.cuda.synchronize()
P_ = {k: v.double().detach().requires_grad_(True) for k, v in O.to_torch(Pnp).items()}
lo, aux = O.training_loss(P_, cfg, x2.double(), y_soft.double(), consm.double(), eps_s.double())
recon = aux["reconstruction"]
go = O.grads_of(lo, P_)
gv = lay.views(gbuf)
print("logit err", (st["logits"].cpu().double() - recon.detach()).abs().max().item())
# dlogit
cm = consm.double(); xl = recon.detach() * cm; t = y_soft.double() * cm
dl_ref = (torch.sigmoid(xl) - t) * cm
print("dlogit err", (st["dlogit"].cpu().double() - dl_ref).abs().max().item(), "max", dl_ref.abs().max().item())
print("rec", st["rec"].item(), aux["reconstruction_loss"].item(), "kl", st["kl"].item(), aux["kl"].item())
for k, r in go.items():
    e = (gv[k].cpu().double() - r).abs().max().item() / (r.abs().max().item() + 1e-12)
    print(f"{k:55s} {e:.3e}  max|g| {r.abs().max().item():.3e}")
