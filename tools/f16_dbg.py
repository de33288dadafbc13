import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2
from oracle import angio_oracle as orc
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.render import render_rays
from nerf_for_angiography_amd.phantomdata.proj_helpers import source_matrix
DEV = "cuda:0"
def make_model(layers, width, precision):
    md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1,
              num_input_channels_views=0, use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5,
              num_img=1, device=torch.device(DEV), precision=precision)
    return CPPN(md).to(DEV)
def grads(m): return {k: p.grad.detach().cpu().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
L, W = int(os.environ.get("L", 2)), int(os.environ.get("W", 64))
for prec in os.environ.get("PRECS", "f16,bf16").split(","):
    torch.manual_seed(3)
    m = make_model(L, W, prec)
    with torch.no_grad():
        m.output_linear[0].weight.mul_(4.0); m.output_linear[0].bias.fill_(-5.0)
    w = 64
    pose = source_matrix(np.array([0, 0, 1500.0]), 30.0, 12.0)
    o_all, d_all = orc.get_rays(pose, w, w, 13.0 * w)
    R = int(os.environ.get("R", 64))
    pick = torch.randperm(w * w)[:R]
    o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
    tgt = torch.rand(o.shape[0])
    cfg = dict(num_early_layers=L, num_filters=W)
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    s = int(os.environ.get("S", 64))
    pix_c, loss_c, grads_c = orc.loss_and_grads(o, d, tgt, cfg, params, near=1400.0, far=1600.0, n_samples=s, convention="acc")
    out = render_rays(m, o.to(DEV), d.to(DEV), s, 1400.0, 1600.0, mode="acc")
    torch.nn.functional.mse_loss(out.rgb_map, tgt.to(DEV)).backward()
    got = grads(m)
    for k, v in grads_c.items():
        a, b = got[k].ravel().astype(np.float64), v.numpy().ravel().astype(np.float64)
        print(prec, f"{k:32s} rel {rel_l2(got[k], v.numpy()):.2e} norm ratio {np.linalg.norm(a)/np.linalg.norm(b):.4f} cos {a@b/np.linalg.norm(a)/np.linalg.norm(b):.5f}")
