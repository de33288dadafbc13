#!/usr/bin/env python3
"""CPU emulation behind the 6-bit H stash ("stash diet"): weight-gradient error of the 8x256 benchmark model when the hidden
activations H_l are stashed as bf6 (e3m2, 2 mantissa bits like bf8/e5m2 but 3 exponent bits) with ONE power-of-two scale per 32-sample group
(and layer) = the E8M0 block scale of v_mfma_scale_f32_32x32x64_f8f6f4, against the bf8 (e5m2) stash of round 2.  dZ' stays bf8 with
stochastic rounding in the kernel; here it is kept exact so that the H format's error is seen alone, then both together (round to nearest).
Test-side tooling (imports the oracle); not collected by pytest."""
import os, sys, math
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import angio_oracle as orc

torch.manual_seed(0)
R, S, F, N = int(os.environ.get("RAYS", 512)), 128, 256, 8
W = H = 512
near, far = 1400.0, 1600.0
pose = orc.source_matrix(np.array([0, 0, 1500.0]), 0.0, 0.0)
o_all, d_all = orc.get_rays(pose, W, H, 13.0 * W)
pick = torch.randperm(W * H, generator=torch.Generator().manual_seed(1234))[:R]
o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
caps = orc.capsule_tree(levels=5, seed=0)
z_gt = torch.linspace(0., 1., 160) * (far - near) + near
tgt = orc.project_mu(lambda p: orc.capsule_mu(p, caps), o, d, z_gt).reshape(-1)
step = (far - near) / S
t = near + (torch.arange(S).float() + 0.5) * step
x = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)


def e3m2(v, scale):
    """v >= 0 -> nearest e3m2 value of v / scale (saturating at 28, subnormal step 1/16), times scale"""
    a = (v / scale).clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp(min=1e-30))).clamp(min=-2.0, max=4.0)
    q = torch.exp2(e - 2.0)
    return torch.round(a / q) * q * scale      # round-half-even on the grid


def e5m2(v, scale=1.0):
    return (v * scale).clamp(-57344, 57344).to(torch.float8_e5m2).float() / scale


def group_scale(h, top):
    """one power of two per 32-sample group: the group's largest value lands in [top/2, top)"""
    g = h.reshape(-1, 32, h.shape[-1]).amax(dim=(1, 2)).clamp(min=1e-20)
    s = torch.exp2(torch.ceil(torch.log2(g / top)))
    return s.repeat_interleave(32)[:, None]


def rel(a, b): return float((a - b).norm() / b.norm())


def study(tag, Ws, bs):
    Hs = [torch.relu(x @ Ws[0].T + bs[0])]
    for l in range(1, N + 1):
        Hs.append(torch.relu(Hs[-1].half().float() @ Ws[l].half().float().T + bs[l]))
    raw = (Hs[N] @ Ws[-1].T + bs[-1]).reshape(R, S)
    sig = torch.sigmoid(raw)
    pix = torch.exp(-(sig * step).sum(-1))
    g = ((-(pix * 2 * (pix - tgt) / R))[:, None] * step * sig * (1 - sig)).reshape(-1, 1)
    J = [None] * (N + 1)
    J[N] = Ws[-1].expand(Hs[N].shape[0], F) * (Hs[N] > 0)
    for l in range(N, 0, -1):
        J[l - 1] = (J[l] @ Ws[l]) * (Hs[l - 1] > 0)
    dZ = [g * J[l] for l in range(N + 1)]
    ref = [dZ[l].T @ Hs[l - 1] for l in range(1, N + 1)]
    print(f"== {tag}: H max per layer " + " ".join(f"{float(Hs[l].max()):.2g}" for l in range(N)))
    # dZ' as the kernel stashes it: g_hat J 2^10 in bf8, g_hat = g / 2^ceil(log2 max|g| of the group)
    gmax = g.abs().reshape(-1, 32).amax(1).clamp(min=1e-30)
    ge = torch.exp2(torch.ceil(torch.log2(gmax))).repeat_interleave(32)[:, None]
    dz8 = [e5m2(dZ[l] / ge, 1024.0) * ge for l in range(N + 1)]
    rows = [("H e5m2 (round 2)", lambda h: e5m2(h))]
    for top in (28.0, 14.0, 56.0):
        rows.append((f"H e3m2, group scale, top {top:g}", lambda h, top=top: e3m2(h, group_scale(h, top))))
    rows.append(("H e3m2, ONE scale per layer", lambda h: e3m2(h, torch.exp2(torch.ceil(torch.log2(h.max() / 28.0))))))
    for name, q in rows:
        e_h = [rel(dZ[l].T @ q(Hs[l - 1]), ref[l - 1]) for l in range(1, N + 1)]
        e_b = [rel(dz8[l].T @ q(Hs[l - 1]), ref[l - 1]) for l in range(1, N + 1)]
        print(f"  {name:34s} alone: " + " ".join(f"{e:.1e}" for e in e_h) + "   with dZ' bf8: " + " ".join(f"{e:.1e}" for e in e_b))


lin = [torch.nn.Linear(3, F)] + [torch.nn.Linear(F, F) for _ in range(N)] + [torch.nn.Linear(F, 1)]
with torch.no_grad():
    lin[-1].weight.mul_(4.0); lin[-1].bias.fill_(-5.0)
Ws = [l.weight.detach().clone() for l in lin]; bs = [l.bias.detach().clone() for l in lin]
study("bench.py's model (default init)", Ws, bs)
# a "trained-like" stand-in: heavier-tailed weights and larger biases (wide spread of activation magnitudes across features)
g = torch.Generator().manual_seed(7)
Wt = [w * torch.exp(0.8 * torch.randn(w.shape[0], 1, generator=g)) for w in Ws[:-1]] + [Ws[-1]]
bt = [b * 4 for b in bs]
study("row-scaled weights (log-normal sigma 0.8), biases x4", Wt, bt)
