#!/usr/bin/env python3
"""Operand-precision study on the CPU (no GPU, no reference import): what the forward pixel error and the hidden-layer
weight-gradient error of the 8x256 benchmark model would be for different operand / stash formats, emulated by rounding the
operands of every contraction to the format and accumulating in fp32 (what an MFMA does).  Used to choose the training
precision of round 2; test-side tooling (it imports the oracle), not collected by pytest."""
import sys, os, math
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import angio_oracle as orc

torch.manual_seed(0)
R, S, F, N = int(os.environ.get("RAYS", 1024)), 128, 256, 8
W = H = 512
near, far = 1400.0, 1600.0
pose = orc.source_matrix(np.array([0, 0, 1500.0]), 0.0, 0.0)
o_all, d_all = orc.get_rays(pose, W, H, 13.0 * W)
pick = torch.randperm(W * H, generator=torch.Generator().manual_seed(1234))[:R]
o, d = o_all.reshape(-1, 3)[pick].float(), d_all.reshape(-1, 3)[pick].float()
caps = orc.capsule_tree(levels=5, seed=0)
z_gt = torch.linspace(0., 1., 160) * (far - near) + near
tgt = orc.project_mu(lambda p: orc.capsule_mu(p, caps), o, d, z_gt).reshape(-1)

# model as bench.py builds it
lin = [torch.nn.Linear(3, F)] + [torch.nn.Linear(F, F) for _ in range(N)] + [torch.nn.Linear(F, 1)]
with torch.no_grad():
    lin[-1].weight.mul_(4.0); lin[-1].bias.fill_(-5.0)
Ws = [l.weight.detach() for l in lin]; bs = [l.bias.detach() for l in lin]
if os.environ.get("TRAINED"):
    pass
step = (far - near) / S
t = near + (torch.arange(S).float() + 0.5) * step
x = (o[:, None, :] + d[:, None, :] * t[None, :, None]).reshape(-1, 3)

def rnd(a, fmt, scale=1.0):
    if fmt == "f32": return a
    if fmt == "bf16": return a.bfloat16().float()
    if fmt == "f16": return a.half().float()
    if fmt == "e4m3": return ((a * scale).clamp(-448, 448).to(torch.float8_e4m3fn).float()) / scale
    if fmt == "e5m2": return ((a * scale).clamp(-57344, 57344).to(torch.float8_e5m2).float()) / scale
    raise ValueError(fmt)

def forward(fmt):
    Hs = []
    h = torch.relu(x @ Ws[0].T + bs[0])          # first layer is always split (fp32-grade)
    Hs.append(h)
    for l in range(1, N + 1):
        h = torch.relu(rnd(h, fmt) @ rnd(Ws[l], fmt).T + bs[l])
        Hs.append(h)
    raw = (h @ Ws[-1].T + bs[-1]).reshape(R, S)
    sig = torch.sigmoid(raw)
    pix = torch.exp(-(sig * step).sum(-1))
    return pix, Hs, sig

def backward(pix, Hs, sig, fmt_b):
    g = (-(pix * 2 * (pix - tgt) / R))[:, None] * step * sig * (1 - sig)       # dL/draw [R,S]
    g = g.reshape(-1, 1)
    dZ = [None] * (N + 1)
    dz = (g @ Ws[-1]) * (Hs[N] > 0)
    dZ[N] = dz
    for l in range(N, 0, -1):
        dh = rnd(dz, fmt_b) @ rnd(Ws[l], fmt_b)
        dz = dh * (Hs[l - 1] > 0)
        dZ[l - 1] = dz
    return dZ

def rel(a, b): return float((a - b).norm() / b.norm())


pix32, H32, sig32 = forward("f32")
def gof(pix, sig):
    return ((-(pix * 2 * (pix - tgt) / R))[:, None] * step * sig * (1 - sig)).reshape(-1, 1)
def jchain(Hs, fmt_b):
    """normalised input-gradient chain J_l = dZ_l / g (starts from w_out), operands rounded to fmt_b"""
    J = [None] * (N + 1)
    j = Ws[-1].expand(Hs[N].shape[0], F) * (Hs[N] > 0)
    J[N] = j
    for l in range(N, 0, -1):
        j = (rnd(j, fmt_b) @ rnd(Ws[l], fmt_b)) * (Hs[l - 1] > 0)
        J[l - 1] = j
    return J
g32 = gof(pix32, sig32)
J32 = jchain(H32, "f32")
dW32 = [(g32 * J32[l]).T @ H32[l - 1] for l in range(1, N + 1)]
print("J abs max per layer", " ".join(f"{float(J32[l].abs().max()):.2e}" for l in range(N + 1)))
print("H abs max per layer", " ".join(f"{float(H32[l].abs().max()):.1f}" for l in range(N + 1)))
for fmt in ["f16"]:
    pix, Hs, sig = forward(fmt)
    g = gof(pix, sig)
    print(f"forward {fmt}: pixel rel-L2 {rel(pix, pix32):.3e}")
    for fb in ["f16"]:
        J = jchain(Hs, fb)
        for fh, sh, fz, sz, lastbf in [("f32", 1, "f32", 1, False), ("f16", 1, "f16", 1, False), ("e4m3", 0.25, "f16", 1, False),
                                       ("e5m2", 1.0, "f16", 1, False), ("e5m2", 1.0, "e5m2", 1024.0, True), ("e4m3", 0.25, "e5m2", 1024.0, True),
                                       ("e4m3", 0.25, "e4m3", 64.0, True), ("e4m3", 1.0 / 64, "e4m3", 4.0, True), ("e4m3", 4.0, "e4m3", 1024.0, True)]:
            errs = []
            for l in range(1, N + 1):
                fzz = "f16" if (lastbf and l == N) else fz
                A = rnd(J[l], fzz, sz)                      # stashed J
                B = (g * 2.0 ** 17) .half().float() * rnd(Hs[l - 1], fh, sh); B = B.half().float() / 2.0 ** 17   # wgrad: (g Ls) f16 x H f16 -> f16
                errs.append(rel(A.T @ B, dW32[l - 1]))
            print(f"  fwd {fmt} Jchain {fb} stash H {fh} J {fz}{' (J_N bf16)' if lastbf else ''}: " + " ".join(f"{e:.1e}" for e in errs))
