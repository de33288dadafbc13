import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from nerf_for_angiography_amd.model.CPPN import CPPN
from nerf_for_angiography_amd.engine import RenderSpec
dev = torch.device("cuda:0")
R, S = int(sys.argv[1]), int(sys.argv[2])
layers, width = int(sys.argv[3]), int(sys.argv[4])
torch.manual_seed(0)
md = dict(num_early_layers=layers, num_late_layers=0, num_filters=width, num_input_channels=3, num_output_channels=1, num_input_channels_views=0,
          use_bias=True, pos_enc="none", pos_enc_basis=5, act_func="relu", fourier_sigma=5, num_img=1, device=dev, precision="f16s8")
m = CPPN(md).to(dev)
o = torch.tensor([[0.0, 0.0, 1500.0]], device=dev).repeat(R, 1)
d = torch.nn.functional.normalize(torch.randn(R, 3, device=dev) * 0.03 + torch.tensor([0, 0, -1.0], device=dev), dim=-1)
tgt = torch.rand(R, device=dev)
spec = RenderSpec(n_rays=R, n_samples=S, origins=o, dirs=d, mode="acc", t_near=1400.0, t_far=1600.0)
eng, prepared = m.engine, m._prepared()
g0 = torch.zeros(eng.param_count, device=dev)
p0 = eng.train_step_mse(prepared, spec, tgt, 1.0 / R, g0, "f16s8")
torch.cuda.synchronize(); print("eager ok", float(p0.sum()), float(g0.abs().sum()), flush=True)
g1 = torch.zeros(eng.param_count, device=dev)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
graph = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(graph, stream=side):
        g1.zero_()
        p1 = eng.train_step_mse(prepared, spec, tgt, 1.0 / R, g1, "f16s8")
torch.cuda.current_stream().wait_stream(side)
print("captured", flush=True)
graph.replay(); torch.cuda.synchronize(); print("replay 1 ok", bool(torch.equal(p1, p0)), bool(torch.equal(g1, g0)), flush=True)
graph.replay(); graph.replay(); torch.cuda.synchronize(); print("replay 3 ok", bool(torch.equal(p1, p0)), bool(torch.equal(g1, g0)), flush=True)
