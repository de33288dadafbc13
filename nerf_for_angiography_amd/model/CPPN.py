"""CPPN — host-side mirror of the reference model class (model/CPPN.py:6-300 upstream).

Same constructor dictionary, attributes, method names and state-dict keys as the reference, so
checkpoints and the training loop carry over.  What differs is where the arithmetic runs: for the
configuration the reference trains (ReLU, no skip block, no view directions, one output channel —
nerf/run_nerf_acc.py:168-183) every Linear's weight and bias is a view into ONE flat fp32 buffer,
and forward/backward on a GPU tensor run in the fused HIP kernels behind include/afx.h.  Other
configurations (tanh / sine / skip block / view head) keep the module semantics through
PyTorch-ROCm operators; they are not on the accelerated path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.optim.optimizer import register_optimizer_step_post_hook


# Optimizer steps the version counters cannot see: torch.optim's FUSED implementations (Adam(fused=True) ...) update the parameters through
# one multi-tensor kernel that bumps no tensor's `_version`, so a cache of re-tiled weights keyed on versions alone goes stale after them
# (seen: 10 fused Adam steps rendered with the initial weights).  A global post-hook counts every optimizer step of the process; the count is
# part of the cache key (CPPN._prepared) - at worst one re-tiling launch (~5 us) per call behind an unrelated optimizer's step.
_OPT_STEPS = [0]


def _count_optimizer_step(optimizer, args, kwargs):
    _OPT_STEPS[0] += 1


register_optimizer_step_post_hook(_count_optimizer_step)


class Sine(nn.Module):
    """act(x; w0) = sin(w0 * x)  (model/CPPN.py:278-300)."""

    def __init__(self, w0: float = 1.0):
        super().__init__()
        self.w0 = w0

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not isinstance(x, torch.Tensor):
            raise TypeError("input to forward() must be torch.xTensor")
        return torch.sin(self.w0 * x)


class _MlpFn(torch.autograd.Function):
    """raw = MLP(points) through afx_mlp_infer / afx_mlp_backward."""

    @staticmethod
    def forward(ctx, model, pts, *params):
        prepared = model._prepared()
        out = model.engine.infer(prepared, pts, model.precision)
        ctx.model = model
        ctx.save_for_backward(pts)
        ctx.n_params = len(params)
        return out.unsqueeze(-1)

    @staticmethod
    def backward(ctx, d_out):
        model = ctx.model
        (pts,) = ctx.saved_tensors
        flat_grad = torch.zeros(model.engine.param_count, dtype=torch.float32, device=pts.device)
        coef_grad = model._coef_grad_buffer()
        with model.engine.encoding_grad(model._flat, coef_grad):
            model.engine.mlp_backward(model._prepared(), pts, d_out.reshape(-1).contiguous(), flat_grad, model.precision)
        return (None, None) + model._fn_grads(flat_grad, coef_grad)


class CPPN(nn.Module):
    """A CPPN model, mapping input coordinates to a multidimensional output (model/CPPN.py:6)."""

    def __init__(self, model_definition: dict) -> None:
        super().__init__()
        md = model_definition
        self.version = "v0.00"
        self.model_definition = md
        self.device = md["device"]
        self.num_early_layers = md["num_early_layers"]
        self.num_late_layers = md["num_late_layers"]
        self.num_filters = md["num_filters"]
        self.num_input_channels = md["num_input_channels"]
        self.num_input_channels_views = md["num_input_channels_views"]
        self.num_output_channels = md["num_output_channels"]
        self.use_bias = md["use_bias"]
        self.use_pos_enc = md["pos_enc"]
        self.num_img = md["num_img"]
        self.mult_img = self.num_img > 1
        self.use_viewdirs = self.num_input_channels_views > 0
        self.enc_fun = None
        self.precision = md.get("precision", "f32")      # arithmetic of the fused kernels (extension)

        self.first_act_func, self.act_func = nn.ReLU(), nn.ReLU()
        self._act_name = md["act_func"]
        if md["act_func"] == "sine":
            self.first_act_func, self.act_func = Sine(w0=md["sine_weights"]), Sine()
        elif md["act_func"] == "tanh":
            self.first_act_func, self.act_func = nn.Tanh(), nn.Tanh()

        n_in, n_in_v = self.num_input_channels, self.num_input_channels_views
        if self.use_pos_enc != "none":
            self.pos_enc_basis = md["pos_enc_basis"]
            n_in = self.num_input_channels * (1 + 2 * self.pos_enc_basis)
            if self.use_viewdirs:
                self.pos_enc_basis_views = md["pos_enc_basis_views"]
                n_in_v = self.num_input_channels_views * (1 + 2 * self.pos_enc_basis_views)
            if self.use_pos_enc == "fourier" and "fourier_sigma" in md:
                self.enc_fun = self.fourier_pos_enc
                self.fourier_sigma = md["fourier_sigma"]
                self.fourier_coefficients = nn.Parameter(
                    torch.randn([self.num_input_channels * self.pos_enc_basis]) * self.fourier_sigma)
                if self.use_viewdirs:
                    self.fourier_coefficients_views = nn.Parameter(
                        torch.randn([self.num_input_channels_views * self.pos_enc_basis_views]) * self.fourier_sigma)
            if self.use_pos_enc == "barf":
                self.enc_fun = self.barf_pos_enc
                self.k_values = torch.repeat_interleave(torch.arange(0., self.pos_enc_basis), self.num_input_channels)
                self.barf_freq = torch.Tensor(2 ** self.k_values * np.pi).to(self.device)
                if self.use_viewdirs:
                    self.k_values_views = torch.repeat_interleave(torch.arange(0., self.pos_enc_basis_views),
                                                                  self.num_input_channels_views)
                    self.barf_freq_views = torch.Tensor(2 ** self.k_values_views * np.pi).to(self.device)
                    self.update_barf_alpha(0, "views")
                self.update_barf_alpha(0, "pts")

        f, bias = self.num_filters, self.use_bias
        # ModuleList += Sequential extends with the Sequential's children, which is what gives the
        # reference its early_pts_layers.{0,2,4,..} state-dict keys.
        early = []
        early += self._layer(n_in, f, bias, self.first_act_func)
        for _ in range(self.num_early_layers):
            early += self._layer(f, f, bias, self.act_func)
        self.early_pts_layers = nn.ModuleList(early)
        if self.num_late_layers > 0:
            self.skip_connection = self._layer(f + n_in, f, bias, self.act_func)
            late = []
            for _ in range(self.num_late_layers - 1):
                late += self._layer(f, f, bias, self.act_func)
            self.late_pts_layers = nn.ModuleList(late)
        if self.use_viewdirs:
            self.views_layers = self._layer(n_in_v + f, f // 2, bias, self.act_func)
            self.alpha_linear = self._layer(f, self.num_output_channels - 1, bias, None)
            self.feature_linear = nn.Linear(f, f)
            self.rgb_linear = self._layer(f // 2, self.num_output_channels - 1, bias, None)
        else:
            self.output_linear = self._layer(f, self.num_output_channels, bias, None)

        self.img1 = nn.Parameter(torch.tensor([0., 0.], dtype=torch.float))
        self.img2 = nn.Parameter(torch.tensor([0., 0.], dtype=torch.float))
        self.store_activations = False
        self.activation_dictionary = {}

        self._engine = None
        self._flat = None
        self._enc_aux_key = None
        self._flatten()

    # ---- construction helpers ------------------------------------------------------------------
    @staticmethod
    def _layer(n_in, n_out, use_bias, activation=None):
        block = [nn.Linear(n_in, n_out, bias=use_bias)]
        if activation:
            block.append(activation)
        return nn.Sequential(*block)

    @property
    def fused(self) -> bool:
        """True when this configuration runs in the fused HIP kernels in BOTH directions (training included): ReLU models at every
        precision; tanh / sine models at precision "f32" (the exact-fp32 chain kernel keeps the activation's slope per element where
        ReLU keeps one bit; the 16-bit kernels are forward-only for them)."""
        return (self._act_name == "relu" or self.precision == "f32") and self.fused_forward

    @property
    def fused_forward(self) -> bool:
        """True when the FORWARD kernels take this configuration: the reference's trained geometry (no skip block, no view
        directions, one output channel) with any of its three activations.  tanh / sine models are evaluated by the kernels
        wherever no gradient is wanted (inference, evaluation renders, density grids: the activation is an epilogue of the same
        chain kernel); with gradients they train in the exact-fp32 kernels (precision "f32", see `fused`) and otherwise keep the
        module's PyTorch-ROCm operators."""
        return ((self._act_name == "relu" or (self._act_name in ("tanh", "sine") and self.use_pos_enc == "none"))
                and self.num_late_layers == 0 and not self.use_viewdirs
                and self.num_output_channels == 1 and self.num_input_channels == 3 and self.use_bias
                and self.num_filters in (64, 128, 256) and 1 <= self.num_early_layers <= 16
                and self.use_pos_enc in ("none", "barf", "fourier")
                and (self.use_pos_enc == "none" or 2 * ((4 + 6 * self.pos_enc_basis) // 2) <= self.num_filters))

    def _linears(self):
        lins = [m for m in self.early_pts_layers if isinstance(m, nn.Linear)]
        return lins + [self.output_linear[0]]

    def _layout(self):
        """(w_off, b_off, rows, cols) per Linear in the flat buffer — same formula as afx_param_layout."""
        f = self.num_filters
        k0 = self.num_input_channels * (1 + 2 * self.pos_enc_basis) if self.use_pos_enc != "none" else self.num_input_channels
        out, off = [], 0
        for rows, cols in [(f, k0)] + [(f, f)] * self.num_early_layers + [(1, f)]:
            out.append((off, off + rows * cols, rows, cols))
            off += rows * cols + rows
        return out, off

    def _flatten(self):
        """Make every Linear's weight/bias a view into one flat fp32 buffer (the C-ABI's parameter layout)."""
        if not self.fused_forward:
            return
        lins = self._linears()
        layout, total = self._layout()
        dev = lins[0].weight.device
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        for lin, (wo, bo, rows, cols) in zip(lins, layout):
            flat[wo:wo + rows * cols].view(rows, cols).copy_(lin.weight.data)
            flat[bo:bo + rows].copy_(lin.bias.data)
            lin.weight.data = flat[wo:wo + rows * cols].view(rows, cols)
            lin.bias.data = flat[bo:bo + rows]
        self._flat = flat

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        if self.use_pos_enc == "barf":
            self.barf_freq = fn(self.barf_freq)
        self._flatten()
        return self

    # ---- fused-kernel plumbing -----------------------------------------------------------------
    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    @property
    def engine(self):
        if self._engine is None:
            from ..engine import Engine
            self._engine = Engine(self.num_filters, self.num_early_layers, self.use_pos_enc,
                                  self.pos_enc_basis if self.use_pos_enc != "none" else 0, act=self._act_name,
                                  act_w0=float(self.model_definition.get("sine_weights", 1.0)) if self._act_name == "sine" else 1.0)
            if self._engine.param_count != self._flat.numel():
                raise RuntimeError("flat parameter layout disagrees with afx_param_layout")
        return self._engine

    def _enc_aux(self):
        dev = self._flat.device
        if self.use_pos_enc == "barf":
            return torch.cat([self.barf_freq.detach().to(dev, torch.float32),
                              self.barf_weights.detach().to(dev, torch.float32)]).contiguous()
        if self.use_pos_enc == "fourier":
            return self.fourier_coefficients.detach().to(dev, torch.float32).contiguous()
        return None

    def invalidate(self):
        """Drop the cached re-tiled weights.  Needed after writes the version counters cannot see: `p.data.copy_(...)`,
        `p.data.normal_()`, initialisers applied through `.data`, EMA code.  (Optimizer steps - every torch.optim step of the process is
        counted, fused implementations included -, `load_state_dict`, in-place ops on the parameters and writes to `flat_params` are
        tracked automatically.)"""
        if self._engine is not None:
            self._engine._prepared.clear()

    def _check_views(self):
        """Every Linear's weight / bias must still alias the flat buffer the kernels read: rebinding `p.data = ...`
        (e.g. torch.nn.utils.vector_to_parameters) detaches it.  Cheap pointer comparison; re-flatten when broken."""
        layout, _ = self._layout()
        base, esz = self._flat.data_ptr(), self._flat.element_size()
        for lin, (wo, bo, rows, cols) in zip(self._linears(), layout):
            if lin.weight.data_ptr() != base + wo * esz or lin.bias.data_ptr() != base + bo * esz:
                self._flatten()
                self.invalidate()
                return

    def _prepared(self):
        """Prepared (re-tiled) weights for the current parameter values.  Cached on the version counters of the
        parameters (bumped by in-place updates / load_state_dict), the process-wide count of optimizer steps (fused optimizers
        bump no version counter) and the version of the flat buffer itself (bumped by direct writes such as a broadcast);
        parameters that no longer alias the flat buffer are re-flattened first.
        Writes through `.data` bump no counter: call `invalidate()` after them."""
        self._check_views()
        aux_key = None
        if self.use_pos_enc == "barf":
            aux_key = float(self.barf_alpha)
        elif self.use_pos_enc == "fourier":
            aux_key = self.fourier_coefficients._version
        key = (self._flat._version, self._flat.data_ptr(), self.precision, aux_key, _OPT_STEPS[0],
               tuple(p._version for p in self._hip_params()))
        cached = self.engine._prepared.get(self.precision)
        if cached is not None and cached[1] == key:
            return cached[0]
        return self.engine.prepare(self._flat, self._enc_aux(), self.precision, key)

    def _hip_params(self):
        out = []
        for lin in self._linears():
            out += [lin.weight, lin.bias]
        return out

    def _coef_trainable(self):
        return self.use_pos_enc == "fourier" and self.fourier_coefficients.requires_grad

    def _fn_params(self):
        """Inputs of the autograd Functions: the Linear parameters, then the fourier coefficients when they train."""
        return self._hip_params() + ([self.fourier_coefficients] if self._coef_trainable() else [])

    def _coef_grad_buffer(self):
        """Zeroed accumulator for d loss / d fourier_coefficients (None when they do not train); the upstream module makes
        them an nn.Parameter (model/CPPN.py:92), so the optimiser updates them."""
        if not self._coef_trainable():
            return None
        if self.precision == "f32":
            raise NotImplementedError("trainable fourier_coefficients need a 16-bit precision of the fused kernels "
                                      "(f16s8, f16, bf16, bf16x3); with 'f32' freeze them: "
                                      "model.fourier_coefficients.requires_grad_(False)")
        return torch.zeros(self.fourier_coefficients.numel(), dtype=torch.float32, device=self._flat.device)

    def _fn_grads(self, flat_grad, coef_grad):
        out = self._split_grad(flat_grad)
        if self._coef_trainable():
            out = out + (coef_grad.view_as(self.fourier_coefficients),)
        return out

    def _split_grad(self, flat_grad):
        layout, _ = self._layout()
        out = []
        for wo, bo, rows, cols in layout:
            out += [flat_grad[wo:wo + rows * cols].view(rows, cols), flat_grad[bo:bo + rows]]
        return tuple(out)

    # ---- reference API -------------------------------------------------------------------------
    def activations(self, store_activations: bool) -> None:
        self.store_activations = store_activations
        if not store_activations:
            self.activation_dictionary = {}

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda and x.shape[-1] == self.num_input_channels:
            if self.fused:
                pts = x.reshape(-1, 3).float().contiguous()
                out = _MlpFn.apply(self, pts, *self._fn_params())
                return out.reshape(*x.shape[:-1], 1)
            if self.fused_forward and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))):
                # tanh / sine: the forward kernels, where no gradient is being recorded
                pts = x.reshape(-1, 3).float().contiguous()
                return self.engine.infer(self._prepared(), pts, self.precision).reshape(*x.shape[:-1], 1)
        return self._forward_ops(x)

    def _forward_ops(self, x: torch.Tensor) -> torch.Tensor:
        """Module semantics through PyTorch operators (model/CPPN.py:166-205): configurations outside the
        fused path, and host tensors."""
        input_pts, input_views = torch.split(x, [self.num_input_channels, self.num_input_channels_views], dim=-1)
        pts_encoded, views_encoded = input_pts, input_views
        if self.use_pos_enc != "none":
            pts_encoded = self.pos_enc(input_pts, self.pos_enc_basis, "pts")
            if self.use_viewdirs:
                views_encoded = self.pos_enc(input_views, self.pos_enc_basis_views, "views")
        values = pts_encoded
        for layer in self.early_pts_layers:
            values = layer(values)
        if self.num_late_layers > 0:
            values = self.skip_connection(torch.cat([pts_encoded, values], dim=-1))
            for layer in self.late_pts_layers:
                values = layer(values)
        if self.use_viewdirs:
            alpha = self.alpha_linear(values)
            values = torch.cat([self.feature_linear(values), views_encoded], -1)
            for layer in self.views_layers:
                values = layer(values)
            return torch.cat([self.rgb_linear(values), alpha], -1)
        return self.output_linear(values)

    def pos_enc(self, values, pos_enc_basis, type):
        """[x, enc_sin(x tiled L times), enc_cos(...)] (model/CPPN.py:207-214); the fused kernels evaluate the same terms."""
        if pos_enc_basis <= 0:
            return values
        tiled = values.repeat(*([1] * (values.dim() - 1)), pos_enc_basis)        # x, y, z, x, y, z, ...
        waves = self.enc_fun(tiled, type)
        return torch.cat((values, *waves), dim=-1)

    def _enc_tables(self, type):
        views = type == "views"
        if self.use_pos_enc == "fourier":
            return (self.fourier_coefficients_views if views else self.fourier_coefficients), None
        return (self.barf_freq_views if views else self.barf_freq), (self.barf_weights_views if views else self.barf_weights)

    def fourier_pos_enc(self, values, type):
        """sin / cos of 2 pi x c with learnable c ~ N(0, sigma^2) (model/CPPN.py:216-222)."""
        coeff, _ = self._enc_tables(type)
        phase = 2 * np.pi * values * coeff
        return [phase.sin(), phase.cos()]

    def barf_pos_enc(self, values, type):
        """w_k sin / cos(2^k pi x) with the coarse-to-fine weights w_k of update_barf_alpha (model/CPPN.py:224-234)."""
        freq, weights = self._enc_tables(type)
        phase = freq.to(values.device) * values
        weights = weights.to(values.device)
        return [weights * phase.sin(), weights * phase.cos()]

    def update_barf_alpha(self, barf_alpha, type):
        """Advance the coarse-to-fine schedule (model/CPPN.py:236-242)."""
        if type == "views":
            self.barf_alpha_views, self.barf_weights_views = barf_alpha, self.barf_coefficients(barf_alpha, self.k_values_views)
        else:
            self.barf_alpha, self.barf_weights = barf_alpha, self.barf_coefficients(barf_alpha, self.k_values)

    def barf_coefficients(self, barf_alpha, k_values):
        """Coarse-to-fine weight per frequency band k for progress alpha (model/CPPN.py:244-259): 0 before the band opens
        (alpha < k + 1), 1 once it is fully open (alpha >= k + 2), a raised-cosine ramp in between - literal incl. the
        3.1415 constant and the (alpha - k + 1) argument of the ramp.  Returned as a fresh nn.Parameter (D8) so
        that state_dict()['barf_weights'] exists, but frozen: it is a schedule, not a weight."""
        k = torch.as_tensor(k_values, dtype=torch.float32)
        opened = barf_alpha - (k + 1)
        ramp = (1 - torch.cos((barf_alpha - k + 1) * 3.1415)) / 2
        w = torch.where(opened < 0, torch.zeros_like(k), torch.where(opened < 1, ramp, torch.ones_like(k)))
        return nn.Parameter(w.to(torch.float32), requires_grad=False)

    def save(self, filename: str, training_information: dict) -> None:
        """Same checkpoint dictionary as the reference (model/CPPN.py:261-276)."""
        torch.save({"version": self.version, "parameters": self.model_definition,
                    "training_information": training_information, "model": self.state_dict()}, f=filename)
