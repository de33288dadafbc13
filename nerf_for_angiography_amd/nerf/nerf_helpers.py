"""Mirror of the reference's nerf/nerf_helpers.py call surface (names, argument order, return arity),
with the arithmetic routed to the HIP library for GPU tensors.  Citations are upstream file:line."""
import numpy as np
import torch

from .. import engine as _engine


def randomize_depth(z_vals, device):
    """Stratified jitter (nerf/nerf_helpers.py:13-22).  With the 1-D z the driver passes, one jitter vector
    is shared by the whole ray batch."""
    mids = .5 * (z_vals[..., 1:] + z_vals[..., :-1])
    upper = torch.cat([mids, z_vals[..., -1:]], -1)
    lower = torch.cat([z_vals[..., :1], mids], -1)
    t_rand = torch.rand(z_vals.shape).to(device)
    return (lower + (upper - lower) * t_rand).to(device)


def get_minibatches(inputs, chunksize=1024 * 8):
    return [inputs[i:i + chunksize] for i in range(0, inputs.shape[0], chunksize)]


def get_predictions(model, flattened_query_points, chunksize, target_img_idx=None):
    """nerf/nerf_helpers.py:31-45.  Each chunk is one fused MLP launch; nothing but [P,1] is materialised."""
    predictions = []
    for batch in get_minibatches(flattened_query_points, chunksize=chunksize):
        if target_img_idx:
            indices = torch.Tensor([target_img_idx]).repeat((batch.shape[0], 1)).to(model.device)
            predictions.append(model(torch.cat((batch, indices), dim=-1)))
        else:
            predictions.append(model(batch))
    return torch.cat(predictions, dim=0)


def cumprod_exclusive(tensor):
    cumprod = torch.cumprod(tensor, -1)
    return torch.cat([torch.ones_like(cumprod[..., :1]), cumprod[..., :-1]], -1)


class _CompositeDenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, dirs, z):
        rgb, depth, w, ent, sig = _engine.composite_dense(raw, dirs, z)
        ctx.save_for_backward(raw, dirs, z, rgb)
        ctx.mark_non_differentiable(depth, w, ent, sig)
        return rgb, depth, w, ent, sig

    @staticmethod
    def backward(ctx, d_rgb, *unused):
        raw, dirs, z, rgb = ctx.saved_tensors
        return _engine.composite_dense_backward(raw, dirs, z, rgb, d_rgb.contiguous()), None, None


def get_ray_entropy(sigmas, rgb_map, threshold=0.4):
    weights_sum = torch.sum(sigmas, dim=-1)
    ray_density = sigmas / (weights_sum.unsqueeze(1) + 1e-10)
    ray_entropy = -torch.sum(ray_density * torch.log(ray_density + 1e-10), dim=-1)
    return ray_entropy * ((1 - rgb_map) > threshold).detach()


def render_volume_density(radiance_field, ray_directions, depth_values, raw_noise_std=0.):
    """nerf/nerf_helpers.py:59-123 -> (rgb_map, depth_map, weights, entropy, [sigma_a, rgb]).

    One output channel on a GPU runs in afx_composite_dense (gradient flows through rgb_map, the quantity
    the loss uses; depth_map/weights/entropy are returned detached).  The reference's quirks are kept:
    last distance 1e10 (D3), ||d|| scaling (D4), depth_map = sum(alpha*z) (D5)."""
    c = radiance_field.shape[-1]
    if c == 1 and radiance_field.is_cuda:
        raw = radiance_field[..., 0].float().contiguous()
        z = depth_values.float().contiguous()
        rgb_map, depth_map, weights, entropy, sigma_a = _CompositeDenseFn.apply(raw, ray_directions.float().contiguous(), z)
        rgb = torch.ones(sigma_a.shape[0], sigma_a.shape[1], 1, device=raw.device)
        return rgb_map, depth_map, weights, entropy, [sigma_a, rgb]
    one_e_10 = torch.tensor([1e10], dtype=ray_directions.dtype, device=ray_directions.device)
    dists = torch.cat((depth_values[..., 1:] - depth_values[..., :-1], one_e_10.expand(depth_values[..., :1].shape)), dim=-1)
    norm_dists = dists * torch.norm(ray_directions[..., None, :], dim=-1)
    if c == 2:
        sigma_a = torch.relu(radiance_field[..., -1])
        rgb = torch.sigmoid(radiance_field[..., :-1])
        alpha = 1. - torch.exp(-sigma_a * dists)
        weights = alpha * cumprod_exclusive(1. - alpha + 1e-10)
        rgb_map = torch.squeeze((weights[..., None] * rgb).sum(dim=-2))
        depth_map = (weights * depth_values).sum(dim=-1)
        alpha_sum = torch.sum(alpha, dim=-1)
        ray_density = alpha / (alpha_sum.unsqueeze(-1) + 1e-10)
        ray_entropy = -torch.sum(ray_density * torch.log(ray_density + 1e-10), dim=-1) * (alpha_sum > 0.7).detach()
        return rgb_map, depth_map, weights, torch.mean(ray_entropy), [sigma_a, rgb]
    sigma_a = torch.relu(torch.mean(radiance_field, dim=-1)) if c > 1 else torch.sigmoid(radiance_field[..., -1])
    rgb = torch.ones(sigma_a.shape[0], sigma_a.shape[1], 1).to(ray_directions.device)
    alpha = torch.exp(-sigma_a * norm_dists)
    weights = (1 - alpha + 1e-10) * cumprod_exclusive(alpha)
    rgb_map = torch.prod(alpha, dim=-1)
    depth_map = (alpha * depth_values).sum(dim=-1)
    return rgb_map, depth_map, weights, get_ray_entropy(sigma_a, rgb_map), [sigma_a, rgb]


def sample_pixel_rays(train_ray_df, img_sample_size, device, weights=None, unseen=False):
    """Weighted sampling without replacement over all pixels of all training projections, then shuffle
    (nerf/nerf_helpers.py:137-150) -> [origins[R,3], directions[R,3], pixel_values[R]] fp32 on device."""
    ray_batch = train_ray_df.sample(n=img_sample_size, weights=weights).sample(frac=1)
    batch_pix_vals = None
    if not unseen:
        batch_pix_vals = torch.from_numpy(ray_batch['pixel_value'].to_numpy()).to(device).float()
    batch_origins = torch.from_numpy(np.asarray(ray_batch['ray_origins'].tolist(), dtype=np.float32)).to(device)
    batch_directions = torch.from_numpy(np.asarray(ray_batch['ray_directions'].tolist(), dtype=np.float32)).to(device)
    return [batch_origins, batch_directions, batch_pix_vals]


def sample_image_rays(train_df, train_ray_df, img_sample_size, device, random=False):
    """nerf/nerf_helpers.py:152-176."""
    proj_id = train_df.sample(n=1).index[0]
    rays = train_ray_df[train_ray_df['image_id'] == proj_id].copy()
    if not random:
        side = int(np.sqrt(img_sample_size))
        pix = torch.zeros(side, side)
        pix[rays['x_position'].tolist(), rays['y_position'].tolist()] = torch.Tensor(rays['pixel_value'].tolist())
        pix = pix.flatten().to(device)
    else:
        rays = rays.sample(n=img_sample_size)
        pix = torch.Tensor(rays['pixel_value'].tolist()).to(device)
    origins = torch.from_numpy(np.asarray(rays['ray_origins'].tolist(), dtype=np.float32)).to(device)
    directions = torch.from_numpy(np.asarray(rays['ray_directions'].tolist(), dtype=np.float32)).to(device)
    return [origins, directions, pix]


def sample_pdf(bins, weights, N_samples, device, u=None):
    """Inverse-CDF sampling (nerf/nerf_helpers.py:197-222).  `u` lets the caller supply the uniform draw
    (parity tests); by default torch.rand, as upstream."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, dim=-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    if u is None:
        u = torch.rand(list(cdf.shape[:-1]) + [N_samples]).to(weights)
    inds = torch.searchsorted(cdf, u.contiguous(), right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_lo, bin_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)


def fine_sampling(depth_values, weights_coarse, ray_origins, ray_directions, coarse_model, fine_model,
                  depth_samples_per_ray_fine, chunksize, u=None):
    """Hierarchical re-sampling and re-rendering (nerf/nerf_helpers.py:178-195) ->
    (rgb_map_fine, depth_map_fine, entropy_fine).  The upstream call at :191 passes a stray positional
    argument to get_predictions and cannot run; it is restated here without it.  On a GPU
    the merged depths come from afx_fine_depths and the fine pass is the fused renderer."""
    from ..render import render_rays
    n_rays = ray_origins.shape[0]
    network = coarse_model if fine_model is None else fine_model
    if weights_coarse.is_cuda and getattr(network, "fused", False):
        if u is None:
            u = torch.rand(n_rays, depth_samples_per_ray_fine, device=weights_coarse.device)
        depth_vals = _engine.fine_depths(depth_values.float().contiguous(), weights_coarse.detach().float().contiguous(), u)
        out = render_rays(network, ray_origins, ray_directions, mode="dense", z=depth_vals, want_aux=True)
        return out.rgb_map, out.depth_map, out.entropy
    pdf_depth_values = depth_values.repeat(n_rays, 1) if depth_values.dim() == 1 else depth_values
    mids = .5 * (pdf_depth_values[..., 1:] + pdf_depth_values[..., :-1])
    samples = sample_pdf(mids, weights_coarse[..., 1:-1], depth_samples_per_ray_fine, ray_origins.device, u=u)
    depth_vals, _ = torch.sort(torch.cat([pdf_depth_values, samples.detach()], -1), -1)
    fine_pts = ray_origins[..., None, :] + ray_directions[..., None, :] * depth_vals[..., :, None]
    radiance = get_predictions(network, fine_pts.reshape((-1, 3)).float(), chunksize)
    rgb_map, depth_map, _, entropy, _ = render_volume_density(radiance.reshape(n_rays, -1, 1), ray_directions, depth_vals)
    return rgb_map, depth_map, entropy


def sample_depth(batch_directions, depth_samples_per_ray_coarse, device):
    """nerf/nerf_helpers.py:245-257."""
    near_thresh, far_thresh = 0, 1
    t_vals = torch.linspace(0., 1., depth_samples_per_ray_coarse).to(device)
    z_vals = near_thresh * (1. - t_vals) + far_thresh * t_vals
    return near_thresh, far_thresh, randomize_depth(z_vals.to(device), device)
