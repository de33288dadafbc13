"""Training driver — mirror of the reference's nerf/run_nerf_acc.py (flags :25-47, hyper-parameters :126-183,
loop :263-441) on the MI355X hot path.

Same command-line flags, model dictionary, ray sampling (`sample_pixel_rays` with `distance_pixel_value`
weights), BARF schedule, learning-rate decay, evaluation cadence, best-PSNR / vessel-PSNR checkpointing and
early stopping.  Differences: the dataset comes from `load_data` (which the reference calls but never defines) or
is synthesised in memory; TensorBoard/pyvista outputs are replaced by a JSONL log.

Three iteration bodies (--march): `dense` (default) is ONE fused launch per ray chunk (`train_step_mse`: ray -> samples ->
MLP -> Beer-Lambert -> MSE -> backward; render + autograd with --precision f32); `grid` is the reference's own body,
run_nerf_acc.py:284-306 - acc_update_n_step for both grids, acc_ray_marching with the occupancy grid (HIP march / visibility
kernels, nerf/occupancy.py), then positions / get_predictions / acc_render_volume_density / mse_loss / backward as ONE fused pass
over the march's packed samples, march included, in one library call (`march_train_step_mse`; f16s8); `grid_ops` is the same body call for call through
the mirrored functions (also what `grid` does at the other precisions).
The training rays live on the GPU: one table (origins, directions, pixel, weight) built once, and every iteration's
batch is drawn there (weighted sampling without replacement, `engine.sample_rays`); --host_sampler restores the
reference's per-iteration pandas draw (`sample_pixel_rays`).

    python -m nerf_for_angiography_amd.nerf.run_nerf_acc --synthetic --n_iters 2000 --num_layers 4 --num_hidden_units 128
"""
from __future__ import annotations

import argparse
import ast
import json
import os
import time

import numpy as np
import torch

from ..engine import RenderSpec
from ..model.CPPN import CPPN
from ..phantomdata import dataset as ds
from .. import engine as _engine
from ..render import render_rays, train_step_mse, march_train_step_mse
from .nerf_helpers import sample_pixel_rays, get_predictions
from .nerf_helpers_acc import acc_ray_marching, acc_render_volume_density, acc_update_n_step
from .occupancy import OccupancyGrid, ContractionType


def build_parser():
    p = argparse.ArgumentParser()
    # the reference's flags (nerf/run_nerf_acc.py:27-34), parsed as strings and cast the same way
    p.add_argument('--limited_size', help='Angle range to sample the projections in')
    p.add_argument('--number_angles', help='Number of projections to sample per axis')
    p.add_argument('--center_point', help='Center point for the angle sampling')
    p.add_argument('--binary', help='Whether images are binary or not')
    p.add_argument('--sampling_strategy', help='What sampling strategy to use, options: frangi, segmentation or random')
    p.add_argument('--data_name', help='Either CT data or LCA data')
    p.add_argument('--num_layers', help='Number of layers for MLP')
    p.add_argument('--num_hidden_units', help='Number of hidden units for MLP')
    # extensions
    p.add_argument('--synthetic', action='store_true', help='synthesise the dataset in memory instead of load_data')
    p.add_argument('--data_root', default='data')
    p.add_argument('--img_size', type=int, default=64)
    p.add_argument('--n_iters', type=int, default=500000)
    p.add_argument('--display_every', type=int, default=500)
    p.add_argument('--sample_size', type=int, default=75, help='rays per dimension per iteration (75^2 = 5625)')
    p.add_argument('--depth_samples', type=int, default=300)
    p.add_argument('--pos_enc', default='none', choices=['none', 'barf', 'fourier'])
    p.add_argument('--precision', default='f16s8', choices=['f32', 'bf16x3', 'bf16', 'f16', 'f16s8'],
                   help='arithmetic of the training step (include/afx.h); f16s8 = f16 with the backward stash kept as bf8')
    p.add_argument('--eval_precision', default='f16', choices=['f32', 'bf16x3', 'bf16', 'f16'])
    p.add_argument('--march', default='dense', choices=['dense', 'grid', 'grid_ops'],
                   help='dense: fused fixed-step march (one launch per ray chunk); grid: the reference loop with the occupancy grid, its body behind '
                        'the march (positions, get_predictions, acc_render_volume_density, mse, backward) as ONE fused packed step at the f16s8 '
                        'precision; grid_ops: the same loop call for call through the mirrored functions')
    p.add_argument('--adam', default='fused', choices=['fused', 'foreach'], help="PyTorch Adam implementation (same update rule)")
    p.add_argument('--host_sampler', action='store_true', help="draw each batch with pandas on the host (the reference's sample_pixel_rays)")
    p.add_argument('--log_dir', default='runs/afx')
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--out_bias_init', type=float, default=-5.0,
                   help='initial output bias. Dense marching has no nerfacc early termination: with the default '
                        'nn.Linear init sigma ~ 0.5 everywhere, the transmittance underflows to 0 and so does its '
                        'gradient; starting from near-empty space (sigmoid(-5) ~ 0.007) keeps training well-posed')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    if device.type != "cuda":
        raise SystemExit("run_nerf_acc: needs an MI355X; there is no CPU fallback")
    limited_size = float(args.limited_size) if args.limited_size is not None else 180.0
    number_angles = float(args.number_angles) if args.number_angles is not None else 4.0
    center_point = ast.literal_eval(args.center_point) if args.center_point is not None else [90, 0]
    binary = args.binary == 'True' if args.binary is not None else False
    sampling_strategy = args.sampling_strategy if args.sampling_strategy is not None else 'segmentation'
    data_name = args.data_name if args.data_name else 'ct'
    num_layers = int(args.num_layers) if args.num_layers else 4
    num_hidden_units = int(args.num_hidden_units) if args.num_hidden_units else 128
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)

    outside = 100
    file_name = f'limited-sparse-{limited_size}-{number_angles}-{center_point}' if binary else \
        f'background-{limited_size}-{number_angles}-{center_point}'
    if args.synthetic:
        proj_df, ray_df = ds.make_synthetic_dataset(ds.angle_grid(limited_size, int(number_angles), center_point),
                                                    img_size=args.img_size, sampling_strategy=sampling_strategy,
                                                    device=device, seed=args.seed)
    else:
        step_size = limited_size / number_angles if number_angles > 0 else limited_size
        proj_df, ray_df, _, _ = ds.load_data(data_name, file_name, False, binary, args.img_size, step_size, args.data_root)

    # held-out projection = last one (run_nerf_acc.py:85-99); [W,H] image indexing as upstream
    test_proj_id = proj_df.index[-1]
    test_ray_df = ray_df[ray_df['image_id'] == test_proj_id].copy()
    # (np.ascontiguousarray: a frame's multi-column to_numpy() is column-major, and .float().to(device) would keep those strides)
    cols = lambda df, stem: torch.from_numpy(np.ascontiguousarray(df[[f'{stem}_x', f'{stem}_y', f'{stem}_z']].to_numpy(), dtype=np.float32)).to(device)
    test_origins, test_directions = cols(test_ray_df, 'ray_origins'), cols(test_ray_df, 'ray_directions')
    test_x = torch.from_numpy(test_ray_df['x_position'].to_numpy().astype('int64')).to(device)
    test_y = torch.from_numpy(test_ray_df['y_position'].to_numpy().astype('int64')).to(device)
    img_width, img_height = int(test_x.max()) + 1, int(test_y.max()) + 1
    test_img = torch.zeros((img_width, img_height), device=device)
    test_img[test_x, test_y] = torch.from_numpy(test_ray_df['pixel_value'].to_numpy()).to(device).float()
    vessel = torch.from_numpy((test_ray_df['distance_pixel_value'] > test_ray_df['distance_pixel_value'].mean()).to_numpy()).to(device)

    train_ray_df = ray_df[ray_df['image_id'] != test_proj_id].copy()
    train_ray_df['ray_origins'] = train_ray_df[['ray_origins_x', 'ray_origins_y', 'ray_origins_z']].to_numpy().tolist()
    train_ray_df['ray_directions'] = train_ray_df[['ray_directions_x', 'ray_directions_y', 'ray_directions_z']].to_numpy().tolist()

    src_pt_z = float(proj_df['src_pt_z'].iloc[0])
    depth_samples_per_ray_coarse = args.depth_samples
    near_thresh, far_thresh = src_pt_z - outside, src_pt_z + outside

    n_iters, early_stop_iters = args.n_iters, 50000
    display_every, save_every = args.display_every, args.display_every * 100
    coarse_lr, decay_rate, decay_steps = 1e-4, 0.1, 500 * 1000
    img_sample_size = args.sample_size ** 2
    start_pos_enc_basis, pos_enc_basis, fourier_sigma = 0, 5, 5
    barf_start, barf_stop = 8000, 250000
    barf_step_size = pos_enc_basis / (barf_stop - barf_start)
    params = {'num_early_layers': num_layers, 'num_late_layers': 0, 'num_filters': num_hidden_units,
              'num_input_channels': 3, 'num_output_channels': 1, 'num_input_channels_views': 0, 'use_bias': True,
              'pos_enc': args.pos_enc, 'pos_enc_basis': pos_enc_basis, 'act_func': 'relu', 'fourier_sigma': fourier_sigma,
              'num_img': 1, 'device': device, 'precision': args.precision}
    coarse_model = CPPN(dict(params)).to(device)
    if coarse_model.use_pos_enc == 'barf':
        coarse_model.update_barf_alpha(start_pos_enc_basis, 'pts')
    if coarse_model.use_pos_enc == 'fourier' and args.precision == 'f32':
        coarse_model.fourier_coefficients.requires_grad_(False)     # the f32 kernels take them as constants; the 16-bit ones train them
    with torch.no_grad():
        coarse_model.output_linear[0].bias.fill_(args.out_bias_init)
    # run_nerf_acc.py:206.  fused=True: PyTorch's single multi-tensor Adam kernel instead of its foreach sequence of ~7 launches - same
    # update, and on the reference's 1.3 ms iteration the difference is 10 % (host-side dispatch: 0.95 -> 0.55 ms with the grid march)
    coarse_optimizer = torch.optim.Adam(list(coarse_model.parameters()), lr=coarse_lr, fused=(args.adam == 'fused'))

    # device-resident ray table (R13): built once; every batch is drawn and gathered on the GPU
    tab_o, tab_d = cols(train_ray_df, 'ray_origins'), cols(train_ray_df, 'ray_directions')
    tab_pix = torch.from_numpy(train_ray_df['pixel_value'].to_numpy()).float().to(device)
    tab_w = torch.from_numpy(train_ray_df['distance_pixel_value'].to_numpy()).float().to(device)

    # occupancy grid of the reference loop (run_nerf_acc.py:196-198); thresholds :68-70
    early_stop_eps, alpha_thre, vessel_alpha_thre = 1e-2, 1e-4, 5e-2
    scene_aabb = torch.tensor([-outside, -outside, -outside, outside, outside, outside], dtype=torch.float32, device=device)
    acc_grid = OccupancyGrid(roi_aabb=scene_aabb, resolution=128, contraction_type=ContractionType.AABB, seed=args.seed).to(device) \
        if args.march != 'dense' else None
    # the reference's second grid (:198,286): same updates at the vessel threshold; it only feeds the exported occupancy volumes (:362-367)
    vessel_acc_grid = OccupancyGrid(roi_aabb=scene_aabb, resolution=128, contraction_type=ContractionType.AABB, seed=args.seed + 1).to(device) \
        if args.march != 'dense' else None
    packed_step = args.march == 'grid' and args.precision == 'f16s8'      # else the operator sequence
    batch_size = 131072

    os.makedirs(args.log_dir, exist_ok=True)
    log = open(os.path.join(args.log_dir, 'train_log.jsonl'), 'a')
    highest_psnr, highest_iter, history = 0.0, 0, []
    if not args.host_sampler:      # the batches of 16 iterations per launch sequence
        ray_batches = _engine.RayBatchSampler(tab_o, tab_d, tab_pix, tab_w, img_sample_size, seed=args.seed, prefetch=16)
    new_lr_coarse = coarse_lr
    loss_coarse = torch.tensor(float('nan'), device=device)
    n_marched = 0
    t_last = time.time()
    for n_iter in range(n_iters + 1):
        coarse_model.train()
        if coarse_model.use_pos_enc == 'barf' and barf_start <= n_iter < barf_stop:
            coarse_model.update_barf_alpha(coarse_model.barf_alpha + barf_step_size, 'pts')
        if args.host_sampler:
            batch_origins, batch_directions, batch_pix_vals = sample_pixel_rays(train_ray_df, img_sample_size, device,
                                                                               weights='distance_pixel_value')
        else:
            batch_origins, batch_directions, batch_pix_vals, _ = ray_batches.draw(n_iter)      # == sample_rays(..., seed, stream_id=n_iter)
        coarse_optimizer.zero_grad()
        if args.march != 'dense':
            # the reference's iteration body, run_nerf_acc.py:284-306
            with torch.no_grad():
                acc_grid.train()
                vessel_acc_grid.train()
                acc_grid = acc_update_n_step(acc_grid, coarse_model, n_iter, occ_thre=alpha_thre)
                vessel_acc_grid = acc_update_n_step(vessel_acc_grid, coarse_model, n_iter, occ_thre=vessel_alpha_thre)
            if packed_step:
                # :287-306 - march, alpha pass, visibility and the fused packed step - as ONE library call (the entry points of the
                # operator branch below, in the same order: ~30 launches that a Python loop issues slower than the GPU runs them)
                loss_k, pred_k, n_kept = march_train_step_mse(coarse_model, acc_grid, scene_aabb, batch_origins, batch_directions,
                                                              depth_samples_per_ray_coarse, near_thresh, far_thresh, early_stop_eps,
                                                              alpha_thre, batch_pix_vals)
                ray_indices = range(n_kept)      # (only its length is used below: the reference steps when the march kept samples)
                if n_kept:
                    loss_coarse, pred = loss_k, pred_k
                    n_marched += n_kept
            else:
                with torch.no_grad():
                    ray_indices, t_starts, t_ends = acc_ray_marching(coarse_model, acc_grid, scene_aabb, batch_origins, batch_directions,
                                                                     depth_samples_per_ray_coarse, near_thresh, far_thresh, early_stop_eps,
                                                                     alpha_thre)
                if len(ray_indices) > 0:
                    positions = batch_origins[ray_indices.long()] + batch_directions[ray_indices.long()] * (t_starts + t_ends) / 2.0
                    predictions = get_predictions(coarse_model, positions, batch_size)
                    pred, _ = acc_render_volume_density(predictions, ray_indices, t_starts, t_ends, img_sample_size,
                                                        depth_samples_per_ray_coarse)
                    loss_coarse = torch.nn.functional.mse_loss(pred, batch_pix_vals)
                    loss_coarse.backward()
                    n_marched += int(len(ray_indices))
        elif args.precision == 'f32':
            pred = render_rays(coarse_model, batch_origins, batch_directions, depth_samples_per_ray_coarse, near_thresh,
                               far_thresh, mode='acc').rgb_map
            loss_coarse = torch.nn.functional.mse_loss(pred, batch_pix_vals)
            loss_coarse.backward()
        else:
            loss_coarse, pred = train_step_mse(coarse_model, RenderSpec(
                n_rays=img_sample_size, n_samples=depth_samples_per_ray_coarse, origins=batch_origins,
                dirs=batch_directions, mode='acc', t_near=near_thresh, t_far=far_thresh), batch_pix_vals)
        if args.march == 'dense' or len(ray_indices) > 0:      # (the reference steps only when the march kept samples, :293)
            coarse_optimizer.step()
        new_lr_coarse = coarse_lr * (decay_rate ** (n_iter / decay_steps))
        for param_group in coarse_optimizer.param_groups:
            param_group['lr'] = new_lr_coarse

        if n_iter % display_every == 0:
            coarse_model.eval()
            keep, coarse_model.precision = coarse_model.precision, args.eval_precision
            with torch.no_grad():
                if args.march != 'dense':          # run_nerf_acc.py:338-349
                    ri_t, ts_t, te_t = acc_ray_marching(coarse_model, acc_grid, scene_aabb, test_origins, test_directions,
                                                        depth_samples_per_ray_coarse, near_thresh, far_thresh, early_stop_eps, alpha_thre)
                    pos_t = test_origins[ri_t.long()] + test_directions[ri_t.long()] * (ts_t + te_t) / 2.0
                    test_pred, _ = acc_render_volume_density(get_predictions(coarse_model, pos_t, batch_size) if len(ri_t) else pos_t[:, :1],
                                                             ri_t, ts_t, te_t, test_origins.shape[0], depth_samples_per_ray_coarse)
                else:
                    test_pred = render_rays(coarse_model, test_origins, test_directions, depth_samples_per_ray_coarse,
                                            near_thresh, far_thresh, mode='acc').rgb_map
            coarse_model.precision = keep
            pred_img = torch.zeros_like(test_img)
            pred_img[test_x, test_y] = test_pred
            mse = torch.nn.functional.mse_loss(pred_img, test_img)
            psnr = float(-10. * torch.log10(mse))
            vessel_psnr = float(-10. * torch.log10(torch.nn.functional.mse_loss(test_pred[vessel], test_img[test_x, test_y][vessel])))
            rec = dict(iter=n_iter, train_loss=float(loss_coarse.detach()), train_psnr=float(-10. * torch.log10(loss_coarse.detach())),
                       test_psnr=psnr, test_vessel_psnr=vessel_psnr, lr=new_lr_coarse,
                       barf_alpha=float(getattr(coarse_model, 'barf_alpha', 0.0)), sec=round(time.time() - t_last, 3),
                       it_per_s=round(display_every / max(time.time() - t_last, 1e-9), 1) if n_iter else 0.0,
                       marched_samples_per_iter=(n_marched // max(display_every, 1)) if args.march != 'dense' else
                       img_sample_size * depth_samples_per_ray_coarse)
            n_marched = 0
            t_last = time.time()
            history.append(rec)
            log.write(json.dumps(rec) + "\n")
            log.flush()
            print(rec, flush=True)
            if not np.isfinite(rec['train_loss']):
                raise FloatingPointError(
                    f"non-finite training loss at iteration {n_iter} (precision {args.precision}): the f16 precisions hold hidden "
                    "activations up to 65504 (include/afx.h); re-run with --precision bf16 (fp32 exponent range) or bf16x3")
            if psnr > highest_psnr:
                highest_psnr, highest_iter = psnr, n_iter
                coarse_model.save(os.path.join(args.log_dir, 'coarsemodel.pth'),
                                  {'epochs': n_iter, 'psnr': psnr, 'vessel_psnr': vessel_psnr})
                if acc_grid is not None:      # the occupancy volumes the reference writes as VTK next to the best model (:362-367,384-385)
                    np.save(os.path.join(args.log_dir, 'acc_grid_binary.npy'), acc_grid.binary.cpu().numpy())
                    np.save(os.path.join(args.log_dir, 'vessel_acc_grid_binary.npy'), vessel_acc_grid.binary.cpu().numpy())
            if n_iter % save_every == 0 and n_iter > 0:
                coarse_model.save(os.path.join(args.log_dir, f'coarsemodel-{n_iter}.pth'), {'epochs': n_iter})
            if n_iter - highest_iter > early_stop_iters:
                print('early stopping at', n_iter)
                break
    log.close()
    return dict(history=history, best_psnr=highest_psnr, best_iter=highest_iter, model=coarse_model,
                optimizer=coarse_optimizer, test_image=test_img, log_dir=args.log_dir, acc_grid=acc_grid,
                vessel_acc_grid=vessel_acc_grid)


if __name__ == "__main__":
    main()
