"""Training entry point (mirrors reference experiment_scripts/train_realestate10k.py + training.py:46-246).

    python experiment_scripts/train_realestate10k.py --experiment_name demo --views 2 --batch_size 12 --synthetic [--gpus N] [--max_steps K]

The reference's loop, with ``training.render_train`` (HIP forward + HIP backward, csrc/car_backward.hip) in the place of
``model(model_input)``: Adam(lr, betas=(0.99, 0.999)) (train_realestate10k.py:93), 192 random query rays per scene (query_sparsity,
train_realestate10k.py:78), L1 image loss (loss_functions.image_loss; --depth adds the reference's per-patch depth-variance term on 32 x 32
pixel patches, loss_functions.py:112-127, and samples the rays as such patches: --query_sparsity must then be a multiple of 1024),
gradient clipping at norm 1 (training.py:130-134), parameters broadcast from rank 0 and gradients all-reduced when --gpus > 1
(train_realestate10k.py:60-62, training.py:21-28: one process per GPU over RCCL), checkpoints ``{'model', 'optimizer'}`` as
``checkpoints/model_current.pth`` / ``model_final.pth`` (training.py:82-84, 244-246) that the eval / render scripts load.

Data: the RealEstate10K training set and its augmenting reader are not available offline, so scenes are synthetic (--synthetic, the
default here: seeded stereo pairs with a smooth random target image per scene; every step draws new rays).  The encoder trains when the
model is built with it (--with_encoder: the pyramid then comes from ``get_z`` under autograd); otherwise the pyramid itself is a leaf
that receives gradients, standing in for the encoder's output.  LPIPS (--lpips) needs the lpips package, which is not installed."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402


def _parser():
    p = common.parser(__doc__)
    p.add_argument("--lr", type=float, default=5e-5)
    p.add_argument("--l2_coeff", type=float, default=0.05)
    p.add_argument("--depth", action="store_true", default=False)
    p.add_argument("--lpips", action="store_true", default=False)
    p.add_argument("--max_steps", type=int, default=20)
    p.add_argument("--steps_til_summary", type=int, default=10)
    p.add_argument("--query_sparsity", type=int, default=192)
    p.set_defaults(batch_size=12, synthetic=True)
    return p


def train(rank, opt):
    import torch
    import torch.distributed as dist
    from cross_attention_renderer_amd import harness, synthetic
    from cross_attention_renderer_amd import training
    from cross_attention_renderer_amd.training import average_gradients, render_train
    if opt.lpips:
        raise SystemExit("--lpips needs the lpips package (not installed here)")
    dev = common.init_rank(rank, opt)
    H, b, R = opt.img_sidelength, opt.batch_size, opt.query_sparsity
    model = common.build_model(opt, dev).train()
    params = [p for p in model.parameters() if p.requires_grad]
    if opt.gpus > 1:                                              # sync_model (train_realestate10k.py:60-62)
        for p in params:
            dist.broadcast(p.data, 0)
    g = torch.Generator().manual_seed(1234 + rank)                 # every rank shuffles on its own (train_realestate10k.py:80-81)
    base = synthetic.stereo_scene(H, b=b, seed=5 + rank, n_view=opt.views)
    z = None
    if model.encoder.__class__.__name__ == "EncoderNotBuilt":
        # torch.channels_last memory: the renderer takes such a level as a view and returns its gradient in the same layout (no copies)
        z = [t.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in synthetic.feature_maps(b, opt.views, H, seed=1 + rank)]
    optimizer = training.make_adam(params, opt.lr)                  # the reference's Adam and parameter group: what the checkpoint stores
    # the stand-in pyramid (no encoder built) is a per-rank leaf with an optimizer of its own, so that the saved 'optimizer' state matches
    # the reference's param groups
    z_optimizer = training.make_adam(z, opt.lr) if z is not None else None
    if opt.depth and R % 1024:
        raise SystemExit("--depth: the reference's depth-variance term works on 32 x 32 pixel patches (loss_functions.py:112-127): "
                         "--query_sparsity must be a multiple of 1024")
    # a smooth random target image per scene: low-frequency colours of the pixel coordinates
    coef = (torch.rand(b, 3, 4, generator=g) * 2 - 1).to(dev)
    ckpt_dir = os.path.join(opt.logging_root, opt.experiment_name, "checkpoints")
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True)
    # the scene (context images, cameras) goes to the device ONCE; every step only draws new rays there.  (Round 3's loop rebuilt the
    # input dict on the host every step — twelve 65 536-element permutations on the CPU and a 19 MB upload of the context images — and
    # spent 100 ms per step in it; render_train itself queues a step in ~33 ms and the GPU needs ~45 ms: profiles/round4_train_step.md.)
    base = harness.to_device(base, dev, opt.cameras)
    grid = synthetic.pixel_grid(H, H).to(dev)
    gdev = torch.Generator(device=dev).manual_seed(4321 + rank)
    t0, losses, t_warm = time.time(), [], None
    for step in range(opt.max_steps):
        if step == min(2, opt.max_steps - 1):                     # steady-state clock: after the first steps' allocations and builds
            torch.cuda.synchronize()
            t_warm = (time.time(), step)
        if opt.depth:                                            # 32 x 32 pixel patches at random corners, row-major inside a patch
            gi = grid.view(H, H, 2)
            corners = torch.randint(0, H - 31, (b, R // 1024, 2), generator=g).tolist()
            uv = torch.stack([torch.cat([gi[y0:y0 + 32, x0:x0 + 32].reshape(1024, 2) for y0, x0 in corners[sc]]) for sc in range(b)])[:, None]
        else:                                                    # R distinct random pixels per scene (query_sparsity), drawn on the device
            uv = torch.stack([grid[torch.randperm(H * H, device=dev, generator=gdev)[:R]] for _ in range(b)])[:, None]   # (b, 1, R, 2)
        inp = {"context": base["context"], "query": dict(base["query"], uv=uv)}
        u = inp["query"]["uv"][:, 0] / (H - 1) * 3.14159
        feats = torch.stack([torch.sin(u[..., 0]), torch.cos(u[..., 1]), torch.sin(u[..., 0] + u[..., 1]), torch.ones_like(u[..., 0])], dim=-1)
        gt_rgb = torch.tanh(torch.einsum("brk,bck->brc", feats, coef))[:, None]                             # (b, 1, R, 3)
        out = model(inp, z=z)                                    # train() mode under autograd = training.render_train (the reference's call, training.py:92)
        loss = (gt_rgb - out["rgb"]).abs().mean()                                                           # loss_functions.image_loss
        if opt.depth:                                            # loss_functions.py:112-127: per-patch depth variance, masked per patch
            d = out["depth_ray"][..., 0].reshape(-1, 1, 32, 32)
            mean = d.mean(dim=-1).mean(dim=-1)[:, None, None]
            dist_ = opt.l2_coeff * torch.pow(d - mean, 2).mean(dim=-1).mean(dim=-1).mean(dim=-1)
            mask = torch.ones_like(dist_)                         # gt['mask']: every synthetic patch counts
            loss = loss + (dist_ * mask).mean()
        optimizer.zero_grad()
        if z_optimizer is not None:
            z_optimizer.zero_grad()
        loss.backward()
        if opt.gpus > 1:
            average_gradients(model)                              # the stand-in pyramid, if any, is per rank: not reduced
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
        optimizer.step()
        if z_optimizer is not None:
            z_optimizer.step()
        losses.append(loss.detach())                              # no .item() here: that would drain the queue every step
        if rank == 0 and (step % opt.steps_til_summary == 0 or step == opt.max_steps - 1):
            print(f"step {step}: loss {losses[-1].item():.5f}  ({(time.time() - t0) / (step + 1) * 1e3:.1f} ms/step since the start, {b} scenes x {R} rays)", flush=True)
            torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict()}, os.path.join(ckpt_dir, "model_current.pth"))
    torch.cuda.synchronize()
    if rank == 0:
        if t_warm is not None and opt.max_steps - t_warm[1] > 0:
            print(f"steady state: {(time.time() - t_warm[0]) / (opt.max_steps - t_warm[1]) * 1e3:.1f} ms per step over the last {opt.max_steps - t_warm[1]} steps "
                  f"(wall clock, checkpoint writes included)")
        torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict()}, os.path.join(ckpt_dir, "model_final.pth"))
        print(f"trained {opt.max_steps} steps: loss {losses[0].item():.5f} -> {losses[-1].item():.5f}; wrote {os.path.join(ckpt_dir, 'model_final.pth')}")
    if opt.gpus > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    opt = _parser().parse_args()
    common.spawn(train, opt)
