"""Shared CLI of the eval / render entry points.  Flag names follow the reference's scripts
(eval_realestate10k.py:34-72, render_realestate10k_traj.py:37-82); configargparse is not installed here, so plain argparse.

Differences from the reference, all additive:
  --gpus N        shards the RAYS of every frame over N processes (RCCL all-gather of the tiles) instead of spawning N
                  identical replicas (eval_realestate10k.py:95-99);
  --synthetic     seeded synthetic stereo pair + feature pyramid (default when --data_root is absent: the datasets and the
                  DPT encoder weights are not available offline);
  --out_dir       where frames (PNG + NPY) and metrics are written.
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parser(description: str) -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--experiment_name", type=str, required=True)
    p.add_argument("--logging_root", type=str, default="logs")
    p.add_argument("--data_root", type=str, default=None, help="directory of scene directories (one *.npz of frames each)")
    p.add_argument("--pose_root", type=str, default=None, help="directory of <scene>.txt camera files (RealEstate10K format)")
    p.add_argument("--checkpoint_path", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--views", type=int, default=2)
    p.add_argument("--model", type=str, default="midas_vit")
    p.add_argument("--img_sidelength", type=int, default=256)
    p.add_argument("--no_sample", action="store_true", default=False)
    p.add_argument("--no_latent_concat", action="store_true", default=False)
    p.add_argument("--no_multiview", action="store_true", default=False)
    p.add_argument("--no_high_freq", action="store_true", default=False)
    p.add_argument("--reconstruct", action="store_true", default=False)
    p.add_argument("--synthetic", action="store_true", default=False)
    p.add_argument("--with_encoder", action="store_true", default=False,
                   help="build the DPT encoder and run get_z on the context images even without a checkpoint (random weights)")
    p.add_argument("--n_frames", type=int, default=8, help="frames of the rendered trajectory")
    p.add_argument("--out_dir", type=str, default=None)
    p.add_argument("--cameras", choices=("host", "gpu", "device"), default="host",
                   help="where the camera matrices live and the 4x4 pose algebra runs: host = matrices stay on the CPU, the reference's "
                        "torch.inverse calls there (strict parity, no device sync; default); gpu = the whole input dict on the GPU as in the "
                        "reference's scripts, the engine downloads the matrices for the same host algebra (strict parity, one small sync per "
                        "new pose); device = car_pose_setup on the GPU (no host work per frame, last-ulp differences)")
    p.add_argument("--port", type=int, default=1492)          # the reference rendezvous port (eval_realestate10k.py:97)
    return p


def build_model(opt, device, with_encoder=None):
    """The renderer; with a checkpoint (or --with_encoder) the multi-view DPT encoder is built too, so that ``get_z`` runs on the
    context images.  Loading follows the reference (strict unless --reconstruct, eval_realestate10k.py:110-118) and FAILS on keys that
    do not match instead of silently leaving layers at their initial values."""
    import torch
    from cross_attention_renderer_amd import synthetic
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    torch.manual_seed(0)
    if with_encoder is None:
        with_encoder = bool(opt.checkpoint_path) or getattr(opt, "with_encoder", False)
    model = CrossAttentionRenderer(no_sample=opt.no_sample, no_latent_concat=opt.no_latent_concat,
                                   no_multiview=opt.no_multiview, no_high_freq=opt.no_high_freq, model=opt.model,
                                   n_view=opt.views, with_encoder=with_encoder).eval()
    if opt.checkpoint_path:
        state = torch.load(opt.checkpoint_path, map_location="cpu")["model"]
        res = model.load_state_dict(state, strict=False)
        bad = [k for k in res.missing_keys if not opt.reconstruct] + list(res.unexpected_keys)
        if bad:
            raise SystemExit(f"checkpoint {opt.checkpoint_path} does not match the model: {len(res.missing_keys)} missing, "
                             f"{len(res.unexpected_keys)} unexpected keys, e.g. {bad[:6]}")
    else:
        with torch.no_grad():                            # un-trained weights: fc_1 is zero-initialised otherwise (resnet_block_fc.py:39)
            g = torch.Generator().manual_seed(0)
            for n_, p_ in model.named_parameters():
                if not n_.startswith("encoder."):
                    p_.add_(0.02 * torch.randn(p_.shape, generator=g))
    model.H = model.W = opt.img_sidelength
    model.pose_route = "device" if getattr(opt, "cameras", "host") == "device" else "host"
    return model.to(device)


def spawn(fn, opt):
    """One process per GPU, NCCL(=RCCL) process group over tcp://127.0.0.1:<port> as in the reference scripts."""
    import torch.multiprocessing as mp
    if opt.gpus > 1:
        mp.spawn(fn, nprocs=opt.gpus, args=(opt,), join=True)
    else:
        fn(0, opt)


def init_rank(rank, opt):
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    if opt.gpus > 1:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{opt.port}", world_size=opt.gpus, rank=rank)
    return torch.device("cuda", rank)
