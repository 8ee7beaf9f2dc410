"""Evaluation entry point (mirrors reference experiment_scripts/eval_realestate10k.py).

    python experiment_scripts/eval_realestate10k.py --experiment_name demo --views 2 --synthetic [--gpus N]

Renders the query view of each item in chunks (eval_realestate10k.py:142-176) and reports PSNR against a target.  In
--synthetic mode the "ground truth" is the render of the same frame by this implementation with chunking disabled, so
the number checks chunk / shard invariance (it must be inf or > 100 dB), not image quality."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402


def evaluate(rank, opt):
    import torch
    from cross_attention_renderer_amd import harness
    dev = common.init_rank(rank, opt)
    model = common.build_model(opt, dev)
    H = opt.img_sidelength
    if opt.data_root and not opt.synthetic:
        raise SystemExit("dataset readers are not built yet (SURVEY.md §8f row 3); use --synthetic")
    psnrs = []
    for item in range(opt.batch_size):
        inp, z = harness.synthetic_pair(H, opt.views, seed=5 + item)
        inp, z = harness.to_device(inp, dev), [t.to(dev) for t in z]
        start = time.time()
        # the reference splits a frame into 9 chunks (18 for 3 views), eval_realestate10k.py:144-149
        n_chunks = 9 if opt.views < 3 else 18
        tile = harness.render_frame(model, inp, z, chunk_rays=-(-H * H // n_chunks), rank=rank, world=opt.gpus)
        torch.cuda.synchronize()
        elapsed = time.time() - start
        ref = harness.render_frame(model, inp, z, chunk_rays=16384)
        rgb, target = (tile[0, :, :3] + 1) / 2, (ref[0, :, :3] + 1) / 2
        psnrs.append(harness.psnr(rgb, target))
        if rank == 0:
            print(f"item {item}: elapsed {elapsed:.3f} s, psnr vs unchunked render {psnrs[-1]:.1f} dB, "
                  f"valid {tile[0, :, 4].mean().item():.3f}")
    if rank == 0:
        print("mean psnr", sum(min(p, 200.0) for p in psnrs) / len(psnrs))


if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(evaluate, opt)
