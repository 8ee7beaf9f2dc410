"""Evaluation entry point (mirrors reference experiment_scripts/eval_realestate10k.py).

    python experiment_scripts/eval_realestate10k.py --experiment_name demo --views 2 --checkpoint_path model.pth \\
        --data_root data_download/realestate/test --pose_root poses/realestate/test.mat [--gpus N]
    python experiment_scripts/eval_realestate10k.py --experiment_name demo --views 2 --synthetic [--gpus N]

Per item (eval_realestate10k.py:131-199): ``z = model.get_z(model_input)`` on the context images, the query view rendered in 9 ray
chunks (18 for three views), PSNR of the rendered against the ground-truth query frame.  Items come from
``dataio.RealEstate10kVis`` (the reference's reader, pinned in tests/test_dataio.py).  LPIPS / SSIM need lpips / skimage, which are
not installed here.  With --gpus N the rays of every item are banded over N processes instead of N replicas evaluating everything.
--synthetic: a seeded pair and feature pyramid (no dataset, no encoder); the target is then the un-chunked render of the same frame,
so the figure checks chunk / shard invariance (inf or > 100 dB), not image quality — and is labelled so."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import common  # noqa: E402

DEFAULT_DATA = ("data_download/realestate/test", "poses/realestate/test.mat")      # eval_realestate10k.py:101-102


def evaluate(rank, opt, default_data=DEFAULT_DATA):
    import random
    import torch
    from cross_attention_renderer_amd import dataio, harness
    dev = common.init_rank(rank, opt)
    H = opt.img_sidelength
    real = not opt.synthetic and (opt.data_root or os.path.isdir(default_data[0]))
    model = common.build_model(opt, dev, with_encoder=True if real else None)
    n_chunks = 9 if opt.views < 3 else 18                    # eval_realestate10k.py:144-149
    psnrs = []
    if real:
        ds = dataio.RealEstate10kVis(opt.data_root or default_data[0], opt.pose_root or default_data[1], num_ctxt_views=opt.views,
                                     num_query_views=1, augment=False)
        random.seed(0)                                       # every rank draws the same query frames
        items = ((ds[i][0], None) for i in range(len(ds)))
    else:
        items = (harness.synthetic_pair(H, opt.views, seed=5 + i) for i in range(opt.batch_size))
    def prepared():
        """Items moved to the device with their pyramids: get_z for the real reader, the seeded stand-in otherwise."""
        for inp, z in items:
            if real:
                inp = {part: {kk: (vv[None] if torch.is_tensor(vv) else vv) for kk, vv in d.items()} for part, d in inp.items()}   # batch of 1
                inp = harness.to_device(inp, dev, opt.cameras)
                with torch.no_grad():
                    z = model.get_z(inp)
            else:
                inp, z = harness.to_device(inp, dev, opt.cameras), [t.to(dev) for t in z]
            yield inp, z
    it = prepared()
    nxt = next(it, None)
    k = -1
    while nxt is not None:
        (inp, z), k = nxt, k + 1
        # the loop knows its next batch (eval_realestate10k.py:131-142): its pyramid is made now and announced, so that its per-pair set-up
        # (car_project_maps) runs on a side stream beside this item's render instead of in front of the next one
        nxt = next(it, None)
        if nxt is not None:
            model.prefetch_pair(nxt[1])
        start = time.time()
        tile = harness.render_frame(model, inp, z, chunk_rays=-(-H * H // n_chunks), rank=rank, world=opt.gpus)
        torch.cuda.synchronize()
        elapsed = time.time() - start
        # the reference's protocol (eval_realestate10k.py:170-181): prediction and target are both composited over 0.5 grey with the
        # rendered valid mask — rays that see neither context view contribute no error — and nothing is clamped
        valid = tile[0, :, 4:5]

        def composite(img):
            return ((img + 1) * 0.5) * valid + 0.5 * (1 - valid)
        if real:
            target, what = inp["query"]["rgb"][0, 0].reshape(-1, 3).to(tile.device), "psnr"
        else:
            target, what = harness.render_frame(model, inp, z, chunk_rays=16384)[0, :, :3], "psnr vs un-chunked render"
        rgb, target = composite(tile[0, :, :3]), composite(target)
        psnrs.append(harness.psnr(rgb, target))
        if rank == 0:
            print(f"item {k}: elapsed {elapsed:.3f} s, {what} {psnrs[-1]:.2f} dB, valid {tile[0, :, 4].mean().item():.3f}")
    if rank == 0 and psnrs:
        print("mean psnr", sum(min(p, 200.0) for p in psnrs) / len(psnrs))


if __name__ == "__main__":
    opt = common.parser(__doc__).parse_args()
    common.spawn(evaluate, opt)
