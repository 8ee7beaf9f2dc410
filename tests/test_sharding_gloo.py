"""world_size-2 ``gloo`` tests (CPU) of the multi-GPU ray-sharding path: band partitioning, the padded all-gather that
reassembles a frame, and the overlapped tile gather used by bench.py.  The render itself is replaced by a deterministic
per-ray function so the test needs no GPU; rays are independent, so this is exactly what sharding relies on."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cross_attention_renderer_amd import sharding as Sh


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_render(uv):
    """(b,1,R,2) -> forward()-shaped dict whose values depend only on the ray itself."""
    x, y = uv[:, 0, :, 0], uv[:, 0, :, 1]
    rgb = torch.stack([x * 0.01, y * 0.02, x + y], dim=-1)[:, None]
    return {"rgb": rgb, "depth_ray": (x - y)[..., None], "valid_mask": ((x + y) % 2)[..., None]}


def _worker(rank, world, port, n_rays, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        uv = torch.rand(2, 1, n_rays, 2, generator=g) * 100
        inp = {"context": {"rgb": torch.zeros(2, 2, 4, 4, 3)}, "query": {"uv": uv, "cam2world": torch.eye(4)}}
        shard, (s, e) = Sh.shard_query(inp, rank, world)
        assert inp["query"]["uv"].shape[2] == n_rays, "caller's dict was mutated"
        tile = Sh.pack_tile(_fake_render(shard["query"]["uv"]))
        full = Sh.gather_rays(tile, n_rays)
        want = Sh.pack_tile(_fake_render(uv))
        assert torch.equal(full, want), "sharded render differs from the unsharded one"
        # overlapped equal-size gather (bench path), twice to exercise buffer reuse
        tg = Sh.TileGather(world, 16, 5, "cpu")
        for it in range(2):
            t = torch.full((16, 5), float(rank * 10 + it))
            tg(t)
            t.zero_()                                   # the caller may overwrite its tile right away
            out = tg.wait()
            for r in range(world):
                assert (out[r] == r * 10 + it).all()
        # config 3's exchange (bench.py --gpus N --config c3): a batch of b = 2 scenes, the rays of BOTH banded over the ranks, every
        # rank's (b, R / N, 5) tile flattened to rows through ONE overlapped gather, reassembled into whole frames
        if n_rays % world == 0:
            tg2 = Sh.TileGather(world, 2 * (n_rays // world), 5, "cpu")
            tg2(tile.reshape(-1, 5))
            assert torch.equal(Sh.assemble_tiles(tg2.wait(), 2, n_rays), want), "the batched tile gather does not reassemble the frames"
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rays", [64, 65, 7])
def test_ray_sharding_world2(n_rays):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n_rays, ret), nprocs=world, join=True)
        assert dict(ret) == {0: 1, 1: 1}


def test_ray_bands_partition_exactly():
    for n in (1, 7, 64, 65536, 147456):
        for w in (1, 2, 3, 4, 8):
            bands = [Sh.ray_band(n, r, w) for r in range(w)]
            assert bands[0][0] == 0 and bands[-1][1] == n
            assert all(bands[i][1] == bands[i + 1][0] for i in range(w - 1))
            sizes = [e - s for s, e in bands]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_tile_gather_over_rccl_single_rank():
    """The RCCL branch of TileGather (all_gather_into_tensor on a side stream, the path bench.py --gpus N takes) with a one-rank
    "nccl" group on the one GPU of the test box: API shapes, stream hand-over and buffer reuse; plus shard_query / gather_rays."""
    if dist.is_initialized():
        pytest.skip("a process group is already up in this process")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
    except Exception as e:                                  # no RCCL in this environment: nothing of ours to test
        pytest.skip(f"RCCL process group cannot be created here: {e}")
    try:
        dev = torch.device("cuda:0")
        tg = Sh.TileGather(1, 4096, 5, dev)
        for it in range(3):
            t = torch.full((4096, 5), float(it + 1), device=dev)
            tg(t)
            t.zero_()                                   # the caller may overwrite its tile right away
            out = tg.wait()
            torch.cuda.synchronize()
            assert out.shape == (1, 4096, 5) and (out == it + 1).all()
        uv = torch.rand(2, 1, 65, 2, device=dev) * 100
        inp = {"context": {"rgb": torch.zeros(2, 2, 4, 4, 3, device=dev)}, "query": {"uv": uv}}
        shard, (s, e) = Sh.shard_query(inp, 0, 1)
        assert (s, e) == (0, 65)
        tile = Sh.pack_tile(_fake_render(shard["query"]["uv"]))
        assert torch.equal(Sh.gather_rays(tile, 65), tile)
    finally:
        dist.destroy_process_group()


# ---- the training loop's gradient all-reduce (reference training.py:21-28) over a world-2 gloo group ----------------------------
def _grad_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cross_attention_renderer_amd.training import average_gradients
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
        frozen = torch.nn.Linear(2, 2)                       # a module whose parameters never get a gradient (latent_avg_* in the renderer)
        model = torch.nn.ModuleList([net, frozen])
        x = torch.full((4, 5), float(rank + 1))
        net(x).sum().backward()
        mine = [p.grad.clone() for p in net.parameters()]
        average_gradients(model)
        # the reference's arithmetic: sum over ranks / world size, per parameter
        for p, g0 in zip(net.parameters(), mine):
            parts = [torch.zeros_like(g0) for _ in range(world)]
            dist.all_gather(parts, g0)
            assert torch.allclose(p.grad, sum(parts) / world, rtol=0, atol=1e-6)
        assert all(p.grad is None for p in frozen.parameters())
        # a branch only ONE rank back-propagated through (an unused layer, an encoder-less rank): the bucket is laid out over every
        # parameter that requires a gradient, so the offsets agree; the other rank receives the average too and the replicas stay in step
        net.zero_grad(set_to_none=True)
        side = torch.nn.Linear(3, 2)
        model2 = torch.nn.ModuleList([net, side, frozen])
        y = net(x).sum()
        if rank == 0:
            y = y + side(torch.ones(1, 3)).sum()
        y.backward()
        mine = [None if p.grad is None else p.grad.clone() for p in model2.parameters()]
        average_gradients(model2)
        for p, g0 in zip(model2.parameters(), mine):
            parts = [torch.zeros(p.shape) for _ in range(world)]
            dist.all_gather(parts, torch.zeros(p.shape) if g0 is None else g0)
            if all(float(t.abs().sum()) == 0 for t in parts) and g0 is None:
                assert p.grad is None                          # nobody had one: stays None, the optimizer skips it (frozen)
            else:
                assert p.grad is not None and torch.allclose(p.grad, sum(parts) / world, rtol=0, atol=1e-6)
        assert all(p.grad is not None for p in side.parameters())
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


def test_average_gradients_world2():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_grad_worker, args=(world, port, ret), nprocs=world, join=True)
        assert sorted(ret.keys()) == [0, 1]


# ---- two ranks on two GPUs over RCCL (self-skips on a one-GPU box; the driver's 8-GPU node runs it) -----------------------------
def _rccl_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from cross_attention_renderer_amd import synthetic as S
        from cross_attention_renderer_amd.models import CrossAttentionRenderer
        from cross_attention_renderer_amd.training import average_gradients
        dev = torch.device("cuda", rank)
        H, P = 64, 16
        torch.manual_seed(0)
        m = CrossAttentionRenderer(model="midas_vit", n_view=2, npoints=P, with_encoder=False).eval()
        S.perturb_parameters(m, seed=0)
        m.H = m.W = H
        m = m.to(dev)
        inp = S.stereo_scene(H, b=1, seed=5)
        z = [t.to(dev) for t in S.feature_maps(1, 2, H, seed=1)]
        n_rays = inp["query"]["uv"].shape[2]
        move = lambda d: {k: {kk: (vv if kk in ("cam2world", "intrinsics") else vv.to(dev)) for kk, vv in v.items()} for k, v in d.items()}
        shard, (s, e) = Sh.shard_query(inp, rank, world)
        tg = Sh.TileGather(world, e - s, Sh.TILE_CHANNELS, dev)
        with torch.no_grad():
            tile = Sh.pack_tile(m(move(shard), z=z))[0]
            tg(tile)                                             # RCCL all_gather_into_tensor on the side stream ...
            again = Sh.pack_tile(m(move(shard), z=z))[0]          # ... while this rank renders its band once more
            banded = tg.wait().reshape(n_rays, Sh.TILE_CHANNELS)
            whole = Sh.pack_tile(m(move(inp), z=z))[0]
        torch.cuda.synchronize()
        assert torch.equal(again, tile)
        assert torch.equal(banded, whole), "two ray bands over RCCL differ from the one-GPU frame"
        # the training loop's gradient all-reduce over RCCL (reference training.py:21-28)
        lin = torch.nn.Linear(4, 3).to(dev)
        lin(torch.full((2, 4), float(rank + 1), device=dev)).sum().backward()
        mine = lin.weight.grad.clone()
        average_gradients(lin)
        other = torch.full((2, 4), float(2 - rank), device=dev).sum(0).expand(3, 4)
        assert torch.allclose(lin.weight.grad, (mine + other) / 2)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ray_bands_and_gradient_all_reduce_over_rccl_two_gpus():
    """Two ranks, two devices, backend "nccl" (= RCCL over xGMI): the banded render of a frame, its tiles exchanged by
    all_gather_into_tensor on the side stream under a second render, equals the one-GPU frame bit for bit; average_gradients over RCCL."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (the build box has one; the driver's multi-GPU node runs this)")
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rccl_worker, args=(world, port, ret), nprocs=world, join=True)
        assert sorted(ret.keys()) == [0, 1]
