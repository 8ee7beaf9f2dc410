"""Throughput of the constructor variants on the STAGED route of the engine (stage kernels of the C ABI sequenced
by engine.py, every layer on the fp32 matrix pipe): rays/s at the C2 shape (256 x 256, 64 samples) in 8192-ray forward calls, so that
what a variant costs next to the fused default route is a number.  Builder-kept figures (the driver benches the default configuration).
Usage (GPU box): python tools/bench_variants.py [variant ...]     variants: default nview3 nview1 no_sample no_latent_concat no_repeat"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cross_attention_renderer_amd import harness, synthetic as S  # noqa: E402
from cross_attention_renderer_amd.models import CrossAttentionRenderer  # noqa: E402

VARIANTS = {"default": dict(n_view=2), "nview3": dict(n_view=3), "nview1": dict(n_view=1), "no_sample": dict(n_view=2, no_sample=True),
            "no_latent_concat": dict(n_view=2, no_latent_concat=True), "no_repeat": dict(n_view=2, repeat_attention=False)}


def main():
    names = sys.argv[1:] or list(VARIANTS)
    dev = torch.device("cuda:0")
    H, P, R, calls = 256, 64, 8192, 6
    a = torch.randn(8192, 8192, device=dev)                       # a second of matrix work first: the first variant must not be timed on a
    t0 = time.perf_counter()                                      # device that is still leaving its idle state
    while time.perf_counter() - t0 < 1.5:
        (a @ a).sum().item()
    del a
    for name in names:
        kw = VARIANTS[name]
        torch.manual_seed(0)
        m = CrossAttentionRenderer(model="midas_vit", npoints=P, with_encoder=False, **kw).eval()
        S.perturb_parameters(m, seed=0, scale=0.02)
        m.H = m.W = H
        m = m.to(dev)
        V = kw["n_view"]
        inp = S.stereo_scene(H, b=1, seed=5, n_view=V)
        z = [t.to(dev) for t in S.feature_maps(1, V, H, seed=1)]
        uv = inp["query"]["uv"]
        chunks = [harness.to_device({"context": inp["context"], "query": dict(inp["query"], uv=uv[:, :, (16 + 32 * k) * H:(16 + 32 * k) * H + R].contiguous())}, dev)
                  for k in range(calls)]
        with torch.no_grad():
            for _ in range(2):
                for c in chunks:                       # warm-up: plan, lattice, workspace, allocator, every chunk's input dict once
                    m(c, z=z)
            route = "one-call (fused)" if m._engine._pair is not None else "staged"      # only the one-call route builds a lattice
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for c in chunks:
                m(c, z=z)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / calls
        print(f"{name:18s} n_view {V}  route {route:16s} {dt * 1e3:8.2f} ms per 8192-ray call  {R / dt:12.0f} rays/s  "
              f"(peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)
        if name == "default":                           # the same configuration forced onto the staged route: what the fusion buys
            m._engine.fuse_samples = False
            with torch.no_grad():
                for c in chunks[:3]:
                    m(c, z=z)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for c in chunks:
                    m(c, z=z)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / calls
            print(f"{'default, staged':18s} n_view {V}  route {'staged':16s} {dt * 1e3:8.2f} ms per 8192-ray call  {R / dt:12.0f} rays/s", flush=True)
        del m, z, chunks
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
