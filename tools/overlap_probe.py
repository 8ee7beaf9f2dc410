"""Does the HBM-bound half of a forward call (attention rounds + per-ray chains) hide behind the power-bound half (the fused per-sample
kernel) of the NEXT batch of rays?  Renders the bench frame in n ray batches, each with its own workspace: phase 1 of every batch on one
stream, phase 2 on a second one behind an event (car_render_forward_phase), and times frames against the single-stream one-call frame.
Usage (GPU box): python tools/overlap_probe.py [n_batches ...]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from cross_attention_renderer_amd import _lib  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine, _ptr  # noqa: E402


def main():
    import __graft_entry__ as ge
    ge.build()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    eng = model._engine = RenderEngine(model)
    inp, z = bench.make_frame(0.5, dev)
    with torch.no_grad():
        ref = model(inp, z=z)
        for _ in range(3):
            model(inp, z=z)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            model(inp, z=z)
        torch.cuda.synchronize()
    base = (time.time() - t0) / 10 * 1e3
    print(f"one call per frame, one stream: {base:.2f} ms per frame")
    b, V, R, P, H = 1, 2, bench.H * bench.H, bench.P, bench.H
    f32 = dict(device=dev, dtype=torch.float32)
    poses = eng._poses(inp, H, b * V, dev)
    uv = inp["query"]["uv"].reshape(b, R, 2).float().contiguous()
    steps = eng._linspace(0.0, 1.0, P, dev)
    d_all = eng._dims(b, R, z)
    plan = eng._plan_for(d_all, dev)
    pair, d_pair = eng._pair_for(plan, z, dev, 0, b, R)
    gmeta_ptr = pair.data_ptr() + 4 * lib.car_gmeta_offset(ctypes.byref(d_pair))
    order = ("rgb", "valid_mask", "depth_ray", "at_wt", "at_wt_max", "coords", "pixel_val")
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ncu = int(os.environ.get("CAR_FUSED_CUS", "0"))            # phase 1 on a stream restricted to the first n compute units (spread over the XCCs)
    if ncu:
        hip = ctypes.CDLL("libamdhip64.so")

        def masked(lo, hi):
            words = (ctypes.c_uint32 * 8)()
            for i in range(lo, hi):
                words[i // 32] |= 1 << (i % 32)
            hs = ctypes.c_void_p()
            assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), 8, words) == 0
            return torch.cuda.ExternalStream(hs.value)
        sa = masked(0, ncu)
        if os.environ.get("CAR_RAYS_MASK"):
            sb = masked(ncu, 256)
        print(f"phase 1 on the first {ncu} compute units" + (", phase 2 on the others" if os.environ.get("CAR_RAYS_MASK") else ", phase 2 unrestricted"))
    for nb in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        rc = R // nb
        d = eng._dims(b, rc, z)
        need = lib.car_workspace_bytes(ctypes.byref(d))
        works = [torch.empty(need // 4, **f32) for _ in range(nb)]
        outs, cis, cos, uvs = [], [], [], []
        for c in range(nb):
            o = {"rgb": torch.empty(b, 1, rc, 3, **f32), "valid_mask": torch.empty(b, rc, 1, **f32), "depth_ray": torch.empty(b, rc, 1, **f32),
                 "at_wt": torch.empty(b * V, rc, P, **f32), "at_wt_max": torch.empty(b * V, rc, 1, device=dev, dtype=torch.int32),
                 "coords": torch.empty(b * V, rc, 9, **f32), "pixel_val": torch.empty(b * V, rc, P, 2, **f32)}
            u = uv[:, c * rc:(c + 1) * rc].contiguous()
            ci = _lib.CarInputs()
            ci.poses, ci.uv, ci.lattice, ci.gmeta, ci.steps = poses.data_ptr(), u.data_ptr(), pair.data_ptr(), gmeta_ptr, steps.data_ptr()
            outs.append(o); uvs.append(u); cis.append(ci); cos.append(_lib.CarOutputs(*[o[k].data_ptr() for k in order]))
        evs = [torch.cuda.Event() for _ in range(nb)]

        def frame():
            for c in range(nb):
                _lib.check(lib.car_render_forward_phase(ctypes.byref(d), _ptr(plan), ctypes.byref(cis[c]), ctypes.byref(cos[c]), _ptr(works[c]),
                                                        works[c].numel() * 4, 1, ctypes.c_void_p(sa.cuda_stream)), "phase 1")
                evs[c].record(sa)
                sb.wait_event(evs[c])
                _lib.check(lib.car_render_forward_phase(ctypes.byref(d), _ptr(plan), ctypes.byref(cis[c]), ctypes.byref(cos[c]), _ptr(works[c]),
                                                        works[c].numel() * 4, 2, ctypes.c_void_p(sb.cuda_stream)), "phase 2")
        torch.cuda.synchronize()
        for _ in range(3):
            frame()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(10):
            frame()
        torch.cuda.synchronize()
        ms = (time.time() - t0) / 10 * 1e3
        rgb = torch.cat([o["rgb"] for o in outs], dim=2)
        err = (rgb - ref["rgb"]).abs().max().item()
        print(f"{nb} batches of {rc} rays, phases on two streams: {ms:.2f} ms per frame ({100 * (ms / base - 1):+.1f} %), rgb max |diff| vs one call {err:.1e}, "
              f"workspaces {nb * need / 2**30:.1f} GiB")
        del works, outs


if __name__ == "__main__":
    main()
