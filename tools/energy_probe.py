"""Round-5 energy table: joules per sample of the fused per-sample kernel and of its timing-only variants, from the
device's energy accumulator / socket-power samples over a ~2 s loop of back-to-back launches of one 8192-ray chunk (1 048 576 samples),
with the delivered shader clock beside every row.  Differences between rows price the components: tap loads (variant 1), A-operand LDS
reads (13), the e-path matrix work of the source passes (5), toggling multiplier inputs (zeroed weights), the first round's partial
sums (product with / without `part`).  The HBM-bound kernels of the tail (attention over rows, over partial sums) are priced the same way.
usage (GPU box): python tools/energy_probe.py [seconds per row]"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
from build_dev import build_dev  # noqa: E402
from cross_attention_renderer_amd import _lib  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine  # noqa: E402
from power_sampler import PowerSampler  # noqa: E402

P_ = ctypes.c_void_p


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    dev_lib = ctypes.CDLL(build_dev())
    fn = dev_lib.car_fused_samples_ablate
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + _lib.SIGNATURES["car_fused_samples"][1]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    eng = model._engine = RenderEngine(model)
    inp, z = bench.make_frame(0.5, dev)
    R = 8192
    uv = inp["query"]["uv"][:, :, 96 * 256: 96 * 256 + R].contiguous()
    chunk = {"context": inp["context"], "query": dict(inp["query"], uv=uv)}
    with torch.no_grad():
        model(chunk, z=z)
    torch.cuda.synchronize()
    d = eng._dims(1, R, z)
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()

    def ws(name):
        _lib.check(lib.car_workspace_find(ctypes.byref(d), name.encode(), ctypes.byref(off), ctypes.byref(cnt)), name)
        return eng._work.data_ptr() + 4 * off.value
    keep = []

    def dptr(t):
        t = t.detach().float().reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.detach().float().contiguous()
        keep.append(t)
        return t.data_ptr()
    w = _lib.CarWeights()
    sd = dict(model.named_parameters())
    for n in _lib.WEIGHT_FIELDS[0]:
        setattr(w, f"{n.replace('.', '_')}_w", dptr(sd[n + ".weight"]))
        setattr(w, f"{n.replace('.', '_')}_b", dptr(sd[n + ".bias"]))
    blob = torch.empty(lib.car_fused_blob_floats(), device=dev)
    bias = torch.empty(lib.car_fused_bias_floats(), device=dev)
    wpt = torch.empty(576 * 4, device=dev)
    st = P_(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.car_fused_pack(ctypes.byref(w), blob.data_ptr(), bias.data_ptr(), wpt.data_ptr(), st), "car_fused_pack")
    blob0 = torch.zeros_like(blob)
    lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.car_lattice_shape(ctypes.byref(d), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)), "car_lattice_shape")
    gmeta = eng._pair.data_ptr() + 4 * lib.car_gmeta_offset(ctypes.byref(d))
    steps = eng._linspace(0.0, 1.0, bench.P, dev)
    pixel_val = torch.empty(2 * R * bench.P * 2, device=dev)
    S = 2 * R * bench.P
    poses = eng._pose_dev if eng._pose_dev is not None else None
    if poses is None:
        from cross_attention_renderer_amd.poses import pack_poses
        poses = pack_poses({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in chunk.items()}, bench.H).to(dev)
    ts = lib.car_fused_tile_steps()

    def fused_args(blob_t):
        return (poses.data_ptr(), ws("rays"), steps.data_ptr(), eng._pair.data_ptr(), lh.value, lw.value, lpad.value, gmeta, wpt.data_ptr(),
                blob_t.data_ptr(), bias.data_ptr(), 1, 2, R, bench.P, bench.H, bench.H, 0, ws("e"), ws("g"), ws("logit"), ws("pt"),
                pixel_val.data_ptr())
    w_out = torch.empty(S, device=dev)
    zb = torch.empty(R, 576, device=dev)
    big = torch.empty(256 * 1024 * 1024, device=dev)              # 1 GB for the plain streaming rows

    rows = [
        ("idle (no launches)", None, 0),
        ("product kernel + first-round partial sums (car_fused_samples_parts)", lambda: lib.car_fused_samples_parts(*fused_args(blob), ws("part"), st), S),
        ("product kernel alone (car_fused_samples)", lambda: lib.car_fused_samples(*fused_args(blob), st), S),
        ("  same, weights zeroed (same instructions, idle multipliers)", lambda: lib.car_fused_samples(*fused_args(blob0), st), S),
        ("  variant 1: no tap loads", lambda: fn(1, *fused_args(blob), st), S),
        ("  variant 13: no A-operand reads from LDS", lambda: fn(13, *fused_args(blob), st), S),
        ("  variant 5: source passes without their matrix work", lambda: fn(5, *fused_args(blob), st), S),
        ("  variant 2: no gather work at all", lambda: fn(2, *fused_args(blob), st), S),
        ("attention over the rows of e (car_attend, 2.4 GB)", lambda: lib.car_attend(ws("logit"), None, 128, ws("e"), 576, 1, 2, R, bench.P, None, 0.0,
                                                                                      w_out.data_ptr(), zb.data_ptr(), 576, 1, None, None, None, None, st), S),
        ("attention over the partial sums (car_attend_parts, 0.3 GB)", lambda: lib.car_attend_parts(ws("logit"), ws("part"), ts, 576, 1, 2, R, bench.P,
                                                                                                   w_out.data_ptr(), zb.data_ptr(), 576, 1, None, None, None, None, st), S),
        ("torch fill of 1 GB (HBM writes)", lambda: (big.fill_(1.0), 0)[1], 0),
        ("torch sum over 1 GB (HBM reads)", lambda: (big.sum(), 0)[1], 0),
    ]
    if os.environ.get("CAR_ENERGY_ROWS") == "closing":
        # round 6's closing table (profiles/round6_fused_closing.md): the product kernel, the development build's kernel (CAR_DEV_FLAGS, e.g.
        # -DCAR_WG_ORDER=0: the ray-major workgroup order of rounds 2-5) and, if built in (CAR_DEV_UNIT=car_fused_w32.hip), the 32x32x16 candidate
        rows = [rows[0], rows[2], ("development build's kernel (CAR_DEV_FLAGS=%s)" % os.environ.get("CAR_DEV_FLAGS", ""), lambda: fn(0, *fused_args(blob), st), S)]
        fn_w32 = getattr(dev_lib, "car_fused_samples_w32", None)
        if fn_w32 is not None:
            fn_w32.restype = ctypes.c_int
            fn_w32.argtypes = _lib.SIGNATURES["car_fused_samples"][1]
            torch.cuda.synchronize()
            blob32 = blob.clone()
            old = blob[:18 * 18 * 512].view(torch.int16).view(18, 18, 2, 64, 8)
            T_, kh_, lane_ = torch.meshgrid(torch.arange(9, device=dev), torch.arange(2, device=dev), torch.arange(64, device=dev), indexing="ij")
            new = old[:, 2 * T_ + ((lane_ % 32) >> 4), :, 16 * (2 * kh_ + lane_ // 32) + lane_ % 16, :]        # (T, kh, lane, ks, hl, e)
            blob32[:18 * 18 * 512].view(torch.int16).view(18, 9, 2, 2, 64, 8).copy_(new.permute(3, 0, 1, 4, 2, 5))
            rows.append(("32x32x16 candidate, two waves per SIMD (tools/probes/car_fused_w32.hip)", lambda: fn_w32(*fused_args(blob32), st), S))
            rows.append(("product kernel alone, again", rows[1][1], S))
    out = []
    for name, call, samples in rows:
        torch.cuda.synchronize()
        n = 0
        with PowerSampler(interval=0.01, skip=0.3) as ps:
            t0 = time.perf_counter()
            if call is None:
                time.sleep(secs)
            else:
                while time.perf_counter() - t0 < secs:
                    for _ in range(20):
                        rc = call()
                        assert rc == 0, (name, lib.car_last_error())
                    n += 20
                    if n % 200 == 0:
                        torch.cuda.synchronize()                 # bound the queue depth
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        s = ps.summary()
        row = {"row": name, "launches": n, "seconds": dt, "ms_per_launch": (dt / n * 1e3) if n else None, **s}
        if n and s.get("available"):
            pw = s.get("energy_mean_w", s["mean_w"])
            row["j_per_launch"] = pw * dt / n
            if samples:
                row["nj_per_sample"] = pw * dt / n / samples * 1e9
        out.append(row)
        print(json.dumps(row), flush=True)
    idle = out[0].get("mean_w")
    print("\n| row | ms / launch | socket W (mean) | W above idle | sclk MHz | J / launch | nJ / sample |")
    print("|---|---|---|---|---|---|---|")
    for r in out:
        f = lambda v, fmt: (fmt % v) if isinstance(v, (int, float)) else "-"
        above = (r.get("energy_mean_w", r.get("mean_w")) - idle) if (idle and r.get("mean_w")) else None
        print(f"| {r['row']} | {f(r.get('ms_per_launch'), '%.3f')} | {f(r.get('energy_mean_w', r.get('mean_w')), '%.0f')} | {f(above, '%.0f')} | "
              f"{f(r.get('mean_sclk_mhz'), '%.0f')} | {f(r.get('j_per_launch'), '%.3f')} | {f(r.get('nj_per_sample'), '%.1f')} |")


if __name__ == "__main__":
    main()
