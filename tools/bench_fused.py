"""Times the fused per-sample kernel (csrc/car_fused.hip) alone on one 8192-ray chunk of the bench frame (256x256x64), for the
product kernel (variant 0) and the timing-only ablation variants of the development build (tools/build_dev.py; results of
variants > 0 are wrong by construction):
  1 no tap loads | 2 no gather work | 3 = 2 + no weight DMA / barriers | 5 the source passes without their matrix work | 4 phase stamps | 20 where a wave waits inside a chunk
  11 no chunk barrier (racy) | 12 = 3 + no A-operand reads | 13 no A-operand reads | 100 the PRODUCT library's kernel, timed the same way | 101 the product launch with the first round's partial sums (what the forward issues)
  400 round 6's candidate with the source passes on 32x32x16 tiles, two waves per SIMD (tools/probes/car_fused_w32.hip; build with CAR_DEV_UNIT=car_fused_w32.hip), compared with 100
(earlier rounds' probes — masked lanes, tap orders, deep tap rings, the full-lattice timing probe — are recorded in profiles/)
Usage (GPU box): python tools/bench_fused.py [variants...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import bench  # noqa: E402
from build_dev import build_dev  # noqa: E402
from cross_attention_renderer_amd import _lib  # noqa: E402
from cross_attention_renderer_amd.engine import RenderEngine  # noqa: E402

P_ = ctypes.c_void_p


def main():
    dev_lib = ctypes.CDLL(build_dev())
    fn = dev_lib.car_fused_samples_ablate
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + _lib.SIGNATURES["car_fused_samples"][1]
    fn_sliced = dev_lib.car_fused_samples_sliced         # variant 1000 + n: the launch cut into kernel launches of n sample groups each
    fn_sliced.restype = ctypes.c_int
    fn_sliced.argtypes = [ctypes.c_int] + _lib.SIGNATURES["car_fused_samples"][1]
    base_path = os.path.join(ROOT, "tools", "_dev", "libcar_base.so")          # variant 300: a saved earlier build of the product library
    fn_base = None
    if os.path.exists(base_path):
        fn_base = ctypes.CDLL(base_path).car_fused_samples
        fn_base.restype = ctypes.c_int
        fn_base.argtypes = _lib.SIGNATURES["car_fused_samples"][1]
    fn_w32 = getattr(dev_lib, "car_fused_samples_w32", None)
    if fn_w32 is not None:
        fn_w32.restype = ctypes.c_int
        fn_w32.argtypes = _lib.SIGNATURES["car_fused_samples"][1]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    model = bench.build_model(dev)
    eng = model._engine = RenderEngine(model)
    inp, z = bench.make_frame(0.5, dev)
    R = 8192
    uv = inp["query"]["uv"][:, :, 96 * 256: 96 * 256 + R].contiguous()
    if "CAR_BENCH_TILE" in os.environ:                 # experiment: rays in 2-D tiles of (rows x cols) pixels instead of row strips
        th, tw = [int(x) for x in os.environ["CAR_BENCH_TILE"].split("x")]
        g = uv.view(1, 1, 32, 256, 2)                  # 32 image rows x 256 columns
        uv = g.view(1, 1, 32 // th, th, 256 // tw, tw, 2).permute(0, 1, 2, 4, 3, 5, 6).reshape(1, 1, R, 2).contiguous()
    chunk = {"context": inp["context"], "query": dict(inp["query"], uv=uv)}
    with torch.no_grad():
        model(chunk, z=z)                              # plan, projected maps, workspace (and the rays of this chunk inside it)
    if eng._pose_dev is None:                          # cameras on the GPU: the engine made the records with car_pose_setup
        from cross_attention_renderer_amd.poses import pack_poses
        eng._pose_dev = pack_poses({k: {kk: vv.cpu() for kk, vv in v.items()} for k, v in chunk.items()}, bench.H).to(dev)
    torch.cuda.synchronize()
    d = eng._dims(1, R, z)
    off, cnt = ctypes.c_size_t(), ctypes.c_size_t()

    def ws(name):
        _lib.check(lib.car_workspace_find(ctypes.byref(d), name.encode(), ctypes.byref(off), ctypes.byref(cnt)), name)
        return eng._work.data_ptr() + 4 * off.value
    # the fused layers, packed once more into buffers of our own (the plan's offsets are private)
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
    # variant 400's blob: the W2 region (18 K steps x 18 tiles of [hi | lo][lane][8 halves]) re-laid for the 32 x 32 x 16 A operand:
    # [K step][32-channel tile T][K half][hi | lo][lane][8 halves], lane l = row l % 32, k = 16 half + 8 (l / 32) + e — a permutation of the bytes
    torch.cuda.synchronize()
    blob32 = blob.clone()
    old = blob[:18 * 18 * 512].view(torch.int16).view(18, 18, 2, 64, 8)
    T_, kh_, lane_ = torch.meshgrid(torch.arange(9, device=dev), torch.arange(2, device=dev), torch.arange(64, device=dev), indexing="ij")
    t_src = 2 * T_ + ((lane_ % 32) >> 4)
    lane_src = 16 * (2 * kh_ + lane_ // 32) + lane_ % 16
    new = old[:, t_src, :, lane_src, :]                                 # index tensors split by a slice: (T, kh, lane, ks, hl, e)
    blob32[:18 * 18 * 512].view(torch.int16).view(18, 9, 2, 2, 64, 8).copy_(new.permute(3, 0, 1, 4, 2, 5))
    lh, lw, lpad = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    _lib.check(lib.car_lattice_shape(ctypes.byref(d), ctypes.byref(lh), ctypes.byref(lw), ctypes.byref(lpad)), "car_lattice_shape")
    gmeta = eng._pair.data_ptr() + 4 * lib.car_gmeta_offset(ctypes.byref(d))
    steps = eng._linspace(0.0, 1.0, bench.P, dev)
    pixel_val = torch.empty(2 * R * bench.P * 2, device=dev)
    S = 2 * R * bench.P
    flop = 2.0 * S * bench.FUSED_MACS
    variants = [int(v) for v in sys.argv[1:]] or [0, 1, 2, 3, 5, 6, 7, 8]
    if os.environ.get("CAR_ZERO"):                       # power probe: the same instruction stream over zeros (weights and / or lattice)
        if "w" in os.environ["CAR_ZERO"]:
            blob.zero_()
        if "l" in os.environ["CAR_ZERO"]:
            eng._pair.zero_()
        torch.cuda.synchronize()
    prod = lib.car_fused_samples                        # variant 100: the product library's kernel, timed the same way
    prod_parts = lib.car_fused_samples_parts            # variant 101: the same launch as the forward issues it (with the first round's partial sums)
    outs = {}
    # power / occupancy probe: CAR_CU_MASK=n runs the launches on a stream restricted to n compute units (every (256 / n)-th one),
    # CAR_R=rays shrinks the launch with it
    Rk = int(os.environ.get("CAR_R", R))
    ext = None
    if os.environ.get("CAR_CU_MASK"):
        ncu = int(os.environ["CAR_CU_MASK"])
        hip = ctypes.CDLL("libamdhip64.so")
        words = (ctypes.c_uint32 * 8)()
        mode = os.environ.get("CAR_CU_MODE", "stride")
        for i in range(ncu):
            c = i * (256 // ncu) if mode == "stride" else i
            words[c // 32] |= 1 << (c % 32)
        hs = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(hs), 8, words)
        assert rc == 0, rc
        ext = torch.cuda.ExternalStream(hs.value)
        st = P_(hs.value)
        flop = flop * Rk / R
        print(f"stream restricted to {ncu} CUs ({mode}), {Rk} rays per launch")
    for v in variants:
        lat = []
        lat_ptr, lat_h, lat_w, lat_pad = eng._pair.data_ptr(), lh.value, lw.value, lpad.value
        for it in range(int(os.environ.get("CAR_LOOP", 7))):
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(ext) if ext is not None else a.record()
            args = (eng._pose_dev.data_ptr(), ws("rays"), steps.data_ptr(), lat_ptr, lat_h, lat_w, lat_pad,
                    gmeta, wpt.data_ptr(), blob.data_ptr(),
                    bias.data_ptr(), 1, 2, Rk, bench.P, bench.H, bench.H, 0, ws("e"), ws("g"), ws("logit"), ws("pt"),
                    pixel_val.data_ptr(), st)
            if v == 400:
                args = args[:9] + (blob32.data_ptr(),) + args[10:]
            rc = fn_w32(*args) if v == 400 else fn_sliced(v - 1000, *args) if v >= 1000 else prod(*args) if v == 100 else prod_parts(*args[:-1], ws("part"), args[-1]) if v == 101 else fn_base(*args) if v == 300 else fn(v, *args)
            b_.record(ext) if ext is not None else b_.record()
            assert rc == 0, dev_lib.car_last_error()
            lat.append((a, b_))
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b_) for a, b_ in lat[2:])
        if v == 4 or "CAR_STAMP_ALL" in os.environ.get("CAR_DEV_FLAGS", ""):
            st8 = pixel_val.view(torch.int64)[: (2 * R * bench.P // 192) * 16].view(-1, 16).cpu().double()
            seq = [0, 1, 2, 3, 4, 7, 8, 5, 6]                     # stamp indices in program order
            names = ["tables+geometry", "first gather + weights 0", "source pass 0", "source pass 1", "K1 over e_1 (+ its stores)", "K1 over e_0 (LDS-DMA rows)",
                     "(nothing: K2 folded away)", "query layer + folded 128 x 128 layer + logit"]
            tot = (st8[:, 6] - st8[:, 0]).mean().item()
            print("phase clock ticks per workgroup (mean over %d groups; s_memtime):" % st8.shape[0])
            for j, n_ in enumerate(names):
                x = (st8[:, seq[j + 1]] - st8[:, seq[j]]).mean().item()
                print(f"   {n_:38s} {x:10.0f}  {100 * x / tot:5.1f} %")
            print(f"   total {tot:.0f} ticks")
        if v == 21:                                         # where the taps go: unique lattice nodes per instruction / workgroup / launch
            nd = pixel_val.view(torch.int32)[: 2 * R * bench.P * 2].view(2, R, bench.P, 2).cpu().long()      # [context view n][ray][step][source]
            lw_ = lw.value
            for sv in (0, 1):
                for n in (0, 1):
                    x = nd[n, :, :, sv]                                    # [ray][step]
                    livef = (x >= 0).float().mean().item()
                    # one tap instruction = 8 consecutive rays at one step (rows r0 .. r0+7 of a wave): unique 128-byte lines among its 8 rows
                    g8 = x.view(R // 8, 8, bench.P).permute(0, 2, 1).reshape(-1, 8)
                    uniq8 = torch.tensor([len(set(r.tolist()) - {-1}) for r in g8[:4096]]).float().mean().item()
                    # the round-3 start tile: 48 consecutive rays x 4 consecutive steps (now 24 x 8, see `shapes` below); nodes touched by its 4 taps (nw, +1, +row, +row+1)
                    wg = x[: (R // 48) * 48].view(R // 48, 48, bench.P // 4, 4).permute(0, 2, 1, 3).reshape(-1, 192)
                    def touched(t):
                        t = t[t >= 0]
                        return torch.unique(torch.cat([t, t + 1, t + lw_, t + lw_ + 1]))
                    uw = sum(len(touched(r)) for r in wg[:512]) / 512
                    # the 32 workgroups an XCD runs at a time: 96 consecutive rays x all steps (two bundles x 16 step groups)
                    rnd = x[:96].view(2, 48, bench.P // 4, 4).permute(0, 2, 1, 3).reshape(32, 192)
                    per_wg = sum(len(touched(r)) for r in rnd)
                    union = len(touched(rnd.reshape(-1)))
                    same_pg = x[:96, :4].reshape(-1)                       # the two bundles at one step group
                    print(f"      one XCD round (96 rays x 64 steps): sum of the 32 workgroups' nodes {per_wg}, union {union} ({union * 2304 / 1e6:.1f} MB, {union * 128 / 1e3:.0f} KB per chunk phase); "
                          f"two neighbouring bundles at one step group: {len(touched(x[:48, :4].reshape(-1)))} + {len(touched(x[48:96, :4].reshape(-1)))} nodes, union {len(touched(same_pg))}")
                    shapes = []
                    for rt, st_ in ((192, 1), (96, 2), (48, 4), (24, 8), (12, 16), (6, 32), (3, 64)):     # workgroup tile: rays x steps
                        t = x[: (R // rt) * rt].view(R // rt, rt, bench.P // st_, st_).permute(0, 2, 1, 3).reshape(-1, 192)
                        sel = t[torch.linspace(0, t.shape[0] - 1, 256).long()]
                        shapes.append(f"{rt}x{st_}: {sum(len(touched(r)) for r in sel) / 256:.0f}")
                    print("      nodes touched per workgroup by tile shape (rays x steps): " + ", ".join(shapes))
                    comp = []
                    for nb, np_ in ((4, 8), (8, 4), (16, 2), (32, 1)):         # an XCD round = nb bundles of 24 rays x np_ groups of 8 steps
                        u = [len(touched(x[24 * nb * k: 24 * nb * (k + 1), 8 * j * np_: 8 * (j + 1) * np_].reshape(-1))) for k in (0, 3) for j in (0, 8 // np_ - 1)]
                        comp.append(f"{24 * nb} rays x {8 * np_} steps: {sum(u) / len(u):.0f}")
                    xr = x.view(32, 256, bench.P)                               # [image row][column][step]
                    for rows_, cols_, np_ in ((4, 24, 8), (8, 12, 8), (4, 48, 4), (8, 24, 4), (16, 12, 4), (8, 96, 1), (32, 24, 1)):   # 2-D pixel blocks
                        u = [len(touched(xr[r0_: r0_ + rows_, c0_: c0_ + cols_, 8 * j * np_: 8 * (j + 1) * np_].reshape(-1)))
                             for r0_, c0_ in ((0, 0), (32 - rows_, 120)) for j in (0, 8 // np_ - 1)]
                        comp.append(f"{rows_} rows x {cols_} cols x {8 * np_} steps: {sum(u) / len(u):.0f}")
                    print("      nodes touched by one XCD round of 6144 samples, by composition: " + "; ".join(comp))
                    tot = touched(x.reshape(-1))
                    print(f"   view {n} source {sv} ({'own, border' if n == sv else 'other, zeros'}): fetching samples {100 * livef:.1f} %, unique nw nodes per 8-row instruction "
                          f"{uniq8:.2f}, nodes touched per workgroup {uw:.0f} of {192 * 4} tap reads, nodes touched by the launch {len(tot)} "
                          f"({len(tot) * 2304 / 1e6:.0f} MB) for {int((x >= 0).sum()) * 4} tap reads: {int((x >= 0).sum()) * 4 / max(len(tot), 1):.1f} reads per node")
        if v == 20:
            w8 = pixel_val.view(torch.int64)[: (2 * R * bench.P // 192) * 12 * 8].view(-1, 8).cpu().double()
            n = w8[:, 3].mean().item()
            m = lambda k: w8[:, k].mean().item() / n
            print(f"source passes, per wave and chunk (mean over {w8.shape[0]} waves, {n:.0f} chunks each; s_memtime ticks): "
                  f"9 x (A-operand reads + 6 MFMAs issued) {m(7):.0f}, DMA pieces {m(5):.0f}, affine {m(6):.0f}, blends incl. the wait for their taps {m(0):.0f}, "
                  f"h rows stored + tap loads issued {m(4):.0f}, chunk-end wait for the weight DMA {m(1):.0f}, barrier {m(2):.0f}")
        if v in (0, 100, 101, 300, 400) or v >= 1000 or 60 <= v <= 69:    # keep the results: the development kernels must equal the product's bit for bit
            outs[v] = [torch.empty(cnt_, device=dev).copy_(eng._work[o_:o_ + cnt_]) for o_, cnt_ in
                       [(lambda n_: (lib.car_workspace_find(ctypes.byref(d), n_.encode(), ctypes.byref(off), ctypes.byref(cnt)), (off.value, cnt.value))[1])(n_)
                        for n_ in ("e", "logit", "pt", "g")]]
            ref = 300 if 300 in outs else 100
            if v != ref and ref in outs:
                for n_, x, y in zip(("e", "logit", "pt", "g"), outs[v], outs[ref]):
                    print(f"   ({v}) vs ({ref}) {n_:6s} max |diff| {(x - y).abs().max().item():.3e}  max |ref| {y.abs().max().item():.3e}  equal {torch.equal(x, y)}")
        print(f"ABL={v}: fused kernel median {ms[len(ms) // 2]:.3f} ms  min {ms[0]:.3f} ms  -> {flop / ms[len(ms) // 2] / 1e9:.1f} TFLOP/s (nominal flops)", flush=True)


if __name__ == "__main__":
    main()
