"""Headline benchmark: rendered rays/s of the epipolar cross-attention render forward (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A *step* renders one full 256x256 query frame (65 536 rays, 64 samples per view, 2 context views = config
"RealEstate10K pair 256x256, 64 samples") of a camera trajectory between the two context views: EVERY step has its own query
pose, so the per-frame pose algebra is inside the timed region, as in the reference's render loop
(render_realestate10k_traj.py:118-137): by default the reference's own torch.inverse calls on the host CPU and a 768-byte pinned
upload queued behind the previous frame (the path whose results the parity tests pin at 1e-4 against the oracle);
``--cameras device`` keeps the cameras on the GPU and runs car_pose_setup there instead (same frame time).  Inputs are synthetic
(seeded stereo pair, N(0,1) feature pyramid, perturbed default-init weights), resident in HBM before the timed region;
``get_z`` (the image encoder) is excluded on both sides, as in BASELINE.md.

With N GPUs ONE frame's rays are banded over the ranks (rank g renders rays [g R/N, (g+1) R/N) of the frame,
pyramid and weights replicated) and the rendered tiles [rgb, depth, valid] are exchanged with one RCCL all-gather per
frame, overlapped with the next frame: strong scaling of a frame.  ``frame_per_rank`` in the JSON line is the other
arrangement (every rank a whole frame of its own), measured right after on fewer steps.

The JSON line also carries
  roofline      : the dominant kernel (the fused per-sample kernel), timed live with HIP events on the launch stream
                  (car_profile_* of the C ABI): algorithmic fp32-equivalent FLOP per launch / mean launch time against the
                  pipes it uses; `bound` names the unit the PMC counters show busiest (profiles/);
  gather_stage  : the stand-alone epipolar gather (car_gather_bilinear, both gathers of a frame chunk) against the HBM roofline
                  with SURVEY.md §8(d)'s algorithmic bytes;
  cpu_baseline  : the CPU oracle (a port of the reference forward, validated against it) on this host's cores over a bounded
                  sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, P, V, CHUNK = 256, 64, 2, 8192
FP32_MFMA_PEAK = 157.3e12          # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)
F16_MFMA_PEAK = 2.5e15             # v_mfma_f32_32x32x16_f16, dense (never the 2:1-sparse marketing figure)
HBM_PEAK = 8.0e12
# matrix-pipe MACs per sample of csrc/car_fused.hip: 2 x 576x288 (e), 576x128 + 128x128 (key), 16x128 + 128x128 (qry)
FUSED_MACS = 2 * 576 * 288 + 576 * 128 + 128 * 128 + 16 * 128 + 128 * 128


def build_model(device):
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=V, npoints=P, with_encoder=False).eval()
    S.perturb_parameters(m, seed=0, scale=0.02)
    m.H = m.W = H
    return m.to(device)


def make_frame(alpha: float, device):
    """Input dict + feature pyramid for one query frame (the same stereo pair, query pose at ``alpha``)."""
    from cross_attention_renderer_amd import synthetic as S
    inp = S.stereo_scene(H, b=1, alpha=alpha, seed=5)
    z = S.feature_maps(1, V, H, seed=1)
    if device is not None:
        inp = {k: {kk: vv.to(device) for kk, vv in v.items()} for k, v in inp.items()}
        z = [t.to(device) for t in z]
    return inp, z


def trajectory(n_frames: int, device, band=None, cameras_on_host: bool = False):
    """``n_frames`` input dicts of the same stereo pair, query pose moving from one context camera towards the other;
    ``band`` = (start, end) restricts the rays (this rank's share of the frame)."""
    base, _ = make_frame(0.5, None)
    frames = []
    for i in range(n_frames):
        inp, _ = make_frame((i + 0.5) / n_frames, None)
        uv = base["query"]["uv"] if band is None else base["query"]["uv"][:, :, band[0]:band[1]].contiguous()
        cam = (lambda t: t) if cameras_on_host else (lambda t: t.to(device))
        frames.append({"context": {k: (cam(v) if k in ("cam2world", "intrinsics") else v.to(device)) for k, v in base["context"].items()},
                       "query": {"cam2world": cam(inp["query"]["cam2world"]), "intrinsics": cam(inp["query"]["intrinsics"]),
                                 "uv": uv.to(device)}})
    return frames


def render_frame(model, inp, z, tile, chunk_rays):
    """One step: the rays of ``inp`` in forward calls of ``chunk_rays`` rays, results packed as [rgb(3), depth, valid]."""
    uv_all = inp["query"]["uv"]
    R = uv_all.shape[2]
    for c0 in range(0, R, chunk_rays):
        chunk = inp if chunk_rays >= R else {"context": inp["context"], "query": dict(inp["query"], uv=uv_all[:, :, c0:c0 + chunk_rays])}
        out = model(chunk, z=z)
        tile[c0:c0 + chunk_rays, 0:3] = out["rgb"][0, 0]
        tile[c0:c0 + chunk_rays, 3:4] = out["depth_ray"][0]
        tile[c0:c0 + chunk_rays, 4:5] = out["valid_mask"][0]
    return tile


def cpu_baseline(rays: int):
    """The CPU oracle on a bounded sample of the same workload (same pair, weights, feature maps)."""
    from oracle import car_oracle as O
    model = build_model(torch.device("cpu"))
    inp, z = make_frame(0.5, None)
    sd = {k: v for k, v in model.state_dict().items()}
    uv = inp["query"]["uv"][:, :, 96 * H: 96 * H + rays].contiguous()
    inp = {"context": inp["context"], "query": dict(inp["query"], uv=uv)}
    cfg = O.RenderConfig(n_view=V, npoints=P, H=H, W=H)
    best = float("inf")
    with torch.no_grad():
        O.render_forward(sd, inp, z, cfg)                      # warm-up
        for _ in range(2):
            t0 = time.perf_counter()
            O.render_forward(sd, inp, z, cfg)
            best = min(best, time.perf_counter() - t0)
    threads = torch.get_num_threads()
    model_name, phys = "unknown", None
    try:
        sockets, per_socket = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model_name == "unknown":
                    model_name = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    sockets.add(line.split(":", 1)[1].strip())
                elif line.startswith("cpu cores") and per_socket is None:
                    per_socket = int(line.split(":", 1)[1])
        if sockets and per_socket:
            phys = len(sockets) * per_socket
    except (OSError, ValueError):
        pass
    return {"value": rays / best, "unit": "rays/s", "cores": threads, "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "kind": "port",
            "sample": f"{rays} rays of the same 256x256x64 frame, best of 2 after warm-up, oracle/car_oracle.py with {threads} torch threads on "
                      f"{model_name} ({phys if phys else '?'} physical cores, {os.cpu_count()} logical CPUs)"}


def gather_stage(model, inp, z, rays: int = CHUNK):
    """SURVEY.md §8(d), second figure: the stand-alone gather stage (car_gather_bilinear over the raw channel-last pyramid, once
    with border and once with zeros padding = a7 + a10) for one chunk of ``rays`` rays of the frame, at the chunk's epipolar sample
    positions.  Algorithmic bytes = both gathered tensors written (2 V P C 4 per ray) + the pyramid once; time = HIP events around
    the two launches, median of 5."""
    eng = model._engine
    dev = inp["query"]["uv"].device
    maps = eng._channel_last(z)
    C = sum(t.shape[3] for t in maps)
    sub = {"context": inp["context"], "query": dict(inp["query"], uv=inp["query"]["uv"][:, :, 96 * H:96 * H + rays].contiguous())}
    grid = model(sub, z=z)["pixel_val"].reshape(V, rays * P, 2).contiguous()
    out = torch.empty(V * rays * P, C, device=dev)
    ev = []
    for _ in range(7):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.gather(maps, grid, rays * P, 0, 0, V, out, C, 0, run=P)   # a7: border
        eng.gather(maps, grid, rays * P, 1, 0, V, out, C, 0, run=P)   # a10: zeros
        b_.record()
        ev.append((a, b_))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in ev[2:])[2]
    nbytes = 2 * out.numel() * 4 + sum(t.numel() * 4 for t in maps)
    return {"kernel": "gather_wave_kernel (car_gather_bilinear), a7 + a10 of one 8192-ray chunk", "bound": "hbm",
            "algorithmic_bytes": nbytes, "ms": ms, "achieved": nbytes / (ms * 1e-3) / 1e12, "peak": HBM_PEAK / 1e12, "unit": "TB/s",
            "frac": nbytes / (ms * 1e-3) / HBM_PEAK,
            "note": "stand-alone stage only: the product path fuses the gather into the per-sample kernel and writes no gathered features"}


def timed_loop(model, frames, z, tile, gather, steps, chunk_rays, dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        render_frame(model, frames[i % len(frames)], z, tile, chunk_rays)
        if gather is not None:
            gather(tile)
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--chunk-rays", type=int, default=65536,
                    help="rays per forward call: the whole frame (or band) by default (a 288 GB GPU does not need the reference render script's 8192-ray chunks; --chunk-rays 8192 reproduces them)")
    ap.add_argument("--no-extras", action="store_true", help="skip the gather-stage and frame-per-rank measurements")
    ap.add_argument("--cameras", choices=("host", "device"), default="host",
                    help="where the camera matrices live: host = the reference's own torch.inverse on the CPU per frame + a 768-byte "
                         "upload (the arithmetic the parity tests pin; default); device = car_pose_setup per frame (no host work)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test the multi-rank code path)")
    ap.add_argument("--device", type=int, default=None, help="device index for every rank (smoke tests of the multi-rank path on one GPU; default LOCAL_RANK)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm device: the render path has no CPU fallback")
    if args.device is not None:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)      # nccl == RCCL on ROCm

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()                                   # one builder per node: concurrent hipcc runs would race on the .so
    if dist is not None:
        dist.barrier()
    from cross_attention_renderer_amd.engine import RenderEngine
    from cross_attention_renderer_amd.sharding import TileGather, ray_band

    model = build_model(dev)
    model._engine = RenderEngine(model)
    _, z = make_frame(0.5, dev)
    R_frame = H * H
    band = ray_band(R_frame, rank, world)
    R = band[1] - band[0]
    n_frames = args.steps + args.warmup
    frames = trajectory(n_frames, dev, band if world > 1 else None, args.cameras == "host")
    tile = torch.empty(R, 5, device=dev)
    gather = TileGather(world, -(-R_frame // world), 5, dev) if world > 1 else None
    if gather is not None and R != -(-R_frame // world):
        sys.exit("bench.py: the frame's rays must divide evenly over the ranks")

    with torch.no_grad():
        for i in range(args.warmup):
            render_frame(model, frames[args.steps + i], z, tile, args.chunk_rays)
            if gather is not None:
                gather(tile)
        if gather is not None:
            gather.wait()
        model._engine.profile(True)                              # stage events from here on (rank-local)
        elapsed = timed_loop(model, frames, z, tile, gather, args.steps, args.chunk_rays, dist)
        stages = model._engine.stage_times()
        model._engine.profile(False)

        # the other multi-GPU arrangement: every rank renders whole frames of its own (replicas), tiles all-gathered
        per_rank = None
        if world > 1 and not args.no_extras:
            k2 = max(2, args.steps // 2)
            full = trajectory(k2, dev, None, args.cameras == "host")
            tile2 = torch.empty(R_frame, 5, device=dev)
            g2 = TileGather(world, R_frame, 5, dev)
            render_frame(model, full[0], z, tile2, args.chunk_rays)
            t2 = timed_loop(model, full, z, tile2, g2, k2, args.chunk_rays, dist)
            per_rank = (k2, t2)

    rank_stages = None
    if dist is not None:
        mine = {}
        for name, ms in stages:
            mine.setdefault(name, []).append(ms)
        mine = {k: sum(v) / len(v) for k, v in mine.items()}
        rank_stages = [None] * world
        dist.all_gather_object(rank_stages, mine)            # every rank's mean stage times: a short scaling curve can be read from one run
        t = torch.tensor([elapsed, per_rank[1] if per_rank else 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t[0].item()
        if per_rank:
            per_rank = (per_rank[0], t[1].item())

    if rank == 0:
        rays_total = R_frame * args.steps
        prof = {}
        try:
            with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
                prof = json.load(f)
        except OSError:
            pass
        by_stage = {}
        for name, ms in stages:
            by_stage.setdefault(name, []).append(ms)
        roof = None
        lat = by_stage.get("fused_samples")
        if lat:
            mean = sum(lat) / len(lat) * 1e-3
            samples = V * R * P * (min(args.chunk_rays, R) / R)
            flop = 2.0 * samples * FUSED_MACS
            pmc = dict(prof.get("fused_samples", {}))
            part = samples / float(V * H * H * P)              # the PMC passes profiled a whole-frame launch: scale to this launch
            for k in ("l1_bytes_per_launch", "bytes_per_launch"):
                if pmc.get(k) is not None:
                    pmc[k] = pmc[k] * part
            if part != 1.0 and pmc.get("source"):
                pmc["source"] += f", scaled by {part:g} to this launch's share of the frame"
            roof = {"bound": "mfma", "limiter": pmc.get("limiter"), "kernel": f"fused_kernel on {int(samples)} samples (geometry, 4-tap gather of the per-texel-projected pyramid on its lattice, e, key, qry, logits; "
                                                                  "f16 matrix pipe, fp16 hi/lo split x3)",
                    "achieved": flop / mean / 1e12, "peak": F16_MFMA_PEAK / 3 / 1e12, "unit": "TFLOP/s", "frac": flop / mean / (F16_MFMA_PEAK / 3),
                    "peak_note": "dense f16 MFMA peak 2500 / 3 products per fp32 term.  What keeps the kernel from it (limiter): MFMAs and ordinary "
                                 "vector instructions share a SIMD's issue, the A operands wait on LDS, and the tap loads stall in the texture "
                                 "path's issue (see ta_busy, mfma_busy, valu_share; DESIGN.md section 4.6)",
                    "frac_of_fp32_pipe_peak": flop / mean / FP32_MFMA_PEAK,
                    "ta_busy": pmc.get("ta_busy"), "mfma_busy": pmc.get("mfma_busy"), "valu_share": pmc.get("valu_share"), "l1_bytes": pmc.get("l1_bytes_per_launch"),
                    "traffic": pmc.get("bytes_per_launch"), "traffic_source": pmc.get("source"),
                    # which of these fields this run measured and which it copied from the committed counter passes
                    "live_fields": ["achieved", "frac", "frac_of_fp32_pipe_peak", "launches", "ms_per_launch", "flop_per_launch"],
                    "static_fields": ["limiter", "ta_busy", "mfma_busy", "valu_share", "l1_bytes", "traffic"],
                    "static": "profiles/traffic.json: rocprofv3 --pmc passes of an earlier run of this kernel (separate passes, not collected here)",
                    "launches": len(lat), "ms_per_launch": mean * 1e3, "flop_per_launch": flop}
        fr = prof.get("frame")
        hbm = None
        if fr and args.chunk_rays >= R and world == 1:
            per_frame = fr["bytes_per_frame"]
            hbm = {"bytes_per_frame": per_frame, "achieved": per_frame / (elapsed / args.steps) / 1e12, "peak": 8.0, "unit": "TB/s",
                   "frac": per_frame / (elapsed / args.steps) / HBM_PEAK, "source": fr["source"],
                   "static": "bytes_per_frame comes from profiles/traffic.json (PMC passes, not collected in this run); the time is this run's"}
        gs = None
        if world == 1 and not args.no_extras:
            with torch.no_grad():
                gs = gather_stage(model, frames[0], z)
        ref_flop = V * P * 2617728 + 791808
        line = {
            "metric": "rendered_rays_per_sec", "value": rays_total / elapsed, "unit": "rays/s",
            "frames_per_sec": args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 (split-f16 x3 MFMA)", "data": "synthetic",
            "config": {"workload": f"256x256 query frame, 64 samples/view, 2 context views (config 2), a new query pose every step (cameras on the {args.cameras}), "
                                   f"{-(-R // args.chunk_rays)} forward call(s) x {min(args.chunk_rays, R)} rays per rank",
                       "rays_per_step": R_frame, "rays_per_step_per_gpu": R,
                       "parallelism": "one GPU" if world == 1 else f"one frame's rays banded over {world} ranks, all-gather of tiles over {'RCCL' if args.backend == 'nccl' else args.backend}"
                                                                      + (f" (all ranks on device {args.device}: a dry run of the code path, no scaling figure)" if args.device is not None else "")},
            "stage_ms": {k: sum(v) / len(v) for k, v in by_stage.items()},
            "stage_ms_per_rank": rank_stages,
            "roofline": roof,
            "hbm": hbm,
            "gather_stage": gs,
            "frame_per_rank": None if per_rank is None else {
                "value": world * R_frame * per_rank[0] / per_rank[1], "unit": "rays/s", "steps": per_rank[0],
                "ms_per_step": per_rank[1] / per_rank[0] * 1e3, "note": "every rank renders whole frames of its own (replicas), tiles all-gathered"},
            # SURVEY.md §8(d) prices the path with the REFERENCE's arithmetic: V*P*2 617 728 + 791 808 FLOP per ray (335.9 MFLOP at
            # P = 64) against the fp32 matrix peak.  This implementation executes about a third of that (first point-MLP layer per
            # texel, value projection after the attention average) and runs it on the f16 pipe, hence a figure above 1.
            "reference_flops": {"flop_per_ray": ref_flop, "equivalent_tflops": rays_total / elapsed * ref_flop / 1e12,
                                "frac_of_fp32_mfma_peak": rays_total / elapsed * ref_flop / FP32_MFMA_PEAK / world},
            # the CPU leg runs on rank 0 of a single-GPU job only (it is a per-host figure and would stall the other ranks)
            "cpu_baseline": cpu_baseline(args.cpu_rays) if (args.cpu_rays > 0 and world == 1) else None,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
