"""Headline benchmark: rendered rays/s of the epipolar cross-attention render forward (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)
    python bench.py --config c3|c4|c5                    (builder-kept lines of the other single-GPU BASELINE configurations)

A *step* renders one full query frame of the configuration (default c2 = BASELINE's headline: "RealEstate10K pair 256x256, 64 samples" =
65 536 rays, 64 samples per view, 2 context views) of a camera trajectory between the two context views: EVERY step has its own query
pose, so the per-frame pose algebra is inside the timed region, as in the reference's render loop
(render_realestate10k_traj.py:118-137): the reference's own torch.inverse calls on the host CPU and a 768-byte pinned upload queued
behind the previous frame (the strict-parity route, engine._poses).  ``--cameras gpu`` hands the whole input dict over on the GPU as the
reference's scripts do (same arithmetic, one small download + stream synchronisation per pose); ``--cameras device`` opts into
car_pose_setup on the GPU.  Inputs are synthetic (seeded stereo pair, N(0,1) feature pyramid, perturbed default-init weights), resident in
HBM before the timed region; ``get_z`` (the image encoder) is excluded on both sides, as in BASELINE.md.

Configurations (sizes of BASELINE.json's configs that fit one GPU):
  c2  256 x 256, 64 samples, 1 scene: one forward call of 65 536 rays per step                        (the driver's line)
  c3  256 x 256, 64 samples, batch of 12 scenes: on one GPU a rank's share of the 8-GPU job (12 x 8192 rays in one call per step); with
      --gpus N the whole frames of all twelve scenes, their rays banded over the N ranks (12 x 65536 / N rays per rank and step)
  c4  256 x 256, 128 samples (gather-bound stress), 1 scene, 65 536 rays per step
  c5  384 x 384, 64 samples, 1 scene, 147 456 rays per step

With N GPUs ONE frame's rays are banded over the ranks (rank g renders rays [g R/N, (g+1) R/N) of the frame,
pyramid and weights replicated) and the rendered tiles [rgb, depth, valid] are exchanged with one RCCL all-gather per
frame, overlapped with the next frame: strong scaling of a frame.  ``frame_per_rank`` in the JSON line is the other
arrangement (every rank a whole frame of its own), measured right after on fewer steps.

The JSON line also carries
  roofline      : the dominant kernel (the fused per-sample kernel), timed live with HIP events on the launch stream
                  (car_profile_* of the C ABI): algorithmic fp32-equivalent FLOP per launch / mean launch time against the
                  pipes it uses; `bound` names the unit the PMC counters show busiest (profiles/);
  gather_stage  : the stand-alone epipolar gather (car_gather_bilinear, both gathers of a frame chunk) against the HBM roofline
                  with SURVEY.md §8(d)'s algorithmic bytes — warmed up, 24 timed pairs, median / min / max;
  pair_setup_ms, lattice_bytes, workspace_bytes, eval_mode : what the restructured path costs outside the timed region — the
                  once-per-stereo-pair projection of the pyramid onto its lattice, the memory it and the per-call workspace hold, and
                  the frame rate when every frame brings a new pair (the eval loop, eval_realestate10k.py:142-161);
  first_round_ab: the same steps with the first attention round over the rows of e (rounds 1-4) instead of the fused kernel's partial sums;
  rank_share    : one call of 1/G of the frame per step (a rank's share at G = 2, 4, 8 GPUs) and the scaling each projects;
  power         : socket power, shader clock and joules per frame, sampled while the timed loop's steps run a second time;
  pose_route    : what handing the cameras over on the GPU costs per frame (the download + synchronisation of the host pose route);
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

V, CHUNK = 2, 8192
CONFIGS = {
    # name: (H, P, scenes per step, rays per scene and step, description)
    "c2": (256, 64, 1, 256 * 256, "256x256 query frame, 64 samples/view, 2 context views (config 2)"),
    "c3": (256, 64, 12, 8192, "batch of 12 scenes at 256x256, 64 samples/view, one rank's band of 8192 rays per scene (config 3's per-rank share at 8 GPUs)"),
    "c4": (256, 128, 1, 256 * 256, "256x256 query frame, 128 samples/view (config 4, gather-bound stress)"),
    "c5": (384, 64, 1, 384 * 384, "384x384 query frame of an UNPOSED pair (seeded R, unit t / 1.2 as the essential-matrix route places them), 64 samples/view (config 5's frame on one GPU)"),
}
H, P = 256, 64                     # the headline configuration's sizes (module-level for tools/ that import this file)
SCENE = "stereo"                   # "unposed" for config 5: the pair of synthetic.unposed_scene (seeded R, unit t)
FP32_MFMA_PEAK = 157.3e12          # v_mfma_f32_32x32x2_f32, dense (MI355X_MICROARCH.md)
F16_MFMA_PEAK = 2.5e15             # v_mfma_f32_32x32x16_f16, dense (never the 2:1-sparse marketing figure)
HBM_PEAK = 8.0e12
# ALGORITHMIC matrix-pipe MACs per sample of the path the fused kernel covers (models.py:333-344, 487-491, 529): 2 x 576x288 (e), 576x128 +
# 128x128 (key), 16x128 + 128x128 (qry); independent of P.  Since round 6 the kernel EXECUTES 128x128 fewer (key_map_2 and query_embed_2 are
# folded into one layer, car_fused_layout.h): FUSED_MACS_EXECUTED; the roofline's `achieved` keeps the algorithmic count, as in every round.
FUSED_MACS = 2 * 576 * 288 + 576 * 128 + 128 * 128 + 16 * 128 + 128 * 128
FUSED_MACS_EXECUTED = FUSED_MACS - 128 * 128


def build_model(device, P_=None, H_=None):
    from cross_attention_renderer_amd import synthetic as S
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    torch.manual_seed(0)
    m = CrossAttentionRenderer(model="midas_vit", n_view=V, npoints=P if P_ is None else P_, with_encoder=False).eval()
    S.perturb_parameters(m, seed=0, scale=0.02)
    m.H = m.W = H if H_ is None else H_
    return m.to(device)


def make_frame(alpha: float, device, H_=None, b=1):
    """Input dict + feature pyramid for one query frame (the same stereo pair(s), query pose at ``alpha``)."""
    from cross_attention_renderer_amd import synthetic as S
    Hh = H if H_ is None else H_
    if SCENE == "unposed":                                             # query = a pose of the unposed demo's rotate_interpolate path
        inp = S.unposed_scene(Hh, frame=int(round(alpha * 75)), seed=5)
    else:
        inp = S.stereo_scene(Hh, b=b, alpha=alpha, seed=5)
    z = S.feature_maps(b, V, Hh, seed=1)
    if device is not None:
        inp = {k: {kk: vv.to(device) for kk, vv in v.items()} for k, v in inp.items()}
        z = [t.to(device) for t in z]
    return inp, z


def trajectory(n_frames: int, device, band=None, cameras_on_host: bool = False, H_=None, b=1):
    """``n_frames`` input dicts of the same stereo pair(s), query pose moving from one context camera towards the other;
    ``band`` = (start, end) restricts the rays (this rank's share of the frame)."""
    base, _ = make_frame(0.5, None, H_, b)
    frames = []
    for i in range(n_frames):
        inp, _ = make_frame((i + 0.5) / n_frames, None, H_, b)
        uv = base["query"]["uv"] if band is None else base["query"]["uv"][:, :, band[0]:band[1]].contiguous()
        cam = (lambda t: t) if cameras_on_host else (lambda t: t.to(device))
        frames.append({"context": {k: (cam(v) if k in ("cam2world", "intrinsics") else v.to(device)) for k, v in base["context"].items()},
                       "query": {"cam2world": cam(inp["query"]["cam2world"]), "intrinsics": cam(inp["query"]["intrinsics"]),
                                 "uv": uv.to(device)}})
    return frames


def render_frame(model, inp, z, tile, chunk_rays):
    """One step: the rays of ``inp`` in forward calls of ``chunk_rays`` rays, results packed as [rgb(3), depth, valid] per scene."""
    uv_all = inp["query"]["uv"]
    R = uv_all.shape[2]
    for c0 in range(0, R, chunk_rays):
        chunk = inp if chunk_rays >= R else {"context": inp["context"], "query": dict(inp["query"], uv=uv_all[:, :, c0:c0 + chunk_rays])}
        out = model(chunk, z=z)
        tile[:, c0:c0 + chunk_rays, 0:3] = out["rgb"][:, 0]
        tile[:, c0:c0 + chunk_rays, 3:4] = out["depth_ray"]
        tile[:, c0:c0 + chunk_rays, 4:5] = out["valid_mask"]
    return tile


def cpu_baseline(rays: int, Hc: int, Pc: int):
    """The CPU oracle on a bounded sample of the same workload (same pair, weights, feature maps)."""
    from oracle import car_oracle as O
    model = build_model(torch.device("cpu"), Pc, Hc)
    inp, z = make_frame(0.5, None, Hc)
    sd = {k: v for k, v in model.state_dict().items()}
    r0 = (3 * Hc // 8) * Hc
    uv = inp["query"]["uv"][:, :, r0: r0 + rays].contiguous()
    inp = {"context": inp["context"], "query": dict(inp["query"], uv=uv)}
    cfg = O.RenderConfig(n_view=V, npoints=Pc, H=Hc, W=Hc)
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
            "sample": f"{rays} rays of the same {Hc}x{Hc}x{Pc} frame, best of 2 after warm-up, oracle/car_oracle.py with {threads} torch threads on "
                      f"{model_name} ({phys if phys else '?'} physical cores, {os.cpu_count()} logical CPUs)"}


def gather_stage(model, inp, z, Hc: int, Pc: int, rays: int = CHUNK, warm: int = 30, pairs: int = 24):
    """SURVEY.md §8(d), second figure: the stand-alone gather stage (car_gather_bilinear over the raw channel-last pyramid, once
    with border and once with zeros padding = a7 + a10) for one chunk of ``rays`` rays of scene 0's frame, at the chunk's epipolar
    sample positions.  Algorithmic bytes = both gathered tensors written (2 V P C 4 per ray) + the pyramid once.  ``warm`` untimed pairs
    first (the clocks of a box that was idle or on its CPU leg settle), then ``pairs`` pairs each between two HIP events: the line
    carries the median, the fastest and the slowest pair, so a noisy box shows as a spread instead of as another number."""
    eng = model._engine
    dev = inp["query"]["uv"].device
    maps = [t[:V] for t in eng._channel_last(z)]
    C = sum(t.shape[3] for t in maps)
    r0 = min((3 * Hc // 8) * Hc, max(0, inp["query"]["uv"].shape[2] - rays))          # c3's frames hold one band of rays only
    sub = {"context": {k: v[:1] for k, v in inp["context"].items()},
           "query": {k: (v[:1, :, r0:r0 + rays].contiguous() if k == "uv" else v[:1]) for k, v in inp["query"].items()}}
    grid = model(sub, z=z if z[0].shape[0] == V else [t[:V] for t in z])["pixel_val"].reshape(V, rays * Pc, 2).contiguous()
    out = torch.empty(V * rays * Pc, C, device=dev)
    ev = []
    for i in range(warm + pairs):
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.gather(maps, grid, rays * Pc, 0, 0, V, out, C, 0, run=Pc)   # a7: border
        eng.gather(maps, grid, rays * Pc, 1, 0, V, out, C, 0, run=Pc)   # a10: zeros
        b_.record()
        if i >= warm:
            ev.append((a, b_))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b_) for a, b_ in ev)
    med = ms[len(ms) // 2]
    nbytes = 2 * out.numel() * 4 + sum(t.numel() * 4 for t in maps)
    frac = lambda t: nbytes / (t * 1e-3) / HBM_PEAK
    return {"kernel": f"gather_wave_kernel (car_gather_bilinear), a7 + a10 of one {rays}-ray chunk", "bound": "hbm",
            "algorithmic_bytes": nbytes, "ms": med, "ms_min": ms[0], "ms_max": ms[-1], "pairs": len(ms), "warmup_pairs": warm,
            "achieved": nbytes / (med * 1e-3) / 1e12, "peak": HBM_PEAK / 1e12, "unit": "TB/s",
            "frac": frac(med), "frac_min": frac(ms[-1]), "frac_max": frac(ms[0]), "spread": (ms[-1] - ms[0]) / med,
            "note": "stand-alone stage only: the product path fuses the gather into the per-sample kernel and writes no gathered features; "
                    "measured before the warm-up and timed steps, after one priming frame"}


def timed_loop(model, frames, z, tile, gather, steps, chunk_rays, dist):
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        render_frame(model, frames[i % len(frames)], z, tile, chunk_rays)
        if gather is not None:
            gather(tile.view(-1, tile.shape[-1]))
    if gather is not None:
        gather.wait()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    return time.perf_counter() - t0


def pair_setup(model, frame, z, reps: int = 3):
    """car_project_maps of the frame's stereo pair(s): the once-per-pair part of the restructured path (first point-MLP layer per
    texel + the merge onto the common lattice), timed by forcing it ``reps`` times (HIP events; median)."""
    eng = model._engine
    ms = []
    for _ in range(reps):
        eng._pair_key = None
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b, R = frame["query"]["uv"].shape[0], frame["query"]["uv"].shape[2]
        plan = eng._plan_for(eng._dims(b, R, z), z[0].device)
        a.record()
        eng._pair_for(plan, z, z[0].device, 0, b, R)
        b_.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b_))
    return sorted(ms)[len(ms) // 2]


def eval_mode(model, frames, z, tile, steps: int, prefetch: bool = True):
    """The eval loop's shape (eval_realestate10k.py:142-161): every frame belongs to a NEW stereo pair, so the pyramid is re-laid channel-last
    and the lattice re-projected for every frame (the pyramid itself — get_z — is excluded as everywhere).  Two copies of the pyramid
    alternate, so every frame's pair differs from the one in place.  ``prefetch``: the loop announces the next pair before it renders the
    current one (model.prefetch_pair, what experiment_scripts/eval_realestate10k.py does): the set-up runs on a side stream beside the
    render; False: the set-up runs inside the next forward, in front of its kernels (rounds 4-5)."""
    pyr = (z, [t.clone() for t in z])
    render_frame(model, frames[0], pyr[0], tile, 1 << 30)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pyr = pyr + ([t.clone() for t in z],)                           # three copies: (in place, announced, next) are distinct tensors, as in the loop
    for i in range(steps):
        if prefetch:
            model.prefetch_pair(pyr[(i + 1) % 3])                   # the loop's order: the next pair is announced, then the current one rendered
        render_frame(model, frames[i % len(frames)], pyr[i % 3], tile, 1 << 30)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    model._engine.drop_prefetched()
    render_frame(model, frames[0], z, tile, 1 << 30)               # the caller's pyramid back in place
    torch.cuda.synchronize()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2",
                    help="which BASELINE configuration a step renders (c2 = the headline, what the driver runs; c3 / c4 / c5: builder-kept lines)")
    ap.add_argument("--cpu-rays", type=int, default=4096, help="rays in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--chunk-rays", type=int, default=1 << 30,
                    help="rays per scene and forward call: the whole frame (or band) by default (a 288 GB GPU does not need the reference render script's 8192-ray chunks; --chunk-rays 8192 reproduces them)")
    ap.add_argument("--no-extras", action="store_true", help="skip the gather-stage, pair-setup, eval-mode, rank-share, pose-route and frame-per-rank measurements")
    ap.add_argument("--cameras", choices=("host", "gpu", "device"), default="host",
                    help="host = camera matrices stay CPU tensors: the reference's own torch.inverse on the CPU per frame + a 768-byte upload, no "
                         "device synchronisation (default); gpu = the whole input dict on the GPU as the reference's scripts hand it over: same "
                         "host arithmetic after one small download per frame; device = car_pose_setup on the GPU (opt-in, last-ulp differences)")
    ap.add_argument("--rays-per-scene", type=int, default=None,
                    help="c3 with N > 1 ranks: rays of every scene's frame that are banded over the ranks (default: the whole 256 x 256 frame; "
                         "smaller values only for dry runs of the code path with all ranks on one device)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to smoke-test the multi-rank code path)")
    ap.add_argument("--device", type=int, default=None, help="device index for every rank (smoke tests of the multi-rank path on one GPU; default LOCAL_RANK)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): launch with "
                 f"python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a ROCm device: the render path has no CPU fallback")
    if args.device is None and world > torch.cuda.device_count():
        sys.exit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} visible device(s)")
    if args.device is not None:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)      # nccl == RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            sys.exit(f"bench.py: the process group reports {dist.get_world_size()} ranks, --gpus asked for {args.gpus}")

    import __graft_entry__ as ge
    if rank == 0:
        ge.build()                                   # one builder per node: concurrent hipcc runs would race on the .so
    if dist is not None:
        dist.barrier()
    from cross_attention_renderer_amd.engine import RenderEngine
    from cross_attention_renderer_amd.sharding import TileGather, ray_band

    Hc, Pc, nb, R_frame, what = CONFIGS[args.config]
    if args.config == "c5":
        global SCENE
        SCENE = "unposed"
    if world > 1 and nb != 1:
        # config 3 on N ranks (BASELINE: "batch_size 12, ray-sharded across 8 GPUs"): the rays of ALL twelve scenes' frames are banded —
        # rank g renders rays [g R / N, (g + 1) R / N) of every scene (shard rays, not scenes: 12 is no multiple of 8) and the
        # (12, R / N, 5) tiles are exchanged by one all-gather per step.  At N = 8 a rank's call is the 12 x 8192 rays the one-GPU line times.
        R_frame = args.rays_per_scene or Hc * Hc
        what = f"batch of 12 scenes at {Hc}x{Hc}, {Pc} samples/view, {R_frame} rays of every scene banded over {world} ranks (config 3)"
    model = build_model(dev, Pc, Hc)
    model.pose_route = "device" if args.cameras == "device" else "host"
    model._engine = RenderEngine(model)
    _, z = make_frame(0.5, dev, Hc, nb)
    band = ray_band(R_frame, rank, world)
    if nb != 1 and world == 1:
        band = ((3 * Hc // 8) * Hc, (3 * Hc // 8) * Hc + R_frame)            # c3 on one GPU: one rank's band of image rows of every scene
    R = band[1] - band[0]
    n_frames = args.steps + args.warmup
    frames = trajectory(n_frames, dev, band if (world > 1 or nb != 1) else None, args.cameras == "host", Hc, nb)
    tile = torch.empty(nb, R, 5, device=dev)
    # one all-gather per step of this rank's (scenes, rays, 5) tile, flattened to rows: (world, nb * R / N, 5) on every rank
    gather = TileGather(world, nb * -(-R_frame // world), 5, dev) if world > 1 else None
    if gather is not None and R != -(-R_frame // world):
        sys.exit("bench.py: the frame's rays must divide evenly over the ranks")
    extras = world == 1 and not args.no_extras

    gs = setup_ms = None
    with torch.no_grad():
        # one priming frame (plan, lattice, workspace: untimed set-up, not one of the W warm-up steps), then the measurements that must
        # not sit between the warm-up and the timed steps (they free and allocate gigabytes: the first frame after them pays the
        # allocator's device mallocs), then W warm-up steps, then the K timed ones
        render_frame(model, frames[0], z, tile, args.chunk_rays)
        eng = model._engine
        lattice_bytes = eng._pair.numel() * 4 if eng._pair is not None else None
        workspace_bytes = eng._work.numel() * 4 if eng._work is not None else None
        if extras:
            gs = gather_stage(model, frames[0], z, Hc, Pc)
            setup_ms = pair_setup(model, frames[0], z)
        for i in range(args.warmup):
            render_frame(model, frames[args.steps + i], z, tile, args.chunk_rays)
            if gather is not None:
                gather(tile.view(-1, 5))
        if gather is not None:
            gather.wait()
        model._engine.profile(True)                              # stage events from here on (rank-local)
        elapsed = timed_loop(model, frames, z, tile, gather, args.steps, args.chunk_rays, dist)
        stages = model._engine.stage_times()
        model._engine.profile(False)

        ev_ms = ev_ms_pf = share = pose = power = ab = None
        if extras:
            # A/B of this round's change to the tail: the same K steps with the first attention round streaming the rows of e
            # (CAR_PHASE_ROWS_FIRST_ROUND: the fused kernel without its partial sums + car_attend over e, the form of rounds 1-4)
            eng.first_round_parts = False
            render_frame(model, frames[0], z, tile, args.chunk_rays)
            eng.profile(True)
            e_rows = timed_loop(model, frames, z, tile, None, args.steps, args.chunk_rays, None)
            st_rows = {}
            for name, ms in eng.stage_times():
                st_rows.setdefault(name, []).append(ms)
            eng.profile(False)
            eng.first_round_parts = True
            render_frame(model, frames[0], z, tile, args.chunk_rays)
            ab = {"rows_of_e": {"ms_per_step": e_rows / args.steps * 1e3, "stage_ms": {k: sum(v) / len(v) for k, v in st_rows.items()}},
                  "partial_sums": {"ms_per_step": elapsed / args.steps * 1e3},
                  "note": "first attention round over the rows of e (rounds 1-4; fused kernel without the partial-sum phase) against the default "
                          "(partial sums per 8-step group left by the fused kernel): what the phase costs the fused kernel and saves the round"}

            # the same K steps once more under a socket-power / shader-clock sampler (tools/power_sampler.py: amdsmi, 5 ms period, and the
            # device's energy accumulator): what the chip's power management delivers under THIS workload, measured beside the timed
            # loop instead of quoted from an earlier profile.  Kept out of the timed region so the sampler thread cannot touch `value`.
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                from power_sampler import PowerSampler
                with PowerSampler(interval=0.005) as ps:
                    e_pw = timed_loop(model, frames, z, tile, None, args.steps, args.chunk_rays, None)
                power = ps.summary()
                power["ms_per_step"] = e_pw / args.steps * 1e3
                if power.get("available"):
                    w_mean = power.get("energy_mean_w", power["mean_w"])
                    power["joule_per_frame"] = w_mean * e_pw / args.steps
                    power["nanojoule_per_sample"] = w_mean * e_pw / args.steps / (nb * V * R * Pc) * 1e9
                    power["note"] = ("socket power and shader clock sampled while the timed loop's K steps ran a second time (whole frame: fused kernel "
                                     "+ attention + per-ray chains); the fused kernel alone by component: profiles/round5_fused_energy.md")
            except Exception as exc:                                  # measurement plumbing must never cost the line
                power = {"available": False, "error": repr(exc)}
            ev_ms = eval_mode(model, frames, z, tile, max(4, min(args.steps, 8)), False)
            ev_ms_pf = eval_mode(model, frames, z, tile, max(4, min(args.steps, 8)), True)
            if nb == 1 and args.chunk_rays >= R:
                # a rank's share of this frame at G GPUs: one call of R / G rays per step, every step a new pose — what bounds the scaling
                # of the banded frame before the (overlapped) all-gather
                share = {}
                for G in (2, 4, 8):
                    kG, rs = max(10, 2 * args.steps), R_frame // G
                    g0 = (3 * G // 8) * rs                                    # a band from the middle of the frame (rank 3 of 8, 1 of 4, 0 of 2)
                    frG = trajectory(kG, dev, (g0, g0 + rs), args.cameras == "host", Hc, nb)
                    tG = torch.empty(nb, rs, 5, device=dev)
                    render_frame(model, frG[0], z, tG, 1 << 30)
                    eG = min(timed_loop(model, frG, z, tG, None, kG, 1 << 30, None) for _ in range(2))     # the faster of two passes: the first
                    # one after a change of the ray count can carry an allocator refill
                    share[f"projected_scaling_{G}"] = (elapsed / args.steps) / (eG / kG)
                    share[f"ms_per_step_{G}"] = eG / kG * 1e3
                    if G == 8:                                                # where an eighth of the frame spends its time (HIP events per stage)
                        eng.profile(True)
                        timed_loop(model, frG, z, tG, None, kG, 1 << 30, None)
                        st8 = {}
                        for name, ms in eng.stage_times():
                            st8.setdefault(name, []).append(ms)
                        eng.profile(False)
                        share["stage_ms_8"] = {k: sum(v) / len(v) for k, v in st8.items()}
                share.update({"rays_per_step": R_frame // 8, "steps": kG, "ms_per_step": share["ms_per_step_8"],
                              "note": "one forward call of 1/G of the frame per step on ONE GPU (a new pose every step), G = 2, 4, 8: frame time / this = the "
                                      "scaling G ranks reach if the tile all-gather hides under the next frame; a projection, not a measurement on G GPUs"})
            if nb != 1 and args.chunk_rays >= R:
                # config 3's job is twelve WHOLE frames per step, their rays banded over the ranks: one GPU renders all of it (the engine splits
                # the call where the workspace does not fit), a rank of G renders 12 x 65536 / G rays per step — the G = 8 share is the line above
                full_R = Hc * Hc
                kF = 2
                frF = trajectory(kF + 1, dev, None, args.cameras == "host", Hc, nb)
                tF = torch.empty(nb, full_R, 5, device=dev)
                render_frame(model, frF[kF], z, tF, 1 << 30)
                eF = timed_loop(model, frF, z, tF, None, kF, 1 << 30, None) / kF
                del tF
                share = {"whole_job_ms_per_step": eF * 1e3, "whole_job_rays_per_step": nb * full_R}
                for G in (2, 4, 8):
                    kG, rs = max(4, args.steps // 2), full_R // G
                    g0 = (3 * G // 8) * rs
                    frG = trajectory(kG, dev, (g0, g0 + rs), args.cameras == "host", Hc, nb)
                    tG = torch.empty(nb, rs, 5, device=dev)
                    render_frame(model, frG[0], z, tG, 1 << 30)
                    eG = min(timed_loop(model, frG, z, tG, None, kG, 1 << 30, None) for _ in range(2)) / kG
                    share[f"projected_scaling_{G}"] = eF / eG
                    share[f"ms_per_step_{G}"] = eG * 1e3
                    del tG, frG
                share.update({"rays_per_step": nb * full_R // 8, "ms_per_step": share["ms_per_step_8"],
                              "note": "config 3 = twelve whole 256 x 256 frames per step; a rank of G renders the band [g R / G, (g + 1) R / G) of every "
                                      "scene in one call: whole job on ONE GPU / this = the scaling G ranks reach if the tile all-gather hides under the "
                                      "next step; a projection, not a measurement on G GPUs"})
                render_frame(model, frames[0], z, tile, args.chunk_rays)         # the line's own workspace back in place
            if args.cameras == "host" and nb == 1 and args.chunk_rays >= R:
                # the same frames with the WHOLE dict on the GPU (the reference scripts' call): host pose route, one download + sync per pose
                kp = max(4, min(args.steps, 10))
                frg = trajectory(kp, dev, None, False, Hc, nb)
                render_frame(model, frg[0], z, tile, 1 << 30)
                syncs = []
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for f in frg:
                    render_frame(model, f, z, tile, 1 << 30)
                    syncs.append(eng.last_pose_sync_ms)
                torch.cuda.synchronize()
                ep = (time.perf_counter() - t0) / kp * 1e3
                pose = {"default": "host algebra (the reference's torch.inverse on the CPU) wherever the cameras live",
                        "cameras_on_gpu_ms_per_step": ep, "cameras_on_host_ms_per_step": elapsed / args.steps * 1e3,
                        "cost_ms_per_step": ep - elapsed / args.steps * 1e3, "download_and_sync_ms": sum(syncs) / len(syncs), "steps": kp,
                        "note": "download_and_sync_ms is host time inside forward (it includes waiting for the previous frame's kernels, which "
                                "the host-camera loop overlaps with queueing the next frame)"}

        # the other multi-GPU arrangement: every rank renders whole frames of its own (replicas), tiles all-gathered
        per_rank = None
        if world > 1 and not args.no_extras and nb == 1:
            k2 = max(2, args.steps // 2)
            full = trajectory(k2, dev, None, args.cameras == "host", Hc, nb)
            tile2 = torch.empty(nb, R_frame, 5, device=dev)
            g2 = TileGather(world, nb * R_frame, 5, dev)
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
        rays_step = nb * R_frame
        rays_total = rays_step * args.steps
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
            samples = nb * V * min(args.chunk_rays, R) * Pc
            flop = 2.0 * samples * FUSED_MACS
            pmc = dict(prof.get("fused_samples", {})) if args.config == "c2" else {}
            part = samples / float(V * 256 * 256 * 64)           # the PMC passes profiled a whole c2-frame launch: scale to this launch
            for k in ("l1_bytes_per_launch", "bytes_per_launch"):
                if pmc.get(k) is not None:
                    pmc[k] = pmc[k] * part
            if part != 1.0 and pmc.get("source"):
                pmc["source"] += f", scaled by {part:g} to this launch's share of the frame"
            roof = {"bound": "mfma", "limiter": pmc.get("limiter"), "kernel": f"fused_kernel on {int(samples)} samples (geometry, 4-tap gather of the per-texel-projected pyramid on its lattice, e, key_map, query_embed, the first round's logits as a bilinear form; "
                                                                  "f16 matrix pipe, fp16 hi/lo split x3; since round 5 the launch also reduces the first attention "
                                                                  "round's value sums per 8-step group from L2 — about 4 % of its time, no matrix work, not counted in the flops)",
                    "achieved": flop / mean / 1e12, "peak": F16_MFMA_PEAK / 3 / 1e12, "unit": "TFLOP/s", "frac": flop / mean / (F16_MFMA_PEAK / 3),
                    "peak_note": "dense f16 MFMA peak 2500 / 3 products per fp32 term.  What keeps the kernel from it (limiter): MFMAs and ordinary "
                                 "vector instructions share a SIMD's issue, the A operands wait on LDS, the tap loads stall in the texture "
                                 "path's issue, and the chip clocks to its power budget under this mix (1.8-2.0 of 2.4 GHz): per compute unit the same "
                                 "stream reaches 0.47 of this peak when a quarter of the chip or less is running (DESIGN.md section 4.6 vi)",
                    "frac_of_fp32_pipe_peak": flop / mean / FP32_MFMA_PEAK,
                    "ta_busy": pmc.get("ta_busy"), "mfma_busy": pmc.get("mfma_busy"), "valu_share": pmc.get("valu_share"), "l1_bytes": pmc.get("l1_bytes_per_launch"),
                    "per_unit_unthrottled_frac": pmc.get("per_unit_unthrottled_frac"),
                    "traffic": pmc.get("bytes_per_launch"), "traffic_source": pmc.get("source"),
                    # which of these fields this run measured and which it copied from the committed counter passes
                    "live_fields": ["achieved", "frac", "frac_executed", "frac_of_fp32_pipe_peak", "launches", "ms_per_launch", "flop_per_launch", "flop_executed_per_launch"],
                    "static_fields": ["limiter", "ta_busy", "mfma_busy", "valu_share", "l1_bytes", "traffic", "per_unit_unthrottled_frac"],
                    "static": "profiles/traffic.json: rocprofv3 --pmc passes of this kernel at config c2 (separate passes by construction, not collected here; null for the other configs)",
                    "static_as_of": pmc.get("as_of"),      # date / build of the counter passes the static fields were copied from
                    "launches": len(lat), "ms_per_launch": mean * 1e3, "flop_per_launch": flop,
                    # what the kernel executes since round 6: key_map_2 and query_embed_2 folded into ONE 128 x 128 layer (the logit is a
                    # bilinear form of the two hidden vectors): 16 384 of the 440 320 algorithmic MACs per sample are not issued
                    "flop_executed_per_launch": 2.0 * samples * FUSED_MACS_EXECUTED,
                    "frac_executed": 2.0 * samples * FUSED_MACS_EXECUTED / mean / (F16_MFMA_PEAK / 3),
                    # the same matrix work without the partial-sum phase (first_round_ab): the figure comparable with rounds 3-4
                    "frac_without_partial_sums": None if not ab or "fused_samples" not in ab["rows_of_e"]["stage_ms"] else
                    flop / (ab["rows_of_e"]["stage_ms"]["fused_samples"] * 1e-3) / (F16_MFMA_PEAK / 3),
                    # the chip under this frame's mix, sampled beside the timed loop (null where the box offers no power interface)
                    "sclk_mhz_live": None if not power else power.get("mean_sclk_mhz"), "socket_w_live": None if not power else power.get("energy_mean_w", power.get("mean_w"))}
            roof["live_fields"] += ["sclk_mhz_live", "socket_w_live", "frac_without_partial_sums"]
        fr = prof.get("frame")
        hbm = None
        if fr and args.chunk_rays >= R and world == 1 and args.config == "c2":
            per_frame = fr["bytes_per_frame"]
            hbm = {"bytes_per_frame": per_frame, "achieved": per_frame / (elapsed / args.steps) / 1e12, "peak": 8.0, "unit": "TB/s",
                   "frac": per_frame / (elapsed / args.steps) / HBM_PEAK, "source": fr["source"],
                   "static": "bytes_per_frame comes from profiles/traffic.json (PMC passes, not collected in this run); the time is this run's",
                   "static_as_of": fr.get("as_of")}
        ref_flop = V * Pc * 2617728 + 791808
        calls = -(-R // args.chunk_rays)
        ms_step = elapsed / args.steps * 1e3
        line = {
            "metric": "rendered_rays_per_sec", "value": rays_total / elapsed, "unit": "rays/s",
            "frames_per_sec": args.steps / elapsed,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 (split-f16 x3 MFMA)", "data": "synthetic",
            "config": {"name": args.config, "builder_kept": args.config != "c2",
                       "workload": f"{what}, a new query pose every step (cameras on the {args.cameras}), "
                                   f"{calls} forward call(s) x {nb} scene(s) x {min(args.chunk_rays, R)} rays per rank",
                       "H": Hc, "samples_per_view": Pc, "scenes_per_step": nb,
                       "rays_per_step": rays_step, "rays_per_step_per_gpu": nb * R,
                       "parallelism": "one GPU" if world == 1 else f"one frame's rays banded over {world} ranks, all-gather of tiles over {'RCCL' if args.backend == 'nccl' else args.backend}"
                                                                      + (f" (all ranks on device {args.device}: a dry run of the code path, no scaling figure)" if args.device is not None else "")},
            "stage_ms": {k: sum(v) / len(v) for k, v in by_stage.items()},
            "stage_ms_per_rank": rank_stages,
            "roofline": roof,
            "hbm": hbm,
            "gather_stage": gs,
            # what the restructuring moved out of the timed region, and what it holds in memory
            "pair_setup_ms": setup_ms, "lattice_bytes": lattice_bytes, "workspace_bytes": workspace_bytes,
            "eval_mode": None if ev_ms is None else {
                "ms_per_step": ev_ms, "value": rays_step / (ev_ms * 1e-3), "unit": "rays/s", "ms_per_step_with_prefetch": ev_ms_pf,
                "note": "every frame brings a new stereo pair (eval_realestate10k.py:142-161): channel-last copies of the pyramid + car_project_maps "
                        "for every frame, inside the forward that first sees the pair; ms_per_step_with_prefetch: the same free-running loop "
                        "announcing the next pair one frame ahead (model.prefetch_pair: the set-up on a side stream) — it competes with the "
                        "running frame's kernels for the same compute units and gains only where the loop synchronises per item, as the eval "
                        "script does (DESIGN.md section 4.3); get_z excluded as in the headline figure"},
            "rank_share": share,
            "first_round_ab": ab,
            "power": power,
            "pose_route": pose,
            "frame_per_rank": None if per_rank is None else {
                "value": world * rays_step * per_rank[0] / per_rank[1], "unit": "rays/s", "steps": per_rank[0],
                "ms_per_step": per_rank[1] / per_rank[0] * 1e3, "note": "every rank renders whole frames of its own (replicas), tiles all-gathered"},
            # SURVEY.md §8(d) prices the path with the REFERENCE's arithmetic: V*P*2 617 728 + 791 808 FLOP per ray (335.9 MFLOP at
            # P = 64) against the fp32 matrix peak.  This implementation executes about a third of that (first point-MLP layer per
            # texel, value projection after the attention average) and runs it on the f16 pipe, hence a figure above 1.
            "reference_flops": {"flop_per_ray": ref_flop, "equivalent_tflops": rays_total / elapsed * ref_flop / 1e12,
                                "frac_of_fp32_mfma_peak": rays_total / elapsed * ref_flop / FP32_MFMA_PEAK / world},
            # the CPU leg runs on rank 0 of a single-GPU job only (it is a per-host figure and would stall the other ranks)
            "cpu_baseline": cpu_baseline(args.cpu_rays, Hc, Pc) if (args.cpu_rays > 0 and world == 1) else None,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
