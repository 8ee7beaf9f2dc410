"""Non-fatal diagnostic sweep on a GPU box: runs every golden case through the HIP path and prints an error table per
output/stage against the oracle and the reference fixture.  Unlike pytest it never stops at the first failure, so one
gpurun call localises every discrepancy.  Usage: python tools/gpu_selfcheck.py [case ...]"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import torch  # noqa: E402

import cases as C  # noqa: E402
from hip_harness import err_stats, run_case  # noqa: E402


def fmt(e):
    return f"max={e['max']:.2e} f>1e-4={e['f1e-4']:.1e} f>1e-3={e['f1e-3']:.1e}"


def main():
    names = sys.argv[1:] or list(C.CASES)
    print("device:", torch.cuda.get_device_name(0))
    for name in names:
        t0 = time.time()
        try:
            c, fx, ora, out = run_case(name)
        except Exception:
            print(f"[{name}] EXCEPTION\n{traceback.format_exc()}")
            continue
        print(f"[{name}] ({time.time() - t0:.1f}s)")
        st, hs = ora["stages"], out["stages"]
        rows = [("lf", hs["rays"][..., 0:6], st["lf"]), ("pixel_val", out["pixel_val"], st["pixel_val"]),
                ("pt", hs["pt"], st["pt"]), ("local_coords", hs["local_coords"], st["local_coords"]),
                ("interp_val", hs["interp_val"], st["interp_val"]), ("at_wt", out["at_wt"], ora["at_wt"])]
        if hs.get("at_wt2") is not None and "at_wt2" in st:
            rows.append(("at_wt2", hs["at_wt2"], st["at_wt2"]))
        b, V = c["b"], c["n_view"]
        zf = st["z_final"]
        rows += [("z_final", hs.get("z_final"), zf.reshape(b, V, *zf.shape[1:])[:, 0]),
                 ("depth_ray", out["depth_ray"], ora["depth_ray"]), ("rgb", out["rgb"], ora["rgb"]),
                 ("rgb vs REF", out["rgb"], fx["out_rgb"]), ("depth vs REF", out["depth_ray"], fx["out_depth_ray"]),
                 ("at_wt vs REF", out["at_wt"], fx["out_at_wt"])]
        for label, a, b_ in rows:
            if a is None or b_ is None:
                continue
            try:
                print(f"    {label:14s} {fmt(err_stats(a, b_))}")
            except Exception as ex:
                print(f"    {label:14s} ERROR {ex}")
        vm = (out["valid_mask"] == ora["valid_mask"]).float().mean().item()
        am = (out["at_wt_max"] == ora["at_wt_max"]).float().mean().item()
        print(f"    valid_mask agreement {vm:.4f}   at_wt_max agreement {am:.4f}")


if __name__ == "__main__":
    main()
