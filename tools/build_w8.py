"""Builds compile-time variants of the 8-wave candidate kernel (tools/probes/car_fused_w8.hip), one library each:
tools/_dev/libw8_<n>.so exporting car_fused_samples_ws (tools/bench_fused.py variant 200 + n).  Development only.
Usage: python tools/build_w8.py 0 1 2 3 ["-DEXTRA ..."]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__ as ge
    ge.build()
    dev = os.path.join(ROOT, "tools", "_dev")
    os.makedirs(dev, exist_ok=True)
    src = os.path.join(ROOT, "tools", "probes", "car_fused_w8.hip")
    objs = [os.path.join(ge.CSRC, "_obj", u.replace(".hip", ".o")) for u in ge.UNITS]
    for v in sys.argv[1:]:
        if v.startswith("-"):
            continue
        extra = [a for a in sys.argv[1:] if a.startswith("-")]
        obj = os.path.join(dev, f"w8_{v}.o")
        r = subprocess.run([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCAR_W8_VAR={v}", *extra, "-c", src, "-o", obj, "-I", os.path.join(ROOT, "include"),
                            "-I", ge.CSRC, *ge.UNITS["car_fused.hip"], "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        if r.returncode:
            print(r.stderr)
            raise SystemExit(1)
        info = [ln.split("remark:")[1].strip().split(" [")[0] for ln in r.stderr.splitlines() if "remark:" in ln and ("Spill" in ln or "VGPRs:" in ln or "Scratch" in ln)]
        subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(dev, f"libw8_{v}.so"), obj, *objs])
        print(f"variant {v}: " + "; ".join(i for i in info if "Function" not in i))


if __name__ == "__main__":
    main()
