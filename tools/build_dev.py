"""Development build of the HIP library with the timing-only ablation variants of the fused per-sample kernel compiled in
(-DCAR_ABLATION: ``car_fused_samples_ablate``).  Never part of the product: ``__graft_entry__.build()`` does not know this file,
the release library has no such symbol and reads no environment variable.  Output: tools/_dev/libcar_dev.so."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV_DIR = os.path.join(ROOT, "tools", "_dev")
DEV_LIB = os.path.join(DEV_DIR, "libcar_dev.so")


def build_dev() -> str:
    """CAR_DEV_FLAGS (environment of the BUILD, e.g. -DCAR_FUSED_WAVES=8): extra flags for the development copy of car_fused.hip."""
    import __graft_entry__ as ge
    ge.build()
    os.makedirs(DEV_DIR, exist_ok=True)
    extra = os.environ.get("CAR_DEV_FLAGS", "").split()
    stamp = os.path.join(DEV_DIR, "flags.txt")
    fresh = os.path.exists(stamp) and open(stamp).read() == " ".join(extra) + (" unit=" + os.environ["CAR_DEV_UNIT"] if os.environ.get("CAR_DEV_UNIT") else "")
    headers = [os.path.join(ge.CSRC, f) for f in os.listdir(ge.CSRC) if f.endswith(".h")]
    dev_units = ("car_fused.hip", "car_gather.hip")            # the units that carry -DCAR_ABLATION variants
    objs = [os.path.join(ge.CSRC, "_obj", u.replace(".hip", ".o")) for u in ge.UNITS if u not in dev_units]
    for unit in dev_units:
        obj = os.path.join(DEV_DIR, unit.replace(".hip", "_dev.o"))
        src = os.path.join(ge.CSRC, unit)
        if ge._stale(obj, [src] + headers) or not fresh:
            subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DCAR_ABLATION", *extra, "-c", src, "-o", obj,
                                   "-I", os.path.join(ROOT, "include"), "-I", ge.CSRC, *ge.UNITS[unit]])
        objs.append(obj)
    # development-only units (kernels under evaluation, tools/probes/): CAR_DEV_UNIT=car_fused_w32.hip
    unit = os.environ.get("CAR_DEV_UNIT")
    if unit:
        src = os.path.join(ROOT, "tools", "probes", unit)
        obj = os.path.join(DEV_DIR, unit.replace(".hip", "_dev.o"))
        if ge._stale(obj, [src] + headers) or not fresh:
            subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, "-c", src, "-o", obj,
                                   "-I", os.path.join(ROOT, "include"), "-I", ge.CSRC, *ge.UNITS["car_fused.hip"]])
        objs.append(obj)
    open(stamp, "w").write(" ".join(extra) + (" unit=" + unit if unit else ""))
    if ge._stale(DEV_LIB, objs):
        subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DEV_LIB, *objs])
    return DEV_LIB


if __name__ == "__main__":
    print(build_dev())
