"""Debug build of the HIP library with -DCAR_BOUNDS (csrc/car_common.h): every LDS-DMA / buffer-load / row-load helper of the kernels
compares the range it is about to read with the extent its launcher passed and TRAPS outside it.  Never part of the product
(``__graft_entry__.build()`` does not know this file).  Output: tools/_dev/libcar_bounds.so — run the out-of-bounds harness on it with
CAR_OOB_FULL_LIB=tools/_dev/libcar_bounds.so python tests/oob_runner.py <family>   (tools/oob_selfcheck.sh does all of it).
``--reintroduce-0d74f26``: also tools/_dev/liboldlin16_bounds.so — today's car_linear16.hip with the bug commit 0d74f26 fixed put back
(three LDS-DMA pieces issued whatever the column group's width: a narrow group's third piece copies from behind its chunk)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DEV = os.path.join(ROOT, "tools", "_dev")


def build(reintroduce: bool = False):
    import __graft_entry__ as ge
    os.makedirs(os.path.join(DEV, "bounds_obj"), exist_ok=True)
    headers = [os.path.join(ge.CSRC, f) for f in os.listdir(ge.CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "car_hip.h")]
    objs, procs = [], []
    for unit, extra in ge.UNITS.items():
        src, obj = os.path.join(ge.CSRC, unit), os.path.join(DEV, "bounds_obj", unit.replace(".hip", ".o"))
        objs.append(obj)
        if ge._stale(obj, [src, *headers, os.path.abspath(__file__)]):
            procs.append(subprocess.Popen([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DCAR_BOUNDS", "-c", src, "-o", obj,
                                           "-I", os.path.join(ROOT, "include"), "-I", ge.CSRC, *extra]))
    if any(p.wait() != 0 for p in procs):
        sys.exit("build_bounds: hipcc failed")
    lib = os.path.join(DEV, "libcar_bounds.so")
    if ge._stale(lib, objs):
        subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs])
    out = [lib]
    if reintroduce:
        cur = open(os.path.join(ge.CSRC, "car_linear16.hip")).read()
        old = cur.replace("p < pieces_of(NT); ++p", "p < kPieces; ++p").replace("if (qs < pieces_of(NT)) stream_issue_piece", "if (qs < kPieces) stream_issue_piece")
        old = old.replace('static_assert(kWaves * pieces_of(NT) <= 3 * 2 * NT,', 'static_assert(true || kWaves * pieces_of(NT) <= 3 * 2 * NT,')
        assert old != cur and old.count("kPieces; ++p") >= 1, "car_linear16.hip no longer has the lines commit 0d74f26 changed"
        srcf = os.path.join(DEV, "linear16_with_0d74f26_reverted.hip")
        open(srcf, "w").write(old)
        for tag, flags in (("", []), ("_bounds", ["-DCAR_BOUNDS"])):
            so = os.path.join(DEV, f"liboldlin16{tag}.so")
            subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", *flags, "-I", os.path.join(ROOT, "include"),
                                   "-I", ge.CSRC, srcf, os.path.join(ge.CSRC, "car_api.hip"), "-o", so])
            out.append(so)
    return out


if __name__ == "__main__":
    print("\n".join(build("--reintroduce-0d74f26" in sys.argv)))
