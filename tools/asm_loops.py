"""Lists the loops of a kernel's assembly (hipcc -S --cuda-device-only) with their MFMA / scratch / LDS / vector-memory instruction counts.
Usage: python tools/asm_loops.py <source.hip> [-Dflags ...]"""
import os
import re
import subprocess
import sys

VALU = r"^\s+v_(?!mfma)"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__ as ge
    src = sys.argv[1]
    out = "/tmp/asm_loops.s"
    subprocess.check_call([ge._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", src, "-o", out, "-I", os.path.join(ROOT, "include"),
                           "-I", ge.CSRC, *ge.UNITS["car_fused.hip"], *sys.argv[2:]])
    lines = open(out).read().splitlines()
    labels = {m.group(1): i for i, ln in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", ln))}
    for i, ln in enumerate(lines):
        m = re.match(r"\s+s_cbranch_\w+ (\.LBB\d+_\d+)", ln)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            cnt = lambda pat: sum(1 for b in body if re.search(pat, b))
            if cnt("mfma") >= 8:
                print(f"loop {m.group(1)} lines {labels[m.group(1)]}-{i}: {len(body)} lines, mfma {cnt('v_mfma')}, scratch {cnt('scratch_')}, ds_read {cnt('ds_read')}, ds_write {cnt('ds_write')}, "
                      f"buffer_load {cnt('buffer_load')}, global {cnt('global_')}, valu {cnt(VALU)}, s_waitcnt {cnt('s_waitcnt')}, s_nop {cnt('s_nop')}")


if __name__ == "__main__":
    main()
