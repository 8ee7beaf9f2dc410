"""The C ABI from a C host: examples/render_frame.c is compiled with plain gcc against include/car_hip.h + libcar_hip.so
(no Python, no torch in the process) and, on a GPU box, renders a frame through car_plan_build / car_project_maps /
car_render_forward."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cross_attention_renderer_amd")
EXE = os.path.join(ROOT, "examples", "_build", "render_frame")


def _compile():
    import __graft_entry__ as ge
    ge.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c11", "-O2", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "examples", "render_frame.c"),
           f"-I{rocm}/include", f"-I{ROOT}/include", f"-L{PKG}", f"-L{rocm}/lib", "-lcar_hip", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{PKG}", f"-Wl,-rpath,{rocm}/lib", "-o", EXE]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_c_host_compiles_and_links():
    assert os.path.exists(_compile())


@pytest.mark.gpu
@pytest.mark.parametrize("H,P", [(64, 32), (256, 64)])
def test_c_host_renders_a_frame(H, P):
    exe = _compile()
    r = subprocess.run([exe, str(H), str(P)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"mean rgb (-?[0-9.]+) \| valid ([0-9.]+) \| mean depth ([0-9.]+) \| bad values (\d+)", r.stdout)
    assert m, r.stdout
    assert int(m.group(4)) == 0 and 0.5 < float(m.group(2)) <= 1.0 and abs(float(m.group(1))) < 5.0
    # same inputs, same kernels: a second process reproduces the line bit for bit (everything after the timing)
    r2 = subprocess.run([exe, str(H), str(P)], capture_output=True, text=True, timeout=600)
    assert r2.stdout.split("| mean rgb")[1] == r.stdout.split("| mean rgb")[1]
