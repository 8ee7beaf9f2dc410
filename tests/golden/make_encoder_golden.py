"""Generate the ``get_z`` fixtures by running the *reference's own* ``CrossAttentionRenderer.get_z`` (models.py:148-188) with its
own multi-view DPT-hybrid encoder (midas/dpt_depth.py, midas/vit.py, midas/blocks.py, vit_models.py) in this container.

    python tests/golden/make_encoder_golden.py                  (timm restated by timm_stub.py: what this container can do)
    python tests/golden/make_encoder_golden.py --timm           (on a machine WITH timm == 0.5.4: the reference on the real package;
                                                                 compares with the committed fixtures, --write replaces them)

The reference builds the encoder's ResNetV2 trunk and transformer blocks from timm 0.5.4, which is not installed here:
``timm_stub.py`` restates those layers (see its header) so that the reference's modules import and run.  Everything else — image
normalisation, the relative-pose embedding, token concatenation across views, position-embedding resize, read-out projection,
re-assembly, RefineNet fusion, ``conv_map`` and the order of the returned levels — is the reference's code executing.  Writes
``tests/golden/getz_<variant>.npz``: the parameter name -> shape table, strided samples and whole-tensor statistics of the three
pyramid levels.  Only data is written; the weights and images are regenerated from seeds (``encoder_cases.py``)."""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

USE_TIMM = "--timm" in sys.argv
if USE_TIMM:
    # the pin SURVEY 8(f2) still lacks: the timm-defined layers (ResNetV2 stem / stages, ViT blocks, hybrid embedding) from timm itself
    try:
        import timm                                    # noqa: E402
    except ImportError:
        sys.exit("make_encoder_golden.py --timm: timm is not installed here (pip install timm==0.5.4 — the version the reference pins, requirements.txt)")
    if timm.__version__ != "0.5.4":
        sys.exit(f"make_encoder_golden.py --timm: timm {timm.__version__} found, the reference pins 0.5.4 (its ResNetV2 / hybrid-ViT definitions changed later)")
else:
    import timm_stub                                   # noqa: E402
    timm_stub.install()                                # before anything of the reference is imported
import encoder_cases as EC                             # noqa: E402
import ref_import                                      # noqa: E402


def main():
    ref = ref_import.load_reference(real_encoder=True)
    from cross_attention_renderer_amd.models import CrossAttentionRenderer
    for variant, kw in EC.VARIANTS.items():
        torch.manual_seed(0)
        model = ref.models.CrossAttentionRenderer(model="midas_vit", n_view=2, **kw).eval()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        ours = CrossAttentionRenderer(model="midas_vit", n_view=2, **kw).eval()
        mine = {k: tuple(v.shape) for k, v in ours.state_dict().items()}
        assert mine == shapes, f"state_dict tables differ: {sorted(set(mine.items()) ^ set(shapes.items()))[:10]}"
        sd = EC.seeded_weights(shapes)
        model.load_state_dict(sd, strict=True)
        inp = EC.context_pair()
        # The fixture holds the reference's result in float64: with seeded (untrained) weights the 16 weight-standardised
        # bottlenecks and 12 transformer blocks amplify fp32 summation-order noise to ~1e-3 of a level's magnitude, which would
        # blur the pin; in float64 two correct implementations agree to ~1e-11.
        model64 = model.double()
        inp64 = {k: {kk: vv.double() for kk, vv in v.items()} for k, v in inp.items()}
        t0 = time.time()
        with torch.no_grad():
            z = model64.get_z(inp64)
        t_ref = time.time() - t0
        assert (model64.H, model64.W) == (EC.H, EC.H)
        ours.load_state_dict(sd, strict=True)
        with torch.no_grad():
            z32 = ours.get_z(inp)
            zo = ours.double().get_z(inp64)
        worst = [((a - b_).abs() / b_.abs().clamp_min(1.0)).max().item() for a, b_ in zip(zo, z)]
        noise = [((a.double() - b_).abs().max() / b_.pow(2).mean().sqrt()).item() for a, b_ in zip(z32, z)]
        print(f"{variant:14s} reference (fp64) {t_ref:5.1f}s  this repo vs reference, fp64: max rel err per level {['%.1e' % w for w in worst]}  "
              f"fp32 run: max err / rms {['%.1e' % w for w in noise]}  shapes {[tuple(t.shape) for t in z]}")
        assert max(worst) < 1e-8, "get_z does not reproduce the reference"
        names = sorted(shapes)
        if USE_TIMM:
            # the committed fixture was made on timm_stub.py: how far is the real package from it?
            old = np.load(EC.fixture_path(variant))
            diff = [float(np.abs(old[f"z{i}"] - s_).max() / max(np.abs(s_).max(), 1e-30)) for i, s_ in enumerate(EC.sample(z))]
            same_table = list(old["names"]) == names and list(old["shapes"]) == [str(shapes[n]) for n in names]
            print(f"{variant:14s} timm {timm.__version__} vs the committed (stub-made) fixture: max |diff| / max per level {['%.1e' % d for d in diff]}, "
                  f"name / shape table {'identical' if same_table else 'DIFFERENT'}")
            if "--write" not in sys.argv:
                continue
        np.savez_compressed(EC.fixture_path(variant), names=np.asarray(names), shapes=np.asarray([str(shapes[n]) for n in names]),
                            stats=EC.stats(z), **{f"z{i}": s_ for i, s_ in enumerate(EC.sample(z))})


if __name__ == "__main__":
    main()
