"""Import the *reference* hot path (read-only, this container only) behind module stubs.

This file is test infrastructure.  It is used by ``make_golden.py`` (and by the optional
``tests/test_oracle_vs_reference.py``, which is skipped when ``/root/reference`` is absent) to
  (1) validate ``oracle/car_oracle.py`` against the real reference, and
  (2) emit the golden vectors committed under ``tests/golden/*.npz``.
Nothing here is imported by the product package, by ``bench.py`` or by any ``-m gpu`` test, and no
reference source travels: only the arrays written by ``make_golden.py`` do.

The reference needs a handful of third-party modules that are not installed in this image
(jaxtyping, timm, torchvision, cv2, matplotlib).  None of them is touched by
``CrossAttentionRenderer.forward(input, z=z)`` (models.py:190-626), so inert stand-ins are injected
into ``sys.modules`` for the duration of the import.  The two hard-coded ``.cuda()`` calls on the
path (geometry.py:320, 398) are neutralised by making ``Tensor.cuda`` the identity.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("CAR_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models.py"))


class _Subscriptable:
    """Stands in for jaxtyping's Float/Int64/Bool: ``Float[Tensor, "a b"]`` -> Tensor."""

    def __class_getitem__(cls, item):
        return torch.Tensor


class _Anything(types.ModuleType):
    """A module whose every attribute is another permissive stub (callable, subscriptable)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _AnythingCallable(f"{self.__name__}.{name}")
        setattr(self, name, child)
        return child


class _AnythingCallable(_Anything):
    def __call__(self, *a, **k):
        return _AnythingCallable(self.__name__ + "()")

    def __getitem__(self, item):
        return self

    def __mro_entries__(self, bases):  # allows `class X(stub.Something):`
        return (object,)


def _install_stubs() -> None:
    jt = types.ModuleType("jaxtyping")
    for n in ("Float", "Int64", "Bool", "Int", "Shaped"):
        setattr(jt, n, _Subscriptable)
    sys.modules.setdefault("jaxtyping", jt)
    for name in (
        "timm", "timm.models", "timm.models.layers", "timm.models.layers.std_conv",
        "timm.models.vision_transformer", "timm.models.vision_transformer_hybrid",
        "timm.models.helpers", "torchvision", "torchvision.transforms", "cv2",
        "matplotlib", "matplotlib.colors", "matplotlib.pyplot",
    ):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _Anything(name)
    # parent/child links so `import a.b.c` style attribute access works
    for name, mod in list(sys.modules.items()):
        if isinstance(mod, _Anything) and "." in name:
            parent, _, child = name.rpartition(".")
            if parent in sys.modules and isinstance(sys.modules[parent], _Anything):
                setattr(sys.modules[parent], child, mod)


class _DummyEncoder(torch.nn.Module):
    """Holder for ``self.encoder``: ``forward(input, z=z)`` never calls the encoder."""

    def __init__(self, *a, **k):
        super().__init__()
        self.pretrained = _AnythingCallable("pretrained")

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("the reference encoder is out of scope for the golden vectors")


_REF = None


def load_reference(real_encoder: bool = False):
    """Returns the reference's ``models``, ``geometry``, ``epipolar`` modules (cached).  ``real_encoder``: keep the reference's own
    DPTDepthModel (needs ``timm_stub.install()`` beforehand) instead of the inert holder used for the render-forward fixtures."""
    global _REF
    if _REF is not None:
        return _REF
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _install_stubs()
    torch.Tensor.cuda = lambda self, *a, **k: self  # geometry.py:320, 398
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        geometry = importlib.import_module("geometry")
        epipolar = importlib.import_module("epipolar")
        models = importlib.import_module("models")
    finally:
        sys.path.remove(REFERENCE_ROOT)
    if not real_encoder:
        models.dpt_depth.DPTDepthModel = _DummyEncoder
    models.UNetEncoder = _DummyEncoder
    _REF = types.SimpleNamespace(models=models, geometry=geometry, epipolar=epipolar)
    return _REF


def build_reference_model(n_view=2, npoints=64, model="midas_vit", H=None, **kw):
    """Instantiates the reference renderer on CPU in eval mode with ``H``/``W`` preset
    (normally set by ``get_z``, models.py:162)."""
    ref = load_reference()
    m = ref.models.CrossAttentionRenderer(model=model, n_view=n_view, npoints=npoints, **kw)
    m.eval()
    if H is not None:
        m.H = m.W = H
    return m
