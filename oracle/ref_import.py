"""Import helper for the REAL reference modules (authoring container only).

TEST INFRASTRUCTURE - never imported by the product path.  `/root/reference` does not exist on the GPU
box; this module is used only by `oracle/make_golden.py` (fixture generation) and by CPU tests that are
skipped when the reference tree is absent.

Per SURVEY.md section 8(c): the reference imports here once two absent third-party modules are stubbed
in `sys.modules` (torchaudio; transformers.utils.model_parallel_utils, removed in transformers 5.x while
the reference pins 4.36.2, setup.py:49).  Nothing from the reference is copied.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REF_ROOT = os.environ.get("ITTS_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "indextts"))


def _stub(name: str, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install():
    """Make `import indextts...` resolve to the reference tree."""
    if not available():
        raise RuntimeError("reference tree not present")
    try:
        import torchaudio  # noqa: F401
    except Exception:
        ta = _stub("torchaudio")
        ta.transforms = _stub("torchaudio.transforms")
        ta.functional = _stub("torchaudio.functional")
    try:
        import transformers.utils.model_parallel_utils  # noqa: F401
    except Exception:
        _stub("transformers.utils.model_parallel_utils",
              assert_device_map=lambda *a, **k: None, get_device_map=lambda *a, **k: None)
    # the product tree ships its own drop-in `indextts` package: make sure the REFERENCE one wins here
    for k in [k for k in sys.modules if k == "indextts" or k.startswith("indextts.")]:
        del sys.modules[k]
    if REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
    sys.path.insert(0, REF_ROOT)


def uninstall():
    for k in [k for k in sys.modules if k == "indextts" or k.startswith("indextts.")]:
        del sys.modules[k]
    if REF_ROOT in sys.path:
        sys.path.remove(REF_ROOT)
