"""Import the *reference* (read-only at /root/reference) in the authoring container.

Used ONLY by `make_golden.py` and by the optional `-m "not gpu"` cross-checks that skip
when /root/reference is absent (it never exists on the GPU box). Nothing here is shipped
as product code; no reference source is copied -- the modules are imported in place.

Recipe: SURVEY.md Appendix C. Four third-party imports of the reference are absent from
this image and are replaced by empty `sys.modules` stubs (they are not on the hot path).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TEDSPAD_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "aux_code"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules.setdefault(name, m)
    return sys.modules[name]


def import_reference():
    """Returns (model_loaders module, NTXentLoss class)."""
    if not reference_available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    tk = _stub("tkinter")
    tix = _stub("tkinter.tix", Tree=object)
    tk.tix = tix
    _stub("segmentation_models_pytorch", UnetPlusPlus=None)
    tv = _stub("torchvision")
    tvm = _stub("torchvision.models", resnet50=None, ResNet50_Weights=None)
    tvv = _stub("torchvision.models.video", r3d_18=None, mvit_v2_s=None)
    tv.models = tvm
    tvm.video = tvv
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import aux_code.model_loaders as ml  # noqa: E402
    from aux_code.nt_xent_original import NTXentLoss  # noqa: E402
    return ml, NTXentLoss
