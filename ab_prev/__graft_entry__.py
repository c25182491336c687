"""Driver entry points: build() compiles the HIP library for gfx950 (cross-compiles without a
GPU); smoke() runs one small invocation of the hot path on cuda:0 and checks it against the
CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build() -> None:
    from ted_spad_amd import build as b
    lib = b.build_library(force=False)
    assert os.path.exists(lib)
    # the checker: oracle/ is pure Python (torch CPU + numpy); compile-check it by importing
    import oracle.conv_ref, oracle.extract_ref, oracle.i3res50_ref, oracle.inception_i3d_ref, oracle.losses_ref, oracle.unet_ref  # noqa: F401,E401
    import ted_spad_amd  # noqa: F401
    from ted_spad_amd import _lib
    _lib.lib()  # loads the .so and resolves every symbol include/tedspad_hip.h declares
    print("built", lib)


def smoke() -> None:
    import torch
    from oracle import i3res50_ref
    from ted_spad_amd import _lib
    from ted_spad_amd.model_loaders import load_ft_model
    from ted_spad_amd.synth import synth_clips, synth_state_dict
    _lib.lib()  # fail loudly if the HIP extension is missing
    assert torch.cuda.is_available(), "smoke() needs cuda:0"
    torch.cuda.set_device(0)
    ft = load_ft_model("largei3d", num_classes=102)
    sd = synth_state_dict(ft.state_dict(), 0)
    ft.load_state_dict(sd, strict=True)
    ft = ft.cuda().eval()
    x = synth_clips(0, 2, (3, 16, 112, 112))
    f = ft.i3d.extract_features(x.cuda()).cpu()
    with torch.no_grad():
        ref = i3res50_ref.extract_features(x, {k[4:]: v for k, v in sd.items() if k.startswith("i3d.")})
    rel = float((f - ref).norm() / ref.norm())
    assert f.shape == (2, 2048, 1, 1, 1) and rel < 1e-3, rel
    print("smoke ok: I3Res50.extract_features on cuda:0, rel-L2 vs CPU oracle = %.3e" % rel)


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
