"""-m gpu: the MI355X I3Res50 against the golden vectors captured from the reference and
against the CPU oracle (feature relative L2 <= 1e-3, BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from ted_spad_amd.synth import synth_clips, synth_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-3  # north_star: "features within 1e-3 relative L2 of the reference PyTorch CPU path"


@pytest.fixture(scope="module")
def net():
    from ted_spad_amd.i3res50 import I3Res50
    m = I3Res50(num_classes=102)
    sd = synth_state_dict(m.state_dict(), 0)
    # the golden weights were generated under the wrapper's key names (`i3d.` prefix)
    from ted_spad_amd.model_loaders import wrapper_i3d
    w = wrapper_i3d(num_classes=102)
    w.load_state_dict(synth_state_dict(w.state_dict(), 0), strict=True)
    return w.cuda().eval()


def test_cfg1_112_vs_golden(net, golden):
    x = synth_clips(0, 1, (3, 16, 112, 112), device="cuda")
    f = net.i3d.extract_features(x)
    assert f.shape == (1, 2048, 1, 1, 1) and f.dtype == torch.float32
    assert rel_l2(f.cpu().reshape(1, 2048), golden["i3res50_feat_112"]) < TOL


def test_224_vs_golden_per_clip(net, golden):
    x = synth_clips(0, 2, (3, 16, 224, 224), device="cuda")
    f = net.i3d.extract_features(x).cpu().reshape(2, 2048)
    for i in range(2):
        assert rel_l2(f[i], golden["i3res50_feat_224"][i]) < TOL


def test_stage_checksums_vs_golden(net, golden_meta):
    x = synth_clips(0, 2, (3, 16, 224, 224), device="cuda")
    taps = {}
    net.i3d._trunk(x, taps=taps)
    for name, (mean, l2) in golden_meta["i3res50_taps_224"].items():
        t = taps[name].buf.double()
        assert abs(float(t.norm()) - l2) < 2e-3 * l2, name
        assert abs(float(t.mean()) - mean) < 2e-3 * abs(mean) + 1e-4, name


def test_batch_invariance_and_determinism(net, monkeypatch):
    """With the built-in tile heuristic (tuner off) only configurations that sum K in the same order run, so a clip's
    feature is BIT-identical whatever batch it is in and from run to run. With the tuner on, the reassociating
    configurations (halo-direct 15/16, split-K stem 21: one f16 rounding step on ~0.1 % of a layer's outputs) may be
    picked for one batch size and not for another: the features then agree to 1e-4 rel-L2 (gate: 1e-3 vs the oracle)."""
    from ted_spad_amd import engine as E
    x = synth_clips(0, 5, (3, 16, 112, 112), device="cuda")
    tuned5 = net.i3d.extract_features(x).flatten(1)
    tuned1 = torch.cat([net.i3d.extract_features(x[i:i + 1]) for i in range(5)]).flatten(1)
    assert rel_l2(tuned5.cpu(), tuned1.cpu()) < 1e-4
    monkeypatch.setattr(E, "AUTOTUNE", False)
    f5 = net.i3d.extract_features(x)
    f1 = torch.cat([net.i3d.extract_features(x[i:i + 1]) for i in range(5)])
    assert torch.equal(f5, f1)
    assert torch.equal(f5, net.i3d.extract_features(x))


def test_bf16_mode_misses_the_gate_f16_meets_it(golden):
    """Documents the precision decision: same kernels, bf16 storage."""
    from ted_spad_amd.model_loaders import wrapper_i3d
    w = wrapper_i3d(num_classes=102, dtype="bf16")
    w.load_state_dict(synth_state_dict(w.state_dict(), 0), strict=True)
    w = w.cuda().eval()
    x = synth_clips(0, 1, (3, 16, 112, 112), device="cuda")
    r = rel_l2(w.i3d.extract_features(x).cpu().reshape(1, 2048), golden["i3res50_feat_112"])
    assert 1e-3 < r < 8e-3
