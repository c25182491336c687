"""Headline benchmark (BASELINE.json): clips/sec of 16x224^2 I3D feature extraction on
1/2/4/8 MI355X, with the feature relative-L2 against the fp32 CPU path and the CPU
extractor timed beside it.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N ranks as a child torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one synthetic UCF-Crime-length video:
225 clip times x 10 crops = 2250 clip-forwards of (3,16,224,224) fp32 (cfg2), inputs
resident in HBM before the timed region. N > 1 (cfg4): that ONE video's clip times are split over the ranks
(`"scaling": "strong"`, `value`), features all-gathered over RCCL and copied to the host (the .npy rows) inside
the timed region; the weak reading (a whole video per rank) is measured right after it and reported as
`weak_scaling` in the same line. Prints ONE JSON line on rank 0. At N = 1 the line also carries `train_cfg3`: the anonymizer training
iteration of BASELINE.json configs[2] (UNet + I3Res50 + CE/triplet, batch 8 x 48 x 112^2), timed per phase after the
headline measurement (`--no-train` skips it, `--train` runs only it).

`--dry-run-cpu` rehearses the N > 1 control flow (process group, sharding, barrier, MAX-reduce of the time, rank-0-only JSON) on the
CPU with the gloo backend and a stub feature extractor: no kernel runs, the numbers mean nothing (tests/test_bench_dry_run.py).
"""
import argparse
import contextlib
import glob
import hashlib
import io
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_CLIP = {"largei3d": 32.829145088, "i3d": 55.575138304}  # BASELINE.md §2 (conv MACs x 2)
MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16/f16 MFMA (MI355X_MICROARCH.md)
TRAIN_TFLOP = {"phase1": 18.040, "phase2": 6.480}                 # BASELINE.md §2, cfg3 (fb branch excluded)
# the privacy branch of train_anonymizer.py:80-84,153-157 on the two VISPR views (2 x 12 images of 3 x 224 x 224, params_anonymization.py:29), counted like
# BASELINE.md §2 (scripts/count_macs.py: hooks on the oracle's functional convs): UNet 61.232 GFLOP and fb (ResNet-50 + MLP) 8.183 GFLOP per image.
# phase 1: fa trained (fwd + dgrad + wgrad = 3 x) under a frozen fb (fwd + dgrad = 2 x); phase 2: fa forward only, fb trained (3 x)
FB_TFLOP = {"phase1": 24 * (3 * 61.232 + 2 * 8.183) * 1e-3, "phase2": 24 * (61.232 + 3 * 8.183) * 1e-3}
UNETPP_GFLOP_PER_FRAME = 24.626        # smp UnetPlusPlus(resnet18), one 3 x 224 x 224 frame: 12 312 788 992 MACs (scripts/count_macs.py on oracle/unetpp_ref.py)


def kernel_sources_sha() -> str:
    """Identity of the kernels a PMC traffic figure belongs to: csrc/ + the launch-sequence files."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "ted_spad_amd", "csrc", "*.h*"))) + [os.path.join(ROOT, "ted_spad_amd", f) for f in ("engine.py", "i3res50.py", "inception_i3d.py")]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def train_sources_sha() -> str:
    """kernel_sources_sha() + the training-side launch sequences."""
    h = hashlib.sha256(kernel_sources_sha().encode())
    for f in ("train_engine.py", "train_nets.py", "train_step.py", "unet.py", "resnet50.py"):
        h.update(open(os.path.join(ROOT, "ted_spad_amd", f), "rb").read())
    return h.hexdigest()[:16]


def bench_train(dev, steps=10, warmup=45, hw=112, with_fb=True):
    """cfg3 on this GPU: ms per phase-1 / phase-2 iteration, algorithmic TFLOP/s.
    with_fb: the WHOLE train_epoch body (train_anonymizer.py:71-198): the privacy branch fb (ResNet-50 + MLP) on the two VISPR views 2 x (12, 3, 224, 224) and the
    NT-Xent term inside both phases (phase 1: loss_fa = -NTXent + 0.7 (CE + 0.1 triplet), fa trained through the frozen fb; phase 2: fb trained on NT-Xent, ft on the
    utility loss), beside the utility-only iteration of the earlier rounds (`*_utility_only`).
    hw = 224: the per-rank batch of cfg5 (8 x 48 x 224^2; its FLOP are 4x cfg3's, every conv scales with the pixels)."""
    from ted_spad_amd import engine as E
    from ted_spad_amd.model_loaders import load_fa_model, load_fb_model, load_ft_model
    from ted_spad_amd.synth import synth_state_dict, synth_tensor, synth_train_video
    from ted_spad_amd.train_step import AnonymizerTrainStep
    with contextlib.redirect_stdout(io.StringIO()):
        fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
        fb = load_fb_model(arch="r50", ssl=True, pretrained=False) if with_fb else None
    fa.load_state_dict(synth_state_dict(fa.state_dict(), 0))
    ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
    fa, ft = fa.to(dev), ft.to(dev)
    video = synth_train_video(0, "bench_train", (8, 48, 3, hw, hw), device=dev)
    fl = (hw / 112.0) ** 2
    labels = torch.randint(1, 102, (8,), device=dev)
    out = {"config": "%s train_anonymizer.py iteration (train_epoch :71-198, both phases): UNet anonymizer + I3Res50 + CE + 0.1 x triplet on a batch of 8 x 48 x %d^2%s, f16 "
                     "activations / fp32 accumulate, Adam steps included" % (
                         "cfg3" if hw == 112 else "cfg5 per-rank", hw,
                         ", privacy branch fb (ResNet-50 + MLP) + NT-Xent (T = 0.1) on the VISPR views 2 x (12, 3, 224, 224)" if with_fb else ", fb branch excluded"), "steps": steps}
    # ---- the utility-only iteration (the figure of rounds 1-3; what profiles/traffic_train_cfg3.json was measured on) -----------------------------------------
    step = AnonymizerTrainStep(fa, ft)
    for name, fn in (("phase1", step.step_fa), ("phase2", step.step_ft)):
        for i in range(4 * warmup):              # until the tile tuner has settled every conv geometry of this phase
            if i >= warmup and not E.tuning_pending():
                break
            fn(video, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            r = fn(video, labels)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        sfx = "_utility_only" if with_fb else ""
        out[name + "_ms" + sfx] = round(ms, 3)
        out[name + "_tflops" + sfx] = round(fl * TRAIN_TFLOP[name] / ms * 1e3, 1)
        out[name + "_mfma_frac" + sfx] = round(fl * TRAIN_TFLOP[name] / ms * 1e3 / MFMA_PEAK_TFLOPS, 4)
        out[name + "_loss" + sfx] = round(float(r["loss_ft"]), 5)
    if with_fb:
        # the alternating loop of rounds 1-3 (both phases per batch, utility terms only): the figure earlier rounds quote as `iteration_ms`
        for _ in range(max(3, warmup // 8)):
            step.step_fa(video, labels); step.step_ft(video, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step.step_fa(video, labels); step.step_ft(video, labels)
        torch.cuda.synchronize()
        out["iteration_ms_utility_only"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
        # ---- the whole train_epoch body: fb + NT-Xent in both phases -----------------------------------------------------------------------------------------
        fb.load_state_dict(synth_state_dict(fb.state_dict(), 0))
        fb = fb.to(dev)
        views = [synth_tensor(0, "vispr_view%d" % v, (12, 3, 224, 224), device=dev) for v in range(2)]
        del step
        step = AnonymizerTrainStep(fa, ft, fb_model=fb)
        for name, fn in (("phase1", lambda: step.step_fa(video, labels, views)), ("phase2", lambda: step.step_ft(video, labels, inputs_vispr=views))):
            for i in range(4 * warmup):
                if i >= max(4, warmup // 4) and not E.tuning_pending():
                    break
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            tf = fl * TRAIN_TFLOP[name] + FB_TFLOP[name]
            out[name + "_ms"] = round(ms, 3)
            out[name + "_tflops"] = round(tf / ms * 1e3, 1)
            out[name + "_mfma_frac"] = round(tf / ms * 1e3 / MFMA_PEAK_TFLOPS, 4)
            out[name + "_loss"] = round(float(r["loss_ft"]), 5)
            out[name + "_loss_fb"] = round(float(r["loss_fb"]), 5)
        out["tflop_per_iteration"] = round(fl * (TRAIN_TFLOP["phase1"] + TRAIN_TFLOP["phase2"]) + FB_TFLOP["phase1"] + FB_TFLOP["phase2"], 3)
        _fa, _ft = step.step_fa, step.step_ft
        step_fa = lambda v, l: _fa(v, l, views)
        step_ft = lambda v, l: _ft(v, l, inputs_vispr=views)
    else:
        step_fa, step_ft = step.step_fa, step.step_ft
    # the reference's loop runs BOTH phases per batch (train_anonymizer.py:87-123 then :137-191): each phase then starts from the
    # other network's fresh weights (frozen-BN folds and 16-bit weight images rebuilt), which the per-phase loops above never pay
    for _ in range(max(3, warmup // 8)):
        step_fa(video, labels); step_ft(video, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fa(video, labels); step_ft(video, labels)
    torch.cuda.synchronize()
    out["iteration_ms"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    # the same loop with the losses handed back as device tensors (AnonymizerTrainStep.lazy_losses): the reference reads `loss.item()` every
    # iteration, a device sync per phase that leaves the GPU idle while Python queues the next phase's first launches; a caller that logs
    # every k-th iteration does not pay it
    step.lazy_losses = True
    for _ in range(3):
        step_fa(video, labels); step_ft(video, labels)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fa(video, labels); step_ft(video, labels)
    torch.cuda.synchronize()
    out["iteration_ms_async_losses"] = round((time.perf_counter() - t0) / steps * 1e3, 3)
    # HBM traffic of one iteration from the PMC passes over the same iteration (scripts/profile_train.sh), quoted only for the kernels it was measured on
    out["traffic"] = None
    tpath = os.path.join(ROOT, "profiles", "traffic_train_cfg3.json")
    if hw == 112 and os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("kernel_sources_sha") == train_sources_sha() and bool(tj.get("with_fb", False)) == bool(with_fb):
            out["traffic"] = tj["traffic_bytes_per_iteration"]
    return out


def bench_anon_extract(dev, n_clips=225, batch=75, steps=3):
    """The extraction the reference's scripts actually run (`anonymized = True`, `arch='unet++'` hard-coded: dali_extraction.py:108,122,169-178,
    st_feature_extraction.py:72): every clip -> fa = UnetPlusPlus(resnet18) on its 16 frames of 224 x 224 -> the Q1 reshape feed -> I3Res50.extract_features.
    clips/s over `n_clips` clips (`batch` per forward), the algorithmic work per clip (16 x UNet++ frame + I3Res50 clip, conv MACs x 2), the fraction of the
    MFMA roofline and the feature's rel-L2 against the CPU oracle (oracle/unetpp_ref -> extract_ref.q1_feed -> i3res50_ref) on 2 clips."""
    from ted_spad_amd import engine as E, extraction
    from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
    from ted_spad_amd.synth import synth_clips, synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        fa, ft = load_fa_model(), load_ft_model("largei3d", num_classes=102)          # load_fa_model's default arch IS 'unet++' (model_loaders.py:17)
    sd_a, sd_t = synth_state_dict(fa.state_dict(), 0), synth_state_dict(ft.state_dict(), 0)
    fa.load_state_dict(sd_a); ft.load_state_dict(sd_t)
    fa, ft = fa.to(dev).eval(), ft.to(dev).eval()
    clips = torch.empty((n_clips, 16, 3, 224, 224), dtype=torch.float32, device=dev)   # the loaders' layout (B, 16, 3, H, W)
    for i in range(0, n_clips, 25):
        k = min(25, n_clips - i)
        clips[i:i + k] = synth_clips(0, k, (3, 16, 224, 224), device=dev, first=i).view(k, 16, 3, 224, 224)
    out = torch.empty((n_clips, 2048), dtype=torch.float32, device=dev)

    nstreams = int(os.environ.get("TEDSPAD_ANON_STREAMS", "2"))

    def step():         # the anonymizer on `batch` clips at a time, the encoder on 75, the 75-clip batches alternating over two streams (as the I3D-only path does)
        extraction.extract_anonymized_clip_features(ft, fa, clips, batch=75, fa_batch=batch, layout="reference", out=out, streams=nstreams)
    with torch.no_grad():
        for i in range(40):                       # until the tile tuner has settled the anonymizer's conv geometries
            step()
            if i >= 2 and not E.tuning_pending():
                break
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        from oracle import extract_ref, i3res50_ref, unetpp_ref
        xs = clips[:2].cpu()
        sdc = {k[4:]: v for k, v in sd_t.items() if k.startswith("i3d.")}
        ref = i3res50_ref.extract_features(extract_ref.q1_feed(xs, lambda fr: unetpp_ref.forward(fr, sd_a)), sdc).flatten(1)
    got = out[:2].cpu()
    rel = (got.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)
    gf = 16 * UNETPP_GFLOP_PER_FRAME + GFLOP_PER_CLIP["largei3d"]
    cps = n_clips / dt
    return {"config": "dali_extraction.py:151-182 as the reference runs it (anonymized = True): fa = unet++ (smp UnetPlusPlus, resnet18 encoder) on 16 frames of 3 x 224 x 224 "
                      "-> Q1 reshape feed -> largei3d extract_features; %d clips, %d per anonymizer forward, 75 per encoder forward, f16 activations / fp32 accumulate, random-init weights" % (n_clips, batch),
            "clips_per_s": round(cps, 1), "ms_per_clip": round(1e3 / cps, 4), "gflop_per_clip": round(gf, 3),
            "gflop_per_clip_parts": {"unetpp_16_frames": round(16 * UNETPP_GFLOP_PER_FRAME, 3), "i3res50_clip": GFLOP_PER_CLIP["largei3d"]},
            "achieved_tflops": round(cps * gf * 1e-3, 1), "frac": round(cps * gf * 1e-3 / MFMA_PEAK_TFLOPS, 4),
            "feature_rel_l2_max": float(rel.max()), "feature_rel_l2_tol": 1e-3, "parity_clips": 2, "steps": steps}


def bench_e2e_uint8(dev, n_clips=2250, batch=375, steps=3, hw=(240, 320)):
    """From DECODED FRAMES to features (SURVEY.md section 8f row 1 -> 8a): uint8 frames (T, 240, 320, 3) resident in HBM -> HybridValPipe's clip sampling (16 frames, every
    2nd, a clip per 32 source frames: dali_extraction.py:62-73) -> val_augmentations (/255, centre crop 0.8, antialiased resize to 224 x 224: :38-50) ->
    I3Res50.extract_features, as extraction.extract_video_features_uint8 runs it: pre-processing writes the persistent stem's 16-bit records in ONE launch per
    batch (tedspad_frames_crop_resize_tp), the stem reads them through its LDS-DMA loader; no fp32 clip batch exists. Beside it: the same clips through the
    fp32 boundary (tedspad_frames_crop_resize per clip into a (n,3,16,224,224) batch -> extract_features), timed on one batch. Parity: 2 clips against the CPU
    oracle (oracle/preprocess_ref -> oracle/i3res50_ref)."""
    from ted_spad_amd import engine as E, extraction, preprocess
    from ted_spad_amd.model_loaders import load_ft_model
    from ted_spad_amd.synth import synth_state_dict
    with contextlib.redirect_stdout(io.StringIO()):
        ft = load_ft_model("largei3d", num_classes=102)
    sd = synth_state_dict(ft.state_dict(), 0)
    ft.load_state_dict(sd)
    ft = ft.to(dev).eval()
    h, w = hw
    t_src = n_clips * 32
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.empty((t_src, h, w, 3), dtype=torch.uint8, device=dev)
    for i in range(0, t_src, 4000):
        k = min(4000, t_src - i)
        frames[i:i + k] = torch.randint(0, 256, (k, h, w, 3), dtype=torch.uint8, device=dev, generator=g)
    out = torch.empty((n_clips, 2048), dtype=torch.float32, device=dev)

    def step():
        extraction.extract_video_features_uint8(ft, frames, batch=batch, streams=2, out=out)
    with torch.no_grad():
        for i in range(60):
            step()
            if i >= 1 and not E.tuning_pending():
                break
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        # the fp32 boundary on one batch: a crop_resize launch per clip into the encoder's batch, then the public extract_features
        box = preprocess.center_crop_box(h, w, int(h * 0.8), int(w * 0.8))
        clips = torch.empty((batch, 3, 16, 224, 224), dtype=torch.float32, device=dev)

        def via_fp32():
            for q in range(batch):
                preprocess.crop_resize(frames[32 * q:32 * q + 32:2].contiguous(), box, (224, 224), out=clips[q], layout="cthw")
            return ft.i3d.extract_features(clips)
        for _ in range(2):
            f32 = via_fp32()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            f32 = via_fp32()
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / steps
        same = bool(torch.equal(f32.flatten(1), out[:batch]))
        from oracle import i3res50_ref, preprocess_ref
        v = frames[:64].cpu().float().unsqueeze(0)                     # clips 0 and 1
        ref_clips = torch.stack([preprocess_ref.val_augmentations(v[:, 32 * i:32 * i + 32:2], 0.8, False, 224, 224)[0].permute(1, 0, 2, 3) for i in range(2)])
        ref = i3res50_ref.extract_features(ref_clips, {k[4:]: x for k, x in sd.items() if k.startswith("i3d.")}).flatten(1)
    got = out[:2].cpu()
    rel = (got.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1)
    cps = n_clips / dt
    return {"config": "dali_extraction.py:38-50,62-73 + large_i3d.py extract_features from uint8 frames resident in HBM: %d x %d x 3 frames, a 16-frame clip (every 2nd frame) "
                      "per 32 source frames -> /255, centre crop 0.8, antialiased resize to 224 x 224 written as the stem's 16-bit records (one launch per %d clips) -> "
                      "I3Res50; %d clips per step, f16 activations / fp32 accumulate, random-init weights" % (h, w, batch, n_clips),
            "clips_per_s": round(cps, 1), "ms_per_clip": round(1e3 / cps, 4), "source_frames_per_s": round(cps * 32, 0),
            "frac": round(cps * GFLOP_PER_CLIP["largei3d"] * 1e-3 / MFMA_PEAK_TFLOPS, 4),
            "via_fp32_clip_clips_per_s": round(batch / dt32, 1), "features_equal_fp32_clip_path": same,
            "feature_rel_l2_max": float(rel.max()), "feature_rel_l2_tol": 1e-3, "parity_clips": 2, "steps": steps}


def bench_train_main(args, dev, world, rank, dry):
    """`--train` at any N (cfg3 at N = 1 / 112^2, cfg5 at N = 8 / --train-hw 224): one process per GPU, per-rank batch 8 x 48 frames, the gradients of the network
    being updated all-reduced over RCCL in per-stage buckets from inside the backward pass (grad_reduce.GradBucketReducer). Reports the iteration time (MAX over
    ranks, barrier on both sides) with the exchange and -- same loop, `exchange = False` -- without it: their difference is the all-reduce's exposed share
    (BASELINE.md section 4, row 4). `--dry-run-cpu`: the same control flow on gloo with a stand-in step that drives the REAL reducer (no kernels)."""
    from ted_spad_amd.grad_reduce import GradBucketReducer

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def timed_loop(fn, steps):
        sync()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        if world > 1:
            dist.barrier()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        return float(dt.item()) / steps * 1e3

    steps = max(args.steps, 5)
    hw = args.train_hw
    if dry:
        # stand-in step: three "stages" whose gradients become final one after the other; every rank contributes rank + 1 so the mean is known
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(n)) for n in (1000, 50, 7, 300, 11)]
        red = GradBucketReducer([params[3:], params[1:3], params[:1]], all_params=params)
        ok = [True]

        def it():
            red.prepare()
            for i in range(len(red.buckets)):
                for p in red.buckets[i]:
                    p.grad.add_(float(rank + 1))
                red.bucket_ready(i)
            red.finish()
            want = (world + 1) / 2.0 if red.exchange else float(rank + 1)
            ok[0] = ok[0] and all(bool((p.grad == want).all()) for p in params) and red.issued == list(range(len(red.buckets)))
        reducers, nbytes = [red], red.nbytes()
        losses = {}
    else:
        from ted_spad_amd import engine as E
        from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
        from ted_spad_amd.synth import synth_state_dict, synth_train_video
        from ted_spad_amd.train_step import AnonymizerTrainStep
        with contextlib.redirect_stdout(io.StringIO()):
            fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
        fa.load_state_dict(synth_state_dict(fa.state_dict(), 0))            # the same initial replica on every rank (the reference's DataParallel broadcast)
        ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
        fa, ft = fa.to(dev), ft.to(dev)
        step = AnonymizerTrainStep(fa, ft, group=dist.group.WORLD if world > 1 else None)
        video = synth_train_video(rank, "bench_train", (8, 48, 3, hw, hw), device=dev)      # a different shard of the batch per rank
        labels = torch.randint(1, 102, (8,), device=dev, generator=torch.Generator(device=dev).manual_seed(rank))
        losses = {}

        def it():
            r1 = step.step_fa(video, labels)
            r2 = step.step_ft(video, labels)
            losses["phase1_loss"], losses["phase2_loss"] = float(r1["loss_ft"]), float(r2["loss_ft"])
        for i in range(4 * 45):                   # until the tile tuner has settled every conv geometry of both phases
            if i >= 45 and not E.tuning_pending():
                break
            it()
        reducers = [step.red_fa, step.red_ft]
        nbytes = sum(r.nbytes() for r in reducers)
        ok = [True]
    for _ in range(2):
        it()
    with_x = timed_loop(it, steps)
    losses_x = dict(losses)              # the losses of the loop WITH the exchange (the one below lets the replicas drift apart at N > 1: local-gradient Adam steps)
    for r in reducers:
        r.exchange = False
    for _ in range(2):
        it()
    without_x = timed_loop(it, steps)
    for r in reducers:
        r.exchange = True
    if rank == 0:
        fl = (hw / 112.0) ** 2 * (TRAIN_TFLOP["phase1"] + TRAIN_TFLOP["phase2"]) * world      # algorithmic TFLOP of one iteration of the whole job
        ach = fl / with_x * 1e3
        cfg = "cfg3" if (hw == 112 and world == 1) else ("cfg5" if hw == 224 else "cfg3-shape, DDP")
        res = {"metric": "%s training iteration" % cfg, "value": round(with_x, 3), "unit": "ms", "n_gpus": world, "steps": steps, "higher_is_better": False,
               "scaling": "weak", "dtype": args.dtype, "data": "synthetic", "samples_per_s": round(8 * world / with_x * 1e3, 2),
               "config": {"workload": ("DRY RUN (CPU, stand-in step, no kernels): " if dry else "") +
                                      "train_anonymizer.py iteration (phase 1 + phase 2, Adam steps included), per-rank batch 8 x 48 x %d^2, global batch %d x 48 x %d^2" % (hw, 8 * world, hw),
                          "parallelism": "data-parallel x%d, RCCL all-reduce of fp32 gradient buckets (per stage, from inside the backward pass)" % world},
               "allreduce": {"bytes_per_rank_per_iteration": nbytes, "iteration_ms_without_exchange": round(without_x, 3),
                             "exposed_ms": round(with_x - without_x, 3), "exposed_frac": round(max(0.0, with_x - without_x) / with_x, 4)}}
        if dry:
            res["dry_run"] = True
        else:
            traffic = None          # per rank, from the PMC passes over the cfg3 iteration (scripts/profile_train.sh); only for the sources it was measured on
            tpath = os.path.join(ROOT, "profiles", "traffic_train_cfg3.json")
            if hw == 112 and os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("kernel_sources_sha") == train_sources_sha():
                    traffic = tj["traffic_bytes_per_iteration"]
            res["roofline"] = {"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS * world, "unit": "TFLOP/s", "frac": round(ach / (MFMA_PEAK_TFLOPS * world), 4),
                               "traffic": traffic, "kernel": "both phases of one iteration on every rank (conv / linear MACs x 2 of BASELINE.md section 2 over the iteration time)"}
            res.update(losses_x)
            res["note"] = "iteration_ms_without_exchange is timed LAST, on the same step object, with local gradients only: the replicas' weights differ afterwards"
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok[0]))
        if rank == 0:
            res["allreduce_ok"] = all(flags)
    elif rank == 0 and dry:
        res["allreduce_ok"] = bool(ok[0])
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _table_hash(table) -> str:
    """Short digest of a tile table (engine.export_tile_table): equal on every rank after the broadcast."""
    import hashlib
    return hashlib.sha1(repr(sorted((k, sorted(v.items(), key=repr)) for k, v in table.items())).encode()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--arch", default="largei3d", choices=["largei3d", "i3d"])
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--batch", type=int, default=375, help="clips per forward. 375 = 6 forwards per 2 250-clip step: layer3's 256 x 256 tiles and the one-frame-per-workgroup "
                                                            "bottleneck kernel then fill 8.97 / 2.93 rounds of the 256 CUs (225 clips: 5.38 / 1.76 rounds, a tenth of each idle); measured +1.2 %% clips/s")
    ap.add_argument("--clip-times", type=int, default=225, help="clip times per GPU (7200 frames / 32)")
    ap.add_argument("--crops", type=int, default=10)
    ap.add_argument("--streams", type=int, default=2, help="HIP streams the clip batches alternate over (fills the tail of one forward with the next)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the cfg3 training-iteration timing appended at N = 1")
    ap.add_argument("--train", action="store_true", help="only the cfg3 training-iteration timing (one JSON line)")
    ap.add_argument("--e2e", action="store_true", help="only the uint8-frames -> features measurement (one JSON line)")
    ap.add_argument("--train-hw", type=int, default=112, help="with --train: frame size (112: cfg3; 224: the per-rank batch of cfg5)")
    ap.add_argument("--no-act-range", action="store_true", help="skip the per-stage max |activation| of one 10-clip forward (f16 head-room: storage saturates at 65504) added to the line at N = 1")
    ap.add_argument("--dry-run-cpu", action="store_true", help="rehearse the multi-process control flow on CPU/gloo with a stub extractor")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong (default at N > 1; cfg4): ONE video of --clip-times clip times split over the N ranks (BASELINE.json north_star: 'clips of a long video "
                         "shard across the 8 GPUs'; ragged shards; a rank's shard runs as equal forwards spread over its streams) -- the weak figure is measured right "
                         "after it and reported as `weak_scaling` in the same line; weak: every rank keeps --clip-times clip times (the video grows with N). "
                         "At N = 1 the two are the same workload (reported as \"weak\")")
    ap.add_argument("--no-weak", action="store_true", help="N > 1, strong: skip the weak-scaling measurement appended to the line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver starts `--gpus 1`: this process has not touched the GPU (importing torch does not) and starts the N ranks as a
        # CHILD `python -m torch.distributed.run` (never an exec), relays rank 0's JSON line (inherited stdout) and exits with the child's code
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        if args.dry_run_cpu:
            env.setdefault("OMP_NUM_THREADS", "1")
        # --standalone: the launcher's own c10d rendezvous on a port IT binds (no probe-close-reuse race with other jobs on the box); 127.0.0.1 because the container's
        # hostname may not resolve
        cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if args.scaling is None:
        args.scaling = "strong" if world > 1 else "weak"
    dry = args.dry_run_cpu
    if dry:
        dev = torch.device("cpu")
        if world > 1:
            dist.init_process_group("gloo")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1:
            dist.init_process_group("nccl", device_id=dev)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    if args.train:
        bench_train_main(args, dev, world, rank, dry)
        return
    if args.e2e:
        print(json.dumps(bench_e2e_uint8(dev)))
        return

    from ted_spad_amd import sharding
    from ted_spad_amd.synth import synth_clips, synth_state_dict

    F = 2048 if args.arch == "largei3d" else 1024
    shape = (3, 16, 224, 224) if not dry else (3, 2, 4, 4)
    if dry:
        sd = None
        proj = torch.linspace(-1.0, 1.0, F).view(1, F)

        def fx(x):          # stub: a deterministic function of the clip (so the gathered rows can be checked), NOT a kernel
            return (x.flatten(1).mean(1, keepdim=True) * proj).view(-1, F, 1, 1, 1)
    else:
        from ted_spad_amd.model_loaders import load_ft_model
        with contextlib.redirect_stdout(io.StringIO()):
            ft = load_ft_model(args.arch, num_classes=102)
        sd = synth_state_dict(ft.state_dict(), 0)
        ft.load_state_dict(sd, strict=True)
        for m in ft.modules():
            if hasattr(m, "compute_dtype"):
                m.compute_dtype = args.dtype
        ft = ft.to(dev).eval()
        fx = ft.extract_features if hasattr(ft, "extract_features") else ft.i3d.extract_features

    def run_workload(scaling, timed_steps, tune):
        """One measurement of the extraction step under `scaling` ("weak": every rank keeps --clip-times clip times; "strong": ONE video of --clip-times
        clip times split over the ranks). Returns the MAX-over-ranks wall time of `timed_steps` steps (barrier + synchronize on both sides), rank 0's host copy of
        the gathered block, the HIP-event records of the forwards and the shard geometry."""
        T_total = args.clip_times * world if scaling == "weak" else args.clip_times
        lo, hi = sharding.shard_range(T_total, rank, world)
        n_local = (hi - lo) * args.crops
        # ---- this rank's shard of the synthetic video, resident in HBM -------------------------------
        clips = torch.empty((n_local,) + shape, dtype=torch.float32, device=dev)
        for i in range(0, n_local, 25):
            k = min(25, n_local - i)
            clips[i:i + k] = synth_clips(0, k, shape, device=dev, first=lo * args.crops + i)
        sync()
        plan = sharding.batch_plan(n_local, args.batch, max(1, args.streams))      # [(first clip, clips)] per forward, spread evenly over the streams

        feats = torch.empty((n_local, F), dtype=torch.float32, device=dev)
        # rank 0's host copy of the gathered (T,10,F) block: pinned, so the device->host copy of a step is an asynchronous 18 MB DMA
        # (a pageable `.cpu()` took ~5 ms of every 144 ms step with the GPU idle); it completes inside the timed region (final synchronize)
        host_feats = torch.empty((T_total, args.crops, F), dtype=torch.float32, pin_memory=not dry) if rank == 0 else None
        ev = []

        def step(timed):
            if dry:
                for i, k in plan:
                    feats[i:i + k] = fx(clips[i:i + k]).flatten(1)
            else:
                main_s = torch.cuda.current_stream()
                if timed:   # HIP events on the launching (main) stream around the fork/join of the forward streams
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(main_s)
                for st in streams:
                    st.wait_stream(main_s)
                for j, (i, k) in enumerate(plan):
                    with torch.cuda.stream(streams[j % len(streams)]):
                        feats[i:i + k] = fx(clips[i:i + k]).flatten(1)
                for st in streams:
                    main_s.wait_stream(st)
                if timed:
                    e1.record(main_s)
                    ev.append((e0, e1, n_local))
            full = sharding.gather_video_features(feats.view(hi - lo, args.crops, F), T_total)
            if rank == 0:                               # the .npy rows reach the host on rank 0
                if copy_stream is None:
                    host_feats.copy_(full.view(T_total, args.crops, F), non_blocking=True)
                else:
                    if full.data_ptr() == feats.data_ptr():     # N = 1: the "gathered" block IS the buffer the next step's forwards write: snapshot it (18 MB on the device)
                        full = full.clone()
                    copy_stream.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(copy_stream):
                        host_feats.copy_(full.view(T_total, args.crops, F), non_blocking=True)
                    full.record_stream(copy_stream)     # the gathered block is freed by the launching stream's allocator: not before the copy ran
                return host_feats
            return full

        tiles_ok, table_hash = None, None
        with torch.no_grad():
            # batch sizes (= conv geometries) any rank will run
            sizes = sorted({k for r in range(world) for _, k in sharding.batch_plan(
                (lambda l_, h_: (h_ - l_) * args.crops)(*sharding.shard_range(T_total, r, world)), args.batch, max(1, args.streams))}, reverse=True)
            if not dry and tune:
                # untimed setup (like cudnn.benchmark's first iterations in the reference): the conv tile configurations are chosen in context during the
                # first ~45 forwards of every conv geometry; do that before the counted warm-up. At N > 1 only rank 0 tunes; its table is broadcast so that
                # every rank runs the same tiles (identical features for identical clips, no start-up skew from N tuning passes).
                from ted_spad_amd import engine as _E
                if rank == 0 or world == 1:
                    for size in sizes:
                        if not _E.AUTOTUNE or size > n_local:
                            continue
                        for i in range(96):                       # > number of tile configurations + TUNE_REPS pruned passes
                            multi = os.environ.get("TEDSPAD_PRIME_MULTI") == "1" or not _E.tuning_pending()
                            with torch.cuda.stream(streams[i % len(streams) if multi and i else 0]):
                                fx(clips[:size])
                            if i >= 48 and not _E.tuning_pending():
                                break
                if world > 1:
                    sync()
                    from ted_spad_amd.extraction import share_tile_choices
                    share_tile_choices(ft)
                    net = ft.i3d if hasattr(ft, "i3d") else ft
                    table_hash = _table_hash(_E.export_tile_table(net.packed()))
            elif dry and world > 1:
                # rehearsal of the tile-table hand-off with stand-in objects (no kernels): rank 0 "decides", every rank must end with the same table
                from ted_spad_amd import engine as _E

                class _Tuned:
                    def __init__(self):
                        self._cfgs = _E._Cfgs()
                fake = {"layer%d.conv" % i: _Tuned() for i in range(3)}
                if rank == 0:
                    for i, o in enumerate(fake.values()):
                        for size in sizes:
                            o._cfgs[(size, 2, 14, 14, 1024, (0, 1, 1), True, None)] = 17 + i
                box = [_E.export_tile_table(fake) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                if rank != 0:
                    _E.import_tile_table(fake, box[0])
                mine = _E.export_tile_table(fake)
                table_hash = _table_hash(mine)
                flags = [None] * world
                dist.all_gather_object(flags, mine == box[0] and all(len(v) == len(sizes) for v in mine.values()))
                tiles_ok = all(flags)
            sync()
            for _ in range(args.warmup):
                step(False)
            if world > 1:
                dist.barrier()
            sync()
            t0 = time.perf_counter()
            for _ in range(timed_steps):
                out = step(True)
            sync()
            if world > 1:
                dist.barrier()
            sync()
            dt = time.perf_counter() - t0
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # what every rank ran, in rank 0's line: a ragged or mistuned rank is visible in the driver's SCALE record
        ranks = None
        if world > 1:
            ranks = [None] * world
            dist.all_gather_object(ranks, {"rank": rank, "clip_times": [lo, hi], "n_local": n_local, "plan": [k for _, k in plan], "tile_table": table_hash,
                                           "ms_per_step": round(dt / max(1, timed_steps) * 1e3, 3)})
        return {"dt": float(tmax.item()), "out": out, "ev": ev, "T_total": T_total, "n_local": n_local, "clips": clips, "plan": plan, "sizes": sizes,
                "tiles_ok": tiles_ok, "ranks": ranks}

    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))] if not dry else []
    # rank 0's device->host copy of a step runs on its own stream, under the next step's forwards (it still completes inside the timed
    # region: the final synchronize waits for every stream); on the launching stream the forward streams of the next step would wait for it
    # (0.7 ms of a 102 ms step at N = 1, N x 18 MB = ~6 ms at N = 8)
    copy_stream = torch.cuda.Stream(device=dev) if (not dry and rank == 0) else None

    W = run_workload(args.scaling, args.steps, tune=True)
    dt, out, ev, T_total, n_local, clips = W["dt"], W["out"], W["ev"], W["T_total"], W["n_local"], W["clips"]
    tiles_ok, plan, sizes, rank_info = W["tiles_ok"], W["plan"], W["sizes"], W["ranks"]
    total_clips = T_total * args.crops * args.steps
    value = total_clips / dt
    # N > 1: the other reading of "clips/s at N GPUs" in the same line -- `value` is cfg4 as BASELINE.json words it (ONE video's clips split over the ranks,
    # total work fixed: "strong"); `weak_scaling` keeps a whole video per rank (per-GPU work fixed), measured right after it with the same barrier / MAX rule
    weak = None
    if world > 1 and args.scaling == "strong" and not args.no_weak:
        W = None
        W2 = run_workload("weak", args.steps, tune=True)
        weak = {"value": round(W2["T_total"] * args.crops * args.steps / W2["dt"], 2), "unit": "clips/s", "ms_per_step": round(1e3 * W2["dt"] / args.steps, 3),
                "clips_per_step": W2["T_total"] * args.crops, "clips_per_forward": sorted({k for _, k in W2["plan"]}, reverse=True),
                "note": "every rank keeps a whole %d-clip-time video (the video grows with N); same barrier + MAX-over-ranks timing, all-gather and D2H inside" % args.clip_times}
        if dry:
            want2 = torch.cat([synth_clips(0, min(25, W2["T_total"] * args.crops - i), shape, first=i).flatten(1).mean(1, keepdim=True)
                               for i in range(0, W2["T_total"] * args.crops, 25)]) if rank == 0 else None
            if rank == 0:
                weak["gather_ok"] = bool(torch.allclose(W2["out"].reshape(-1, F), want2 * proj))
        del W2

    if rank != 0:
        if world > 1:
            dist.barrier()              # rank 0 checks its first clips against the CPU oracle after the timed region (below)
            dist.destroy_process_group()
        return

    res = {"metric": "clips/sec (16x224^2 I3D features)", "value": round(value, 2), "unit": "clips/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
           "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": ("DRY RUN (CPU, stub extractor, no kernels): " if dry else "") +
                                  ("cfg2" if world == 1 else "cfg4 (cfg2's video clip-sharded over %d ranks)" % world if args.scaling == "strong" else "cfg2 per rank") +
                                  " dali_extraction path: %s extract_features, %d clip times x %d crops = %d clip-forwards of "
                                  "3x16x224x224 %s per step, random-init weights" % (args.arch, T_total if args.scaling == "strong" else args.clip_times, args.crops,
                                                                                     T_total * args.crops if args.scaling == "strong" else n_local,
                                                                                     "in total (split over the ranks)" if args.scaling == "strong" and world > 1 else "per GPU"),
                      "global_batch": args.batch * world, "clips_per_step": T_total * args.crops,
                      "clips_per_forward": sorted({k for _, k in plan} if world == 1 else set(sizes), reverse=True),
                      "parallelism": "clip-sharded x%d + RCCL all-gather of (T,10,F) features" % world}}
    if weak is not None:
        res["weak_scaling"] = weak
    if rank_info:
        res["ranks"] = rank_info

    if dry:
        # the stub's rows are a known function of the global clip index: the gathered (T, crops, F) block must be complete and ordered
        want = torch.cat([synth_clips(0, min(25, T_total * args.crops - i), shape, first=i).flatten(1).mean(1, keepdim=True) for i in range(0, T_total * args.crops, 25)])
        res["dry_run"] = True
        res["gather_ok"] = bool(torch.allclose(out.reshape(-1, F), want * proj))
        if world > 1:
            res["tile_table_ok"] = bool(tiles_ok)
        print(json.dumps(res))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- roofline of the conv stack: algorithmic FLOPs of a forward / its device time (HIP events) ---
    fwd_ms = sum(a.elapsed_time(b) for a, b, _ in ev)
    fwd_clips = sum(n for _, _, n in ev)
    achieved = fwd_clips * GFLOP_PER_CLIP[args.arch] / fwd_ms  # GFLOP / ms == TFLOP/s
    n_fwd = len(ev) * max(1, len(plan))
    traffic = traffic_all = None
    tpath = os.path.join(ROOT, "profiles", "traffic_cfg2.json")   # PMC result of the same command (scripts/profile_bench.sh)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        # a traffic figure is only quoted for the kernels it was measured on: the profile stores the hash of the kernel sources
        if (tj.get("arch") == args.arch and tj.get("batch") == args.batch and tj.get("dtype") == args.dtype and
                {k for _, k in plan} == {args.batch} and          # ... and for the forward size it was measured on (a rank of a strong-scaling run has smaller forwards)
                tj.get("kernel_sources_sha") == kernel_sources_sha()):
            traffic = tj["conv_traffic_bytes_per_forward"]
            traffic_all = tj.get("all_traffic_bytes_per_forward")
    # sustained shader clock of THIS box under matrix load (DVFS lowers it below the 2.4 GHz the nominal peak assumes): reported beside the nominal peak, never instead
    clock_mhz = None
    try:
        import ctypes as _C
        from ted_spad_amd import _lib as _L
        tbuf = torch.zeros(2 * 1024, dtype=torch.int64, device=dev)
        for _ in range(2):      # ~20 ms each at 1024 workgroups x 4 waves x 4 x 60000 MFMAs; the second one is read
            _L.check(_L.lib().tedspad_clock_probe(60000, 1024, tbuf.data_ptr(), _C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tedspad_clock_probe")
        torch.cuda.synchronize()
        t = tbuf.view(-1, 2).double().cpu()
        clock_mhz = float((100.0 * t[:, 0] / t[:, 1]).median())
    except Exception as e:          # a measurement aid only
        print("clock probe failed: %s" % e, file=sys.stderr)
    res["roofline"] = {"bound": "mfma", "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                       "traffic_all_kernels": traffic_all,      # `traffic` counts the conv kernels of a forward; this one every kernel between its first and last launch (pools, fix-up, average pool)
                       "clock_mhz_under_mfma_load": None if clock_mhz is None else round(clock_mhz, 1),
                       "peak_at_clock": None if clock_mhz is None else round(MFMA_PEAK_TFLOPS * clock_mhz / 2400.0, 1),
                       "kernel": "conv stack of one batch forward (conv_stem_pt_kernel + conv_p8 / conv_patch / conv_flat / conv_igemm / conv_pw "
                                 "launches; layout, max-pool and average-pool passes included in the time)",
                       "ms_per_forward": round(fwd_ms / n_fwd, 3), "clips_per_forward": max([k for _, k in plan] or [0]), "streams": len(streams)}

    # ---- CPU baseline + parity on a bounded sample: the oracle on this box's host cores ------------
    if not args.no_cpu_baseline:      # the parity check runs at every N (the other ranks wait at the final barrier); the CPU extractor is TIMED at N = 1 only
        from oracle import i3res50_ref, inception_i3d_ref
        ncpu = os.cpu_count() or 1
        cores = min(ncpu, 32)  # measured on the GPU box's host (128 hardware threads): 8/16/32/64/128 threads -> 6.6/9.2/10.2/7.7/4.0 clips/s
        torch.set_num_threads(cores)
        xs = clips[:10].cpu()
        if args.arch == "largei3d":
            sdc = {k[4:]: v for k, v in sd.items() if k.startswith("i3d.")}
            cpu_fx = lambda x: i3res50_ref.extract_features(x, sdc)
        else:
            cpu_fx = lambda x: inception_i3d_ref.extract_features(x, sd)
        with torch.no_grad():
            ref = cpu_fx(xs)  # warm-up (also the parity reference)
            if world == 1:
                t0 = time.perf_counter()
                for _ in range(3):
                    cpu_fx(xs)
                cdt = time.perf_counter() - t0
        ref = ref.flatten(1)
        got = out.reshape(-1, F)[:10]
        rel = ((got.double() - ref.double()).norm(dim=1) / ref.double().norm(dim=1))
        if world == 1:
            res["cpu_baseline"] = {"value": round(30.0 / cdt, 3), "unit": "clips/s", "cores": cores, "kind": "port",
                                   "sample": "oracle (fp32 torch CPU restatement) on the first 10 clips (one 10-crop group), 1 warm-up + 3 timed passes; "
                                             "%d of the host's %d hardware threads (the fastest thread count measured on this host type)" % (cores, ncpu)}
        res["feature_rel_l2_max"] = float(rel.max())
        res["feature_rel_l2_tol"] = 1e-3
    if not args.no_act_range and world == 1 and args.arch == "largei3d":
        taps = {}
        with torch.no_grad():
            ft.i3d._trunk(clips[:10], taps=taps)
        res["act_absmax"] = {k: round(float(v.buf.float().abs().max()), 3) for k, v in taps.items()}   # f16 saturates at 65504
        # ... and the device-side counter of clamped stores (tedspad_count_saturated over the stage outputs of the production forward)
        ft.i3d.check_saturation = True
        with torch.no_grad():
            ft.i3d.extract_features(clips[:10])
        sat, bad = ft.i3d.saturation_counts()
        ft.i3d.check_saturation = False
        res["act_saturated"] = {"at_f16_max": sat, "non_finite": bad, "clips": 10}
    if world == 1 and not args.no_train:
        del clips, W
        torch.cuda.empty_cache()
        res["anon_extract"] = bench_anon_extract(dev)
        torch.cuda.empty_cache()
        res["e2e_uint8"] = bench_e2e_uint8(dev)
        torch.cuda.empty_cache()
        res["train_cfg3"] = bench_train(dev)
    if os.environ.get("TEDSPAD_TILE_PICKS"):          # diagnostic: the tile configurations the tuner settled on over everything this run measured
        from ted_spad_amd import engine as _E2
        res["tile_picks"] = {str(k): v for k, v in sorted(_E2.TILE_PICKS.items())}
    print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
