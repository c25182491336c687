"""-m gpu: the RCCL calls of the N > 1 paths (feature all-gather, gradient all-reduce) on device tensors. A GPU box has
one MI355X, so this is a 1-rank `nccl` (= RCCL) process group in a child process: it checks that the backend comes
up in this environment and that the exact collective calls of sharding.gather_video_features /
train_step.allreduce_mean_grads accept our device tensors; the multi-rank logic is covered over gloo on CPU
(tests/test_sharding_gloo.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _worker(port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from ted_spad_amd.train_step import allreduce_mean_grads
        send = torch.arange(225 * 10 * 16, dtype=torch.float32, device="cuda").view(225, 10, 16)
        recv = torch.empty_like(send)
        dist.all_gather_into_tensor(recv, send.contiguous())          # the call of sharding.gather_video_features
        ok = bool(torch.equal(recv, send))
        ps = [torch.nn.Parameter(torch.ones(5, 3, device="cuda")), torch.nn.Parameter(torch.ones(7, device="cuda"))]
        for i, p in enumerate(ps):
            p.grad = torch.full_like(p, float(i + 1))
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat)                                          # the call of train_step.allreduce_mean_grads
        allreduce_mean_grads(ps)                                       # world 1: returns early, gradients untouched
        ok = ok and float(flat.sum()) == 15 * 1 + 7 * 2 and float(ps[1].grad[0]) == 2.0
        q.put("ok" if ok else "mismatch")
    except Exception as e:  # noqa: BLE001
        q.put("error: %r" % (e,))
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_on_device_tensors():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(port, q))
    p.start()
    p.join(180)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL worker hung")
    assert q.get(timeout=5) == "ok"


def _reducer_worker(port, q):
    """AnonymizerTrainStep with group = WORLD over a 1-rank nccl (= RCCL) group, its bucketed reducers forced to launch their asynchronous all-reduces: both phases
    against the same steps with group = None, in deterministic mode (bit-equal gradients, losses and updated parameters)."""
    import contextlib
    import io
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        from ted_spad_amd import engine as E
        from ted_spad_amd.model_loaders import load_fa_model, load_ft_model
        from ted_spad_amd.synth import synth_state_dict, synth_train_video
        from ted_spad_amd.train_step import AnonymizerTrainStep
        E.set_deterministic(True)
        video = synth_train_video(0, "rccl_video", (2, 48, 3, 32, 32)).cuda()
        labels = torch.tensor([5, 77]).cuda()

        def run(group):
            with contextlib.redirect_stdout(io.StringIO()):
                fa, ft = load_fa_model(arch="unet"), load_ft_model("largei3d", num_classes=102)
            fa.load_state_dict(synth_state_dict(fa.state_dict(), 0)); ft.load_state_dict(synth_state_dict(ft.state_dict(), 0))
            ft.i3d.drop_p = 0.0
            step = AnonymizerTrainStep(fa.cuda(), ft.cuda(), group=group)
            issued = []
            if group is not None:
                for r in (step.red_fa, step.red_ft):
                    r.force = True
            o1 = step.step_fa(video, labels)
            g1 = {k: p.grad.clone() for k, p in fa.named_parameters() if p.grad is not None}
            if group is not None:
                issued.append((list(step.red_fa.issued), len(step.red_fa.buckets)))
            o2 = step.step_ft(video, labels)
            g2 = {k: p.grad.clone() for k, p in ft.named_parameters() if p.grad is not None}
            if group is not None:
                issued.append((list(step.red_ft.issued), len(step.red_ft.buckets)))
            w = {k: v.detach().clone() for k, v in list(fa.state_dict().items()) + list(ft.state_dict().items())}
            return (o1["loss_fa"], o2["loss_ft"]), g1, g2, w, issued

        la, a1, a2, wa, _ = run(None)
        lb, b1, b2, wb, issued = run(dist.group.WORLD)
        ok = la == lb and all(torch.equal(a1[k], b1[k]) for k in a1) and all(torch.equal(a2[k], b2[k]) for k in a2) and all(torch.equal(wa[k], wb[k]) for k in wa)
        ok = ok and len(a1) > 20 and len(a2) > 100 and all(sorted(i) == list(range(n)) and n >= 2 for i, n in issued)
        q.put("ok" if ok else "mismatch: losses %r %r, buckets %r" % (la, lb, issued))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put("error: %r\n%s" % (e, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_reducer_runs_against_rccl():
    """grad_reduce.GradBucketReducer's asynchronous per-bucket all-reduce (launched from inside the backward pass) against the nccl backend itself -- one rank
    (a GPU box has one MI355X), the collectives forced on: the sums of one rank are the rank's own gradients, so both phases must give the bits of the
    group = None run. (World 2 semantics: tests/test_sharding_gloo.py.)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_reducer_worker, args=(port, q))
    p.start()
    p.join(420)
    if p.is_alive():
        p.kill()
        pytest.fail("RCCL worker hung")
    assert q.get(timeout=5) == "ok"
