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
