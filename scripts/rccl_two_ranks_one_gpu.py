"""Can two RCCL ranks share the one GPU of a box? (The driver's N > 1 runs have one GPU per rank; this only tells whether a 2-rank rehearsal is possible on one device.)
Usage: python scripts/rccl_two_ranks_one_gpu.py  -- prints OK or the refusal."""
import os, sys, subprocess
if len(sys.argv) > 1:
    import torch, torch.distributed as dist
    rank = int(sys.argv[1])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    try:
        dist.init_process_group("nccl", rank=rank, world_size=2, device_id=torch.device("cuda", 0))
        x = torch.full((4,), float(rank + 1), device="cuda")
        dist.all_reduce(x)
        torch.cuda.synchronize()
        print("rank", rank, "OK", x.tolist(), flush=True)
        dist.destroy_process_group()
    except Exception as e:
        print("rank", rank, "REFUSED:", str(e).splitlines()[0][:200], flush=True)
        sys.exit(3)
    sys.exit(0)
import socket
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = str(s.getsockname()[1]); s.close()
ps = [subprocess.Popen([sys.executable, __file__, str(r), port]) for r in range(2)]
rc = 0
for p in ps:
    try:
        rc |= p.wait(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill(); rc |= 4
print("two RCCL ranks on one GPU:", "OK" if rc == 0 else "not possible (exit %d)" % rc)
