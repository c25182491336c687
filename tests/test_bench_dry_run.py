"""CPU: bench.py's multi-process control flow (process group, clip sharding, barrier, MAX-reduce of the elapsed time, rank-0-only JSON
line) rehearsed with 2 gloo ranks and a stub extractor (`--dry-run-cpu`), so that the N > 1 branch has run before the driver's
8-GPU node runs it over RCCL. No kernel runs here; the numbers mean nothing."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_single_process_dry_run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-cpu", "--steps", "2", "--warmup", "1", "--clip-times", "5",
                        "--crops", "2", "--batch", "4"], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["gather_ok"] and lines[0]["steps"] == 2


def test_two_rank_dry_run_prints_one_line_from_rank0():
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "3", "--warmup", "1",
           "--clip-times", "5", "--crops", "2", "--batch", "4", "--scaling", "weak"]       # 5 clip times per rank, 10 clips in forwards of at most 4
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                                  # rank 0 only
    j = lines[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["clips_per_step"] == 2 * 5 * 2
    assert j["gather_ok"] is True                                     # every rank's rows arrived, in order
    assert j["value"] > 0 and abs(j["value"] - j["config"]["clips_per_step"] * j["steps"] / (j["ms_per_step"] * j["steps"] / 1e3)) < 1e-2 * j["value"]
    assert j["tile_table_ok"] is True                                 # rank 0's tile choices reached every rank unchanged (engine.export / import_tile_table)


def test_two_rank_strong_scaling_dry_run_splits_one_video():
    """`--scaling strong`: ONE video of --clip-times clip times split over the ranks (ragged: 7 clip times over 2 ranks = 4 + 3), the gathered block complete."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1",
           "--clip-times", "7", "--crops", "2", "--batch", "4", "--scaling", "strong"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1
    j = j[0]
    assert j["scaling"] == "strong" and j["n_gpus"] == 2 and j["config"]["clips_per_step"] == 7 * 2
    assert j["gather_ok"] is True and j["tile_table_ok"] is True


def test_eight_rank_strong_scaling_dry_run_has_the_cfg4_shards():
    """cfg4's shape at N = 8: ONE video of 225 clip times over 8 ranks = 29 / 29 / 29 / 29 / 29 / 29 / 29 / 22 clip times (x crops clips), every rank's shard, forward
    plan and tile-table digest in rank 0's line (`ranks`), the gathered block complete and ordered, the tile table identical everywhere."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-cpu", "--steps", "1", "--warmup", "0",
           "--clip-times", "225", "--crops", "2", "--scaling", "strong", "--no-weak"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1, r.stdout
    j = j[0]
    assert j["n_gpus"] == 8 and j["scaling"] == "strong" and j["config"]["clips_per_step"] == 225 * 2
    assert j["gather_ok"] is True and j["tile_table_ok"] is True
    rk = j["ranks"]
    assert [x["rank"] for x in rk] == list(range(8))
    assert [x["clip_times"][1] - x["clip_times"][0] for x in rk] == [29] * 7 + [22] and rk[0]["clip_times"][0] == 0 and rk[7]["clip_times"][1] == 225
    assert all(x["n_local"] == (x["clip_times"][1] - x["clip_times"][0]) * 2 and sum(x["plan"]) == x["n_local"] for x in rk)
    assert len({x["tile_table"] for x in rk}) == 1 and rk[0]["tile_table"]


def test_plain_python_launch_with_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` exactly as `--gpus 1` is started (no torch.distributed.run, WORLD_SIZE unset): the parent starts the two ranks as a child
    process, relays ONE JSON line and exits 0. N > 1 defaults to strong scaling (cfg4: one video split over the ranks) with the weak figure in the same line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1", "--clip-times", "7",
                        "--crops", "2", "--batch", "4"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1, r.stdout
    j = j[0]
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["config"]["clips_per_step"] == 7 * 2 and j["gather_ok"] is True
    w = j["weak_scaling"]
    assert w["clips_per_step"] == 2 * 7 * 2 and w["value"] > 0 and w["gather_ok"] is True


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_batch_plan_spreads_a_shard_evenly_over_the_streams():
    from ted_spad_amd.sharding import batch_plan
    assert batch_plan(2250, 375, 2) == [(i * 375, 375) for i in range(6)]                    # cfg2 at N = 1: unchanged
    assert batch_plan(290, 375, 2) == [(0, 145), (145, 145)]                                 # cfg4 at N = 8: 29 clip times x 10 crops, one forward per stream
    assert batch_plan(220, 375, 2) == [(0, 110), (110, 110)]                                 # ... the last rank's 22 clip times
    assert batch_plan(1130, 375, 2) == [(0, 283), (283, 283), (566, 283), (849, 281)]        # N = 2: 113 clip times
    assert batch_plan(40, 375, 2) == [(0, 40)] and batch_plan(0, 375, 2) == []               # too small to split; an empty shard launches nothing
    for n in (1, 31, 64, 65, 290, 999, 2250):
        for st in (1, 2, 3):
            pl = batch_plan(n, 375, st)
            assert sum(k for _, k in pl) == n and all(k <= 375 for _, k in pl) and [i for i, _ in pl] == [sum(k for _, k in pl[:j]) for j in range(len(pl))]


def test_two_rank_training_dry_run_drives_the_real_reducer():
    """`bench.py --train --gpus 2 --dry-run-cpu`: the distributed training entry (cfg5's launch line) with a stand-in step around the REAL GradBucketReducer:
    buckets all-reduced in the order they become final, means correct on every rank, one JSON line with the exchange's exposed share."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--train", "--train-hw", "224", "--dry-run-cpu", "--steps", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_lines(r.stdout)
    assert len(j) == 1
    j = j[0]
    assert j["n_gpus"] == 2 and j["dry_run"] and j["allreduce_ok"] is True and j["higher_is_better"] is False
    assert j["allreduce"]["bytes_per_rank_per_iteration"] > 0 and "exposed_ms" in j["allreduce"] and j["metric"].startswith("cfg5")


def test_tile_table_export_import_roundtrip():
    from ted_spad_amd import engine as E

    class T:
        def __init__(self):
            self._cfgs = E._Cfgs()

    class Pair:
        def __init__(self):
            self.pc = T()
    a, b = {"x": T(), "y": Pair()}, {"x": T(), "y": Pair()}
    a["x"]._cfgs[(1, 2, "k")] = 25
    a["x"]._cfgs[(3,)] = {"cands": [1, 2]}          # still tuning: not exported
    a["y"].pc._cfgs[(9, 9)] = 33
    tab = E.export_tile_table(a)
    assert tab == {"x": {(1, 2, "k"): 25}, "y.pc": {(9, 9): 33}}
    assert E.import_tile_table(b, tab) == 2 and E.export_tile_table(b) == tab
