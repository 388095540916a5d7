"""bench.py's launch contract: --gpus N must agree with the launcher's WORLD_SIZE, starts N ranks itself when there is no
launcher, and prints one JSON line with the keys the driver reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--batch", "16", "--chunk", "8", "--pool", "2",
         "--height", "384", "--width", "512", "--no-configs"]


def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "must agree" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_one_rank_line_has_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "gather"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["value"] > 0 and "workload" in d["config"]
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["achieved"] > 0
    assert d["gather"]["records"] == d["strips_per_step"] > 0


@pytest.mark.gpu
def test_two_ranks_started_by_bench_itself_rehearsal_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher: bench.py starts torch.distributed.run on itself.  LFA_BENCH_REHEARSAL=1 lets the two
    ranks share this box's one GPU and talk over gloo; the real run is one rank per GPU over RCCL (same code path otherwise:
    rank-dependent inputs, barrier + max-over-ranks timing, gather of the per-strip records)."""
    env = dict(os.environ, LFA_BENCH_REHEARSAL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    # without --no-configs: the N > 1 branch also runs configs[3] (3072x4096x3 images) on every rank with its own record
    # gather -- the code the driver's 8-GPU run goes through -- here with 4 images per rank and one timed step
    small = [a for a in SMALL if a != "--no-configs"] + ["--config-batch", "4", "--config-chunk", "2", "--config-steps", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + small, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["gather"]["records"] == d["strips_per_step"] > 0 and d["gather"]["backend"].startswith("gloo")
    c3 = d["configs"]["configs[3]"]
    assert c3["n_gpus"] == 2 and c3["images"] == 4 and c3["scaling"] == "weak"
    assert c3["gather"]["records"] == c3["strips_per_step"] == 2 * 4 * 8 and c3["gather"]["backend"].startswith("gloo")
    assert c3["images_per_s"] > 0 and 0 < c3["e2e_frac_hbm"] < 1


_RCCL_ONE_RANK = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
dist.init_process_group(backend="nccl", device_id=dev)
torch.cuda.set_device(0)
# the collectives and operand types bench.py uses between ranks, on device tensors over RCCL
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert float(t.item()) == 1.5
k = torch.tensor([7], dtype=torch.int64, device=dev); dist.all_reduce(k, op=dist.ReduceOp.SUM); assert int(k.item()) == 7
c = torch.arange(5, dtype=torch.int32, device=dev); out = [torch.zeros(5, dtype=torch.int32, device=dev)]
dist.all_gather(out, c); assert torch.equal(out[0], c)
b = torch.arange(136, dtype=torch.uint8, device=dev); ob = [torch.zeros(136, dtype=torch.uint8, device=dev)]
dist.all_gather(ob, b); assert torch.equal(ob[0], b)
dist.barrier(); torch.cuda.synchronize()
dist.destroy_process_group()
print("rccl one-rank ok")
"""


@pytest.mark.gpu
def test_rccl_backend_takes_the_collectives_of_the_record_gather():
    """The rehearsal above talks over gloo (two ranks cannot share one GPU under RCCL).  What can be checked on a one-GPU box:
    that the "nccl" (= RCCL) backend initialises with `device_id` as bench.py does and accepts the collectives and operand
    types of the timing reductions and of shard.gather_strip_records (float64 MAX, int64 SUM, all_gather of int32 counts and
    of uint8 record bytes) on device tensors."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK, str(port)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl one-rank ok" in r.stdout, r.stderr[-3000:]
