"""RCCL on the one GPU a test box has: the `nccl` branch of dp.init_from_env with a single-rank communicator, an
all-reduce of the flat gradient buffer through it, and bench.py's own multi-rank launcher path (`--gpus 1` children
are not spawned; the launcher itself is covered on CPU by tests/test_dist_cpu.py).  Runs in a subprocess so that the
process group never leaks into the pytest process."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, torch
sys.path.insert(0, os.environ["EPN_ROOT"])
from epn_pointcloud_amd import dp
import torch.distributed as dist
rank, local_rank, world = dp.init_from_env(backend="nccl", force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and (rank, world) == (0, 1)
dev = torch.device("cuda", local_rank)
lin = torch.nn.Linear(8, 4).to(dev)
gb = dp.GradBuckets([list(lin.parameters())], world, hooks=False)
gb.zero()
lin(torch.ones(3, 8, device=dev)).sum().backward()
before = gb.flat.clone()
work = dist.all_reduce(gb.flat, op=dist.ReduceOp.SUM, async_op=True)     # RCCL kernel on the flat gradient bucket
work.wait()
torch.cuda.synchronize()
assert torch.equal(gb.flat, before)                                      # one rank: sum == itself
dp.broadcast_parameters(lin)
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_single_rank_rccl_allreduce(gpu):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
               EPN_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("extra,launch", [([], "hipgraph"), (["--no-graph"], "eager")])
def test_bench_two_ranks_rehearsal_on_one_gpu(gpu, extra, launch):
    """`python bench.py --gpus 2` end to end through its own launcher (no torchrun): two rank processes, the HIP graph
    captured before the communicator exists, parameter broadcast, flat-bucket gradient all-reduce, barrier + max-over-ranks
    timing, ONE JSON line from rank 0.  Both ranks share the single GPU of the test box and talk over gloo
    (EPN_DP_SHARE_GPU / EPN_DP_BACKEND: a rehearsal of the control flow, not a measurement; RCCL itself is exercised by the
    single-rank test above)."""
    import json
    env = dict(os.environ, EPN_DP_SHARE_GPU="1", EPN_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "EPN_DP_CHILD"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8 and out["config"]["parallelism"] == "dp2"
    assert out["value"] > 0 and out["config"]["launch"] == launch     # eager: per-stage all-reduce from backward hooks
    # a multi-rank line explains itself (review item 7): every rank's own step time, the collective phase as the compute stream
    # sees it with its bus bandwidth, and -- graph mode -- the same step without the collective, timed in the same run on
    # every rank at once: value / (N x that) is the efficiency against the rank program
    d = out["dp"]
    lo, hi = d["per_rank_ms_per_step"]
    assert 0 < lo <= hi <= 1.05 * out["ms_per_step"] + 1.0
    assert d["allreduce_ms"] > 0 and d["bus_gbps"] > 0 and d["allreduce_mb"] > 20 and d["allreduce_per_step"] >= 1
    assert len(d["per_rank_allreduce_ms"]) == 2
    if launch == "hipgraph":
        assert d["allreduce_per_step"] == 1 and 0 < d["per_rank_no_comm_ms"][0] <= d["per_rank_no_comm_ms"][1]
        assert 0.2 < d["eff_vs_rank_program"] < 1.5, d            # (two ranks time-slicing one GPU over gloo: not a measurement)
    else:
        assert d["eff_vs_rank_program"] is None


def test_two_rank_gradients_equal_single_process_on_the_real_network(gpu, tmp_path):
    """Two ranks x 4 clouds against one process x 8 clouds on a small separable-SO3 backbone with InstanceNorm (statistics
    are per cloud, so the split is exact; BatchNorm statistics are per replica, as in the reference's DataParallel): the
    flat gradient bucket of rank 0 after the hook-driven all-reduce equals the single-process gradients at 1e-5.  Eager
    path with the skip branch on the side stream (EPN_SKIP_STREAM=1): the collective of a bucket must wait for gradients
    produced on BOTH streams.  Ranks share the test box's one GPU and talk over gloo (control flow, not a measurement)."""
    import torch
    from epn_pointcloud_amd import dp
    worker = [sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), str(tmp_path)]
    env = dict(os.environ, EPN_DP_SHARE_GPU="1", EPN_DP_BACKEND="gloo", EPN_SKIP_STREAM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "EPN_DP_CHILD"):
        env.pop(k, None)
    assert dp.launch(2, worker, env=env, timeout=600) == 0
    assert dp.launch(1, worker, env=env, timeout=600) == 0
    two = torch.load(tmp_path / "w2.pt")["flat"]
    one = torch.load(tmp_path / "w1.pt")["flat"]
    assert one.abs().max() > 0
    assert (two - one).abs().max().item() <= 1e-5 * max(1.0, one.abs().max().item())

