"""Rank program of tests/test_dist_cpu.py::test_launcher_gradbuckets_two_ranks: started by epn_pointcloud_amd.dp.launch
(the self-spawning path of `python bench.py --gpus N`), gloo on CPU.  argv: output directory [pack] -- "pack" runs the form
bench.py's replayed step uses: no hooks, gradients gathered by GradBuckets.pack(), ONE all-reduce of the flat buffer."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from epn_pointcloud_amd import dp  # noqa: E402


def main():
    out = sys.argv[1]
    rank, _, world = dp.init_from_env(backend="gloo")
    torch.manual_seed(0)                                     # same seed: replicas start identical (as bench.py does)
    stage0 = torch.nn.Linear(6, 5)
    stage1 = torch.nn.Linear(5, 3)
    model = torch.nn.Sequential(stage0, torch.nn.ReLU(), stage1)
    dp.broadcast_parameters(model)
    # buckets in the order backward completes them: last stage first
    pack = len(sys.argv) > 2 and sys.argv[2] == "pack"
    gb = (dp.GradBuckets([list(stage1.parameters()), list(stage0.parameters())], world, hooks=False, collect="pack") if pack
          else dp.GradBuckets([list(stage1.parameters()), list(stage0.parameters())], world, hooks=True))
    data = torch.arange(7 * 6, dtype=torch.float32).view(7, 6) / 10.0
    lo, hi = dp.shard_batch(7, rank, world)
    grads = []
    for step in range(2):                                    # two steps: zero() / hook re-arming
        gb.zero()
        loss = model(data[lo:hi] + step).square().sum() / 7.0 * world     # finish() averages over ranks
        loss.backward()
        gb.pack()                                            # (no-op in the accumulate form)
        n = gb.finish(one_collective=pack)
        grads.append([p.grad.clone() for p in model.parameters()])
    torch.save({"grads": grads, "collectives": n, "views": all(p.grad.data_ptr() >= gb.flat.data_ptr() for p in
                                                                model.parameters())}, os.path.join(out, f"r{rank}.pt"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
