"""Rank program of tests/test_gpu_dist.py::test_two_rank_gradients_equal_single_process_on_the_real_network: one training
step of a small separable-SO3 backbone (InstanceNorm: per-cloud statistics, as the rotation / 3DMatch models use) on this
rank's shard of 8 clouds, gradients all-reduced through dp.GradBuckets from backward hooks (the eager multi-GPU path:
the skip branch of every block runs on the library's side stream).  Started by dp.launch; world = 1 gives the reference.
argv: output directory."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import epn_pointcloud_amd  # noqa: E402
from epn_pointcloud_amd import dp, schedule as S  # noqa: E402


def main():
    out = sys.argv[1]
    rank, local_rank, world = dp.init_from_env()
    dev = dp.local_device(local_rank)
    torch.cuda.set_device(dev)
    epn_pointcloud_amd.install_vgtk_alias()
    torch.manual_seed(7)
    layers = S.scaled(S.cls_so3net_schedule(256)[:4], 4)          # 1 -> 16 -> 16 -> 32 -> 32, two stages
    model = S.HotPathBackbone(layers, norm=None, model="reg").to(dev).train()
    dp.broadcast_parameters(model)
    pts = S.synthetic_clouds(8, 256, dev, seed=99)
    lo, hi = dp.shard_batch(8, rank, world)
    gb = dp.GradBuckets(dp.stage_buckets(model), max(world, 1), hooks=True)
    gb.zero()
    x = model(pts[lo:hi])
    loss = x.feats.float().square().sum() / (8.0 * x.feats[0].numel()) * world       # finish() averages over ranks
    loss.backward()
    gb.finish()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"flat": gb.flat.cpu(), "world": world}, os.path.join(out, f"w{world}.pt"))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
