"""N > 1 path on CPU: world_size-2 gloo processes exercise the batch sharding and the bucketed gradient
all-reduce that bench.py uses over RCCL (epn_pointcloud_amd/dp.py).  No GPU, no HIP library involved."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from epn_pointcloud_amd import dp
    r, lr, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    if rank == 1:                      # replicas diverge on purpose; broadcast must repair it
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    dp.broadcast_parameters(model)
    data = torch.arange(7 * 6, dtype=torch.float32).view(7, 6) / 10.0
    lo, hi = dp.shard_batch(7, rank, world)
    # gradients as views of one flat buffer, no hooks, one all-reduce per bucket issued by finish() (the accumulate form
    # without overlap), with the collective phase timed the way bench.py reports it for world > 1
    layers = [m for m in model if isinstance(m, torch.nn.Linear)]
    gb = dp.GradBuckets([list(layers[1].parameters()), list(layers[0].parameters())], world, hooks=False, collect="accumulate")
    gb.record_timing = True
    gb.zero()
    loss = model(data[lo:hi]).square().sum() / 7.0 * world  # global-mean loss, shard-local sum; finish() averages over ranks
    loss.backward()
    nb = gb.finish()
    ms = gb.collective_ms()
    assert len(ms) == 1 and ms[0] >= 0.0 and gb.timings == []
    torch.save({"grads": [p.grad.clone() for p in model.parameters()], "buckets": nb, "shard": (lo, hi)},
               os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["shard"] == (0, 4) and r1["shard"] == (4, 7)
    assert r0["buckets"] == 2                                 # one all-reduce per bucket
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    data = torch.arange(7 * 6, dtype=torch.float32).view(7, 6) / 10.0
    (model(data).square().sum() / 7.0).backward()
    for g0, g1, p in zip(r0["grads"], r1["grads"], model.parameters()):
        assert torch.equal(g0, g1)
        assert torch.allclose(g0, p.grad, atol=1e-6)


def test_launcher_gradbuckets_two_ranks(tmp_path):
    """The launcher bench.py uses when started without torchrun (dp.launch: N rank processes with the torchrun
    environment) + GradBuckets: gradients are views of one flat buffer, each stage's all-reduce is issued from a
    backward hook, two consecutive steps, results equal to the single-process gradients."""
    import sys as _sys
    from epn_pointcloud_amd import dp
    rc = dp.launch(2, [_sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)], timeout=600)
    assert rc == 0
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["collectives"] == 2 and r0["views"]            # one all-reduce per stage bucket, no copies
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    data = torch.arange(7 * 6, dtype=torch.float32).view(7, 6) / 10.0
    for step in range(2):
        model.zero_grad()
        (model(data + step).square().sum() / 7.0).backward()
        for g0, g1, p in zip(r0["grads"][step], r1["grads"][step], model.parameters()):
            assert torch.equal(g0, g1)
            assert torch.allclose(g0, p.grad, atol=1e-6)


def test_launcher_gradbuckets_pack_form_two_ranks(tmp_path):
    """The form bench.py's replayed step uses (round 5): autograd hands fresh gradient tensors to p.grad (what a single rank
    does), GradBuckets.pack() gathers them into the flat buffer with one multi-tensor copy, finish(one_collective=True)
    averages the WHOLE buffer with one all-reduce; p.grad are views of the flat buffer afterwards (what Adam reads)."""
    import sys as _sys
    from epn_pointcloud_amd import dp
    rc = dp.launch(2, [_sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path), "pack"], timeout=600)
    assert rc == 0
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["collectives"] == 1 and r0["views"]
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    data = torch.arange(7 * 6, dtype=torch.float32).view(7, 6) / 10.0
    for step in range(2):
        model.zero_grad()
        (model(data + step).square().sum() / 7.0).backward()
        for g0, g1, p in zip(r0["grads"][step], r1["grads"][step], model.parameters()):
            assert torch.equal(g0, g1)
            assert torch.allclose(g0, p.grad, atol=1e-6)


def test_gradbuckets_pack_single_process():
    """pack(): parameters without a gradient contribute zeros, a second step does not accumulate onto the first, and a world
    of one issues no collective unless forced (no process group exists here)."""
    from epn_pointcloud_amd import dp
    torch.manual_seed(1)
    a, b, unused = torch.nn.Linear(4, 3), torch.nn.Linear(3, 2), torch.nn.Linear(2, 2)
    gb = dp.GradBuckets([list(b.parameters()), list(a.parameters()) + list(unused.parameters())], 1, hooks=False, collect="pack")
    x = torch.randn(5, 4)
    for step in range(2):
        gb.zero()
        assert all(p.grad is None for p in a.parameters())
        b(a(x + step)).sum().backward()
        want = [p.grad.clone() for p in list(b.parameters()) + list(a.parameters())]
        gb.pack()
        assert gb.finish(one_collective=True) == 0
        got = [p.grad for p in list(b.parameters()) + list(a.parameters())]
        assert all(torch.equal(w, g) for w, g in zip(want, got))
        assert all(p.grad.data_ptr() >= gb.flat.data_ptr() for p in gb.params)
        assert all(float(p.grad.abs().sum()) == 0.0 for p in unused.parameters())
    assert gb.flat.numel() == sum(p.numel() for p in gb.params)


def test_shard_batch_covers_everything():
    from epn_pointcloud_amd import dp
    for gb in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [dp.shard_batch(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_launcher_stops_the_other_ranks_when_one_dies(tmp_path):
    """dp.launch polls its children: rank 1 exits with code 3 while rank 0 would block for a minute (as a rank stuck in a
    collective would); the launcher terminates rank 0 and returns 3 within seconds.  The watchdog path: every rank hangs,
    timeout=2 -> 124."""
    import sys as _sys
    import time
    from epn_pointcloud_amd import dp
    prog = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "open(os.path.join(sys.argv[1], f'started{r}'), 'w').close()\n"
            "sys.exit(3) if r == 1 and sys.argv[2] == 'die' else time.sleep(60)\n")
    t0 = time.monotonic()
    rc = dp.launch(2, [_sys.executable, "-c", prog, str(tmp_path), "die"], timeout=600)
    assert rc == 3 and time.monotonic() - t0 < 30
    assert (tmp_path / "started0").exists() and (tmp_path / "started1").exists()
    t0 = time.monotonic()
    rc = dp.launch(2, [_sys.executable, "-c", prog, str(tmp_path), "hang"], timeout=2)
    assert rc == 124 and time.monotonic() - t0 < 30
