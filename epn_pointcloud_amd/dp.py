"""Data-parallel glue for the hot path: one process per GPU, batch sharded across ranks, the ONLY collective
is the gradient all-reduce (RCCL over xGMI; torch backend "nccl").  The reference's nn.DataParallel
(vgtk/vgtk/app/trainer.py:153-160) is replaced by this.  Clouds are independent, so there is no activation
or index exchange.

    launch(...)          self-spawn N ranks when no torchrun environment is present (python bench.py --gpus N)
    init_from_env(...)   torchrun-style env -> process group
    GradBuckets          gradients live in ONE flat buffer cut into per-stage buckets: no cat, no copy-back; a
                         bucket's all-reduce is issued from a backward hook as soon as its last gradient is
                         written, i.e. it overlaps with the rest of the backward pass
"""
import contextlib
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, argv=None, env=None, timeout=None):
    """Run `argv` (default: this very command line) as `world` rank processes on this node, one per GPU, with the
    torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT).  Rank 0 inherits
    stdout.  The children are POLLED: the first one that exits non-zero (or the watchdog: `timeout` seconds; None = no
    limit, which is the default -- a stand-in for torchrun must not kill a long job; bench.py and the tests pass one)
    gets the others terminated -- a dead rank must not leave its peers blocked inside a collective
    until some outer limit kills the job -- and its exit code is returned (124 for the watchdog); 0 when all ranks succeed.
    What `python -m torch.distributed.run --nproc-per-node N` would do, for callers that start the program without a
    launcher."""
    import time
    argv = list(argv) if argv is not None else [sys.executable] + sys.argv
    timeout = float("inf") if timeout is None else float(timeout)
    port = free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 EPN_DP_CHILD="1")
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen(argv, env=e, stdout=None if r == 0 else subprocess.DEVNULL))

    def stop_all():
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()                               # exactly the processes started above, never a pattern
                p.wait()

    deadline = time.monotonic() + timeout
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            r, c = bad[0]
            print(f"[dp.launch] rank {r} exited with code {c}; stopping the other ranks", file=sys.stderr)
            stop_all()
            return abs(c) or 1
        if all(c == 0 for c in codes):
            return 0
        if time.monotonic() > deadline:
            print(f"[dp.launch] watchdog: ranks still running after {timeout:.0f} s; stopping them", file=sys.stderr)
            stop_all()
            return 124
        time.sleep(0.05)


@contextlib.contextmanager
def _native_stdout_to_stderr(active=True):
    """File descriptor 1 points at stderr inside the block (C stdio flushed on both edges): what native libraries print to
    stdout while a communicator is created lands on stderr."""
    if not active:
        yield
        return
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def env_world():
    """(rank, local_rank, world) from the torchrun environment, without touching torch.distributed."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_from_env(backend=None, force=False):
    """torchrun-style env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT) -> (rank, local_rank, world).
    The process group is created for world > 1 (or world == 1 with force=True: a single-rank RCCL communicator, used
    to exercise the nccl branch on one GPU)."""
    rank, local_rank, world = env_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # EPN_DP_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks (see local_device)
            backend = os.environ.get("EPN_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # (a forced single-rank group rendezvouses with itself: any free port; real launchers always set one)
        os.environ.setdefault("MASTER_PORT", str(free_port()) if world == 1 else "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)                       # one process per GPU, bound before RCCL init
            kwargs["device_id"] = torch.device("cuda", local_rank)
        with _native_stdout_to_stderr(backend == "nccl"):
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
            if backend == "nccl":
                # create the communicator NOW (a first collective), while stdout is parked: RCCL prints a five-line version
                # banner to C stdout when it does, and a program whose stdout is parsed (bench.py: one JSON line) must not
                # carry it -- block-buffered C stdio would even emit it at exit, after everything else
                t = torch.zeros(1, device=torch.device("cuda", local_rank))
                dist.all_reduce(t)
                torch.cuda.synchronize()
    return rank, local_rank, world


def local_device(local_rank):
    """cuda:<local_rank>; with EPN_DP_SHARE_GPU=1 every rank uses cuda:0 (a rehearsal of the multi-rank control flow --
    launcher, capture before communicator, broadcast, bucketed all-reduce, max-over-ranks timing -- on a one-GPU box,
    with EPN_DP_BACKEND=gloo since RCCL refuses two ranks on one device; never a measurement)."""
    return torch.device("cuda", 0 if os.environ.get("EPN_DP_SHARE_GPU") == "1" else local_rank)


def shard_batch(global_batch, rank, world):
    """Contiguous, balanced shard [start, stop) of `global_batch` clouds for `rank`."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


class GradBuckets:
    """All gradients of `params` as views into one flat fp32 buffer, cut into buckets (given as lists of parameters,
    e.g. one per backbone stage, in the order backward produces them last-to-first).

    * no concatenation and no copy-back: autograd accumulates straight into the views (`p.grad` is pre-set, so the
      AccumulateGrad nodes add in place; call zero() at the start of a step);
    * overlap: with hooks=True a post-accumulate hook counts a bucket's parameters down and issues its asynchronous
      all-reduce the moment the last one is written -- the deeper stages' backward kernels keep running underneath;
      finish() waits for all of them and applies the 1/world average.  With hooks=False (a step replayed as one HIP
      graph has no host-side hook points) finish() issues the all-reduces itself: the whole cls model is 31 MB, ~0.4 ms
      over xGMI against a >100 ms step.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): few large buckets, not NCCL-style 25 MB chunks."""

    def __init__(self, buckets, world, hooks=True, collect="accumulate", force_collectives=False):
        """collect="accumulate": `p.grad` are views of the flat buffer for the whole step, autograd's AccumulateGrad nodes
        add into them (one small kernel per parameter + the zero fill of zero()).  collect="pack": autograd keeps handing
        fresh gradient tensors to `p.grad` (the single-rank program, no accumulate kernels) and pack() gathers them into
        the flat buffer with ONE multi-tensor copy (torch._foreach_copy_) at the end of backward -- graph-capturable; the
        form bench.py's replayed step uses.  force_collectives=True issues the all-reduces on a world of one as well (the
        rank program measured on the single GPU a bench box has)."""
        assert collect in ("accumulate", "pack")
        self.world = world
        self.collect = collect
        self.force = bool(force_collectives)
        self.buckets = [[p for p in b if p.requires_grad] for b in buckets]
        self.buckets = [b for b in self.buckets if b]
        params = [p for b in self.buckets for p in b]
        if not params:
            raise ValueError("no trainable parameters")
        dev, total = params[0].device, sum(p.numel() for p in params)
        self.params = params
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.spans, self.views, off = [], [], 0
        for b in self.buckets:
            start = off
            for p in b:
                self.views.append(self.flat[off:off + p.numel()].view_as(p))
                if collect == "accumulate":
                    p.grad = self.views[-1]
                off += p.numel()
            self.spans.append((start, off))
        self._pending = [len(b) for b in self.buckets]
        self._works = []
        self._handles = []
        self.record_timing = False      # finish(): keep (start, end) of every step's collective phase in self.timings
        self.timings = []
        hooks = hooks and collect == "accumulate" and (world > 1 or self.force)
        if hooks:
            for bi, b in enumerate(self.buckets):
                for p in b:
                    self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))
        self.hooks = hooks

    def _make_hook(self, bi):
        def hook(_p):
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self._issue(bi)
        return hook

    def _issue(self, bi):
        a, b = self.spans[bi]
        if self.flat.is_cuda:
            # ProcessGroupNCCL orders the collective after the stream that is current HERE only; a block's skip branch
            # runs (and accumulates its gradients) on the library's side stream (schedule.FusedSeparableBlock,
            # EPN_SKIP_STREAM), so gradients of this very bucket may still be in flight there
            from . import ops
            side = ops.side_stream_if_any(self.flat.device)
            if side is not None:
                torch.cuda.current_stream(self.flat.device).wait_stream(side)
        self._works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def zero(self):
        """Start of a step.  accumulate: zero the flat buffer (autograd adds into its views); pack: drop the gradient
        tensors so that autograd hands over fresh ones (no fill, no accumulate kernels)."""
        if self.collect == "pack":
            for p in self.params:
                p.grad = None
        else:
            self.flat.zero_()
        self._pending = [len(b) for b in self.buckets]

    def pack(self):
        """collect="pack", end of backward: the step's gradient tensors -> the flat buffer in one multi-tensor copy, and
        `p.grad` := the views (what the all-reduce averages and the optimizer reads).  A parameter that received no
        gradient this step contributes zeros.  No-op in the accumulate form."""
        if self.collect != "pack":
            return
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(g if g.dtype == v.dtype else g.to(v.dtype))
            p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)

    def _span_all(self):
        return [(self.spans[0][0], self.spans[-1][1])]

    def finish(self, one_collective=False):
        """Complete this step's all-reduces (issuing them first when no hooks ran) and average.  one_collective=True (no
        hooks: nothing to overlap with) sends the whole flat buffer as ONE all-reduce -- xGMI is point-to-point, a ring
        step costs its latency per collective.  Returns the number of collectives of the step."""
        if self.world <= 1 and not self.force:
            return 0
        rec = None
        if self.record_timing:
            # what the step pays for its collectives as the COMPUTE stream sees it: from the moment they are issued (after the
            # side stream's gradients have landed) to the moment the stream may go on -- device time on CUDA tensors (events
            # on the current stream), host time otherwise.  bench.py turns it into ms and bus bandwidth per step.
            if self.flat.is_cuda:
                rec = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                rec[0].record(torch.cuda.current_stream(self.flat.device))
            else:
                import time
                rec = time.perf_counter()
        if not self.hooks:
            if one_collective:
                saved, self.spans = self.spans, self._span_all()
                try:
                    self._issue(0)
                finally:
                    self.spans = saved
            else:
                for bi in range(len(self.buckets)):
                    self._issue(bi)
        else:
            for bi, n in enumerate(self._pending):      # a bucket whose parameters got no gradient this step
                if n > 0:
                    self._issue(bi)
        n = len(self._works)
        for w in self._works:
            w.wait()
        self._works = []
        if rec is not None:
            if self.flat.is_cuda:
                rec[1].record(torch.cuda.current_stream(self.flat.device))
                self.timings.append(rec)
            else:
                import time
                self.timings.append((time.perf_counter() - rec) * 1e3)
        if self.world > 1 or self.force:
            self.flat.div_(self.world)
        self._pending = [len(b) for b in self.buckets]
        return n

    def collective_ms(self):
        """Milliseconds each recorded finish() spent between issuing its collectives and being allowed to continue (needs the
        device to be idle: synchronises); clears the record."""
        out = []
        for t in self.timings:
            if isinstance(t, tuple):
                t[1].synchronize()
                out.append(t[0].elapsed_time(t[1]))
            else:
                out.append(t)
        self.timings = []
        return out

    def remove_hooks(self):
        for h in self._handles:
            h.remove()
        self._handles, self.hooks = [], False


def stage_buckets(model):
    """Bucket layout for the shipped networks: the head first, then the backbone stages last-to-first -- the order in
    which backward completes them -- so every bucket's all-reduce starts as early as possible."""
    buckets, seen = [], set()

    def take(mod):
        ps = [p for p in mod.parameters() if id(p) not in seen]
        seen.update(id(p) for p in ps)
        if ps:
            buckets.append(ps)

    if hasattr(model, "outblock"):
        take(model.outblock)
    if hasattr(model, "backbone"):
        for stage in reversed(list(model.backbone)):
            take(stage)
    take(model)                                   # whatever is left
    return buckets


def broadcast_parameters(module, src=0):
    """Replicas must start identical (DataParallel replicates from device 0 every step)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
