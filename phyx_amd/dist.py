"""One process per GPU: rendezvous, per-step barrier and max-over-ranks reduction for bench.py.

torch.distributed is plumbing only (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in the CPU
tests).  The hot path shards by island, so no body or joint data ever crosses ranks; the only collectives
are the per-step barrier (a 4-byte all-reduce, BASELINE.json north_star) and the final reduction of the
timing / unit counters.
"""
import os


class Single:
    """world_size == 1: no torch import, no collective."""
    rank, world_size, local_rank = 0, 1, 0

    def barrier(self):
        pass

    def step_barrier(self):
        pass

    def reduce_max(self, x):
        return float(x)

    def reduce_sum(self, x):
        return float(x)

    def stream_hook(self, stream_ptr):
        return None

    def shutdown(self):
        pass


class Group:
    def __init__(self, backend):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ["RANK"])
        self.world_size = int(os.environ["WORLD_SIZE"])
        self.local_rank = int(os.environ.get("LOCAL_RANK", self.rank))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        self.backend = backend
        if backend == "nccl":
            torch.cuda.set_device(self.local_rank)
            self.device = torch.device("cuda", self.local_rank)
        else:
            self.device = torch.device("cpu")
        dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world_size)
        self._flag = torch.zeros(1, dtype=torch.int32, device=self.device)

    def _sync(self):
        if self.backend == "nccl":
            self.torch.cuda.synchronize(self.device)

    def barrier(self):
        self.dist.barrier()
        self._sync()

    def step_barrier(self):
        """The per-step exchange of the island-sharded solve: one 4-byte all-reduce (done / error flag)."""
        self._flag.zero_()
        self.dist.all_reduce(self._flag, op=self.dist.ReduceOp.MAX)
        self._sync()
        return int(self._flag.item())

    def stream_hook(self, stream_ptr):
        """hook(step, phase) for Solver.bench: the per-step exchange (a 4-byte all-reduce) lives ON THE SOLVER'S OWN STREAM.
        Phase 0 (step queued) starts it asynchronously behind the step; phase 1 (the next step's local preparation is
        queued, its sweeps are not) makes the stream wait for it — so the exchange overlaps the preparation, the sweeps
        of step s+1 start only after step s of every rank, and the host never blocks.
        (gloo has no streams: there phase 0 is the blocking all-reduce.)"""
        if self.backend != "nccl":
            return lambda step, phase: self.step_barrier() if phase == 0 else None
        ext = self.torch.cuda.ExternalStream(stream_ptr, device=self.device)
        flag = self.torch.zeros(1, dtype=self.torch.int32, device=self.device)
        self.torch.cuda.synchronize(self.device)
        pending = []

        def hook(step, phase):
            with self.torch.cuda.stream(ext):
                if phase == 0:
                    pending.append(self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX, async_op=True))
                else:
                    while pending:
                        pending.pop(0).wait()          # the solver's stream waits for the exchange; the host does not
        self._hook_keep = (ext, flag, pending)
        return hook

    def _reduce(self, x, op):
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=op)
        self._sync()
        return float(t.item())

    def reduce_max(self, x):
        return self._reduce(x, self.dist.ReduceOp.MAX)

    def reduce_sum(self, x):
        return self._reduce(x, self.dist.ReduceOp.SUM)

    def shutdown(self):
        try:
            self.dist.destroy_process_group()
        except Exception:
            pass


def init(n_gpus, backend="nccl", force=False):
    """Returns the process group wrapper; a plain single-process object when WORLD_SIZE is 1/unset."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws == 1 and not force:
        if n_gpus != 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ..." % (n_gpus, n_gpus, n_gpus))
        return Single()
    if ws != n_gpus and not force:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (ws, n_gpus))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    return Group(backend)


def shard_columns(total_columns, rank, world_size):
    """Contiguous slab of stack columns (= islands) owned by `rank`: (first_column, column_count)."""
    base, extra = divmod(total_columns, world_size)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)
