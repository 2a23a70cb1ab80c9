"""In-process stand-in for the process group of tangram_amd.sharded (TEST INFRASTRUCTURE): several shards of one problem
run as threads of one process on ONE device and meet at the three exchange points of a step.  Reductions sum the ranks'
buffers in rank order (a fixed order, like a ring all-reduce is for a fixed topology)."""
import threading

import torch


class LocalGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class LocalComm:
    same_process = True          # (peer transport: the ranks exchange raw device pointers instead of hipIpc handles)

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def barrier(self):
        self.g.barrier.wait()

    def _sync(self, t):
        if t.is_cuda:
            torch.cuda.synchronize(t.device)

    def all_reduce(self, t):
        self.g.slots[self.rank] = t
        self._sync(t)
        self.g.barrier.wait()
        total = self.g.slots[0].clone()
        for r in range(1, self.world):
            total += self.g.slots[r]
        self._sync(t)
        self.g.barrier.wait()            # everybody has read every slot
        t.copy_(total)
        self._sync(t)
        self.g.barrier.wait()

    def all_gather_into_tensor(self, out, t):
        self.g.slots[self.rank] = t
        self._sync(t)
        self.g.barrier.wait()
        out.copy_(torch.cat([s.reshape(-1) for s in self.g.slots]))
        self._sync(t)
        self.g.barrier.wait()

    def all_gather(self, outs, t):
        self.g.slots[self.rank] = t
        self._sync(t)
        self.g.barrier.wait()
        for r in range(self.world):
            outs[r].copy_(self.g.slots[r])
        self._sync(t)
        self.g.barrier.wait()


def run_ranks(world, fn):
    """fn(comm) on `world` threads; returns the list of results in rank order, re-raises the first failure."""
    group = LocalGroup(world)
    out, err = [None] * world, [None] * world

    def work(r):
        try:
            out[r] = fn(LocalComm(group, r))
        except BaseException as e:      # noqa: BLE001 -- unblock the peers, then report
            err[r] = e
            group.barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in err:
        if e is not None:
            raise e
    return out
