"""
TEST-ONLY in-process collective: lets several HIP_Backend objects act as the ranks of one sample-sharded job inside ONE
process on ONE GPU (each rank on its own Python thread, its own library context).  It implements the small interface
HIP_Backend accepts in place of a torch.distributed group -- ``rank``, ``world_size``, ``all_reduce_sum(tensor)``, ``all_gather(tensor)`` -- with
a fixed-rank-order sum, so the ranks' results are bit-identical to each other like RCCL's are.
"""
import threading

import torch


class LocalGroup:
    def __init__(self, world_size: int):
        self.world_size = world_size
        self._slots = [None] * world_size
        self._barrier = threading.Barrier(world_size)
        self.calls = 0

    def member(self, rank: int) -> 'LocalRank':
        return LocalRank(self, rank)


class LocalRank:
    def __init__(self, group: LocalGroup, rank: int):
        self.group, self.rank, self.world_size = group, rank, group.world_size

    def all_reduce_sum(self, t: torch.Tensor) -> None:
        g = self.group
        g._slots[self.rank] = t.detach().clone()
        g._barrier.wait()                    # every rank has deposited its buffer
        total = g._slots[0].clone()
        for r in range(1, g.world_size):     # same order on every rank
            total += g._slots[r]
        t.copy_(total)
        if self.rank == 0:
            g.calls += 1
        g._barrier.wait()                    # nobody overwrites a slot that is still being read


    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """The ranks' tensors one behind the other, in rank order."""
        g = self.group
        g._slots[self.rank] = t.detach().clone()
        g._barrier.wait()
        out = torch.stack([g._slots[r] for r in range(g.world_size)])
        if self.rank == 0:
            g.calls += 1
        g._barrier.wait()
        return out


def run_ranks(world_size: int, fn):
    """Runs fn(rank, collective) on one thread per rank; returns the results in rank order, re-raises the first error."""
    group = LocalGroup(world_size)
    out, err = [None] * world_size, [None] * world_size

    def body(rank):
        try:
            out[rank] = fn(rank, group.member(rank))
        except BaseException as exc:  # noqa: BLE001
            err[rank] = exc
            group._barrier.abort()

    threads = [threading.Thread(target=body, args=(r,)) for r in range(world_size)]
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
    return out, group
