"""CPU stand-in for `nestfit_amd.comm.RcclComm` (test infrastructure): the same communicator
interface over torch.distributed's gloo backend, so that the N > 1 logic of the product (stripes,
the padded all-gather of per-pixel records, reductions) runs on hosts without a GPU."""
import numpy as np


class GlooComm:
    def __init__(self):
        import torch.distributed as dist
        self._dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allgather(self, x):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).ravel().copy())
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t)
        return np.concatenate([o.numpy() for o in out])

    def allreduce(self, x, op='sum'):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).copy())
        ops = {'sum': self._dist.ReduceOp.SUM, 'max': self._dist.ReduceOp.MAX, 'min': self._dist.ReduceOp.MIN}
        self._dist.all_reduce(t, op=ops[op])
        return t.numpy()

    def barrier(self):
        self._dist.barrier()

    def close(self):
        pass
