"""Data-parallel collectives of the pretraining step: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).  Three exchanges per step:
  gather_embeddings  all-gather of the packed contrastive embeddings, rank-major  (pretrain_model.py:290)
  scatter_grad       its transpose: reduce-scatter(sum) of dL/dE_all back to the owning rank
  allreduce_mean     mean of the bf16 gradient buffer over ranks                    (pretrain_model.py:329)
"""
import torch
import torch.distributed as dist


class Comm:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def gather_embeddings(self, E, E_all):
        """E [R,H] -> E_all [world,R,H], block r = rank r's E."""
        dist.all_gather_into_tensor(E_all.view(-1), E.reshape(-1), group=self.group)
        return E_all

    def scatter_grad(self, dE_all, out):
        """out [R,H] = sum over ranks of their dE_all[self.rank]."""
        if self.backend == 'nccl':
            dist.reduce_scatter_tensor(out.view(-1), dE_all.view(-1), op=dist.ReduceOp.SUM, group=self.group)
        else:                                   # gloo has no reduce_scatter: all-reduce and keep the own block
            tmp = dE_all.float()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            out.copy_(tmp[self.rank].to(out.dtype))
        return out

    def allreduce_mean(self, flat):
        if self.backend == 'nccl':
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            tmp = flat.float()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
            flat.copy_((tmp / self.world).to(flat.dtype))
        return flat

    def allreduce_mean_async(self, flat):
        """Non-blocking form for gradient buckets: returns a handle with .wait() (nccl: runs on RCCL's stream after the
        producing kernels, overlapping the compute stream), or None when the backend completed it synchronously."""
        if self.backend == 'nccl':
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        self.allreduce_mean(flat)
        return None
