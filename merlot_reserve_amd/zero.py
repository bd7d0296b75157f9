"""Adam moments partitioned over the data-parallel ranks ("Adam sharding": pretrain/train_fixres.py:178-199,
finetune/optimization.py:148-171).

The reference slices every gradient leaf eight ways along its first axis, lets device i of a group of eight keep and
update only slice i of mu / nu, all-gathers the resulting UPDATES (fp32) inside the group and applies weight decay,
schedule and learning rate to the replicated fp32 parameters.  Same contract here, on the flat buffers of params.py:

  * the partition follows the gradient BUCKETS (trainer.Trainer._make_buckets): bucket [lo, hi) is cut into `world` chunks of
    c = ceil((hi - lo) / world / 2048) * 2048 elements and rank r owns [lo + r c, min(lo + (r + 1) c, hi)) -- every rank has
    optimizer work for every bucket, so the per-bucket overlap with backward keeps its balance;
  * mu and nu exist only for the owned ranges (2 x 2 B / parameter / world instead of 2 x 2 B / parameter);
  * after the bucket's all-reduce each rank runs the fused Adam chain on its chunk (fp32 master + bf16 working copy of that
    chunk), then the chunks of the fp32 MASTER are all-gathered (4 B / parameter: what the reference moves as updates), and the
    bf16 working copy of the whole bucket is re-derived from it -- parameters stay replicated and valid on every rank, so
    checkpoints, `state.params` and evaluation need no gather.

Arithmetic per element is the replicated chain's, so a sharded run equals the replicated one bit for bit
(tests/test_dist_gpu.py::test_two_rank_sharded_adam_equals_replicated).  The moments of a checkpoint are the gathered ones
(`full_moments`), a loaded checkpoint is cut back into the shards (`store_moments`): files are interchangeable between modes.
"""
import torch

from . import ops

ALIGN = 2048


class MomentShards:
    def __init__(self, params, buckets, comm):
        """buckets: [(key, lo, hi)] covering the flat buffers; comm: dist.Comm / dist.NativeComm (world >= 1)."""
        self.p, self.comm = params, comm
        W, r = comm.world, comm.rank
        self.table, off, cmax = {}, 0, 0
        for key, lo, hi in buckets:
            assert lo % ALIGN == 0 and hi % ALIGN == 0 and lo < hi
            c = -(-(hi - lo) // (W * ALIGN)) * ALIGN
            mlo = min(lo + r * c, hi)
            mhi = min(mlo + c, hi)
            self.table[key] = (lo, hi, c, mlo, mhi, off)
            off += mhi - mlo
            cmax = max(cmax, c)
        dev = params.device
        self.owned = off
        self.mu = torch.zeros(max(off, 1), dtype=torch.bfloat16, device=dev)
        self.nu = torch.zeros(max(off, 1), dtype=torch.bfloat16, device=dev)
        self.send = torch.zeros(cmax, dtype=torch.float32, device=dev)
        self.recv = torch.zeros(W * cmax, dtype=torch.float32, device=dev)
        if getattr(params, 'mu', None) is not None:          # a store built with full moments: carry them over, then release them
            self.store_moments(params.mu, params.nu)
        params.mu = params.nu = None

    def update(self, key, adam):
        """The bucket's averaged gradients are final: `adam(mlo, mhi, mu, nu)` on the owned chunk, then the bucket's parameters
        from every owner.  Enqueued on the current stream; capturable with the library's RCCL communicator."""
        lo, hi, c, mlo, mhi, off = self.table[key]
        p, W, n = self.p, self.comm.world, mhi - mlo
        if n > 0:
            adam(mlo, mhi, self.mu[off:off + n], self.nu[off:off + n])
        if W > 1:
            send = self.send[:c]
            if n > 0:
                send[:n].copy_(p.master[mlo:mhi])
            self.comm.allgather_flat(send, self.recv[:W * c])
            p.master[lo:hi].copy_(self.recv[:hi - lo])
            if p.master.is_cuda:
                ops.cast_params(p.master[lo:hi], p.work[lo:hi])
            else:                                    # (host tensors: the world-size-2 gloo test of the partition logic)
                p.work[lo:hi].copy_(p.master[lo:hi].to(torch.bfloat16))
        p.update_transposed(lo, hi)

    # ---- checkpoint form ----
    def full_moments(self):
        """(mu, nu) of the whole model on the host: a COLLECTIVE (every rank calls it)."""
        p, W = self.p, self.comm.world
        out = [torch.zeros(p.total, dtype=torch.bfloat16) for _ in range(2)]
        send16, recv16 = self.send.view(torch.bfloat16), self.recv.view(torch.bfloat16)
        for full, shard in zip(out, (self.mu, self.nu)):
            for lo, hi, c, mlo, mhi, off in self.table.values():
                n = mhi - mlo
                if W == 1:
                    full[lo:hi] = shard[off:off + n].cpu()
                    continue
                send = send16[:c]
                send.zero_()
                if n > 0:
                    send[:n].copy_(shard[off:off + n])
                self.comm.allgather_flat(send, recv16[:W * c])
                full[lo:hi] = recv16[:hi - lo].cpu()
        return out[0], out[1]

    def store_moments(self, mu_full, nu_full):
        for shard, full in ((self.mu, mu_full), (self.nu, nu_full)):
            for lo, hi, c, mlo, mhi, off in self.table.values():
                if mhi > mlo:
                    shard[off:off + mhi - mlo].copy_(full[mlo:mhi])
