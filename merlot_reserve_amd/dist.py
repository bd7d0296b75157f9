"""Data-parallel collectives of the pretraining step: one process per GPU.  Four exchanges per step:
  gather_embeddings  all-gather of the packed contrastive embeddings, rank-major  (pretrain_model.py:290)
  scatter_grad       its transpose: reduce-scatter(sum) of dL/dE_all back to the owning rank
  allreduce_mean     mean of a bf16 gradient bucket over ranks                      (pretrain_model.py:329)
  allreduce_mean_f32 mean of the fp32 metrics                                       (pretrain_model.py:336)

Two transports with the same methods; every method enqueues on torch's CURRENT stream and orders itself after the work
already on it:
  NativeComm  the library's own RCCL communicator (mr_comm_* of include/mreserve_hip.h) over xGMI.  Its calls are plain
              stream-ordered launches, so they are captured into the step's hipGraph with the kernels around them:
              what bench.py uses on the GPUs.
  Comm        torch.distributed ("gloo" for the CPU tests and for two test processes sharing one GPU; "nccl" as a
              fallback).  Not capturable: the step then runs eagerly.
"""
import ctypes as C

import torch
import torch.distributed as dist


class Comm:
    capturable = False

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

    def allreduce_mean_f32(self, t):
        return self.allreduce_mean(t)

    def allgather_flat(self, src, dst):
        """dst [world * n] = every rank's src [n], rank-major (any dtype): the parameter chunks of zero.MomentShards."""
        assert dst.numel() == self.world * src.numel() and dst.dtype == src.dtype
        dist.all_gather_into_tensor(dst, src.contiguous(), group=self.group)
        return dst

    def close(self):
        pass


class NativeComm:
    """RCCL communicator owned by libmreserve_hip.so (one per process, bound to the current device)."""
    capturable = True
    backend = 'rccl-native'

    def __init__(self, rank=0, world=1, unique_id=None, device=None):
        from . import _lib
        self._lib = _lib
        self._h = None
        lib = _lib.load()
        if unique_id is None:
            assert world == 1, 'world > 1: rank 0 creates the id (NativeComm.new_unique_id) and every rank passes the same bytes'
            unique_id = self.new_unique_id()
        assert len(unique_id) == 128
        self.rank, self.world = rank, world
        # the communicator binds to the device that is current inside mr_comm_init: make that the caller's device, and keep it
        # for the collectives (two ranks that both initialise on device 0 hang inside RCCL)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:                    # 'cuda' without an index: the current device, made explicit (the collectives compare indices)
            self.device = torch.device('cuda', torch.cuda.current_device())
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        with torch.cuda.device(self.device):
            _lib.check(lib.mr_comm_init(rank, world, buf, C.byref(h)), 'mr_comm_init')
        self._h = h
        assert lib.mr_comm_world(h) == world and lib.mr_comm_rank(h) == rank

    @staticmethod
    def new_unique_id():
        from . import _lib
        buf = (C.c_char * 128)()
        _lib.check(_lib.load().mr_comm_unique_id(buf), 'mr_comm_unique_id')
        return bytes(buf)

    @classmethod
    def from_torch_distributed(cls, device, group=None):
        """Bootstrap over an initialised torch.distributed group (any backend).  SYMMETRIC: every rank makes the same sequence of
        control-plane collectives whatever fails where, and either every rank returns a communicator or every rank raises --
        (1) rank 0 creates the id and broadcasts (ok, id | error text); (2) every rank initialises; (3) a MIN all-reduce of the
        per-rank outcome.  (A rank-0 failure used to skip the broadcast the other ranks were waiting in.)"""
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        obj_dev = torch.device(device) if dist.get_backend(group) == 'nccl' else None
        # (0) everything that can fail LOCALLY -- loading the library, resolving the device -- happens before any rank enters RCCL's
        # collective initialisation, and its outcome is agreed on first: a rank that dies here must not leave the others inside
        # ncclCommInitRank, from which no control-plane message can call them back
        ready, why = 1, ''
        try:
            from . import _lib
            _lib.load()
            dev_ = torch.device(device)
            if dev_.type != 'cuda' or (dev_.index is not None and dev_.index >= torch.cuda.device_count()):
                raise RuntimeError(f'{device} is not a visible GPU')
        except Exception as e:                                       # noqa: BLE001 -- reported to every rank below
            ready, why = 0, f'{type(e).__name__}: {e}'
        flag0 = torch.tensor([ready], device=obj_dev)
        dist.all_reduce(flag0, op=dist.ReduceOp.MIN, group=group)
        if int(flag0) == 0:
            raise RuntimeError(f'the native RCCL communicator cannot be initialised on some rank (this rank: {why or "ready"})')
        box = [None]
        if rank == 0:
            try:
                box = [(True, cls.new_unique_id())]
            except Exception as e:                                   # noqa: BLE001 -- reported to every rank below
                box = [(False, f'{type(e).__name__}: {e}')]
        dist.broadcast_object_list(box, src=0, group=group, device=obj_dev)
        ok, payload = box[0]
        if not ok:
            raise RuntimeError(f'rank 0 could not create the RCCL unique id: {payload}')
        comm, err = None, ''
        try:
            comm = cls(rank, world, payload, device=device)
        except Exception as e:                                       # noqa: BLE001
            err = f'{type(e).__name__}: {e}'
        flag = torch.tensor([1 if comm is not None else 0], device=obj_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag) == 0:
            if comm is not None:
                comm.close()
            raise RuntimeError(f'RCCL communicator initialisation failed on some rank (this rank: {err or "ok"})')
        return comm

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                                            # noqa: BLE001 -- interpreter shutdown
            pass

    def _stream(self):
        assert torch.cuda.current_device() == self.device.index, f'NativeComm is bound to {self.device}; current device is cuda:{torch.cuda.current_device()}'
        return torch.cuda.current_stream().cuda_stream

    def gather_embeddings(self, E, E_all):
        assert E.is_contiguous() and E_all.is_contiguous() and E.dtype in (torch.bfloat16, torch.float32) and E_all.dtype == E.dtype
        assert E_all.numel() == self.world * E.numel()
        n16 = E.numel() * (E.element_size() // 2)          # an all-gather moves bytes: fp32 rows travel as pairs of 16-bit elements
        self._lib.check(self._lib.load().mr_allgather(self._h, E.data_ptr(), E_all.data_ptr(), n16, self._stream()), 'mr_allgather')
        return E_all

    def scatter_grad(self, dE_all, out):
        assert out.is_contiguous() and dE_all.is_contiguous() and out.dtype == dE_all.dtype and dE_all.numel() == self.world * out.numel()
        if out.dtype == torch.float32:                      # the fp32 training step
            self._lib.check(self._lib.load().mr_reducescatter_sum_f32(self._h, dE_all.data_ptr(), out.data_ptr(), out.numel(), self._stream()),
                            'mr_reducescatter_sum_f32')
            return out
        assert out.dtype == torch.bfloat16
        self._lib.check(self._lib.load().mr_reducescatter_sum(self._h, dE_all.data_ptr(), out.data_ptr(), out.numel(), self._stream()),
                        'mr_reducescatter_sum')
        return out

    def allreduce_mean(self, flat):
        assert flat.is_contiguous()
        if flat.dtype == torch.float32:
            return self.allreduce_mean_f32(flat)
        assert flat.dtype == torch.bfloat16
        self._lib.check(self._lib.load().mr_allreduce_mean_bf16(self._h, flat.data_ptr(), flat.numel(), self._stream()), 'mr_allreduce_mean_bf16')
        return flat

    def allgather_flat(self, src, dst):
        assert src.is_contiguous() and dst.is_contiguous() and dst.dtype == src.dtype and dst.numel() == self.world * src.numel()
        assert src.element_size() % 2 == 0
        n16 = src.numel() * (src.element_size() // 2)      # an all-gather moves bytes: counted in 16-bit elements
        self._lib.check(self._lib.load().mr_allgather(self._h, src.data_ptr(), dst.data_ptr(), n16, self._stream()), 'mr_allgather')
        return dst

    def allreduce_mean_f32(self, t):
        assert t.is_contiguous() and t.dtype == torch.float32
        self._lib.check(self._lib.load().mr_allreduce_mean_f32(self._h, t.data_ptr(), t.numel(), self._stream()), 'mr_allreduce_mean_f32')
        return t

    def close(self):
        if getattr(self, '_h', None) is not None:
            self._lib.load().mr_comm_destroy(self._h)
            self._h = None
