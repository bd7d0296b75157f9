"""The N > 1 path on CPU: world_size 2, gloo.  Covers the three collectives of merlot_reserve_amd/dist.py and the
reference semantics they implement (rank-major all_gather, its transpose, mean of gradients), plus the oracle's
virtual-device loss that the GPU test of the multi-rank engine is checked against."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from merlot_reserve_amd.dist import Comm
    comm = Comm()
    R, H = 10, 16
    g = torch.Generator().manual_seed(100 + rank)
    E = torch.randn(R, H, generator=g).to(torch.bfloat16)
    E_all = torch.zeros(world, R, H, dtype=torch.bfloat16)
    comm.gather_embeddings(E, E_all)
    ok = True
    for r in range(world):
        ref = torch.randn(R, H, generator=torch.Generator().manual_seed(100 + r)).to(torch.bfloat16)
        ok &= torch.equal(E_all[r], ref)                         # block r = rank r (all_gather(...).reshape(-1, H))
    dE_all = torch.randn(world, R, H, generator=g).to(torch.bfloat16)
    out = torch.zeros(R, H, dtype=torch.bfloat16)
    comm.scatter_grad(dE_all, out)
    tot = torch.zeros(R, H)
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        torch.randn(R, H, generator=gr)
        tot += torch.randn(world, R, H, generator=gr).to(torch.bfloat16)[rank].float()
    ok &= torch.allclose(out.float(), tot, atol=2e-2)
    flat = torch.full((64,), float(rank + 1), dtype=torch.bfloat16)
    comm.allreduce_mean(flat)
    ok &= bool((flat.float() == (1 + world) / 2).all())
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def test_collectives_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_virtual_device_loss_matches_single_device_when_replicated():
    """loss_fn_given_preds with 2 virtual devices holding the SAME preds: every positive keeps its logit, the negatives
    are duplicated, so lse grows by exactly log 2 for each direction -> each objective's loss grows by log 2."""
    from oracle import ref_torch as R
    g = torch.Generator().manual_seed(0)
    mk = lambda n: torch.nn.functional.normalize(torch.randn(n, 32, generator=g), dim=-1) * 1.6
    preds = {'imgs_to_audio': {'x': mk(8), 'y': mk(8)},
             'text_to_audio': {'x': mk(6), 'y': mk(6), 'y_extra': mk(18)},
             'stuff_to_span': {'x': mk(16), 'y': mk(16), '_sources': torch.randint(-1, 3, (16,), generator=g)}}
    l1, i1 = R.loss_fn_given_preds([preds])
    for rank in (0, 1):
        l2, i2 = R.loss_fn_given_preds([preds, preds], rank=rank)
        for k in ('imgs_to_audio', 'text_to_audio', 'stuff_to_span'):
            assert abs(float(i2[k]) - float(i1[k]) - float(torch.log(torch.tensor(2.0)))) < 1e-5
