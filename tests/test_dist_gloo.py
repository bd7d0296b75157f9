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


# ---- partitioned Adam moments (merlot_reserve_amd/zero.py; pretrain/train_fixres.py:178-199, finetune/optimization.py:148-171) ----
def _toy_adam(p, lr=0.01):
    """An elementwise stand-in for the fused chain (the HIP kernel needs a GPU): what the partition must reproduce is WHO updates
    WHAT, which any elementwise rule shows."""
    def adam(lo, hi, mu, nu):
        g = p.grad[lo:hi].float()
        m = (0.9 * mu.float() + 0.1 * g).to(torch.bfloat16)
        v = (0.98 * nu.float() + 0.02 * g * g).to(torch.bfloat16)
        mu.copy_(m)
        nu.copy_(v)
        p.master[lo:hi] -= lr * m.float() / (v.float().sqrt() + 1e-3)
        p.work[lo:hi] = p.master[lo:hi].to(torch.bfloat16)
    return adam


def _toy_store(total):
    class Store:
        pass
    p = Store()
    p.device, p.total = torch.device('cpu'), total
    g = torch.Generator().manual_seed(7)
    p.master = torch.randn(total, generator=g)
    p.work = p.master.to(torch.bfloat16)
    p.grad = torch.zeros(total, dtype=torch.bfloat16)
    p.mu = torch.zeros(total, dtype=torch.bfloat16)
    p.nu = torch.zeros(total, dtype=torch.bfloat16)
    p.touched = []
    p.update_transposed = lambda lo=0, hi=None: p.touched.append((lo, hi))
    return p


# ragged on purpose: 5 blocks over 2 ranks (3 + 2), 1 block (rank 1 owns nothing), 4 blocks (2 + 2)
_BUCKETS = [('a', 0, 5 * 2048), ('b', 5 * 2048, 6 * 2048), ('c', 6 * 2048, 10 * 2048)]


def _shard_worker(rank, world, port, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from merlot_reserve_amd.dist import Comm
    from merlot_reserve_amd.zero import MomentShards
    total = _BUCKETS[-1][2]
    ref, p = _toy_store(total), _toy_store(total)
    sh = MomentShards(p, _BUCKETS, Comm())
    assert p.mu is None and p.nu is None
    owned = sum(mhi - mlo for _lo, _hi, _c, mlo, mhi, _o in sh.table.values())
    for step in range(3):
        g = torch.randn(total, generator=torch.Generator().manual_seed(50 + step)).to(torch.bfloat16)     # the averaged gradient: the same on every rank
        ref.grad.copy_(g)
        p.grad.copy_(g)
        for key, lo, hi in _BUCKETS:
            _toy_adam(ref)(lo, hi, ref.mu[lo:hi], ref.nu[lo:hi])
            sh.update(key, _toy_adam(p))
    mu, nu = sh.full_moments()
    ok = dict(master=torch.equal(p.master, ref.master), work=torch.equal(p.work, ref.work), mu=torch.equal(mu, ref.mu), nu=torch.equal(nu, ref.nu),
              transposed=p.touched[:3] == [(lo, hi) for _k, lo, hi in _BUCKETS])
    # a checkpoint's moments go back into the shards
    sh2 = MomentShards(_toy_store(total), _BUCKETS, Comm())
    sh2.store_moments(mu, nu)
    ok['reload'] = torch.equal(sh2.mu[:sh2.owned], sh.mu[:sh.owned]) and torch.equal(sh2.nu[:sh2.owned], sh.nu[:sh.owned])
    ret[rank] = dict(ok=ok, owned=owned, table={k: v[3:5] for k, v in sh.table.items()})
    dist.destroy_process_group()


def test_partitioned_moments_world2_gloo():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_shard_worker, args=(world, 29700 + (os.getpid() % 1000), ret), nprocs=world, join=True)
    for r in range(world):
        assert all(ret[r]['ok'].values()), (r, ret[r]['ok'])
    assert ret[0]['owned'] + ret[1]['owned'] == _BUCKETS[-1][2]
    assert ret[0]['table'] == {'a': (0, 3 * 2048), 'b': (5 * 2048, 6 * 2048), 'c': (6 * 2048, 8 * 2048)}
    assert ret[1]['table'] == {'a': (3 * 2048, 5 * 2048), 'b': (6 * 2048, 6 * 2048), 'c': (8 * 2048, 10 * 2048)}
