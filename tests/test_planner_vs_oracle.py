"""Host logic: the planner's integer decisions against the oracle's literal restatement of the reference's index code
(CPU only).  Everything here is bit-exact: gather codes, attention mask, selections, pooling targets."""
import numpy as np
import pytest
import torch

from merlot_reserve_amd.config import Dims
from merlot_reserve_amd.planner import VOCAB, build_plan
from oracle import ref_torch as R
from tests.util import oracle_batch, oracle_draws, tiny_setup, tree_to


def _segsum(table, indptr, idx):
    out = torch.zeros(len(indptr) - 1, table.shape[1], dtype=table.dtype)
    for i in range(len(indptr) - 1):
        if indptr[i + 1] > indptr[i]:
            out[i] = table[idx[indptr[i]:indptr[i + 1]].astype(np.int64)].sum(0)
    return out


MULTI = dict(num_audio2text_seqs=2, num_text2audio_seqs=3, num_text_seqs=2)       # pretrain_model.py:99-137 with more than one sequence per kind


@pytest.mark.parametrize('data_flags', [None, MULTI], ids=['stock', 'multi_seq'])
def test_plan_matches_oracle_indexing(data_flags):
    for seed in (3, 4, 5):
        cfg, store, batch, splits, z = tiny_setup(B=2, seed=seed, data_flags=data_flags)
        d = Dims(cfg, 2)
        plan = build_plan(batch, d, splits, z)
        params = tree_to(store.master_tree(), torch.float64)
        osp, oz = oracle_draws(splits, z)
        out, dbg = R.pretrain_forward(params, cfg, oracle_batch(batch, torch.float64), osp, oz, return_debug=True)

        # joint input assembled through the gather codes == the oracle's concatenated x
        H = d.H
        table = torch.cat([params['token_encoder']['Embed_0']['embedding'],
                           dbg['audio_seq'].reshape(-1, H), dbg['imgs_seq'].reshape(-1, H)])
        x = _segsum(table, plan['joint_gather_indptr'], plan['joint_gather_idx'])
        assert torch.equal(x, dbg['joint_x'].reshape(-1, H))

        # attention mask from one code per position
        c = torch.from_numpy(plan['joint_code'].reshape(d.Nj, d.Sj).astype(np.int64))
        allowed = (c[:, :, None] == c[:, None, :]) & (c[:, :, None] >= 0)
        assert torch.equal(allowed, dbg['joint_mask'])
        assert (c[0] >= 8).any(), 'the augmented video-source split must be present in this test'

        # "rotary" table == sin -/+ cos of the reference's sinusoids
        sinus = R.construct_rotary_sinusoids(dbg['joint_coords'])          # [N, 2, S, 32] (cos, sin), pairs repeated
        cos_t, sin_t = sinus[:, 0], sinus[:, 1]
        tab = np.where(np.arange(32) % 2 == 0, sin_t - cos_t, sin_t + cos_t).reshape(-1, 32)
        assert np.allclose(plan['joint_rot'], tab.astype(np.float32), atol=1e-6)
        # and applying it is apply_rotary
        q = torch.randn(d.Nj, d.Sj, 2, 64, dtype=torch.float64)
        ref = R.apply_rotary(q, torch.as_tensor(sinus))
        mine = q.clone()
        mine[..., :32] *= torch.as_tensor(tab).reshape(d.Nj, d.Sj, 1, 32)
        assert torch.allclose(ref, mine, atol=1e-12)

        # selections
        assert np.array_equal(plan['idx_sort'], dbg['idx_sort'].numpy())
        assert np.array_equal(plan['best_sp'], dbg['best_sp'].numpy())
        assert np.array_equal(plan['t2sp_src'], out['stuff_to_span']['_sources'].numpy())

        # pooled rows: segment sums of the oracle's head output reproduce the oracle's x's (after normalisation)
        pooled = _segsum(dbg['joint_head'].reshape(-1, H), plan['pool_indptr'], plan['pool_idx'])
        n1, n2 = 2 * d.nseg, 2 * d.ntrg
        scales = torch.exp(torch.clamp(params['contrastive_scales'], max=float(np.log(100.0))) / 2)
        for sl, key, si in ((slice(0, n1), 'imgs_to_audio', 0), (slice(n1, n1 + n2), 'text_to_audio', 1),
                            (slice(n1 + n2, None), 'stuff_to_span', 2)):
            assert torch.allclose(R.unit_normalize(pooled[sl]) * scales[si], out[key]['x'], atol=1e-12)
        acls = _segsum(dbg['audio_cls'].reshape(-1, H), plan['acls_indptr'], plan['acls_idx'])
        ref_y = torch.cat([out['text_to_audio']['y'], out['text_to_audio']['y_extra']])
        assert torch.allclose(R.unit_normalize(acls) * scales[1], ref_y, atol=1e-12)

        # inverted lists are exact transposes of the forward lists
        n_rows = d.Nj * d.Sj
        fw = np.full(n_rows, -1)
        for i in range(plan['n_pool']):
            fw[plan['pool_idx'][plan['pool_indptr'][i]:plan['pool_indptr'][i + 1]]] = i
        for r in range(n_rows):
            lst = plan['poolT_idx'][plan['poolT_indptr'][r]:plan['poolT_indptr'][r + 1]]
            assert (len(lst) == 0 and fw[r] == -1) or (len(lst) == 1 and lst[0] == fw[r])
        # every non-PAD token position lands in exactly one embedding list
        total = plan['embT_indptr'][-1]
        g = np.full(n_rows, -1, dtype=np.int64)
        has = np.diff(plan['joint_gather_indptr']) > 0
        g[has] = plan['joint_gather_idx']
        ntok = ((g > 0) & (g < VOCAB)).sum() + (plan['span_gather_idx'] > 0).sum()
        assert total == ntok
        assert plan['embT_indptr'][1] == 0, 'PAD row carries no list (its gradient is exactly zero)'
