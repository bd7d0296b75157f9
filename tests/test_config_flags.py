"""Config flags of the reference's pretrainer that change the program (pretrain/pretrain_model.py:61-63, 99, 124, 146-148;
mreserve/modeling.py:598): built ones are honoured, the others are refused -- never silently ignored.  CPU only."""
import copy

import pytest

from merlot_reserve_amd.config import Dims, load_config, tiny_config


def test_stock_configs_pass():
    for name in ('base', 'large'):
        d = Dims(load_config(name), 4)
        assert d.no_vision is False and d.nh * 64 == d.H


def test_no_vision_is_a_flag_of_the_step():
    cfg = tiny_config()
    cfg['model']['no_vision'] = True
    assert Dims(cfg, 2).no_vision is True


@pytest.mark.parametrize('section,key,value', [('model', 'size_per_head', 32)])
def test_unbuilt_flags_are_refused(section, key, value):
    cfg = copy.deepcopy(tiny_config())
    cfg[section][key] = value
    with pytest.raises(NotImplementedError):
        Dims(cfg, 2)


def test_do_rotary_false_adds_the_learned_position_table():
    """pretrain_model.py:146-148 -> modeling.py:335-341: the joint tower of a config without rotary learns `pe` [seq_len, H]."""
    from merlot_reserve_amd.params import param_specs
    cfg = copy.deepcopy(tiny_config())
    assert 'joint_transformer/pe' not in [s[0] for s in param_specs(cfg)] and Dims(cfg, 2).do_rotary
    cfg['model']['do_rotary'] = False
    spec = [s for s in param_specs(cfg) if s[0] == 'joint_transformer/pe']
    assert len(spec) == 1 and spec[0][1] == (cfg['data']['seq_len'], cfg['model']['hidden_size']) and not Dims(cfg, 2).do_rotary


def test_sequences_per_kind_change_the_joint_batch():
    """num_audio2text_seqs / num_text2audio_seqs / num_text_seqs (pretrain_model.py:99, 124; dataloader.py:649): rows of the joint batch, masked targets
    and text spans per record follow."""
    cfg = copy.deepcopy(tiny_config())
    d1 = Dims(cfg, 2)
    cfg['data'].update(num_audio2text_seqs=2, num_text2audio_seqs=3, num_text_seqs=2)
    d = Dims(cfg, 2)
    assert d.Nj == 2 * (2 * 2 + 1 + 2 + 2 * 3) and d1.Nj == 2 * (2 + 1 + 1 + 2)
    assert d.ntrg == 3 * d1.ntrg and d.ntext_spans == d1.ntrg * 5 + 2 * d1.budget


def test_trainer_and_model_entry_points_refuse_too():
    """The flags are checked where every program starts (Dims), so the reference-API entry points cannot miss them."""
    from merlot_reserve_amd import pretrain_model as PM
    cfg = copy.deepcopy(tiny_config())
    cfg['model']['size_per_head'] = 32
    with pytest.raises(NotImplementedError):
        PM.MerlotReservePretrainer.from_config(cfg, device='cpu')
