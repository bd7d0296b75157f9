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


@pytest.mark.parametrize('section,key,value', [('model', 'do_rotary', False), ('model', 'size_per_head', 32)])
def test_unbuilt_flags_are_refused(section, key, value):
    cfg = copy.deepcopy(tiny_config())
    cfg[section][key] = value
    with pytest.raises(NotImplementedError):
        Dims(cfg, 2)


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
    cfg['model']['do_rotary'] = False
    with pytest.raises(NotImplementedError):
        PM.MerlotReservePretrainer.from_config(cfg, device='cpu')
