"""Configuration: the reference's two-level YAML schema (data / model / device / optimizer; pretrain/configs/*.yaml)
plus a tiny configuration with the same structure for tests."""
import copy
import os

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))


def load_config(name_or_path):
    path = name_or_path
    if name_or_path in ('base', 'large'):
        path = os.path.join(HERE, 'configs', f'{name_or_path}.yaml')
    with open(path, 'r') as f:
        return yaml.load(f, yaml.FullLoader)


def tiny_config(hidden_size=128, grid=(4, 6), num_segments=4, seq_len=48, lang_seq_len=32):
    """Same structure as base.yaml at toy sizes: 2 segment groups x 2 frames, 12 audio spans (3 targets per masked
    stream), 12 text spans of which 8 are used, and a joint seq_len LONGER than text + vision so that the
    padding branch of prepare_multimodal_inputs (modeling.py:731-739) is exercised."""
    cfg = copy.deepcopy(load_config('base'))
    d, m = cfg['data'], cfg['model']
    d.update(num_segments=num_segments, num_segment_groups=2, num_audio_subsegments=3, seq_len=seq_len,
             lang_seq_len=lang_seq_len, num_text_spans_to_include=8, text_span_budget=6, mask_rate=0.25)
    m.update(hidden_size=hidden_size, joint_num_layers=2, audio_num_layers=2, vit_num_layers=2, span_num_layers=1,
             output_grid=list(grid))
    cfg['optimizer'].update(num_train_steps=1000, num_warmup_steps=10)
    return cfg


RESADAPT_GRIDS = [[18, 32], [24, 24]]        # pretrain/train_fixres.py:78: alternated over processes


def resadapt_config(name, grid=None, rank=0):
    """pretrain/train_fixres.py:78-90, 141-144: the resolution-adaptation stage.  Grid = `grid`, or the reference's
    per-process choice possible_res[rank % 2]; joint seq_len re-derived (lang + 8 * h * w / 4); data-augmentation keys and
    the short schedule (75k steps, 15k warm-up, final_lr_scale 0, learning rate x 0.02) as the reference sets them.
    Every rank keeps the same parameter shapes, so ranks on different grids still all-reduce gradients together."""
    cfg = load_config(name)
    grid = list(grid) if grid is not None else list(RESADAPT_GRIDS[rank % len(RESADAPT_GRIDS)])
    cfg['model']['output_grid'] = grid
    d = cfg['data']
    d['random_scale_max'] = max(min(grid) / max(grid) * 16 / 9, 1.0) + 0.1
    d['shrink_both_sides'] = False
    d['random_scale_min'] = 1.0
    d['max_text_seq_len'] = 1024
    d['do_flip_if_vertical'] = False
    per_group = d['num_segments'] // d['num_segment_groups']
    d['seq_len'] = d['lang_seq_len'] + per_group * (grid[0] * grid[1]) // (cfg['model']['vit_pooling_ratio'] ** 2)
    o = cfg['optimizer']
    o['num_train_steps'] = 75000
    o['final_lr_scale'] = 0.0
    o['num_warmup_steps'] = 15000
    o['learning_rate'] = 0.02 * o['learning_rate']
    return cfg


class Dims:
    """Static shapes of one device's step, derived from config and B = records per device."""

    def __init__(self, config, B):
        d, m = config['data'], config['model']
        self.B = B
        # Flags of the reference's pretrainer that change the program.  Built: no_vision (pretrain/pretrain_model.py:61-63: the pooled
        # vision sequence is multiplied by 0 on its way into the joint tower); since round 5 do_rotary = False (:146-148 drops the joint
        # coordinates, which sends TransformerEncoder to its learned `pe`, mreserve/modeling.py:335-341) and more than one audio2text /
        # text2audio / random-text sequence per record (:99-110, :124-135 tile the inputs).  Not built -- a config that sets it would train a
        # different model, so it is refused instead of ignored: heads of another width than 64 (mreserve/modeling.py:598).
        if m.get('size_per_head', 64) != 64:
            raise NotImplementedError('model.size_per_head other than 64 is not implemented (the attention kernels are written for 64)')
        self.no_vision = bool(m.get('no_vision', False))
        self.do_rotary = bool(m.get('do_rotary', True))       # False: the joint tower gets no coordinates and learns `pe` instead (pretrain_model.py:146-148)
        self.H = m['hidden_size']
        self.nh = self.H // 64
        self.gh, self.gw = m['output_grid']
        self.hw = self.gh * self.gw
        self.pr = m['vit_pooling_ratio']
        self.hw4 = self.hw // (self.pr ** 2)
        self.pp3 = m['vit_patch_size'] ** 2 * 3
        self.nseg = d['num_segments']
        self.ngroups = d['num_segment_groups']
        self.nspg = self.nseg // self.ngroups
        self.nas = d['num_audio_subsegments']
        self.nspans = self.nseg * self.nas
        # sequences per record and kind (pretrain_model.py:99, 124; dataloader.py:500-501, 649): the stock configs use one of each
        self.n_a2t, self.n_t2a, self.n_text = d.get('num_audio2text_seqs', 1), d.get('num_text2audio_seqs', 1), d.get('num_text_seqs', 1)
        assert self.n_a2t >= 1 and self.n_t2a >= 1 and self.n_text >= 1, 'every stream of the joint batch needs at least one sequence per record'
        self.ntrg1 = int(self.nspans * d['mask_rate'])          # masked audio spans per sequence
        self.ntrg = self.ntrg1 * self.n_t2a                     # text -> audio targets per record (pretrain_model.py:178)
        self.lang = d['lang_seq_len']
        self.seq_len = d['seq_len']
        self.vis_len = self.nspg * self.hw4
        assert self.lang + self.vis_len <= self.seq_len
        self.a_raw = m['audio_seq_length']
        self.a_patch = m['audio_patch_size']
        self.a_len = self.a_raw // self.a_patch
        self.a_tok = m['audio_token_length']
        self.a_pool = self.a_raw // (self.a_tok * self.a_patch)
        self.span_len = m['text_span_length']
        self.n_inc = d['num_text_spans_to_include']
        self.budget = d['text_span_budget']
        self.ntext_spans = self.ntrg1 * (self.n_t2a + self.n_a2t) + self.budget * self.n_text      # rows of text_spans per record (62 for base)
        self.Lv, self.La, self.Lj, self.Ls = m['vit_num_layers'], m['audio_num_layers'], m['joint_num_layers'], m['span_num_layers']
        # sequence counts
        self.Nv = B * self.nseg                # images
        self.Sv = self.hw + 1
        self.Na = B * self.nspans              # audio clips
        self.Sa = self.a_len + 1
        self.rows_a2t, self.rows_t2a = self.ngroups * self.n_a2t, self.ngroups * self.n_t2a      # per record: sequence-major, group-minor (tile)
        self.Nj = B * (self.rows_a2t + 1 + self.n_text + self.rows_t2a)   # joint sequences: a2t + matching + random + t2a
        self.Sj = self.seq_len
        self.Ns = B * self.n_inc               # selected spans
        self.Ss = self.span_len + 1
