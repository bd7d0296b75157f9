"""The reference's import names resolve to the build: what pretrain/train.py:15-18 and
finetune/vcr/qa_qar_joint_finetune.py:15,20-21 import.  CPU: nothing is launched."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_training_and_finetuning_names_resolve():
    import finetune.optimization as fo
    import mreserve.checkpoint as ck
    import mreserve.modeling as mm
    import pretrain.optimization as po
    import pretrain.pretrain_model as pm
    import merlot_reserve_amd.finetune as F
    import merlot_reserve_amd.pretrain_model as PM
    assert pm.MerlotReservePretrainer is PM.MerlotReservePretrainer and pm.train_step is PM.train_step
    assert pm.loss_fn_given_preds is PM.loss_fn_given_preds and po.construct_train_state is PM.construct_train_state
    assert fo.finetune_train_step is F.finetune_train_step and fo.construct_finetuning_train_state is F.construct_finetuning_train_state
    assert callable(ck.load_checkpoint) and callable(ck.save_checkpoint) and callable(ck.bf16_to_f32) and callable(ck.f32_to_bf16)
    assert mm.MerlotReserve.__module__ == 'merlot_reserve_amd.modeling'


def test_alias_packages_are_reexports_only():
    """mreserve/, pretrain/, finetune/ at the repo root hold no code of their own (and never touch the oracle)."""
    for pkg in ('mreserve', 'pretrain', 'finetune'):
        for f in os.listdir(os.path.join(ROOT, pkg)):
            if f.endswith('.py'):
                src = open(os.path.join(ROOT, pkg, f)).read()
                body = re.sub(r'""".*?"""', '', src, flags=re.S)
                assert 'oracle' not in body
                for line in body.splitlines():
                    line = line.strip()
                    assert (not line or line.startswith(('from merlot_reserve_amd', 'import merlot_reserve_amd', '#', '__all__', 'def __getattr__', 'return getattr'))
                            or line[0] in '()' or line.endswith((',', ')')) ), (pkg, f, line)
