"""`finetune.optimization` of the reference (finetune/optimization.py) under its own import name: the VCR finetuning step of
merlot_reserve_amd.finetune."""
from merlot_reserve_amd.finetune import construct_finetuning_train_state, finetune_train_step   # noqa: F401
