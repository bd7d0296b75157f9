"""`pretrain.pretrain_model` of the reference (pretrain/pretrain_model.py) under its own import name (`from
pretrain.pretrain_model import *` in pretrain/train.py:15): merlot_reserve_amd.pretrain_model."""
from merlot_reserve_amd.pretrain_model import MerlotReservePretrainer, loss_fn_given_preds, train_step   # noqa: F401

__all__ = ['MerlotReservePretrainer', 'loss_fn_given_preds', 'train_step']
