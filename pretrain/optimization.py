"""`pretrain.optimization` of the reference (pretrain/optimization.py) under its own import name."""
from merlot_reserve_amd.finetune import lr_scale_linearwarmup_lineardecay                       # noqa: F401
from merlot_reserve_amd.pretrain_model import construct_train_state                             # noqa: F401
from merlot_reserve_amd.trainer import lr_scale_linearwarmup_cosinedecay                        # noqa: F401
