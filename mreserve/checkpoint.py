"""`mreserve.checkpoint` of the reference (mreserve/checkpoint.py) under its own import name: merlot_reserve_amd.checkpoint."""
from merlot_reserve_amd.checkpoint import (bf16_to_f32, f32_to_bf16, latest_checkpoint, load_checkpoint, log_param_shapes,   # noqa: F401
                                           save_checkpoint, tree_map_nested_keys)
