"""`mreserve.modeling` of the reference (mreserve/modeling.py) under its own import name: the build's merlot_reserve_amd.modeling."""
from merlot_reserve_amd.modeling import *                                   # noqa: F401,F403
from merlot_reserve_amd.modeling import (AudioTransformer, MerlotReserve, PretrainedMerlotReserve, SpanTransformer,   # noqa: F401
                                         TokenEmbedder, TransformerEncoder, VisionTransformer, apply_rotary,
                                         construct_rotary_sinusoids, get_encoder, get_rotary_coordinates,
                                         get_rotary_coordinates_2d, multimodal_rotary_coords, one_hot_pool, unit_normalize)
