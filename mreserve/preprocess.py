"""`mreserve.preprocess` of the reference (mreserve/preprocess.py) under its own import name: the integer / layout side
(merlot_reserve_amd.preprocess); `encoder` is resolved lazily (the tokenizer vocabulary is the user's file)."""
from merlot_reserve_amd import preprocess as _p
from merlot_reserve_amd.preprocess import (END, LTOVPOOL, MASK, MASKAUDIO, PADDING, RESETCTX, START, patchify,   # noqa: F401
                                           preprocess_video, video_to_segments)


def __getattr__(name):
    return getattr(_p, name)
