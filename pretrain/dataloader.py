"""`pretrain.dataloader` of the reference (pretrain/dataloader.py) under its own import name: the TensorFlow-free reader of
merlot_reserve_amd/records.py.  `dataset_parser`, `handle_batch`, `make_dataset` and `input_fn_builder` keep the reference's roles; arguments that
named TensorFlow / JAX objects are replaced by their plain counterparts (a numpy Generator for the random draws; rank / world instead of
jax.process_index() / process_count(); one process per GPU, so handle_batch has no num_devices axis)."""
from merlot_reserve_amd.records import dataset_parser, handle_batch, input_fn_builder, make_dataset   # noqa: F401
from merlot_reserve_amd.records import load_and_resize_img, load_audio, mask_tokens, pad_tokens_to_fixed_size, select_tokens   # noqa: F401

__all__ = ['dataset_parser', 'handle_batch', 'input_fn_builder', 'make_dataset', 'load_and_resize_img', 'load_audio', 'mask_tokens', 'pad_tokens_to_fixed_size', 'select_tokens']
