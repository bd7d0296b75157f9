"""Import-name drop-in for the reference's `mreserve` package: `from mreserve.modeling import PretrainedMerlotReserve`,
`from mreserve.preprocess import preprocess_video, encoder, MASK`, `from mreserve.checkpoint import load_checkpoint` resolve to the
MI355X build (merlot_reserve_amd).  Re-exports only: no code of its own."""
