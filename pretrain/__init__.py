"""Import-name drop-in for the reference's `pretrain` package: `pretrain.pretrain_model`, `pretrain.optimization`."""
