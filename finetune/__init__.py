"""Import-name drop-in for the reference's `finetune` package: `finetune.optimization` (finetune/optimization.py)."""
