"""Drop-in for the reference's OARSegmentation/Models/Nets/base_blocks.py (the classes the networks instantiate).
Namespace-package portion (no __init__.py): OARSegmentation.config etc. keep resolving to the reference's files."""
from dose_prediction_amd.models.base_blocks import ModifiedUnetrUpBlock, ModifiedUnetOutBlock, MultiUnetBasicBlock  # noqa: F401
