"""Run-time configuration of the HIP path."""
import os

import torch

_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32,
           "bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp16": torch.float16, "f16": torch.float16, "float16": torch.float16}
_compute_dtype = _DTYPES[os.environ.get("DOSE_HIP_DTYPE", "fp32").lower()]


def set_compute_dtype(dtype):
    """torch.float32: parity mode (exact-fp32 MFMA); torch.bfloat16: benchmark mode (bf16 MFMA, fp32 accumulate);
    torch.float16: fp16 storage + fp16 MFMA with fp32 accumulation (BASELINE.json configs[4])."""
    global _compute_dtype
    if isinstance(dtype, str):
        dtype = _DTYPES[dtype.lower()]
    if dtype not in (torch.float32, torch.bfloat16, torch.float16):
        raise ValueError("compute dtype must be float32, bfloat16 or float16")
    _compute_dtype = dtype


def compute_dtype():
    return _compute_dtype
