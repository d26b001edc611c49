"""Dependency-free reader for the safetensors container, used only when the `safetensors` package is absent.

Layout: 8-byte little-endian header length, a JSON header ({name: {dtype, shape, data_offsets}} plus an optional
"__metadata__" string map), then the raw little-endian tensor bytes.  Same call surface as
`safetensors.safe_open(...)` as far as lora.py needs it: .metadata(), .keys(), .get_tensor().
"""
import json
import struct

import numpy as np
import torch

_NP = {"F32": np.float32, "F16": np.float16, "F64": np.float64, "I64": np.int64, "I32": np.int32, "U8": np.uint8}


class _Archive:
    def __init__(self, meta, entries, blob, device):
        self._meta, self._entries, self._blob, self._device = meta, entries, blob, device

    def metadata(self):
        return self._meta

    def keys(self):
        return self._entries.keys()

    def get_tensor(self, key):
        info = self._entries[key]
        lo, hi = info["data_offsets"]
        raw = self._blob[lo:hi]
        if info["dtype"] == "BF16":
            t = torch.frombuffer(bytearray(raw), dtype=torch.bfloat16)
        else:
            t = torch.from_numpy(np.frombuffer(raw, dtype=_NP[info["dtype"]]).copy())
        return t.reshape(info["shape"]).to(self._device)


def safe_open(filename, framework="pt", device="cpu"):
    if framework != "pt":
        raise ValueError("`framework` must be 'pt'")
    with open(filename, "rb") as f:
        (hlen,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hlen).decode("utf-8"))
        blob = f.read()
    meta = header.pop("__metadata__", {})
    return _Archive(meta, header, blob, device)
