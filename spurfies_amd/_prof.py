"""Optional per-launch HIP-event timing of selected C-ABI calls (bench.py's roofline figures).  Events are recorded on torch's
current stream, which is the stream every spf_* call is launched on (`_lib.stream_ptr()`)."""
from __future__ import annotations

import torch

_records = None
_tags = None


def start(tags=None):
    """tags: only spans with these tags are timed (None = all)."""
    global _records, _tags
    _records, _tags = [], (None if tags is None else set(tags))


def active() -> bool:
    return _records is not None


class span:
    """with span('tag', **meta): <launch>   — meta values may be device tensors (counts or 0/1 masks: their sum is read at stop())."""

    def __init__(self, tag, **meta):
        self.tag, self.meta = tag, meta

    def __enter__(self):
        self.on = _records is not None and (_tags is None or self.tag in _tags)
        if self.on:
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on and _records is not None:
            self.e1.record()
            _records.append((self.tag, self.e0, self.e1, self.meta))
        return False


def stop():
    """-> [{'tag', 'ms', **meta}] since start(); synchronises."""
    global _records
    rec, _records = _records or [], None
    torch.cuda.synchronize()
    out = []
    for tag, e0, e1, meta in rec:
        row = {"tag": tag, "ms": e0.elapsed_time(e1)}
        for k, v in meta.items():
            row[k] = float(v.sum().item()) if torch.is_tensor(v) else v        # device counts / masks: summed
        out.append(row)
    return out
