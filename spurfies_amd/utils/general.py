"""Host glue with the reference's names (spurfies/utils/general.py:10-59)."""
import importlib

import torch


def get_class(kls: str):
    module, _, name = kls.rpartition(".")
    return getattr(importlib.import_module(module), name)


def split_input(model_input, total_pixels, n_pixels=10000):
    """Chunk the pixel axis so that a full image fits (train.py:414-428 uses 500, eval 512)."""
    out = []
    dev = model_input["uv"].device
    for idx in torch.split(torch.arange(total_pixels, device=dev), n_pixels, dim=0):
        d = dict(model_input)
        d["uv"] = torch.index_select(model_input["uv"], 1, idx)
        for key in ("object_mask", "rgb"):
            if key in d:
                d[key] = torch.index_select(model_input[key], 1, idx)
        out.append(d)
    return out


def merge_output(res, total_pixels, batch_size):
    merged = {}
    for key, first in res[0].items():
        if first is None:
            continue
        if first.dim() == 1:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, 1) for r in res], 1).reshape(batch_size * total_pixels)
        elif first.dim() == 2:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, r[key].shape[-1]) for r in res], 1).reshape(batch_size * total_pixels, -1)
        elif first.dim() == 3:
            merged[key] = torch.cat([r[key].reshape(batch_size, -1, r[key].shape[-2], r[key].shape[-1]) for r in res], 1).reshape(
                batch_size * total_pixels, -1, first.shape[-1])
        else:
            raise NotImplementedError
    return merged
