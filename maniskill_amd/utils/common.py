"""Tensor / dict helpers shared by the env layer (counterpart of mani_skill/utils/common.py)."""
from collections import defaultdict
from typing import Dict, Optional, Sequence

import numpy as np
import torch


def torch_clone_dict(data):
    if isinstance(data, torch.Tensor):
        return data.clone()
    return {k: torch_clone_dict(v) if isinstance(v, (dict, torch.Tensor)) else v for k, v in data.items()}


def _batch(x):
    if isinstance(x, dict):
        return {k: _batch(v) for k, v in x.items()}
    if isinstance(x, str):
        return x
    if isinstance(x, torch.Tensor):
        return x[None, :]
    if isinstance(x, np.ndarray):
        return x.reshape(1, 1) if x.shape == () else x[None, :]
    if isinstance(x, list) and len(x) == 1:
        return [x]
    if isinstance(x, (float, int, bool, np.bool_)):
        return np.array([[x]])
    return x


def batch(*args):
    """adds a leading dimension to every leaf"""
    out = [_batch(a) for a in args]
    return out[0] if len(args) == 1 else tuple(out)


def _unbatch(x):
    if isinstance(x, dict):
        return {k: _unbatch(v) for k, v in x.items()}
    if isinstance(x, str):
        return x
    if isinstance(x, torch.Tensor):
        return x.squeeze(0)
    if isinstance(x, np.ndarray):
        if np.iterable(x) and x.shape[0] == 1:
            return x.squeeze(0)
        return x
    if isinstance(x, list) and len(x) == 1:
        return x[0]
    return x


def unbatch(*args):
    out = [_unbatch(a) for a in args]
    return out[0] if len(args) == 1 else tuple(out)


def dict_merge(dct: dict, merge_dct: dict):
    """in-place recursive merge (sapien_env.py:254-258 merges user sim_config into the default)"""
    for k, v in merge_dct.items():
        if k in dct and isinstance(dct[k], dict) and isinstance(v, dict):
            dict_merge(dct[k], v)
        else:
            dct[k] = v


def merge_dicts(ds: Sequence[Dict], asarray=False):
    ret = defaultdict(list)
    for d in ds:
        for k in d:
            ret[k].append(d[k])
    ret = dict(ret)
    if asarray:
        ret = {k: np.concatenate(v) for k, v in ret.items()}
    return ret


def to_tensor(array, device=None):
    if isinstance(array, dict):
        return {k: to_tensor(v, device=device) for k, v in array.items()}
    if isinstance(array, torch.Tensor):
        ret = array.to(device) if device is not None else array
    elif isinstance(array, np.ndarray):
        if array.dtype == np.uint16:
            array = array.astype(np.int32)
        elif array.dtype == np.uint32:
            array = array.astype(np.int64)
        ret = torch.from_numpy(np.ascontiguousarray(array)).to(device)
    else:
        if isinstance(array, list) and len(array) > 0 and isinstance(array[0], np.ndarray):
            array = np.array(array)
        ret = torch.tensor(array, device=device)
    if ret.dtype == torch.float64:
        ret = ret.to(torch.float32)
    return ret


def to_cpu_tensor(array):
    if isinstance(array, dict):
        return {k: to_cpu_tensor(v) for k, v in array.items()}
    if isinstance(array, np.ndarray):
        ret = torch.from_numpy(array)
        return ret.float() if ret.dtype == torch.float64 else ret
    if isinstance(array, torch.Tensor):
        return array.cpu()
    return torch.tensor(array).cpu()


def to_numpy(array, dtype=None):
    if isinstance(array, dict):
        return {k: to_numpy(v, dtype=dtype) for k, v in array.items()}
    if isinstance(array, torch.Tensor):
        array = array.detach().cpu().numpy()
    elif isinstance(array, (list, tuple)) and len(array) > 0 and isinstance(array[0], dict):
        return [to_numpy(a, dtype) for a in array]
    else:
        array = np.asarray(array) if not isinstance(array, np.ndarray) else array
    return array.astype(dtype) if dtype is not None else array


def flatten_state_dict(state_dict: dict, use_torch=False, device=None):
    """recursive hstack in insertion order; 1-D tensors become columns (common.py:195-263)"""
    parts = []
    for key, value in state_dict.items():
        if isinstance(value, dict):
            s = flatten_state_dict(value, use_torch=use_torch, device=device)
            s = None if (s.nelement() if isinstance(s, torch.Tensor) else s.size) == 0 else s
        elif isinstance(value, (tuple, list)):
            s = None if len(value) == 0 else (to_tensor(value, device=device) if use_torch else value)
        elif isinstance(value, (bool, np.bool_, int, np.int32, np.int64)):
            s = to_tensor(int(value), device=device) if use_torch else int(value)
        elif isinstance(value, (float, np.float32, np.float64)):
            s = to_tensor(np.float32(value), device=device) if use_torch else np.float32(value)
        elif isinstance(value, np.ndarray):
            if value.ndim > 2:
                raise AssertionError(f"The dimension of {key} should not be more than 2.")
            s = None if value.size == 0 else (to_tensor(value, device=device) if use_torch else value)
        elif isinstance(value, torch.Tensor):
            s = value[:, None] if value.dim() == 1 else value
        else:
            raise TypeError(f"Unsupported type: {type(value)}")
        if s is not None:
            parts.append(s)
    if use_torch:
        return torch.hstack(parts) if parts else torch.empty(0, device=device)
    return np.hstack(parts) if parts else np.empty(0)


def flatten_dict_keys(d: dict, prefix=""):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flatten_dict_keys(v, prefix + k + "/"))
        else:
            out[prefix + k] = v
    return out


def normalize_vector(x: torch.Tensor, eps=1e-6):
    n = torch.linalg.norm(x, dim=1)
    n = torch.where(n < eps, torch.ones_like(n), n)
    return x / n[:, None]


def compute_angle_between(x1: torch.Tensor, x2: torch.Tensor):
    """angle (rad) between batched vectors; zero-length vectors count as zero (common.py:300-305)"""
    n1, n2 = torch.linalg.norm(x1, dim=1), torch.linalg.norm(x2, dim=1)
    a = x1 / torch.where(n1 < 1e-6, torch.ones_like(n1), n1)[:, None]
    b = x2 / torch.where(n2 < 1e-6, torch.ones_like(n2), n2)[:, None]
    return torch.arccos(torch.clip((a * b).sum(1), -1, 1))


def np_normalize_vector(x, eps=1e-6):
    x = np.asarray(x)
    n = np.linalg.norm(x)
    return np.zeros_like(x) if n < eps else x / n


def np_compute_angle_between(x1, x2):
    a, b = np_normalize_vector(x1), np_normalize_vector(x2)
    return float(np.arccos(np.clip(np.dot(a, b), -1, 1)))


def quat_diff_rad(a: torch.Tensor, b: torch.Tensor):
    a = a / torch.norm(a, dim=1, keepdim=True)
    b = b / torch.norm(b, dim=1, keepdim=True)
    return 2 * torch.acos(torch.clamp(torch.abs((a * b).sum(1)), 0.0, 1.0))


def index_dict_array(x, idx, inplace=True):
    if isinstance(x, (np.ndarray, list, torch.Tensor)):
        return x[idx]
    if isinstance(x, dict):
        if inplace:
            for k in x:
                x[k] = index_dict_array(x[k], idx, inplace)
            return x
        return {k: index_dict_array(v, idx, inplace) for k, v in x.items()}
    return x
