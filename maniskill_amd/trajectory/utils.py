"""Readers for recorded trajectories (counterpart of mani_skill/trajectory/utils/__init__.py:
`load_h5_data`, `index_dict`, `dict_to_list_of_dicts`)."""
import numpy as np


def _insert(tree, path, value):
    node = tree
    for p in path[:-1]:
        node = node.setdefault(p, {})
    node[path[-1]] = value


def load_h5_data(path: str) -> dict:
    """nested dict {traj_i: {obs, actions, terminated, truncated, success, rewards, env_states{...}}} from a
    `.h5` (needs h5py) or the `.npz` twin this build writes when h5py is absent"""
    out = {}
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            for k in z.files:
                _insert(out, k.split("/"), z[k])
        return out
    import h5py  # noqa: F401  (reference container)

    def rec(g):
        return {k: (rec(v) if hasattr(v, "keys") else v[()]) for k, v in g.items()}

    with h5py.File(path, "r") as f:
        return rec(f)


def index_dict(x, i):
    if isinstance(x, dict):
        return {k: index_dict(v, i) for k, v in x.items()}
    return x[i]


def dict_to_list_of_dicts(x):
    first = x
    while isinstance(first, dict):
        first = next(iter(first.values()))
    return [index_dict(x, i) for i in range(len(first))]
