"""Tiny pyhocon.ConfigTree look-alike: the reference reads its model config through
get_int/get_float/get_bool/get_list/get_string/get_config and attribute access
(spurfies/model/pointneus_disent.py:31-42,110-114; spurfies/train.py:27-31)."""
from __future__ import annotations


class Conf(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        for key, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Conf):
                self[key] = Conf(v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def _lookup(self, key, default):
        cur = self
        for part in str(key).split("."):
            if not isinstance(cur, dict) or part not in cur:
                return default
            cur = cur[part]
        return cur

    def get_int(self, k, default=None):
        v = self._lookup(k, default)
        return None if v is None else int(v)

    def get_float(self, k, default=None):
        v = self._lookup(k, default)
        return None if v is None else float(v)

    def get_bool(self, k, default=None):
        return bool(self._lookup(k, default))

    def get_list(self, k, default=None):
        v = self._lookup(k, default)
        return None if v is None else list(v)

    def get_string(self, k, default=None):
        return self._lookup(k, default)

    def get_config(self, k, default=None):
        return self._lookup(k, default)


def default_model_conf(near: float = 0.5, **over) -> Conf:
    """Effective DTU training values: config/vol/dtu_pn.yaml:23-44 overlaid by config/ours.yaml:22-24."""
    c = Conf(feature_vector_size=64, scene_bounding_sphere=3.0, initialize_colors=True, k=8, r=2, rbf=45, vox_res=300,
             max_shading_pts=80, density=Conf(params_init=Conf(beta=0.1), beta_min=0.0001),
             ray_sampler=Conf(far=4.5, near=near, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1,
                              beta_iters=10, max_total_iters=5))
    for k, v in over.items():
        c[k] = v
    return c
