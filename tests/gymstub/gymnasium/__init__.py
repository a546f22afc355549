"""Stand-in for the `gymnasium` package (absent from the build image), just wide enough for the registration path of
gym_guided_vision/__init__.py:88-101 and env.py's base class: Env with reset(seed=, options=) seeding np_random, spaces.Box / Dict,
envs.registration.register / registry, make(id, **kwargs) resolving "module:Class" entry points.  TEST INFRASTRUCTURE ONLY
(tests/test_gym_registration.py puts tests/gymstub on sys.path in a subprocess)."""
import importlib

import numpy as np

from . import spaces  # noqa: F401
from .envs.registration import register, registry  # noqa: F401

__version__ = "0.0-stub"


class Env:
    metadata = {"render_modes": []}
    _np_random = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.default_rng(seed)

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    def close(self):
        pass


def make(id, **kwargs):
    spec = registry[id]
    mod, cls = spec.entry_point.split(":")
    kw = dict(spec.kwargs)
    kw.update(kwargs)
    env = getattr(importlib.import_module(mod), cls)(**kw)
    env.spec = spec
    return env
