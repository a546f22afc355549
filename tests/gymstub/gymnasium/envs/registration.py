from dataclasses import dataclass, field

registry = {}
calls = []          # every register() call as it was made, for the test


@dataclass
class EnvSpec:
    id: str
    entry_point: str
    nondeterministic: bool = False
    kwargs: dict = field(default_factory=dict)


def register(id, entry_point=None, nondeterministic=False, kwargs=None, **other):
    calls.append(dict(id=id, entry_point=entry_point, nondeterministic=nondeterministic, kwargs=dict(kwargs or {}), other=other))
    registry[id] = EnvSpec(id, entry_point, nondeterministic, dict(kwargs or {}))
