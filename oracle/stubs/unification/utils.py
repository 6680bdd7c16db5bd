"""API-subset shim of `unification.utils`."""
from contextlib import suppress


def transitive_get(key, d):
    with suppress(TypeError):
        while key in d:
            key = d[key]
    return key
