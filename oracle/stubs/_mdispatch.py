"""Tiny multiple-dispatch helper used by the third-party API shims in this directory.

TEST INFRASTRUCTURE ONLY.  These shims restate the *public APIs* of the
pythological packages `cons`, `etuples` and `logical-unification` (unpinned
dependencies of the reference, pyproject.toml:49-59) which are not installed in
this offline image.  They exist solely so that the reference Aesara can be
imported in the authoring container to generate golden vectors; nothing in the
product path imports them.
"""
import itertools


class Dispatcher:
    def __init__(self, name):
        self.name = name
        self._sigs = []  # list of (signature tuple, fn), later registrations win ties
        self._cache = {}

    def add(self, signature, fn):
        alts = [s if isinstance(s, tuple) else (s,) for s in signature]
        for sig in itertools.product(*alts):
            self._sigs.append((tuple(sig), fn))
        self._cache.clear()

    def register(self, *signature):
        def deco(fn):
            self.add(signature, fn)
            return fn

        return deco

    @staticmethod
    def _dist(cls, target):
        try:
            return cls.__mro__.index(target)
        except ValueError:
            # abstract base classes (e.g. collections.abc.Mapping) registered virtually
            return len(cls.__mro__) + 1 if issubclass(cls, target) else None

    def resolve(self, types):
        fn = self._cache.get(types)
        if fn is not None:
            return fn
        best = None
        for order, (sig, f) in enumerate(self._sigs):
            if len(sig) != len(types):
                continue
            ds = []
            for t, s in zip(types, sig):
                d = self._dist(t, s)
                if d is None:
                    break
                ds.append(d)
            else:
                key = (sum(ds), -order)
                if best is None or key < best[0]:
                    best = (key, f)
        if best is None:
            raise NotImplementedError(
                f"{self.name}: no implementation for {[t.__name__ for t in types]}"
            )
        self._cache[types] = best[1]
        return best[1]

    def __call__(self, *args):
        return self.resolve(tuple(type(a) for a in args))(*args)
