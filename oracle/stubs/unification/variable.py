"""API-subset shim of `unification.variable` (see ../_mdispatch.py header)."""


class Var:
    __slots__ = ("token", "__weakref__")
    _id = 1
    _refs = {}

    def __new__(cls, token=None, prefix=""):
        if token is None:
            token = f"{prefix}_{Var._id}"
            Var._id += 1
        obj = cls._refs.get(token, None)
        if obj is None:
            obj = object.__new__(cls)
            obj.token = token
            cls._refs[token] = obj
        return obj

    def __str__(self):
        return f"~{self.token}"

    __repr__ = __str__

    def __eq__(self, other):
        if type(self) == type(other):
            return self.token == other.token
        return NotImplemented

    def __hash__(self):
        return hash((type(self), self.token))


def var(*args, **kwargs):
    return Var(*args, **kwargs)


def isvar(x):
    return isinstance(x, Var)
