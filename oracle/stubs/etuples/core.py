"""API-subset shim of `etuples.core` (see ../_mdispatch.py header)."""
from collections.abc import Mapping, Sequence

from unification.core import _reify, _unify, _unify_seq

_NULL = object()


class InvalidExpression(Exception):
    pass


class ExpressionTuple(Sequence):
    __slots__ = ("_evaled_obj", "_tuple")
    null = _NULL

    def __init__(self, seq=None, evaled_obj=_NULL):
        self._tuple = tuple(seq) if seq is not None else tuple()
        self._evaled_obj = evaled_obj

    @property
    def evaled_obj(self):
        if len(self._tuple) == 0:
            raise InvalidExpression("Empty expression.")
        if self._evaled_obj is _NULL:
            from .dispatch import apply

            op = self._tuple[0]
            if isinstance(op, ExpressionTuple):
                op = op.evaled_obj
            args = [
                a.evaled_obj if isinstance(a, ExpressionTuple) else a
                for a in self._tuple[1:]
            ]
            self._evaled_obj = apply(op, ExpressionTuple(args))
        return self._evaled_obj

    @evaled_obj.setter
    def evaled_obj(self, obj):
        raise ValueError("Value of evaluated expression cannot be set!")

    def __getitem__(self, key):
        tr = self._tuple[key]
        if isinstance(key, slice):
            return type(self)(tr)
        return tr

    def __len__(self):
        return len(self._tuple)

    def __iter__(self):
        return iter(self._tuple)

    def __add__(self, x):
        return type(self)(self._tuple + tuple(x))

    def __radd__(self, x):
        return type(self)(tuple(x) + self._tuple)

    def __eq__(self, other):
        if isinstance(other, ExpressionTuple):
            return self._tuple == other._tuple
        return NotImplemented

    def __hash__(self):
        return hash(self._tuple)

    def __repr__(self):
        return f"ExpressionTuple({self._tuple!r})"

    def __str__(self):
        return f"e({', '.join(map(str, self._tuple))})"


def etuple(*args, **kwargs):
    return ExpressionTuple(args, evaled_obj=kwargs.pop("evaled_obj", _NULL))


_unify.add((ExpressionTuple, ExpressionTuple, Mapping), _unify_seq)
for _t in (tuple, list):
    _unify.add((ExpressionTuple, _t, Mapping), _unify_seq)
    _unify.add((_t, ExpressionTuple, Mapping), _unify_seq)


def _reify_ExpressionTuple(o, s):
    res = []
    for x in o._tuple:
        res.append((yield _reify(x, s)))
    yield ExpressionTuple(res)


_reify.add((ExpressionTuple, Mapping), _reify_ExpressionTuple)
