"""API-subset shim of `cons.core` (see ../_mdispatch.py header)."""
from collections import OrderedDict
from collections.abc import Iterator

from _mdispatch import Dispatcher


class ConsError(ValueError):
    pass


_car = Dispatcher("_car")
_cdr = Dispatcher("_cdr")


def _car_seq(x):
    if len(x) == 0:
        raise ConsError("Not a cons pair.")
    return x[0]


def _cdr_seq(x):
    if len(x) == 0:
        raise ConsError("Not a cons pair.")
    return type(x)(x[1:]) if not isinstance(x, tuple) else x[1:]


for _t in (tuple, list):
    _car.add((_t,), _car_seq)
    _cdr.add((_t,), _cdr_seq)
_car.add((OrderedDict,), lambda x: _car_seq(list(x.items())))
_cdr.add((OrderedDict,), lambda x: _cdr_seq(list(x.items())))


def car(x):
    try:
        return _car(x)
    except NotImplementedError:
        raise ConsError("Not a cons pair.")


def cdr(x):
    try:
        return _cdr(x)
    except NotImplementedError:
        raise ConsError("Not a cons pair.")
