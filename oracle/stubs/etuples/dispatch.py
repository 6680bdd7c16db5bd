"""API-subset shim of `etuples.dispatch`."""
from _mdispatch import Dispatcher
from cons.core import ConsError, _car, _cdr

from .core import ExpressionTuple, etuple

apply = Dispatcher("apply")
apply.add((object, object), lambda rator, rands: rator(*rands))

rator = _car
rands = _cdr


def etuplize(x, shallow=False, return_bad_args=False, convert_ConsPairs=True):
    if isinstance(x, ExpressionTuple):
        return x
    try:
        op, args = _car(x), _cdr(x)
    except (ConsError, NotImplementedError):
        if return_bad_args:
            return x
        raise TypeError(f"x is neither a non-str Sequence nor term: {type(x)}")
    if not shallow:
        op = etuplize(op, return_bad_args=True)
        args = tuple(etuplize(a, return_bad_args=True) for a in args)
    return etuple(op, *args, evaled_obj=x)
