"""API-subset shim of `unification.core`: generator/trampoline based unify & reify."""
from collections import deque
from collections.abc import Generator, Mapping

from _mdispatch import Dispatcher

from .utils import transitive_get as walk
from .variable import Var, isvar


def assoc(s, u, v):
    s = dict(s)
    s[u] = v
    return s


def stream_eval(z):
    """Evaluate a (possibly nested) generator coroutine without recursion."""
    if not isinstance(z, Generator):
        return z
    stack = deque([z])
    z_args, z_out = None, None
    while stack:
        z = stack[-1]
        try:
            z_out = z.send(z_args)
            if isinstance(z_out, Generator):
                stack.append(z_out)
                z_args = None
            else:
                z_args = z_out
        except StopIteration:
            stack.pop()
    return z_out


_unify = Dispatcher("_unify")
_reify = Dispatcher("_reify")


def _unify_object(u, v, s):
    yield s if u == v else False


_unify.add((object, object, Mapping), _unify_object)


def _unify_Var_object(u, v, s):
    u_w = walk(u, s)
    v_w = walk(v, s) if isvar(v) else v
    if u_w == v_w:
        yield s
    elif isvar(u_w):
        yield assoc(s, u_w, v_w)
    elif isvar(v_w):
        yield assoc(s, v_w, u_w)
    else:
        yield _unify(u_w, v_w, s)


_unify.add((Var, (Var, object), Mapping), _unify_Var_object)
_unify.add((object, Var, Mapping), _unify_Var_object)


def _unify_seq(u, v, s):
    if len(u) != len(v):
        yield False
        return
    for uu, vv in zip(u, v):
        s = yield _unify(uu, vv, s)
        if s is False:
            return
    yield s


for _a in (tuple, list):
    for _b in (tuple, list):
        _unify.add((_a, _b, Mapping), _unify_seq)


def unify(u, v, s=None):
    if s is None:
        s = {}
    return stream_eval(_unify(u, v, s))


def _reify_object(o, s):
    yield o


_reify.add((object, Mapping), _reify_object)


def _reify_Var(o, s):
    o_w = walk(o, s)
    if o_w is o:
        yield o_w
    else:
        yield _reify(o_w, s)


_reify.add((Var, Mapping), _reify_Var)


def _reify_seq(o, s):
    res = []
    for x in o:
        res.append((yield _reify(x, s)))
    yield type(o)(res)


_reify.add((tuple, Mapping), _reify_seq)
_reify.add((list, Mapping), _reify_seq)


def reify(e, s):
    return stream_eval(_reify(e, s))
