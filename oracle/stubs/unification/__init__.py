from .core import assoc, reify, unify  # noqa: F401
from .variable import Var, isvar, var  # noqa: F401
