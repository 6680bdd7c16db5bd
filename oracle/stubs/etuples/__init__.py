from .core import ExpressionTuple, etuple  # noqa: F401
from .dispatch import apply, etuplize, rands, rator  # noqa: F401
