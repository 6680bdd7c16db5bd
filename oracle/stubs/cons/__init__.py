from .core import ConsError, car, cdr  # noqa: F401
