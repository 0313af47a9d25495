from .eval import generative_restrictiveness  # noqa: F401
