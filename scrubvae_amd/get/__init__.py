from .model import model  # noqa: F401
