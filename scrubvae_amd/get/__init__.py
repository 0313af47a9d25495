from .model import model  # noqa: F401
from .eval import latents  # noqa: F401
