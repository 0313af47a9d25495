from . import disentangle
from . import residual
