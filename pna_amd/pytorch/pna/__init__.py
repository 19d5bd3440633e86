from .layer import PNALayer, PNATower  # noqa: F401
from .aggregators import AGGREGATORS  # noqa: F401
from .scalers import SCALERS  # noqa: F401
