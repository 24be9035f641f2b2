from .mappo import MAPPO
from .cmappo import CMAPPO

__all__ = ["MAPPO", "CMAPPO"]
