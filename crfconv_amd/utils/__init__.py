"""Module aliases the reference exposes from utils/__init__.py:9-10, plus its running metrics (:3)."""
from . import cpp_subsampling, nearest_neighbors
from .metrics import runningScore, runningScoreShapeNet

__all__ = ['cpp_subsampling', 'nearest_neighbors', 'runningScore', 'runningScoreShapeNet']
