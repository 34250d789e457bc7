"""Module aliases the reference exposes from utils/__init__.py:9-10."""
from . import cpp_subsampling, nearest_neighbors

__all__ = ['cpp_subsampling', 'nearest_neighbors']
