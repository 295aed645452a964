"""Signed-distance initial data used by the path's drivers: shapeCylinder
(reference InitialConditions/cylinder.py:8) and shapeSphere (sphere.py:8).  Host-side, NumPy."""
import numpy as np

from .utilities import numel, warn

__all__ = ["shapeCylinder", "shapeSphere"]


def _center(grid, center):
    if center is None or not np.any(center):
        return np.zeros((grid.dim, 1))
    center = np.asarray(center, dtype=np.float64)
    if numel(center) == 1:
        return center.item() * np.ones((grid.dim, 1))
    return center.reshape(-1, 1)


def _check(data):
    if np.all(data < 0) or np.all(data > 0):
        warn('Implicit surface not visible because function has single sign on grid')


def shapeCylinder(grid, axis_align=[], center=None, radius=1):
    """sqrt(sum_{i != axis_align}(x_i - c_i)^2) - r   (cylinder.py:48-59)."""
    center = _center(grid, center)
    ignore = axis_align if isinstance(axis_align, (list, tuple)) else [axis_align]
    data = np.zeros(grid.shape)
    for i in range(grid.dim):
        if i not in ignore:
            data += (grid.xs[i] - center[i]) ** 2
    data = np.sqrt(data) - radius
    _check(data)
    return data


def shapeSphere(grid, center=None, radius=1):
    """sqrt(sum_i (x_i - c_i)^2) - r   (sphere.py:50-57)."""
    center = _center(grid, center)
    data = (grid.xs[0] - center[0]) ** 2
    for i in range(1, grid.dim):
        data = data + (grid.xs[i] - center[i]) ** 2      # not in place: xs may be sparse (low_mem grids)
    data = np.sqrt(data) - radius
    _check(data)
    return data
