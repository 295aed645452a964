"""createGrid / processGrid: the grid Bundle the path consumes (reference
Grids/create_grid.py:13, Grids/process_grid.py:12).  Host-side setup, NumPy only."""
import numpy as np

from .boundary import addGhostExtrapolate, addGhostPeriodic
from .utilities import (Bundle, cell, error, isfield, iscell, isscalar, isvector, numel,
                        to_column_mat, isColumnLength, warn)

__all__ = ["createGrid", "processGrid"]


def createGrid(grid_min, grid_max, N, pdDims=None, process=True, low_mem=False):
    """create_grid.py:13.  Axis i is periodic iff i == pdDims (create_grid.py:61-65); as in the
    reference a falsy pdDims (None, [], and also 0) means "no periodic axis" (:34), and the
    periodic axis' max is NOT shrunk -- callers do that themselves (Notes/rcbrt.ipynb cell 3).
    Extension: a list/tuple/array of several axes marks all of them periodic."""
    multi = isinstance(pdDims, (list, tuple, np.ndarray)) and len(pdDims) > 0
    if not multi and not pdDims:
        pdDims = []
    grid_min, grid_max = np.asarray(grid_min, dtype=np.float64), np.asarray(grid_max, dtype=np.float64)
    if isscalar(N):
        N = N * np.ones(grid_min.shape).astype(np.int64)
    N = np.asarray(N)
    if grid_min.ndim == 1:
        grid_min, grid_max = grid_min.reshape(-1, 1), grid_max.reshape(-1, 1)
    if N.ndim == 1:
        N = N.reshape(-1, 1)
    if not isvector(grid_min) or not isvector(grid_max) or not isvector(N):
        error('grid_min, grid_max, N must all be vectors!')
    assert numel(grid_min) == numel(grid_max), 'grid min and grid_max must have the same number of elements!'
    assert numel(grid_min) == numel(N), 'grid min, grid_max, and N must have the same number of elements!'
    grid_min, grid_max, N = to_column_mat(grid_min), to_column_mat(grid_max), to_column_mat(N).astype(np.int64)
    g = Bundle(dict(dim=len(grid_min), min=grid_min, max=grid_max, N=N, bdry=cell(len(grid_min), 1)))
    g.axis_align = pdDims
    for i in range(g.dim):
        periodic = (i in list(pdDims)) if multi else bool(np.any(i == pdDims))
        g.bdry[i] = addGhostPeriodic if periodic else addGhostExtrapolate
    if process:
        g = processGrid(g, sparse_flag=bool(low_mem))
    return g


def processGrid(gridIn, data=None, sparse_flag=False):
    """process_grid.py:12: fill in dx (:185), vs (:204), xs (:234), bdry, bdryData (:264), axis,
    shape (:293) and check consistency.  Accepts a Bundle (or a scalar dim, as the reference)."""
    if isinstance(gridIn, Bundle):
        g = gridIn
    elif isscalar(gridIn):
        g = Bundle(dict(dim=int(gridIn)))
    else:
        error('Unknown format for gridIn parameter')
    defaultMin, defaultMax, defaultN = 0, 1, 101
    if not isfield(g, 'dim'):
        error('grid structure must contain dim field')
    if g.dim < 1:
        error('dimension must be positive')
    if not isfield(g, 'min'):
        g.min = defaultMin * np.ones((g.dim, 1))
    elif isscalar(g.min):
        g.min = float(np.asarray(g.min).item()) * np.ones((g.dim, 1))
    elif not isColumnLength(g.min, g.dim):
        error('min field is not column vector of length dim or a scalar')
    if not isfield(g, 'max'):
        g.max = defaultMax * np.ones((g.dim, 1))
    elif isscalar(g.max):
        g.max = float(np.asarray(g.max).item()) * np.ones((g.dim, 1))
    elif not isColumnLength(g.max, g.dim):
        error('max field is not column vector of length dim or a scalar')
    if np.any(g.max <= g.min):
        error('max bound must be strictly greater than min bound in all dimensions')
    if isfield(g, 'N'):
        if isscalar(g.N):
            g.N = int(np.asarray(g.N).item()) * np.ones((g.dim, 1), dtype=np.int64)
        if np.any(g.N <= 0):
            error('number of grid cells must be strictly positive')
        if not isColumnLength(g.N, g.dim):
            error('N field is not column vector of length dim or a scalar')
    if isfield(g, 'dx'):
        if isscalar(g.dx):
            g.dx = float(np.asarray(g.dx).item()) * np.ones((g.dim, 1))
        if np.any(g.dx <= 0):
            error('grid cell size dx must be strictly positive')
        if not isfield(g, 'N'):
            g.N = np.rint((g.max - g.min) / g.dx).astype(np.int64) + 1
    elif isfield(g, 'N'):
        g.dx = np.divide(g.max - g.min, g.N - 1)                       # :185
    else:
        warn('Neither fields dx nor dN is present, so use default N and infer dx')
        g.N = defaultN * np.ones((g.dim, 1), dtype=np.int64)
        g.dx = np.divide(g.max - g.min, g.N - 1)
    if isfield(g, 'vs'):
        if not iscell(g.vs) or len(g.vs) != g.dim:
            error('vs field is not column cell vector of length dim: %d' % g.dim)
        for i in range(g.dim):
            if not isColumnLength(g.vs[i], g.N[i, 0]):
                error('vs cell entry is not correctly sized vector')
    else:
        g.vs = [np.expand_dims(np.linspace(g.min[i, 0].item(), g.max[i, 0].item(),
                                           num=int(g.N[i, 0])), 1) for i in range(g.dim)]   # :204
    for i in range(g.dim):
        if g.N[i, 0] != len(g.vs[i]):
            error('Inconsistent grid size in dimension %d' % i)
    if not isfield(g, 'xs'):
        g.xs = np.meshgrid(*g.vs, indexing='ij', sparse=sparse_flag)                            # :234
    if isfield(g, 'bdry'):
        if callable(g.bdry):
            g.bdry = [g.bdry for _ in range(g.dim)]
        elif len(g.bdry) != g.dim:
            error('bdry field is not column cell vector of length dim: %d' % g.dim)
    else:
        g.bdry = [addGhostPeriodic for _ in range(g.dim)]                                       # default :15-16
    if isfield(g, 'bdryData'):
        if not iscell(g.bdryData):
            g.bdryData = [g.bdryData for _ in range(g.dim)]
        elif len(g.bdryData) != g.dim:
            error('bdryData field is not column cell vector of length dim: %d' % g.dim)
    else:
        g.bdryData = [None for _ in range(g.dim)]                                               # :264
    if g.dim in (2, 3):
        g.axis = np.zeros((1, 2 * g.dim), dtype=np.float64)
        for i in range(g.dim):
            g.axis[0, 2 * i:2 * i + 2] = [g.min[i, 0].item(), g.max[i, 0].item()]
    else:
        g.axis = []
    Nshape = tuple(int(x) for x in np.asarray(g.N).ravel())
    shape = Nshape + (1,) if g.dim == 1 else Nshape                                             # :293
    if isfield(g, 'shape') and tuple(g.shape) != shape:
        error('shape and N fields do not agree')
    g.shape = shape
    if data is not None:
        if np.ndim(data) != len(g.shape):
            error('data parameter does not agree in dimension with grid')
        if tuple(np.shape(data)) != tuple(g.shape):
            error('data parameter does not agree in array size with grid')
    return g
