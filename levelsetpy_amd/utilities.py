"""MATLAB-isms the path relies on: Bundle, cell, size, expand, isfield, ... --
host-side mirror of the reference's Utilities/matlab_utils.py (names and
behaviour kept; written from the observable behaviour of each helper)."""
import logging
import sys
import time

import numpy as np

__all__ = ["Bundle", "cell", "iscell", "isbundle", "isfield", "size", "numel", "ndims", "numDims",
           "length", "expand", "zeros", "ones", "strcmp", "error", "info", "warn", "cputime",
           "eps", "realmax", "realmin", "DEFAULT_ORDER", "isscalar", "isvector",
           "isColumnLength", "to_column_mat", "omin", "omax"]

logger = logging.getLogger("levelsetpy_amd")

realmin = sys.float_info.min
realmax = sys.float_info.max          # matlab_utils.py:34
eps = sys.float_info.epsilon          # matlab_utils.py:35
DEFAULT_ORDER = "C"                   # matlab_utils.py:36


class Bundle(object):
    """struct-like attribute bag (matlab_utils.py:41-57)."""

    def __init__(self, dicko=None):
        for k, v in (dicko or {}).items():
            object.__setattr__(self, k, v)

    def __dtype__(self):
        return Bundle

    def __len__(self):
        return len(self.__dict__)

    def keys(self):
        return list(self.__dict__.keys())

    def __repr__(self):
        return "Bundle(%s)" % ", ".join(sorted(self.__dict__))


def cell(n, dim=1):
    return [np.nan for _ in range(n)]


def iscell(c):
    return isinstance(c, list)


def isbundle(b):
    return isinstance(b, Bundle)


def isfield(b, field):
    return field in b.__dict__


def size(A, dim=None):
    if isinstance(A, list):
        A = np.asarray(A)
    return tuple(A.shape) if dim is None else A.shape[dim]


def numel(A):
    if isinstance(A, list):
        A = np.asarray(A)
    return int(np.size(A)) if isinstance(A, np.ndarray) or np.isscalar(A) else int(A.numel())


def numDims(A):
    if isinstance(A, list):
        A = np.asarray(A)
    return A.ndim


ndims = numDims


def length(A):
    if isinstance(A, list):
        A = np.asarray(A)
    return max(A.shape)


def expand(x, ax):
    if isinstance(x, np.ndarray):
        return np.expand_dims(x, ax)
    return x.unsqueeze(ax)


def zeros(rows, cols=None, dtype=np.int64):
    if cols is not None:
        return np.zeros((rows, cols), dtype=dtype)
    return np.zeros(rows if isinstance(rows, tuple) else (rows, rows), dtype=dtype)


def ones(rows, cols=None, dtype=np.int64):
    return np.ones((rows, cols) if cols is not None else (rows, rows), dtype=dtype)


def strcmp(a, b):
    return a == b


def error(arg):
    """matlab_utils.py:134-137: raises ValueError."""
    assert isinstance(arg, str)
    raise ValueError(arg)


def info(arg):
    logger.info(arg)


def warn(arg):
    logger.warning(arg)


def cputime():
    return time.time()


def isscalar(x):
    if isinstance(x, np.ndarray):
        return x.size == 1
    return not isinstance(x, list)


def isvector(x):
    m, n = x.shape
    return m == 1 or n == 1


def isColumnLength(x1, x2):
    if isinstance(x1, list):
        x1 = np.expand_dims(np.asarray(x1), 1)
    return x1.ndim == 2 and x1.shape[0] == x2 and x1.shape[1] == 1


def to_column_mat(A):
    n, m = A.shape
    return A.T if n < m else A


def omin(y, ylast):
    """Elementwise minimum (the intended operator; the reference's version collapses to a
    scalar, matlab_utils.py:90-100 -- SURVEY Appendix D)."""
    return np.minimum(y, np.reshape(ylast, np.shape(y)))


def omax(y, ylast):
    return np.maximum(y, np.reshape(ylast, np.shape(y)))
