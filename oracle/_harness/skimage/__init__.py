"""Stub for `skimage` -- TEST INFRASTRUCTURE ONLY.  The reference's Visualization
package imports skimage.measure at import time (Visualization/mesh_implicit.py:9);
nothing on the HJI hot path uses it."""
from . import measure  # noqa: F401
