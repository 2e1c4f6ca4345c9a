"""muygpys_amd -- MI355X-native (HIP, gfx950) implementation of the MuyGPyS local-GP hot path.

Only the path named in BASELINE.json lives here: gather -> pairwise/crosswise distances ->
Matern/RBF kernel -> nugget -> per-neighbourhood factorisation -> posterior mean / variance,
sigma_sq and the LOOCV loss, exposed behind MuyGPyS's backend-dispatch surface so that it
selects as ``MUYGPYS_BACKEND=hip``.  See DESIGN.md / INTEGRATION.md.
"""

__version__ = "0.1.0"
