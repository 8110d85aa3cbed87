"""Drop-in module for the reference's ``code/centroidal_mpc_vertices.py``: put this repository's
root ahead of the reference's ``code/`` directory on ``sys.path`` and
``import centroidal_mpc_vertices`` (code/simulation.py:6) resolves to the MI355X solver while
``centroidal_mpc_vertices.centroidal_mpc(...)`` / ``.solve(current, t)`` keep their signatures."""
import cmpc_amd  # noqa: F401  (registers the package alias)
from cmpc_amd.centroidal_mpc_vertices import centroidal_mpc  # noqa: F401
