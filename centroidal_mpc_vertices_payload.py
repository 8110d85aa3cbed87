"""Drop-in module for the reference's ``code/centroidal_mpc_vertices_payload.py``
(code/simulation_payload.py:6): same class name, payload gains k1, k2 = 7, 1."""
import cmpc_amd  # noqa: F401
from cmpc_amd.centroidal_mpc_vertices import centroidal_mpc_payload as centroidal_mpc  # noqa: F401
