"""MI355X-native batched centroidal-MPC solver (hot path of the reference's
``centroidal_mpc_vertices.centroidal_mpc.solve``).

The directory name is not a valid Python identifier; import it through the
repo-root alias module ``cmpc_amd`` (``import cmpc_amd``), which loads this
package under that name.
"""
__all__ = ["footstep_planner_vertices", "foot_trajectory_generator", "functions"]
