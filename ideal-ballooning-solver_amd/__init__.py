"""MI355X-native batched ideal-ballooning eigen-solver (hot path of rahulgaur104/ideal-ballooning-solver).

Importable as `ibs_amd` (see ibs_amd.py at the repository root; this directory's name is not a
valid Python identifier).
"""
from ._lib import IbsError, LIB_PATH, MEM_DEVICE, MEM_HOST, SYMBOLS  # noqa: F401
from .solver import Context, ScanPlan, default_context  # noqa: F401
from .operators import gamma_ball_full, dPdrho_of, uniform_spacing, make_obj_w_grad, theta_grid, NearestSigmaWarning  # noqa: F401
from .scan import BallooningScan, shard_surfaces, gather_surfaces, gather_rows_tensor, pick_start, append_history, GEO_ORDER  # noqa: F401
from .geometry import SurfaceTables  # noqa: F401
from .config import ScanConfig, load_params_dict, theta_grid_for, create_history_placeholders, PARAMS_KEYS  # noqa: F401
from .lbfgsb import minimize2  # noqa: F401
from .objective import ballooning_objective, dof_fd_gradient, dof_steps, shard_dofs, allreduce_dof_vector, AdjointStep  # noqa: F401
