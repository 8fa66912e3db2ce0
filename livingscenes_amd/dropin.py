"""install(): register this package's mirrors under the reference's import names so that the reference's entry
scripts (eval_3rscan.py / eval_flyingshape.py: `from lib_more.more_solver import More_Solver`,
`from lib_more.pose_estimation import *`, `from model_utils import ...`, `from lib_math import torch_se3`) resolve to
the MI355X implementation without editing them.  See INTEGRATION.md."""
import sys


def install():
    from . import lib_math, lib_more, model_utils
    from .lib_math import torch_se3
    from .lib_more import matcher_new, more_solver, pose_estimation
    sys.modules.update({
        "model_utils": model_utils,
        "lib_math": lib_math, "lib_math.torch_se3": torch_se3,
        "lib_more": lib_more, "lib_more.matcher_new": matcher_new,
        "lib_more.pose_estimation": pose_estimation, "lib_more.more_solver": more_solver,
    })
    return sys.modules["lib_more.more_solver"].More_Solver
