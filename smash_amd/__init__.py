"""smash_amd -- MI355X-native forward + adjoint solver for smash's distributed rainfall-runoff path.

The package is a thin host-side mirror of the reference's ``mw_forward`` boundary over libsmashx
(HIP kernels behind the C ABI of include/smashx.h).  Importing it does not need a GPU; calling the
solver does, and fails loudly without one.
"""
from .solver import (Solver, forward, forward_b, forward_d, gradient_test, hyper_forward, hyper_forward_b, hyper_forward_d,  # noqa: F401
                     invalidate_forcing, scalar_product_test)
from .types import (Hyper_ParametersDT, Hyper_StatesDT, Input_DataDT, MeshDT, Optimize_SetupDT, OutputDT,  # noqa: F401
                    ParametersDT, SetupDT, StatesDT)
from ._lib import SmashxError  # noqa: F401
from .optimize import optimize_lbfgsb  # noqa: F401
