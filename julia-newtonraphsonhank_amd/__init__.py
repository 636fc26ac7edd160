"""MI355X-native sequence-space Newton–Raphson solver for heterogeneous-agent models.

Drop-in for the JVP hot path of vasudeva-ram/Julia-NewtonRaphsonHANK: the reference's call surface
(ModelParser / BackwardIteration / ForwardIteration / JVP / NewtonRaphsonHANK) with the household
block implemented as hand-written HIP kernels for gfx950 behind a C ABI (include/hank_hip.h).
"""
from .dual import Dual
from .GeneralStructures import (ComputationalSpec, HeterogeneityDimension, SequenceModel, SteadyStateSpec,
                                Variable, JVP, RayleighQuotient, assemble_full_xMat, double_exponential,
                                generate_exog_paths, get_RouwenhorstDiscretization, invariant_dist,
                                make_DoubleExponentialGrid, n_total, rouwenhorst_discretization, shift_lag,
                                shift_lead, var_names, vars_of_type)
from .ModelParser import build_model_from_yaml, compile_residuals, detect_max_lag_lead, transform_expr
from . import KrusellSmith
from .KrusellSmith import ValueFunction, exogenousZ
from .Aggregation import Residuals
from .BackwardIteration import BackwardIteration, household_block
from .ForwardIteration import ForwardIteration, make_endogenous_transition, transition_step
from .SteadyState import SSAssembler, SteadyState, find_ss, get_SteadyStates
from .NewtonRaphson import LinearizedFunction, NewtonRaphsonHANK, make_fullFunction, y_Iteration
from .SteadyStateJacobian import getSteadyStateJacobian
from .hip import HouseholdBlock, HankHIPError, KnotsNotSortedError, DomainError, NoDeviceError

__all__ = [n for n in dir() if not n.startswith("_")]
