"""gokalman_amd -- MI355X-native batched Kalman engine behind gokalman's filter interface.

The product is the C-ABI library (include/gokalman_amd.h, csrc/*.hip).  This package is
the host-side binding used by the tests and the benchmark: `batch.FilterBatch` mirrors the
reference's LDKF / NLDKF / Estimate surface over N filters at once.
"""
from . import _capi as capi  # noqa: F401
from .batch import (FilterBatch, Estimate, StaleEstimateError, MonteCarloRuns, MonteCarloRun, MonteCarloEstimate, ShardedBatch,  # noqa: F401
                    new_monte_carlo_runs, new_chi_square, van_loan)
from ._capi import KalmanError  # noqa: F401
