"""Value estimators on the MI355X kernels: the modules of torchrl.objectives.value (same names), the functional
entry points (``functional``) and the cross-GPU statistics exchange."""
from torchrl_amd.objectives.value import functional  # noqa: F401
from torchrl_amd.objectives.value.advantages import GAE, TDLambdaEstimate, TDLambdaEstimator, VTrace  # noqa: F401
from torchrl_amd.objectives.value.advantages import TD0Estimate, TD0Estimator, TD1Estimate, TD1Estimator  # noqa: F401
from torchrl_amd.objectives.value.advantages import KeyedModule, RolloutData, ValueEstimatorBase  # noqa: F401
from torchrl_amd.objectives.value.stats_exchange import StatsExchange  # noqa: F401,E402

__all__ = ["functional", "GAE", "TDLambdaEstimator", "TDLambdaEstimate", "VTrace", "TD0Estimator", "TD0Estimate",
           "TD1Estimator", "TD1Estimate", "ValueEstimatorBase", "KeyedModule", "RolloutData", "StatsExchange"]
