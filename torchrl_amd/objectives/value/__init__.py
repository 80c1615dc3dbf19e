from torchrl_amd.objectives.value import functional  # noqa: F401
from torchrl_amd.objectives.value.advantages import (  # noqa: F401
    GAE,
    KeyedModule,
    RolloutData,
    TD0Estimate,
    TD0Estimator,
    TD1Estimate,
    TD1Estimator,
    TDLambdaEstimate,
    TDLambdaEstimator,
    ValueEstimatorBase,
    VTrace,
)
from torchrl_amd.objectives.value.stats_exchange import StatsExchange  # noqa: F401,E402
