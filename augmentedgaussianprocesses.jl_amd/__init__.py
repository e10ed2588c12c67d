"""augmentedgaussianprocesses.jl_amd -- MI355X-native engine for the SVGP + AnalyticVI/AnalyticSVI hot path of
AugmentedGaussianProcesses.jl, behind the reference's own API names (SVGP / train! / predict_y / ...).

The directory name contains a dot, so import it through the top-level alias module `agp_amd`
(`import agp_amd as AGP`), which loads this package under that name.

Host side mirrors the reference's Julia interface (same names, argument meaning, error behaviour); all numerics run
in hand-written HIP kernels behind the C ABI of include/agp_hip.h.  PyTorch is used only for device memory,
streams and torch.distributed (RCCL).
"""
from .kernels import (  # noqa: F401
    ARDTransform,
    ExponentialKernel,
    Matern32Kernel,
    Matern52Kernel,
    ScaleTransform,
    SqExponentialKernel,
    with_lengthscale,
)
from .likelihoods import (  # noqa: F401
    BayesianSVM,
    GaussianLikelihood,
    HeteroscedasticLikelihood,
    LaplaceLikelihood,
    LogisticLikelihood,
    LogisticSoftMaxLikelihood,
    NegBinomialLikelihood,
    PoissonLikelihood,
    StudentTLikelihood,
    loglikelihood,
)
from .svgp import (  # noqa: F401
    ADAM,
    Descent,
    Momentum,
    ELBO,
    MOSVGP,
    SVGP,
    AnalyticSVI,
    AnalyticVI,
    RobbinsMonro,
    objective,
    objective_enqueue,
    objective_fetch,
    SideObjective,
    predict_f,
    predict_y,
    proba_y,
    train_,
)
from .capi import AGPError  # noqa: F401
from .persistence import load_trained_model, save_trained_model  # noqa: F401
from .inducingpoints import KmeansAlg, RandomSubset, inducingpoints  # noqa: F401
from .online import (  # noqa: F401
    OIPS,
    OnlineSVGP,
    online_objective,
    online_predict_f,
    online_predict_y,
    online_proba_y,
    train_online,
)
