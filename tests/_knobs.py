"""Which of the library's fallback-forcing environment variables (include/agp_hip.h, "Environment") is set for this run.
tools/suite_with_fallbacks.sh runs the GPU suite once per fallback; the numerical assertions hold on every path, the assertions on
the PATH COUNTERS ("the step rode on the next launch", "the fused product was used") describe the default configuration only."""
import os

_DEFAULTS = {"AGP_CHOL_DAG": None, "AGP_CHAIN_SPLIT": None, "AGP_STEP_PROLOGUE": None, "AGP_STEP_EPILOGUE": None,
             "AGP_PF_INKERNEL": None, "AGP_KERNELMATRIX_VALU": None, "AGP_HYPER_GK_FUSED": None, "AGP_SPLIT_MERGED": None}


def forced(*names):
    """True when one of the named variables is set (to anything): the path it switches off cannot be asserted on."""
    for n in names:
        assert n in _DEFAULTS, n
        if os.environ.get(n) not in (None, ""):
            return True
    return False


def no_task_graph():
    return os.environ.get("AGP_CHOL_DAG") == "0"


def no_prologue():
    return no_task_graph() or os.environ.get("AGP_STEP_PROLOGUE") == "0"
