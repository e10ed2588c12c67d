"""KernelFunctions.jl kernel objects accepted by SVGP (reference re-exports them:
src/AugmentedGaussianProcesses.jl:31).  They are descriptors only -- evaluation happens in k_kernelmatrix (HIP).

Julia                                            here
  SqExponentialKernel()                           SqExponentialKernel()
  k ∘ ScaleTransform(s)                           k @ ScaleTransform(s)
  k ∘ ARDTransform(v)                             k @ ARDTransform(v)
  2.0 * k                                         2.0 * k
  with_lengthscale(k, l)                          with_lengthscale(k, l)     (== k ∘ ScaleTransform(1/l))
"""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np

from . import capi


class ScaleTransform:
    def __init__(self, s: float = 1.0):
        if not s > 0:
            raise ValueError("ScaleTransform: s must be positive")
        self.s = float(s)


class ARDTransform:
    def __init__(self, v):
        self.v = np.asarray(v, dtype=np.float64).copy()
        if not np.all(self.v > 0):
            raise ValueError("ARDTransform: scales must be positive")


class Kernel:
    _kind = None
    _name = "kernel"

    def __init__(self):
        self.variance = 1.0
        self.transform = None
        # structure of the Julia object this stands for: `sigma2 * k` is a ScaledKernel with a trainable sigma2, a bare kernel
        # has none (the reference's hyper step is structural: autotuning.jl:99-118, autotuning_utils.jl:47-67)
        self.has_variance = False

    def __matmul__(self, t):  # k ∘ t
        if not isinstance(t, (ScaleTransform, ARDTransform)):
            raise TypeError("only ScaleTransform / ARDTransform compose on this path")
        k = copy.deepcopy(self)
        if k.transform is not None:
            raise ValueError("kernel already has a transform")
        k.transform = t
        return k

    def __rmul__(self, a):  # a * k
        if not (isinstance(a, (int, float)) and a > 0):
            raise ValueError("kernel variance must be a positive scalar")
        k = copy.deepcopy(self)
        k.variance = k.variance * float(a)
        k.has_variance = True
        return k

    def scales(self, D: int) -> np.ndarray:
        if self.transform is None:
            return np.ones(D)
        if isinstance(self.transform, ScaleTransform):
            return np.full(D, self.transform.s)
        if len(self.transform.v) != D:
            raise ValueError("ARDTransform dimension does not match the data dimension")
        return self.transform.v.copy()

    def desc(self, D: int):
        """(KernelDesc, keepalive) for the C ABI."""
        d = capi.KernelDesc()
        d.kind = self._kind
        d.variance = self.variance
        d.has_variance = 1 if self.has_variance else 0
        d.has_transform = 0 if self.transform is None else 1
        keep = None
        if isinstance(self.transform, ARDTransform):
            keep = (C.c_double * D)(*self.scales(D))
            d.ard = 1
            d.scale = 1.0
            d.ard_scales_host = C.cast(keep, C.POINTER(C.c_double))
        else:
            d.ard = 0
            d.scale = 1.0 if self.transform is None else self.transform.s
            d.ard_scales_host = None
        return d, keep

    def __repr__(self):
        t = ""
        if isinstance(self.transform, ScaleTransform):
            t = f" ∘ ScaleTransform({self.transform.s})"
        elif isinstance(self.transform, ARDTransform):
            t = f" ∘ ARDTransform({self.transform.v.tolist()})"
        v = "" if self.variance == 1.0 else f"{self.variance} * "
        return f"{v}{self._name}(){t}"


class SqExponentialKernel(Kernel):
    _kind = capi.K_SQEXP
    _name = "SqExponentialKernel"


class Matern52Kernel(Kernel):
    _kind = capi.K_MATERN52
    _name = "Matern52Kernel"


class Matern32Kernel(Kernel):
    _kind = capi.K_MATERN32
    _name = "Matern32Kernel"


class ExponentialKernel(Kernel):
    _kind = capi.K_EXPONENTIAL
    _name = "ExponentialKernel"


def with_lengthscale(k: Kernel, l):
    if np.ndim(l) == 0:
        return k @ ScaleTransform(1.0 / float(l))
    return k @ ARDTransform(1.0 / np.asarray(l, dtype=np.float64))
