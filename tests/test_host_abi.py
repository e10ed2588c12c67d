"""CPU tests of the host-side mirror and the C-ABI library surface (no compute calls, no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "agp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(agp_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    from agp_amd import capi

    names = _header_functions()
    assert len(names) >= 30
    L = C.CDLL(capi.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"libagp_hip.so does not export {n} declared in include/agp_hip.h"
    # and the Python binding table covers exactly the header
    assert sorted(capi.SYMBOLS) == names
    assert capi.lib().agp_version() >= 100


def test_struct_layouts_match_header():
    from agp_amd import capi

    assert C.sizeof(capi.KernelDesc) == 40  # + has_variance / has_transform
    assert C.sizeof(capi.LikDesc) == 24
    assert C.sizeof(capi.SvgpDesc) == 4 * 4 + 3 * 8 + 24 + 3 * 8 + 8


def test_missing_library_fails_loudly(monkeypatch, built):
    from agp_amd import capi

    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libagp_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.lib()


def test_no_product_import_of_oracle():
    pkg = os.path.join(ROOT, "augmentedgaussianprocesses.jl_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.replace("oracle/", "").lower() or f == "__init__.py" and False, (
                    f"{f} mentions the oracle: the product path must not use it")


def test_kernel_objects():
    import agp_amd as AGP

    k = 2.0 * (AGP.SqExponentialKernel() @ AGP.ScaleTransform(10.0))
    assert k.variance == 2.0 and np.allclose(k.scales(3), 10.0)
    d, keep = k.desc(3)
    assert (d.kind, d.ard, d.variance, d.scale) == (0, 0, 2.0, 10.0)
    # the structure of the kernel object decides what the hyper step may touch (autotuning.jl:99-118)
    assert (d.has_variance, d.has_transform) == (1, 1)
    bare, _ = AGP.SqExponentialKernel().desc(3)
    assert (bare.has_variance, bare.has_transform, bare.variance, bare.scale) == (0, 0, 1.0, 1.0)
    wl, _ = AGP.with_lengthscale(AGP.Matern32Kernel(), 2.0).desc(3)
    assert (wl.has_variance, wl.has_transform) == (0, 1)
    ka = AGP.Matern52Kernel() @ AGP.ARDTransform([1.0, 2.0])
    d, keep = ka.desc(2)
    assert d.kind == 1 and d.ard == 1 and [d.ard_scales_host[i] for i in range(2)] == [1.0, 2.0]
    with pytest.raises(ValueError):
        ka.desc(3)
    kl = AGP.with_lengthscale(AGP.SqExponentialKernel(), 4.0)
    assert np.allclose(kl.scales(2), 0.25)
    assert "ScaleTransform" in repr(k)


def test_likelihood_labels_mirror_reference_tests():
    # test/likelihood/multiclass.jl:1-40 through the product-side host logic
    import agp_amd as AGP
    from agp_amd import likelihoods as LK

    y = [1, 2, 3, 1, 1, 2, 3]
    l = AGP.LogisticSoftMaxLikelihood(3)
    LK.create_mapping(l, y)
    assert sorted(l.class_mapping) == [1, 2, 3] and l.ind_mapping == {1: 1, 2: 2, 3: 3}
    assert np.array_equal(LK.create_one_hot(l, y[:3]), np.eye(3, dtype=bool))
    with pytest.raises(RuntimeError):
        LK.create_mapping(AGP.LogisticSoftMaxLikelihood(2), y)
    y = ["b", "a", "c", "a", "a"]
    l = AGP.LogisticSoftMaxLikelihood(3)
    Y = LK.treat_labels(y, l)
    assert l.class_mapping == ["b", "a", "c"] and l.ind_mapping == {"b": 1, "a": 2, "c": 3}
    assert np.array_equal(Y, np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 1, 0], [0, 1, 0]], bool))
    assert np.array_equal(LK.class_indices(Y), [0, 1, 2, 1, 1])
    l = AGP.LogisticSoftMaxLikelihood(["a", "b", "c"])
    assert l.ind_mapping == {"a": 1, "b": 2, "c": 3} and l.n_latent == 3
    # Bernoulli labels (classification.jl:29-44)
    assert np.array_equal(LK.treat_labels(np.array([0, 1, 1]), AGP.LogisticLikelihood()), [-1, 1, 1])
    assert np.array_equal(LK.treat_labels(np.array([True, False]), AGP.LogisticLikelihood()), [1, -1])
    with pytest.raises(ValueError):
        LK.treat_labels(np.array([1, 2, 3]), AGP.LogisticLikelihood())
    with pytest.raises(ValueError):
        AGP.StudentTLikelihood(0.4)


def test_constructor_checks_and_repr():
    # SVGP.jl:45-49 ; test/inference/analyticVI.jl:1-20
    import agp_amd as AGP

    Z = np.random.default_rng(0).random((5, 2))
    k = AGP.SqExponentialKernel()
    with pytest.raises(TypeError):
        AGP.SVGP(k, AGP.GaussianLikelihood(), "not an inference", Z)
    with pytest.raises(RuntimeError):
        AGP.SVGP(k, object(), AGP.AnalyticVI(), Z)
    # SVGP.jl:39-42,51-65: optimiser defaults to ADAM(0.01), Bool -> ADAM(0.001) / nothing ; Zoptimiser defaults to nothing
    d = AGP.SVGP(k, AGP.GaussianLikelihood(), AGP.AnalyticVI(), Z)
    assert d.k_opt.eta == 0.01 and d.z_opt is None
    t = AGP.SVGP(k, AGP.GaussianLikelihood(), AGP.AnalyticVI(), Z, optimiser=True, Zoptimiser=True)
    assert t.k_opt.eta == 0.001 and t.z_opt.eta == 0.001
    f = AGP.SVGP(k, AGP.GaussianLikelihood(), AGP.AnalyticVI(), Z, optimiser=False)
    assert f.k_opt is None
    m = AGP.SVGP(k, AGP.LogisticSoftMaxLikelihood(3), AGP.AnalyticSVI(10), Z)
    assert m.n_latent == 3 and len(m.kernels) == 3 and m.kernels[0] is not m.kernels[1]
    assert repr(AGP.AnalyticVI()) == "Analytic Variational Inference"
    assert repr(AGP.AnalyticSVI(10)) == "Analytic Stochastic Variational Inference"
    i = AGP.AnalyticSVI(10)
    assert i.stoch and i.batchsize == 10 and i.rho == 1.0 and i.n_iter == 0
    with pytest.raises(ValueError):
        AGP.RobbinsMonro(0.4)
    assert "Sparse Variational Gaussian Process" in repr(m)


def test_no_gpu_no_cpu_fallback(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import agp_amd as AGP

    Z = np.random.default_rng(0).random((5, 2))
    m = AGP.SVGP(AGP.SqExponentialKernel(), AGP.GaussianLikelihood(), AGP.AnalyticVI(), Z)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AGP.train_(m, np.zeros((10, 2)), np.zeros(10), 1)


def test_event_regression_svm_likelihood_mirrors():
    """constructors / labels of the remaining augmented likelihoods (laplace.jl:17-30, bayesiansvm.jl:19-23, poisson.jl:16-24,
    negativebinomial.jl:22-27, heteroscedastic.jl:17-47, event.jl:7-13)."""
    import numpy as np

    import agp_amd as AGP
    from agp_amd import capi
    from agp_amd.likelihoods import treat_labels

    lap = AGP.LaplaceLikelihood(3.0)
    assert lap.a == pytest.approx(1 / 9) and lap.p == 0.5 and lap.lik_desc().kind == capi.LIK_LAPLACE
    assert repr(lap) == "Laplace likelihood (β=3.0)"
    assert AGP.HeteroscedasticLikelihood(2.0).n_latent == 2
    assert repr(AGP.PoissonLikelihood(5.0)) == "Poisson Likelihood (λ = 5.0)"
    assert repr(AGP.NegBinomialLikelihood(10)) == "Negative Binomial Likelihood (r = 10)"
    assert np.array_equal(treat_labels(np.array([0, 1, 1]), AGP.BayesianSVM()), [-1.0, 1.0, 1.0])
    with pytest.raises(ValueError):
        treat_labels(np.array([0, 2]), AGP.BayesianSVM())
    with pytest.raises(TypeError):  # "For event count target(s) should be integers"
        treat_labels(np.array([1.0, 2.0]), AGP.PoissonLikelihood(2.0))
    assert treat_labels(np.array([3, 0, 2]), AGP.NegBinomialLikelihood(4)).dtype == np.float64
    for bad in (lambda: AGP.LaplaceLikelihood(0.0), lambda: AGP.PoissonLikelihood(-1.0),
                lambda: AGP.NegBinomialLikelihood(0), lambda: AGP.HeteroscedasticLikelihood(0.0)):
        with pytest.raises(ValueError):
            bad()


def test_header_is_plain_c_and_a_c_host_can_bind_it(built, tmp_path):
    """include/agp_hip.h must be consumable by a C compiler (it is what a cgo / ccall / JNI stub binds): syntax-check it as C99
    with warnings as errors, then build a tiny C host that dlopens the library and resolves + calls the GPU-free entry points."""
    import shutil
    import subprocess

    from agp_amd import capi

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    hdr = os.path.join(ROOT, "include", "agp_hip.h")
    subprocess.run([gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", hdr], check=True)
    src = tmp_path / "host.c"
    src.write_text(r'''
#include <dlfcn.h>
#include <stdio.h>
#include "agp_hip.h"
typedef int32_t (*version_fn)(void);
typedef agp_status (*info_fn)(agp_comm*, int32_t*, int32_t*, int32_t*);
int main(int argc, char** argv) {
  void* h = dlopen(argv[1], RTLD_NOW);
  if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
  version_fn v = (version_fn)dlsym(h, "agp_version");
  info_fn ci = (info_fn)dlsym(h, "agp_comm_info");
  if (!v || !ci || !dlsym(h, "agp_svgp_cavi_step_multi") || !dlsym(h, "agp_comm_init")) return 3;
  agp_svgp_desc d;   /* the structs are usable from C */
  d.flags = AGP_FLAG_STALE_K;
  agp_kernel_desc k;
  k.has_variance = 1; k.has_transform = 0;
  if (ci(0, 0, 0, 0) != AGP_ERR_INVALID) return 4;   /* argument checking works without a GPU */
  printf("%d %d %d %d\n", (int)v(), (int)sizeof(d), (int)sizeof(k), (int)d.flags + k.has_variance);
  return 0;
}
''')
    exe = tmp_path / "host"
    subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-ldl"],
                   check=True)
    env = dict(os.environ)
    # the library links libamdhip64 only; let the loader find the copy torch ships (or /opt/rocm)
    import torch

    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(os.path.dirname(torch.__file__), "lib"), "/opt/rocm/lib",
                                              env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([str(exe), capi.LIB_PATH], check=True, capture_output=True, text=True, env=env).stdout.split()
    assert int(out[0]) >= 100 and int(out[1]) == C.sizeof(capi.SvgpDesc) and int(out[2]) == C.sizeof(capi.KernelDesc)


def test_c_client_compiles_and_links_against_the_header_and_library(built, tmp_path):
    """tests/c_host/abi_smoke.c is a plain-C client of include/agp_hip.h (gcc -std=c11): it must compile without the C++ front end
    and every entry point it calls must resolve in libagp_hip.so (it is RUN by the GPU suite, tests/test_gpu_round3.py)."""
    import subprocess

    pkg = os.path.join(ROOT, "augmentedgaussianprocesses.jl_amd")
    exe = tmp_path / "abi_smoke"
    cc = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
          os.path.join(ROOT, "tests", "c_host", "abi_smoke.c"), "-o", str(exe), "-L", pkg, "-lagp_hip", "-L", "/opt/rocm/lib",
          "-lamdhip64", "-lm", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert os.path.getsize(exe) > 0


def test_bench_fails_loudly_without_a_gpu(built):
    """The product path has no CPU fallback: on a box without a GPU `bench.py` must stop with an error, not print a line; asked for
    more GPUs than the node shows, it says so before starting any rank."""
    import subprocess
    import sys

    import torch

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "GPU" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 2 and "--gpus 2 but this node shows 0 GPU(s)" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
