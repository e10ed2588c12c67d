/*
 * agp_hip.h -- C ABI of libagp_hip.so : MI355X (gfx950) engine for the SVGP + AnalyticVI/AnalyticSVI
 * hot path of AugmentedGaussianProcesses.jl (reference paths are relative to /root/reference).
 *
 * The reference has no FFI; this header is what a `ccall` shim binds (see INTEGRATION.md and
 * julia/AGPHip.jl).  Every entry point names the reference method(s) it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; no exceptions cross the boundary; every call returns agp_status.
 *   - all data pointers are DEVICE pointers unless the parameter name ends in `_host`.
 *   - element type T is double (AGP_F64) or float (AGP_F32), fixed per handle / per call.
 *   - inputs X, Z are POINT-MAJOR: point i occupies X[i*ldx .. i*ldx+D) (Julia `ColVecs` memory order;
 *     a `RowVecs` N x D column-major matrix is permuted once by the shim at upload).
 *   - m x m outputs are dense row-major with leading dimension m (all are symmetric or explicitly
 *     lower-triangular, so Julia column-major readers see the transpose == the same matrix / upper factor).
 *   - caller owns all data buffers; the library owns the opaque handles and their device workspaces.
 *   - work is enqueued asynchronously on the ctx stream; calls that return host scalars or a
 *     data-dependent status (refresh_K, elbo, check_status, ctx_sync) synchronise that stream.
 *
 * Environment (all the library reads; 17 variables, each read once per process unless said otherwise).  The first eight switch a
 * default path off for the FALLBACK that also exists on its own -- the GPU suite is run once with each of them, AGP_CHAIN_SPLIT both
 * ways (tools/suite_with_fallbacks.sh, profiles/r05_fallback_suites.txt); AGP_CHOL_GROUP, AGP_CHOL_LOOKAHEAD and the test hook
 * AGP_DAG_TEST_ABORT are exercised by tests of their own.  The A/B levers of earlier rounds are gone (docs/DESIGN_LOG.md has their numbers).
 *   AGP_CHOL_DAG=0|1          never / always factor with the one-launch tile task graph (default: up to 32 block columns, i.e. m <= 2048;
 *                             beyond, and after a lost dependency, plain launches per block column / blocked panels)
 *   AGP_CHAIN_SPLIT=0|1       the task graph as one kernel / as chain kernel + tile kernel (default: two kernels from 600 tiles,
 *                             except fp64 launches that carry the natural-gradient step as their prologue: those split only when
 *                             forced.  Round 6: the step's stream starts the tile kernel only when every chain workgroup of the
 *                             launch is resident -- DagSync::here, k_wait_here -- which removed the one way a split launch lost a
 *                             dependency on its own, docs/DESIGN_LOG.md sections 14 and 15)
 *   AGP_STEP_PROLOGUE=0       the natural-gradient step of a single-latent CAVI step as a kernel of its own (k_syrk_tn<SY_ETA2>)
 *                             instead of the prologue of the next step's task-graph launch
 *   AGP_STEP_EPILOGUE=0       the row statistics + local update as a kernel of their own instead of the launch's epilogue
 *   AGP_PF_INKERNEL=0         look-ahead stream handed over by events instead of polled words in signal memory
 *   AGP_KERNELMATRIX_VALU=1   kernel matrices by the direct-difference VALU kernel instead of the MFMA form (the path of D > 128)
 *   AGP_HYPER_GK_FUSED=0      hyper-gradient with kappa' H and K^-1 Sigma K^-1 (the form of handles without a prologue launch:
 *                             several latents, batch-sharded, online, stale-K) instead of the one product C (Sigma K^-1)
 *   AGP_SPLIT_MERGED=0        batch-parallel step: eta step and row statistics as kernels of their own (k_eta2_from_packed)
 *   AGP_GEMM_TALL=0|1         (round 6) never / wherever the shape allows: the kappa-type products on 128 x 64 C tiles (k_gemm_nt_tall) instead of
 *                             64 x 64 ones (default: fp64 products of more than 1100 64-tiles, i.e. C5's kappa GEMM)
 *   AGP_CHOL_GROUP=n          block columns per group of the blocked factorisation beyond the task graph (default 8; 1 = per column)
 *   AGP_CHOL_LOOKAHEAD=0      ... without its side stream
 *   AGP_DAG_TEST_ABORT=1      test hook: every task-graph launch of a CAVI step is treated as having lost a dependency (the in-stream
 *                             fallback k_chol_safe / k_safe_rowstats redoes it)
 *   AGP_DAG_TEST_OVERSUBSCRIBE=1  test hook (with AGP_DAG_TEST_ABORT=1): the fallback's grid is made too large to be resident at once, so
 *                             its grid barrier runs into its limit (8 s) and the step ends in status -3 / AGP_ERR_HIP instead of a hang
 *   AGP_SPLIT_OVERLAP=1       (read at every step) batch-parallel statistics travel in block-column groups next to the next
 *                             factorisation; see "multi-GPU" below
 *   AGP_FORCE_SPLIT=1         diagnostic: the batch-parallel step sequence with a one-rank communicator, its collective issued
 *   AGP_ALLOW_PARTIAL_SHARD=1 diagnostic: a latent-sharded multi-output handle may step without its communicator (bench.py c5, one GPU)
 *   AGP_RCCL_PATH=<file>      the librccl.so to bind when the process holds none (agp_comm_init)
 */
#ifndef AGP_HIP_H
#define AGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t agp_status;
enum {
  AGP_OK = 0,
  AGP_ERR_INVALID = 1,     /* bad argument */
  AGP_ERR_NOT_POSDEF = 2,  /* Julia PosDefException from cholesky(K + jitt*I)   src/gpblocks/latentgp.jl:206 */
  AGP_ERR_NEG_KTILDE = 3,  /* error("K~ has negative values")                   src/gpblocks/latentgp.jl:213 */
  AGP_ERR_BAD_BATCH = 4,   /* batch-size check                                  src/training/training.jl:27-29 */
  AGP_ERR_UNSUPPORTED = 5, /* implemented(likelihood, inference) == false       src/models/SVGP.jl:48-49 */
  AGP_ERR_LABELS = 6,      /* treat_labels! ArgumentError                       src/likelihood/classification.jl:36-44 */
  AGP_ERR_HIP = 7,         /* HIP runtime failure (message in agp_last_error) */
  AGP_ERR_NOMEM = 8
};

enum { AGP_F64 = 0, AGP_F32 = 1 };

/* KernelFunctions.jl kernels (call sites src/gpblocks/latentgp.jl:202-212): sigma2 * base(||s .* (x-y)||) */
enum { AGP_K_SQEXP = 0, AGP_K_MATERN52 = 1, AGP_K_MATERN32 = 2, AGP_K_EXPONENTIAL = 3 };

/* likelihoods with closed-form augmented updates (implemented(l, ::AnalyticVI) == true):
 * src/likelihood/{gaussian,logistic,studentt,logisticsoftmax,laplace,bayesiansvm,poisson,negativebinomial,heteroscedastic}.jl */
enum {
  AGP_LIK_GAUSSIAN = 0,
  AGP_LIK_LOGISTIC = 1,
  AGP_LIK_STUDENTT = 2,
  AGP_LIK_LOGISTICSOFTMAX = 3,
  AGP_LIK_MULTIOUTPUT = 4, /* MOSVGP handle: task likelihoods are installed by agp_svgp_set_multioutput */
  AGP_LIK_LAPLACE = 5,         /* LaplaceLikelihood(beta)            laplace.jl:17-25          p0 = beta */
  AGP_LIK_BAYESIANSVM = 6,     /* BayesianSVM()                      bayesiansvm.jl:19-23 */
  AGP_LIK_POISSON = 7,         /* PoissonLikelihood(lambda)          poisson.jl:16-24          p0 = initial lambda (state) */
  AGP_LIK_NEGBINOMIAL = 8,     /* NegBinomialLikelihood(r)           negativebinomial.jl:22-27 p0 = r */
  AGP_LIK_HETEROSCEDASTIC = 9  /* HeteroscedasticLikelihood(lambda)  heteroscedastic.jl:17-47  p0 = initial lambda (state);
                                  n_latent must be 2 (latent 0 = f, latent 1 = g) */
};

/* ELBO variants: Appendix-A Q2 of SURVEY.md (src/likelihood/logistic.jl:82 uses dot(theta, mu)) */
enum { AGP_ELBO_CORRECTED = 0, AGP_ELBO_REFERENCE = 1 };

/* agp_svgp_desc.flags.
 * AGP_FLAG_STALE_K ("reference_compat_stale_K"): inside one train! the reference never recomputes
 * the Cholesky of K_ZZ after a hyper-parameter step -- compute_kernel_matrices (src/training/training.jl:187-208) only does
 * so while isHPupdated(inference), which update_hyperparameters! for sparse models never sets again (the call is commented
 * out, src/hyperparameter/autotuning.jl:41-46).  kappa = Knm / K then mixes the NEW kernel and Z (Knm, kdiag) with the OLD
 * factor, and natural_gradient! uses the old inv(K), until train! ends (compute_Ks, training.jl:107) or restarts without a
 * state.  The hyper-gradient itself is taken with fresh matrices (ELBO(m, x, y, mu0, ks, Zs, state) recomputes them,
 * src/functions/ELBO.jl:15-21).  With the flag a hyper step leaves the step-side matrices (L, inv(K), K\mu0) as they are and
 * only an explicit agp_svgp_refresh_K (what the host calls where train! starts and ends) recomputes them; without it
 * (default) K is refreshed before the next step. */
enum { AGP_FLAG_STALE_K = 1 };

/* matrices readable through agp_svgp_get_matrix (for parity tests and the shim's state export) */
enum {
  AGP_MAT_L = 0,      /* m x m lower Cholesky factor of K_ZZ + jitter I          (state.kernel_matrices.K) */
  AGP_MAT_KINV = 1,   /* m x m inv(K)                                             (analyticVI.jl:179) */
  AGP_MAT_KNM = 2,    /* B x m                                                    (latentgp.jl:210) */
  AGP_MAT_KAPPA = 3,  /* B x m                                                    (latentgp.jl:211) */
  AGP_VEC_KTILDE = 4, /* B                                                        (latentgp.jl:212) */
  AGP_VEC_MEAN_F = 5, /* B   kappa*mu   (value used by the last local update)     (latentgp.jl:179) */
  AGP_VEC_VAR_F = 6,  /* B                                                        (latentgp.jl:189) */
  AGP_VEC_THETA = 7,  /* B   local variable theta                                 (likelihood local_updates!) */
  AGP_VEC_C = 8,      /* B   local variable c (Laplace: b).  Heteroscedastic: latent 0 -> phi, latent 1 -> c */
  AGP_VEC_GAMMA = 9,  /* B   LogisticSoftMax gamma_k ; Poisson gamma ; Heteroscedastic: latent 0 -> gamma, latent 1 -> sigg */
  AGP_VEC_ALPHA = 10  /* B   LogisticSoftMax alpha (shared by all latents) */
};

typedef struct agp_ctx agp_ctx;
typedef struct agp_svgp agp_svgp;

typedef struct {
  int32_t kind;                /* AGP_K_* */
  int32_t ard;                 /* 0: ScaleTransform(scale)   1: ARDTransform(ard_scales_host[0..D)) */
  double variance;             /* sigma2 of `sigma2 * k` (1 if none) */
  double scale;                /* s of ScaleTransform(s) ; with_lengthscale(k, l) == scale 1/l */
  const double* ard_scales_host; /* host pointer, length D, read at call time (only if ard) */
  /* Structure of the kernel OBJECT, which decides what the hyper-parameter step may touch: the reference differentiates the
   * kernel structurally (Zygote NamedTuple, autotuning.jl:99-118 -> update_kernel!, autotuning_utils.jl:47-67), so only
   * parameters that exist are stepped.  has_variance: the kernel is `sigma2 * k` (a ScaledKernel); has_transform: it is
   * `k o ScaleTransform / ARDTransform` (with_lengthscale included).  A bare SqExponentialKernel() has neither: its
   * hyper step leaves the kernel untouched (Z may still move).  Evaluation ignores both flags. */
  int32_t has_variance;
  int32_t has_transform;
} agp_kernel_desc;

typedef struct {
  int32_t kind;    /* AGP_LIK_* */
  int32_t n_class; /* LogisticSoftMax: K (= number of latent GPs); else 1 */
  double p0;       /* Gaussian: sigma2 ; StudentT: nu ; Laplace: beta ; NegBinomial: r ; Poisson / Heteroscedastic: lambda_0 */
  double p1;       /* StudentT: sigma ; Gaussian: learning rate of the optional noise optimiser -- GaussianLikelihood(sigma2;
                      opt_noise = ADAM(p1)), gaussian.jl:18-23,56-72; 0 = noise fixed.  With it sigma2 is STATE (stepped by every
                      local update in log space, read / set with agp_svgp_get/set_lik_param) */
} agp_lik_desc;

typedef struct {
  int32_t dtype;       /* AGP_F64 / AGP_F32  (SVGP(...; T=Float64) src/models/SVGP.jl:43) */
  int32_t n_latent;    /* latents held by THIS handle (latent-parallel ranks hold a slice) */
  int32_t latent_offset; /* global index of this handle's first latent (class index for LogisticSoftMax) */
  int32_t stochastic;  /* 0 AnalyticVI (Descent(1)) ; 1 AnalyticSVI (RobbinsMonro) analyticVI.jl:44-52 */
  int64_t m;           /* inducing points per latent */
  int64_t D;           /* input dimension */
  int64_t max_batch;   /* largest B ever passed to step / elbo */
  agp_lik_desc lik;
  double jitter;       /* <= 0 : reference default 1e-4 (f64) / 1e-3 (f32)  src/functions/utils.jl:8-9 */
  double rm_kappa;     /* RobbinsMonro kappa (0.51)  src/inference/optimisers.jl:6 */
  double rm_tau;       /* RobbinsMonro tau   (1)     */
  int32_t elbo_mode;   /* AGP_ELBO_* */
  int32_t flags;       /* AGP_FLAG_* */
} agp_svgp_desc;

/* ---- context ------------------------------------------------------------------------------------- */
int32_t agp_version(void);
/* hip_stream: a hipStream_t shared with the caller (AMDGPU.jl / torch) or NULL for the default stream */
agp_status agp_ctx_create(int32_t device, void* hip_stream, agp_ctx** out);
agp_status agp_ctx_destroy(agp_ctx* ctx);
agp_status agp_ctx_sync(agp_ctx* ctx);
/* Diagnostics (round 6): how many task-graph factorisation launches of this context lost a tile dependency and were re-run by their
 * in-stream fallback (k_chol_safe / k_safe_rowstats) since the context was created.  0 on a GPU this process has to itself -- a
 * non-zero count means steps that cost milliseconds instead of 0.3 and a trajectory that is correct to rounding but no longer the
 * bitwise one; bench.py prints it as `task_graph_fallbacks`.  Synchronises the context's stream. */
agp_status agp_ctx_task_graph_fallbacks(agp_ctx* ctx, int64_t* n_host);
const char* agp_last_error(agp_ctx* ctx);

/* ---- building blocks (unit parity) ---------------------------------------------------------------- */
/* kernelmatrix(k, X, Y) / kernelmatrix(k, X) (y == NULL -> symmetric)   src/gpblocks/latentgp.jl:206,210
 * idx (nullable, int64[n]) gathers rows of X: row i of the result uses X[idx[i]] (the view(X, minibatch) of
 * src/training/training.jl:54).  out is n x p row-major with leading dimension ldo. */
agp_status agp_kernelmatrix(agp_ctx* ctx, int32_t dtype, const agp_kernel_desc* k, const void* x, int64_t n,
                            int64_t ldx, const int64_t* idx, const void* y, int64_t p, int64_t ldy, int64_t D,
                            void* out, int64_t ldo);
/* cholesky(A + jitter*I) in place, lower factor (strict upper zeroed).  *info_host = 0 ok, k>0: leading minor
 * k not positive definite (LAPACK potrf convention == Julia PosDefException.info).  src/gpblocks/latentgp.jl:206 */
agp_status agp_potrf_jitter(agp_ctx* ctx, int32_t dtype, void* a, int64_t lda, int64_t n, double jitter,
                            int32_t* info_host);
/* inv(A) for SPD A via Cholesky (inv(K::Cholesky) analyticVI.jl:179 ; -inv(eta2)/2 inference.jl:26) */
agp_status agp_spd_inverse(agp_ctx* ctx, int32_t dtype, const void* a, int64_t lda, int64_t n, void* ainv,
                           int64_t ldi, double* logdet_host, int32_t* info_host);
/* X = B / cholesky(A)  (Knm / K, two triangular solves in the reference, latentgp.jl:211); b is r x n */
agp_status agp_solve_right_spd(agp_ctx* ctx, int32_t dtype, const void* a, int64_t lda, int64_t n, const void* b,
                               int64_t ldb, int64_t r, void* x, int64_t ldx, int32_t* info_host);
/* ---- inducing-point selection (the step before the path) ----------------------------------------------------------------
 * `inducingpoints(KmeansAlg(m), X)` is how every reference example / test picks Z (test/testingtools.jl:66,
 * docs/examples/gpclassification.jl:47, docs/src/userguide.md:140-143).  The algorithm is third party and unvendored
 * (InducingPoints.jl kmeans_ip = AFK-MC2 seeding + Clustering.kmeans!(X, C; tol)); the deterministic part -- Lloyd
 * iterations from given seeds -- runs here, the random seeding stays with the caller (it needs the caller's RNG).
 *
 * agp_nearest_center: labels_out[i] = argmin_j ||x_i - c_j||^2 (ties -> smaller j), mind_out[i] = that squared distance
 *   (either output may be NULL); device pointers, row-major, x is n x D (ldx), centers m x D (ldc); D <= 128.
 * agp_kmeans: centers (in: seeds, out: result) ; Clustering.kmeans! control flow: assign, then repeat { centres <- cluster
 *   means (an emptied cluster keeps its centre) ; assign ; stop when |cost change| < tol } at most max_iter times.
 *   labels_out / counts_out: nullable device int32[n] / int32[m] of the final assignment.  Synchronises. */
agp_status agp_nearest_center(agp_ctx* ctx, int32_t dtype, const void* x, int64_t n, int64_t ldx, int64_t D,
                              const void* centers, int64_t ldc, int64_t m, int32_t* labels_out, void* mind_out);
agp_status agp_kmeans(agp_ctx* ctx, int32_t dtype, const void* x, int64_t n, int64_t ldx, int64_t D, void* centers,
                      int64_t ldc, int64_t m, int32_t max_iter, double tol, int32_t* labels_out, int32_t* counts_out,
                      int32_t* iters_host, double* objective_host, int32_t* converged_host);
/* MFMA microbenchmark: TFLOP/s of back-to-back v_mfma_{f64,f32}_16x16x4 (roofline ceiling measurement) */
agp_status agp_mfma_peak(agp_ctx* ctx, int32_t dtype, double* tflops_host);

/* ---- SVGP model handle ---------------------------------------------------------------------------- */
/* SVGP(kernel, likelihood, AnalyticVI()/AnalyticSVI(B), Z)  src/models/SVGP.jl:33-80 ;
 * posterior init mu=0, Sigma=I, eta1=0, eta2=-I/2            src/gpblocks/posterior.jl:29-37 */
agp_status agp_svgp_create(agp_ctx* ctx, const agp_svgp_desc* desc, agp_svgp** out);
agp_status agp_svgp_destroy(agp_svgp* h);
/* per-latent kernel and inducing points (each latent owns a copy, latentgp.jl:63-68); marks K stale */
agp_status agp_svgp_set_kernel(agp_svgp* h, int32_t latent, const agp_kernel_desc* k);
agp_status agp_svgp_set_Z(agp_svgp* h, int32_t latent, const void* z, int64_t ldz);
agp_status agp_svgp_get_Z(agp_svgp* h, int32_t latent, void* z, int64_t ldz);
/* prior mean values at Z (NULL == ZeroMean, src/mean/zeromean.jl:17) */
agp_status agp_svgp_set_prior_mean(agp_svgp* h, int32_t latent, const void* mu0);
/* compute_K for every latent with a stale K: K = k(Z,Z)+jitt I = L L', inv(K)   latentgp.jl:205-207.
 * Synchronises; AGP_ERR_NOT_POSDEF if any factorisation failed. */
agp_status agp_svgp_refresh_K(agp_svgp* h);
/* RobbinsMonro counters (state_eta1/state_eta2, src/training/states.jl:63-70); n starts at 1 */
agp_status agp_svgp_set_opt_state(agp_svgp* h, int64_t n);
agp_status agp_svgp_get_opt_state(agp_svgp* h, int64_t* n_host);

/* update_parameters!(model::SVGP, state, x, y)  src/training/training.jl:140-144 : one CAVI step on the
 * minibatch X[idx[0..B)] for all latents of this handle.
 *   y   : T[N] (+-1 labels / real targets) ; LogisticSoftMax: int32[N] 0-based class index
 *   idx : int64[B] device, or NULL for rows 0..B-1 (full batch)
 *   rho : N / B   (training.jl:30)
 * Asynchronous; data-dependent failures (K~ <= 0, non-SPD -2*eta2) are latched and reported by
 * agp_svgp_check_status / the next synchronising call.
 * Scheduling note (round 3, invisible through the ABI): for a single-latent handle in a training loop (look-ahead or full batch,
 * up to 16 block columns of 64) the natural-gradient step of this minibatch (natural_gradient! + global_update!,
 * analyticVI.jl:143-180, 229-246) is not enqueued by this call but rides as the PROLOGUE of the NEXT step's factorisation launch;
 * every other entry point of the handle first takes a pending step with the stand-alone kernel, so eta, Sigma, the ELBO, the
 * hyper-gradient and predictions always see the completed step (AGP_STEP_PROLOGUE=0 switches the scheduling off).  x, y, idx of a
 * step are not read again after the call that follows it. */
agp_status agp_svgp_cavi_step(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx,
                              int64_t B, double rho);
/* Diagnostics of the scheduling described above: number of agp_svgp_cavi_step calls on this handle, and how many of them had their
 * natural-gradient part taken as the prologue of the following step's factorisation launch (the rest took it with the stand-alone
 * kernel).  bench.py reads them to decide what the dominant launch contained.  Host counters, no synchronisation. */
agp_status agp_svgp_step_counters(agp_svgp* h, int64_t* n_steps_host, int64_t* n_prologue_host);
/* ... and of the hyper-parameter iteration (round 4): number of hyper-gradient evaluations on this handle, and how many of them
 * formed G_K from ONE m^3 product -- C (Sigma K^-1) with C = kappa' diag(w) kappa + K^-1 / 4 left behind by the prologue of the
 * factorisation launch -- instead of kappa' H and K^-1 Sigma K^-1 (update_hyperparameters!, autotuning.jl:86-140).  Host counters. */
agp_status agp_svgp_hyper_counters(agp_svgp* h, int64_t* n_grad_host, int64_t* n_gk_fused_host);
/* The same step in phases, for multi-GPU runs:
 *   step_local  : compute_kappa + mean_f/var_f (+ c_k for LogisticSoftMax)   latentgp.jl:209-215,171-189
 *   lsm_*       : LogisticSoftMax cross-latent fixed point, logisticsoftmax.jl:65-72:
 *                 lsm_gamma writes gamma_k for local latents and their sum into the `gsum` buffer (T[B]);
 *                 the caller all-reduces gsum across latent-parallel ranks; lsm_alpha sets alpha = 1 + gsum.
 *                 (called twice, as the reference loops twice)
 *   step_stats  : theta, grad_E_mu, grad_E_Sigma and the batch statistics
 *                 stats = [ kappa'(rho g1) (mp) | rho kappa' diag(g2) kappa, lower 64x64 tiles packed block column by block
 *                 column: tile (i, j <= i) of the nt x nt grid (nt = mp / 64) at offset mp + (j nt - j(j-1)/2 + i - j) * 4096,
 *                 row-major inside the tile ] per latent
 *                 -- the buffer a batch-parallel run all-reduces (analyticVI.jl:168,179)
 *   step_global : natural-gradient step + (mu, Sigma) refresh      analyticVI.jl:229-246, inference.jl:25-28 */
agp_status agp_svgp_step_local(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx,
                               int64_t B, double rho);
/* Hyper-parameter / inducing-point step: update_hyperparameters!(m, state, x, y)  src/hyperparameter/autotuning.jl:86-140
 * = reverse mode of ELBO(model, x, y, mu0, kernels, Zs, state) (src/functions/ELBO.jl:15-21; (mu, Sigma, local variables)
 * fixed, AugmentedKL ignored) w.r.t. the kernel parameters (variance, ScaleTransform / ARDTransform scales) and Z,
 * hand-derived instead of Zygote, on the minibatch of the LAST cavi_step; then ADAM ascent with the positive kernel
 * parameters stepped in log space (update_kernel!, autotuning_utils.jl:63-67) and Z directly (update_Z!, :70-76).
 * K is marked stale and refreshed by the next step (the reference leaves it stale inside train!).
 *   hyper_configure : optimiser=ADAM(kernel_eta) / Zoptimiser=ADAM(z_eta) of SVGP(...)  (SVGP.jl:39-42); 0 disables
 *   hypergrad       : the gradient only (parity): dscale_host[D] per input dimension (a ScaleTransform's single
 *                     parameter receives their sum), dZ device T[m][D] (nullable)
 *   get_kernel      : current variance and per-dimension scales */
agp_status agp_svgp_hyper_configure(agp_svgp* h, int32_t opt_kernel, double kernel_eta, int32_t opt_Z, double z_eta,
                                    double adam_b1, double adam_b2, double adam_eps);
/*   hyper_rule      : which Optimisers.jl rule `optimiser` / `Zoptimiser` are (the reference hands whatever rule it is given to
 *                     Optimisers.apply, src/hyperparameter/autotuning_utils.jl:47-82): ADAM (default; eta and the moments from
 *                     hyper_configure), Descent(eta): dx' = eta dx, Momentum(eta, rho): vel = rho vel + eta dx, dx' = vel.
 *                     The optimiser state exported by agp_svgp_hyper_opt_state holds the velocity in the first-moment slot. */
enum { AGP_OPT_ADAM = 0, AGP_OPT_DESCENT = 1, AGP_OPT_MOMENTUM = 2 };
agp_status agp_svgp_hyper_rule(agp_svgp* h, int32_t kernel_rule, double kernel_rho, int32_t z_rule, double z_rho);
agp_status agp_svgp_hypergrad(agp_svgp* h, int32_t latent, double* dvariance_host, double* dscale_host, void* dZ);
agp_status agp_svgp_hyper_step(agp_svgp* h);
/* the optimiser half of hyper_step with a caller-supplied gradient in the layout of agp_svgp_hypergrad (dZ: device m x D,
 * nullable).  hypergrad + hyper_apply == hyper_step; in between a multi-GPU driver can sum the gradient over latents and
 * all-reduce it over ranks (tied-Z mode: one kernel and one Z shared by all latents, BASELINE.json config 4). */
agp_status agp_svgp_hyper_apply(agp_svgp* h, int32_t latent, const double* dvariance_host, const double* dscale_host,
                                const void* dZ);
agp_status agp_svgp_get_kernel(agp_svgp* h, int32_t latent, double* variance_host, double* scales_host);
/* ADAM moments of a latent's kernel-parameter optimiser (hyperopt_state.state_k of the reference's state): host arrays of
 * 1 + D doubles (entry 0 variance, then scales; a ScaleTransform uses entry 1), set = 0 reads, set = 1 writes.  Lets a
 * streaming model (OnlineSVGP: a new handle per batch) and a resumed run continue the optimiser instead of restarting it. */
agp_status agp_svgp_hyper_opt_state(agp_svgp* h, int32_t latent, int32_t set, double* k_m_host, double* k_v_host,
                                    int32_t* k_step_host);
/* Multi-output model  MOSVGP(kernel, likelihoods, inference, Zs; Aoptimiser)  src/models/MOSVGP.jl:33-115 on a handle
 * created with lik.kind = AGP_LIK_MULTIOUTPUT: the handle's n_latent latent GPs are mixed into n_task outputs
 * f_t = sum_q A[t][q] f_q (mean_f / var_f / grad mixing: src/models/single_and_multi_output_utils.jl:24-84).
 *   liks_host : n_task likelihoods (Gaussian / Logistic / StudentT: one latent function per task)
 *   A_host    : n_task x n_latent row-major mixing weights (rows normalised, MOSVGP.jl:101-104)
 *   y passed to the step / ELBO calls is then point-major T[N][n_task] (task t's target of point i at y[i*n_task + t])
 *   adam_eta > 0 enables update_A! (ADAM ascent + projection on the unit sphere, lines 87-118); <= 0: Aoptimiser=false
 * predict_f / predict_y / proba_y outputs become T[n_task][n_t] (Bernoulli tasks: predict_y -> 1.0 / 0.0). */
agp_status agp_svgp_set_multioutput(agp_svgp* h, int32_t n_task, const agp_lik_desc* liks_host, const double* A_host,
                                    double adam_eta, double adam_b1, double adam_b2, double adam_eps);
agp_status agp_svgp_get_A(agp_svgp* h, double* A_host);
/* Latent-sharded multi-output model (C5 = 16 latents over 8 GPUs): every rank's handle owns the latents
 * [desc.latent_offset, desc.latent_offset + desc.n_latent) of q_total.  Call agp_svgp_mo_shard BEFORE set_multioutput; A_host /
 * get_A are then n_task x q_total (replicated on every rank, update_A! runs redundantly and stays bit-identical).  The mixing
 * (single_and_multi_output_utils.jl:24-84) needs every latent's (mean_f, var_f) on the minibatch; they travel through one
 * exchange buffer T[2][q_total][Bp] (mo_fbuf_ptr; own rows filled, the others zero, so an all-reduce(sum) over the ranks is
 * the all-gather).  One training step:
 *     step_local -> all-reduce(fbuf) -> mo_mix -> step_stats -> step_global
 * ELBO (fresh_local = 0) and hypergrad / hyper_step need the mixed means under the UPDATED posterior:
 *     mo_refresh_f -> all-reduce(fbuf) -> elbo / hypergrad     (elbo returns this rank's share: sum the scalars over ranks)
 * predict_f returns this rank's PARTIAL mix sum_{q owned} A[t][q]^p f_q: all-reduce it; predict_y / proba_y are then
 * finished in place by mo_predict_from_f (mode 0: out0 = mixed mean_f -> predict_y ; mode 1: out0, out1 = mixed mean_f,
 * var_f -> proba_y's two outputs; the Gauss-Hermite rule is only read for Bernoulli / NegBinomial tasks in mode 1). */
agp_status agp_svgp_mo_shard(agp_svgp* h, int32_t q_total);
agp_status agp_svgp_mo_fbuf_ptr(agp_svgp* h, void** ptr, int64_t* count);
agp_status agp_svgp_mo_mix(agp_svgp* h);
agp_status agp_svgp_mo_refresh_f(agp_svgp* h);
agp_status agp_svgp_mo_predict_from_f(agp_svgp* h, int64_t n_t, int32_t mode, void* out0, void* out1,
                                      const double* gh_nodes_host, const double* gh_weights_host, int32_t n_nodes);
/* Optional look-ahead: compute Knm / kappa of the NEXT minibatch (compute_kappa, latentgp.jl:209-215) on a second,
 * library-owned stream so it overlaps the current step's latency-bound factorisation.  The next cavi_step /
 * step_local called with the same (x, ldx, idx, B) adopts the result; any other call simply ignores it.
 * CONTRACT: the look-ahead is recognised by POINTER IDENTITY of (x, ldx, idx, B).  Between this call and the step that adopts it
 * the index buffer must stay allocated AND unchanged (the step gathers nothing again: it takes the rows the look-ahead read), and so
 * must the rows of x it names.  A host that refills one index buffer per iteration therefore either alternates between two
 * buffers or calls agp_svgp_invalidate_data (which also drops a pending look-ahead) after refilling.  Kernel / Z / state changes
 * (set_kernel, set_Z, hyper steps, refresh_K) drop it by themselves. */
agp_status agp_svgp_prefetch(agp_svgp* h, const void* x, int64_t ldx, const int64_t* idx, int64_t B);
agp_status agp_svgp_lsm_gamma(agp_svgp* h);
agp_status agp_svgp_lsm_alpha(agp_svgp* h);
agp_status agp_svgp_lsm_gsum_ptr(agp_svgp* h, void** ptr, int64_t* count);
agp_status agp_svgp_step_stats(agp_svgp* h);
agp_status agp_svgp_stats_ptr(agp_svgp* h, void** ptr, int64_t* count);
agp_status agp_svgp_step_global(agp_svgp* h);
/* Measurement hook (bench.py roofline): when enabled, every factorisation sequence of a CAVI step (the launches of
 * the dominant kernel: k_chol_dag, or the launches of k_chol_step) is bracketed by HIP events recorded on the ctx stream.  timing_read
 * synchronises, returns the number of bracketed kernel launches and their summed duration, and resets.
 * on = 1: every sequence; on = n > 1: every n-th sequence (the two event records cost a C2 step about 16 us). */
agp_status agp_svgp_timing_enable(agp_svgp* h, int32_t on);
agp_status agp_svgp_timing_read(agp_svgp* h, int64_t* n_launches_host, double* total_ms_host);
/* returns and clears the latched asynchronous failure (synchronises) */
agp_status agp_svgp_check_status(agp_svgp* h);

/* ELBO(model, state, y)  src/inference/analyticVI.jl:255-274
 *   fresh_local = 0 : on the kernel matrices and local variables left by the last cavi_step (x, y, idx, B
 *                     must be that step's) -- `objective(model, state, y)` of training.jl:76
 *   fresh_local = 1 : external ELBO(model, X, y) of src/functions/ELBO.jl:32-47 : recompute kappa on this
 *                     batch, re-initialise local variables, one local update, then the ELBO; rho explicit.
 *                     Round 6: when the SAME batch (pointer identity of x and idx, same B / ldx) is evaluated again with no training
 *                     step, kernel / Z change or agp_svgp_invalidate_data in between -- a handle used for monitoring only -- K_nm and
 *                     kappa of that batch are kept, like the full-batch kappa cache of the step. */
agp_status agp_svgp_elbo(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                         double rho, int32_t fresh_local, double* elbo_host);
/* The same evaluation WITHOUT a host round trip (convergence monitoring: `objective(model, state, y)` every few iterations of
 * train!, training.jl:71-90, evaluated while the next iterations are already enqueued).  enqueue puts the evaluation into the
 * stream and returns a ticket (up to 8 in flight); the value lands in mapped host memory behind an event.  fetch returns it:
 * wait = 1 blocks until it is there; wait = 0 only reports (*ready = 0 / 1).  A fetched ticket is closed.  Models whose ELBO has
 * host-side pieces (several latents, multi-output, streaming prior, AGP_FLAG_STALE_K) are evaluated synchronously inside enqueue.
 * agp_svgp_elbo_terms keeps referring to the last SYNCHRONOUS evaluation. */
agp_status agp_svgp_elbo_enqueue(agp_svgp* h, const void* x, int64_t ldx, const void* y, const int64_t* idx, int64_t B,
                                 double rho, int32_t fresh_local, int32_t* ticket);
agp_status agp_svgp_elbo_fetch(agp_svgp* h, int32_t ticket, int32_t wait, double* elbo_host, int32_t* ready);
/* the three terms of the last agp_svgp_elbo call, ELBO = rho * terms[0] - terms[1] - rho * terms[2]: unscaled data term
 * (expec_loglikelihood), Gaussian KL (+ extraKL), unscaled augmented KL.  A batch-parallel driver sums terms 0 and 2 over the
 * minibatch shards and counts the (replicated) Gaussian KL once. */
agp_status agp_svgp_elbo_terms(agp_svgp* h, double* terms_host);
/* batch-parallel run: this handle sees shard `rank` of `world` of every minibatch.  Changes the ELBO -- the reference's
 * once-per-evaluation terms (LogisticSoftMax `sum(log, first(beta))` logisticsoftmax.jl:138, the scalar-iteration terms of
 * the Laplace GIGEntropy in AGP_ELBO_REFERENCE mode) are then counted by rank 0 only -- and the hyper-gradient, whose replicated
 * Gaussian-KL part enters with weight 1 / world so that the all-reduced gradient counts it once (agp_svgp_hyper_step_multi; the
 * plain agp_svgp_hyper_step is refused on such a handle).  The batch-mode *_multi calls take rank and world from their
 * communicator themselves, so this call is only needed by hosts that drive the phases (step_local / step_stats / ...) by hand;
 * a handle that is re-created must be told again. */
agp_status agp_svgp_set_batch_shard(agp_svgp* h, int32_t rank, int32_t world);

/* state export / import : VarPosterior(mu, Sigma, eta1, eta2)  src/gpblocks/posterior.jl:21-27 ; any pointer
 * may be NULL.  set_state installs (eta1, eta2) and re-derives (mu, Sigma) (inference.jl:25-28). */
agp_status agp_svgp_get_state(agp_svgp* h, int32_t latent, void* mu, void* sigma, void* eta1, void* eta2);
agp_status agp_svgp_set_state(agp_svgp* h, int32_t latent, const void* eta1, const void* eta2);
/* `cap`: capacity of `out` in rows (AGP_MAT_*) or elements (AGP_VEC_*).  The B-sized outputs refer to the batch of the LAST
 * step / ELBO evaluation (agp_svgp_last_batch) and are refused (AGP_ERR_INVALID) when cap is smaller than that batch;
 * AGP_VEC_ALPHA is state that outlives a batch: min(cap, max_batch) elements are copied. */
agp_status agp_svgp_get_matrix(agp_svgp* h, int32_t latent, int32_t which, void* out, int64_t ldo, int64_t cap);
/* number of points of the last step_local / cavi_step / elbo batch (0 before the first) */
agp_status agp_svgp_last_batch(agp_svgp* h, int64_t* B_host);
/* Data contract of the AnalyticVI (full-batch) kappa cache: Knm / kappa of the last step are reused when the next step is
 * called with the same (x, ldx, idx, B) POINTERS, so X[idx] must not change in place while cached.  A host that refills a
 * buffer (streaming batches through one allocation, mutating a tensor) calls this first.  Stochastic handles never cache. */
agp_status agp_svgp_invalidate_data(agp_svgp* h);
/* init_state(model)  src/training/states.jl:1-9 -- what train! does when it is called WITHOUT a state (training.jl:41-45):
 * local variables as new (LogisticSoftMax alpha = K), RobbinsMonro counters back to 1, new hyper-optimiser (ADAM) states, data
 * caches dropped.  The posterior (eta1, eta2), kernels and Z belong to the model and are kept.  A host that resumes with the
 * state of a previous train! simply does not call it. */
agp_status agp_svgp_init_state(agp_svgp* h);

/* _predict_f (sparse)  src/training/predictions.jl:25-50 : streams over n_t test points without materialising
 * K_*m.  mu_out / var_out : T[n_latent][n_t] (var_out NULL -> cov=false).  n_t = 0 is a successful no-op in every predict_* /
 * proba_y entry point (the reference returns empty arrays). */
agp_status agp_svgp_predict_f(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* mu_out, void* var_out);
/* predict_f(model, X_test; cov=true, diag=false)  predictions.jl:45-49 : full posterior covariance
 *   cov = K** + jitt I - K*m (K^-1 - K^-1 Sigma K^-1) Km*   per latent, cov_out : T[n_latent][n_t][n_t] row-major (mu_out as
 * predict_f).  A multi-output handle returns the MIXED outputs (predictions.jl:52-92): mu_out T[n_task][n_t] = sum_q A[t][q] mu_q,
 * cov_out T[n_task][n_t][n_t] = sum_q A[t][q]^2 cov_q.  K*m IS materialised here (n_t x m), so n_t <= 8192 (AGP_ERR_INVALID
 * beyond). */
agp_status agp_svgp_predict_f_cov(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* mu_out, void* cov_out);
/* predict_y  predictions.jl:178-198 : regression (Gaussian, StudentT, Laplace, Heteroscedastic) -> T[n_t] mean ;
 * logistic / BayesianSVM -> int32[n_t] (mu_f > 0) ; Poisson / NegBinomial -> T[n_t] expected count (predictions.jl:211) ;
 * LogisticSoftMax -> int32[n_t] argmax_k mu_f,k (0-based LOCAL latent index + latent_offset) */
agp_status agp_svgp_predict_y(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, void* y_out);
/* ---- OnlineSVGP (src/models/OnlineSVGP.jl, src/training/onlinetraining.jl) ------------------------------------------------
 * A streaming model is a sequence of handles: when a batch arrives the caller snapshots the current posterior, picks the
 * new inducing points (InducingPoints.OIPS on its side, with agp_kernelmatrix for the kernel values), creates a handle
 * for the new Z and installs the previous posterior as its prior.  Only AnalyticVI() (stochastic = 0) is accepted: the
 * reference's stochastic branch is dead code (onlinetraining.jl:52 uses an undefined name).
 *
 * agp_svgp_online_snapshot: save_old_gp! (onlinetraining.jl:170-180): invDa_out (m x m, ld ldi) = -2 eta2 - inv(K),
 *   eta1_out (m) = eta1, *prevLa_host = (-logdet Sigma + logdet K - mu'eta1)/2.  Device outputs; synchronises.
 * agp_svgp_set_online_prior: previous_gp of the opt state + Z_a of the latent.  za = NULL: first batch (Z_a empty:
 *   kappa_a = I, K_ab = 0, K~_a = 0; pass invDa = I, prev_eta1 = 0, prevLa = 0 as init_opt_state does, states.jl:85-97;
 *   ma must equal m).  From then on every step uses the online natural gradient (analyticVI.jl:183-203)
 *     eta1 = K\mu0 + kappa' grad_E_mu + kappa_a' eta1_a ;  eta2 = -(kappa' diag(grad_E_Sigma) kappa + kappa_a' invD_a kappa_a/2 + inv(K)/2)
 *   and agp_svgp_elbo subtracts extraKL (KLdivergences.jl:30-54).
 * agp_svgp_adopt_local: dst takes src's local variables and expectation gradients (same likelihood / max_batch / ctx):
 *   the first iteration on a new batch updates the local variables under the OLD inducing points (compute_old_matrices,
 *   onlinetraining.jl:78-104): run agp_svgp_step_local on the old handle, then on the new one
 *   agp_svgp_step_local + agp_svgp_adopt_local(new, old) + agp_svgp_step_stats(fused) + agp_svgp_step_global. */
agp_status agp_svgp_online_snapshot(agp_svgp* h, int32_t latent, void* invDa_out, int64_t ldi, void* eta1_out,
                                    double* prevLa_host);
agp_status agp_svgp_set_online_prior(agp_svgp* h, int32_t latent, const void* za, int64_t ldza, int64_t ma, const void* invDa,
                                     int64_t ldi, const void* prev_eta1, double prevLa);
agp_status agp_svgp_adopt_local(agp_svgp* dst, agp_svgp* src);
/* the whole first iteration on a new batch (onlinetraining.jl:78-104) in one call: h_old.step_local(x, y) [+ the
 * LogisticSoftMax fixed point] ; h_new.step_local ; adopt_local(h_new, h_old) ; h_new natural gradient + global update */
agp_status agp_svgp_online_first_step(agp_svgp* h_new, agp_svgp* h_old, const void* x, int64_t ldx, const void* y,
                                      int64_t B);
/* Gauss-Hermite rule used INSIDE training by PoissonLikelihood's lambda update (expectation(logistic, mu, sigma2),
 * src/functions/utils.jl:16-19 ; same nodes as predictions.jl:4: x*sqrt2, w/sqrt(pi)).  Must be called before the first
 * step of a Poisson handle; other likelihoods ignore it. */
agp_status agp_svgp_set_quadrature(agp_svgp* h, const double* gh_nodes_host, const double* gh_weights_host,
                                   int32_t n_nodes);
/* likelihood state: the lambda of Poisson (poisson.jl:78) / Heteroscedastic (heteroscedastic.jl:95), which every local
 * update re-estimates; other likelihoods: get returns p0, set is AGP_ERR_INVALID.  get synchronises. */
agp_status agp_svgp_get_lik_param(agp_svgp* h, double* value_host);
/* LogisticSoftMax: the Gamma shape alpha of the local variables is the one piece of local state the reference carries from
 * one minibatch to the next (logisticsoftmax.jl:65-72 starts its fixed point from the previous alpha); read it with
 * agp_svgp_get_matrix(AGP_VEC_ALPHA), restore it here (device pointer, n <= max_batch) when resuming a saved model. */
agp_status agp_svgp_set_lsm_alpha(agp_svgp* h, const void* alpha, int64_t n);
agp_status agp_svgp_set_lik_param(agp_svgp* h, double value);
/* proba_y  predictions.jl:225-247 + compute_proba : Gaussian / StudentT / Laplace / Heteroscedastic -> (mean, var) ;
 * BayesianSVM / Poisson / NegBinomial -> (E[link(f)], Var) by Gauss-Hermite like logistic ; logistic -> (p, var) by
 * Gauss-Hermite with the caller's nodes/weights (predictions.jl:4 : x*sqrt2, w/sqrt(pi), 100 nodes) ;
 * LogisticSoftMax -> out0 = T[n_t][K] normalised logistic(mu_f) (multiclass.jl:96-117), out1 unused. */
agp_status agp_svgp_proba_y(agp_svgp* h, const void* xt, int64_t ldx, int64_t n_t, const double* gh_nodes_host,
                            const double* gh_weights_host, int32_t n_nodes, void* out0, void* out1);

/* ---- multi-GPU: one process per GPU, collectives behind the ABI --------------------------------
 * The path shards in two ways and both reduce to in-place SUM all-reduces of library-owned device buffers:
 *   AGP_SHARD_LATENT : this handle holds the latent slice [latent_offset, latent_offset + n_latent) of the model, every rank
 *                      sees the same minibatch.  LogisticSoftMax: sum_k gamma_k (a B-vector) twice per step
 *                      (src/likelihood/logisticsoftmax.jl:65-72); latent-sharded multi-output: the (mean_f, var_f) exchange
 *                      buffer once per step (src/models/single_and_multi_output_utils.jl:24-84); otherwise no collective.
 *   AGP_SHARD_BATCH  : every rank holds all latents and sees ITS SHARE of the minibatch (idx, B are the local share,
 *                      rho = N / B_total); the batch statistics [kappa'(rho g1) | rho kappa' diag(g2) kappa] (one triangle,
 *                      packed as 64x64 tiles: mp + nt(nt+1)/2 * 4096 elements per latent, 4.46 MB at m = 1024 f64) are
 *                      all-reduced once per step (src/inference/analyticVI.jl:168,179), then every rank applies the identical
 *                      global step.
 * agp_comm_unique_id : rank 0 draws the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks over whatever host
 *                      channel exists (MPI, a file, Julia's Distributed, torch.distributed's store).
 * agp_comm_init      : ncclCommInitRank on ctx's device; collectives are enqueued on ctx's stream (RCCL over xGMI).
 * agp_comm_init_callback : the host supplies the all-reduce instead (must sum `count` elements of `dtype` in place at the
 *                      DEVICE pointer `buf`, ordered after the work already enqueued on `hip_stream`; return 0 on success).
 *                      `hip_stream` is the ctx's stream, or -- AGP_SPLIT_OVERLAP=1 -- a stream of the communicator's own.
 * AGP_SPLIT_OVERLAP=1 (environment, default off, read at every step; batch-parallel single-latent steps that ride on the task-graph
 *                      launches): the statistics are all-reduced as AGP_SPLIT_OVERLAP_GROUPS (default 4) contiguous ranges -- groups
 *                      of block columns -- one after the other on the communicator's own stream, and the next step's factorisation
 *                      starts on the first group while the others travel (its tile workgroups wait for their column's group).
 *                      Same results bit for bit; every rank must use the same setting (the call sequence differs).
 * All ranks must issue the same sequence of *_multi calls. */
typedef struct agp_comm agp_comm;
enum { AGP_COMM_ID_BYTES = 128 };
enum { AGP_SHARD_LATENT = 0, AGP_SHARD_BATCH = 1 };
typedef int32_t (*agp_allreduce_fn)(void* user, void* buf, int64_t count, int32_t dtype, void* hip_stream);
agp_status agp_comm_unique_id(uint8_t* id_host /* [AGP_COMM_ID_BYTES] */);
agp_status agp_comm_init(agp_ctx* ctx, int32_t rank, int32_t world, const uint8_t* id_host, agp_comm** out);
agp_status agp_comm_init_callback(agp_ctx* ctx, int32_t rank, int32_t world, agp_allreduce_fn fn, void* user,
                                  agp_comm** out);
agp_status agp_comm_destroy(agp_comm* comm);
agp_status agp_comm_info(agp_comm* comm, int32_t* rank_host, int32_t* world_host, int32_t* is_rccl_host);
/* sum all-reduce of a device buffer in place on the ctx stream (what the *_multi calls use; exposed for host drivers) */
agp_status agp_comm_allreduce(agp_comm* comm, void* buf, int64_t count, int32_t dtype);
/* accounting since the last read: number of collectives, bytes reduced (per rank), and -- when timing was enabled with
 * agp_comm_timing(comm, n) -- their summed duration from HIP events on the ctx stream (synchronises).  n = 1 brackets every
 * collective, n > 1 every n-th one (the two event records cost the stream ~20 us each time); the sum is scaled to all calls. */
agp_status agp_comm_timing(agp_comm* comm, int32_t on);
agp_status agp_comm_stats(agp_comm* comm, int64_t* n_calls_host, int64_t* bytes_host, double* ms_host);
/* Diagnostic (one-GPU boxes): a stand-in for an xGMI ring all-reduce that occupies the chip the way RCCL's ring kernel does -- n_wg
 * workgroups of n_threads that must ALL be resident to finish (grid barrier, the buffer moved through their registers, grid
 * barrier), lasting at least min_us.  Sum over one rank: the buffer keeps its values.  Enqueued on `stream` (a hipStream_t; the
 * stream an agp_allreduce_fn callback is handed).  stuck_dev (nullable, device int32): incremented by every workgroup whose bounded
 * wait for the others ran out (~1 s) -- the stand-in then gives up instead of hanging the device.  One n_wg per process. */
agp_status agp_comm_standin_allreduce(void* buf, int64_t count, int32_t dtype, void* stream, int32_t n_wg, int32_t n_threads,
                                      double min_us, int32_t* stuck_dev);

/* update_parameters!(model, state, x, y) (src/training/training.jl:140-158) of a sharded model: agp_svgp_cavi_step with the
 * exchange points above carried out on `comm`.  comm == NULL or world == 1 degenerates to the single-GPU step through the
 * same phase sequence (a handle that owns only a slice of a multi-output model's latents refuses comm == NULL).
 * Poisson / Heteroscedastic likelihoods (AGP_SHARD_BATCH only; the two heteroscedastic latents are coupled point-wise and share a
 * handle): lambda is re-estimated from sums over the WHOLE minibatch (poisson.jl:78, heteroscedastic.jl:94) -- three doubles
 * [S0, S1, B_local] are all-reduced between the local update's partial sums and its finish.
 * Scheduling note (round 3, invisible through the ABI, like agp_svgp_cavi_step's): a single-latent AGP_SHARD_BATCH step over several
 * ranks in a training loop (next minibatch announced with agp_svgp_prefetch) leaves its eta step PENDING on the all-reduced
 * statistics; the next step's factorisation launch takes it as its prologue.  Every other entry point completes it first. */
agp_status agp_svgp_cavi_step_multi(agp_svgp* h, agp_comm* comm, int32_t mode, const void* x, int64_t ldx, const void* y,
                                    const int64_t* idx, int64_t B, double rho);
/* ELBO(model, state, y) of the last minibatch of a sharded run, identical on every rank (analyticVI.jl:255-297):
 * latent mode sums the ranks' shares (shared per-point terms are counted by the owner of latent 0); batch mode sums the data
 * and augmented-KL terms of the shards and counts the replicated Gaussian KL once (rank and world of the shard are taken from
 * the communicator). */
agp_status agp_svgp_elbo_multi(agp_svgp* h, agp_comm* comm, int32_t mode, double* elbo_host);
/* update_hyperparameters! (src/hyperparameter/autotuning.jl:86-140) of a sharded model.
 *   tied = 0 : every latent optimises its own kernel and Z (the reference's deep copies, latentgp.jl:63-68).  Latent-sharded: no
 *              collective, except that a sharded multi-output model first re-exchanges mean_f under the updated posterior.
 *              Batch-sharded (the handle has seen a batch-mode call or agp_svgp_set_batch_shard): per latent, the gradient
 *              (1 + D + m*D doubles; data part summed over the shards, Gaussian-KL part weighted 1 / world) is all-reduced and
 *              every rank takes the identical ADAM step -- the replicas stay bit-identical.
 *   tied = 1 : ONE kernel and ONE Z shared by all latents (an opt-in extension; BASELINE.json config 4's "all-reduce on the
 *              Z hyper-grad"): the gradients of the local latents are summed, all-reduced (1 + D + m*D doubles), and the same
 *              ADAM step is applied to every latent on every rank. */
agp_status agp_svgp_hyper_step_multi(agp_svgp* h, agp_comm* comm, int32_t tied);
/* predict_f / predict_y / proba_y of a latent-sharded multi-output model: partial mixes all-reduced (n_task x n_t each), the
 * task likelihoods finished in place.  what: 0 predict_f (mu_out, var_out nullable), 1 predict_y (mu_out), 2 proba_y
 * (mu_out, var_out; Gauss-Hermite rule as in agp_svgp_proba_y).  Outputs T[n_task][n_t]. */
agp_status agp_svgp_predict_multi(agp_svgp* h, agp_comm* comm, int32_t what, const void* xt, int64_t ldx, int64_t n_t,
                                  void* mu_out, void* var_out, const double* gh_nodes_host, const double* gh_weights_host,
                                  int32_t n_nodes);

#ifdef __cplusplus
}
#endif
#endif /* AGP_HIP_H */
