/* A reference-side client of the C ABI WITHOUT Python: plain C (gcc), include/agp_hip.h, device buffers from the HIP runtime.
 *
 *   abi_smoke <case.bin>
 *
 * runs  create -> set_kernel -> set_Z -> refresh_K -> 10 x (cavi_step [+ prefetch of the next minibatch]) -> elbo -> get_state ->
 * predict_f  on the inputs of a golden fixture (tests/golden/logistic_m64_svi.npz, dumped to a flat file by
 * tests/test_gpu_round3.py) and compares with the fixture's expected arrays -- what `train!` + `predict_f` of the reference do
 * (src/training/training.jl:13-111, src/training/predictions.jl:25-50).  Exit status 0 = every comparison within tolerance.
 *
 * File layout (little endian): int64 N, D, m, B, iters, nt ; double scale, variance ;
 *   double X[N*D] (point-major), y[N] (+-1), Z[m*D] ; int64 idx[iters*B] ;
 *   expected: double eta1[m], eta2[m*m], mu[m], Sigma[m*m] (after `iters` steps), elbo_last, Xt[nt*D], pred_mu[nt], pred_var[nt]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "agp_hip.h"

#define HIP_OK(e)                                                                      \
  do {                                                                                 \
    hipError_t _e = (e);                                                               \
    if (_e != hipSuccess) {                                                            \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define AGP(e)                                                                                     \
  do {                                                                                             \
    agp_status _s = (e);                                                                           \
    if (_s != AGP_OK) {                                                                            \
      fprintf(stderr, "agp status %d (%s) at %s:%d\n", (int)_s, agp_last_error(ctx), __FILE__, __LINE__); \
      return 3;                                                                                    \
    }                                                                                              \
  } while (0)

static double rel_err(const double* a, const double* b, int64_t n) {
  double num = 0.0, den = 1e-300;
  for (int64_t i = 0; i < n; ++i) {
    const double d = fabs(a[i] - b[i]);
    if (d > num) num = d;
    if (fabs(b[i]) > den) den = fabs(b[i]);
  }
  return num / den;
}

static void* rd(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "short read\n");
    exit(4);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc < 2) return 64;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 65;
  int64_t hd[6];
  double ks[2];
  if (fread(hd, 8, 6, f) != 6 || fread(ks, 8, 2, f) != 2) return 66;
  const int64_t N = hd[0], D = hd[1], m = hd[2], B = hd[3], iters = hd[4], nt = hd[5];
  double* X = rd(f, 8 * N * D);
  double* y = rd(f, 8 * N);
  double* Z = rd(f, 8 * m * D);
  int64_t* idx = rd(f, 8 * iters * B);
  double* e1x = rd(f, 8 * m);
  double* e2x = rd(f, 8 * m * m);
  double* mux = rd(f, 8 * m);
  double* Sgx = rd(f, 8 * m * m);
  double* elx = rd(f, 8);
  double* Xt = rd(f, 8 * nt * D);
  double* pmx = rd(f, 8 * nt);
  double* pvx = rd(f, 8 * nt);
  fclose(f);

  agp_ctx* ctx = NULL;
  hipStream_t stream;
  HIP_OK(hipSetDevice(0));
  HIP_OK(hipStreamCreate(&stream));  /* the caller's stream: the library orders all its work on it */
  if (agp_ctx_create(0, (void*)stream, &ctx) != AGP_OK) return 67;
  printf("agp_version %d\n", (int)agp_version());

  double *dX, *dy, *dZ, *dXt, *dmu, *dSg, *de1, *de2, *dpm, *dpv;
  int64_t* didx;
  HIP_OK(hipMalloc((void**)&dX, 8 * N * D));
  HIP_OK(hipMalloc((void**)&dy, 8 * N));
  HIP_OK(hipMalloc((void**)&dZ, 8 * m * D));
  HIP_OK(hipMalloc((void**)&didx, 8 * iters * B));
  HIP_OK(hipMalloc((void**)&dXt, 8 * nt * D));
  HIP_OK(hipMalloc((void**)&dmu, 8 * m));
  HIP_OK(hipMalloc((void**)&dSg, 8 * m * m));
  HIP_OK(hipMalloc((void**)&de1, 8 * m));
  HIP_OK(hipMalloc((void**)&de2, 8 * m * m));
  HIP_OK(hipMalloc((void**)&dpm, 8 * nt));
  HIP_OK(hipMalloc((void**)&dpv, 8 * nt));
  HIP_OK(hipMemcpy(dX, X, 8 * N * D, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dy, y, 8 * N, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dZ, Z, 8 * m * D, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(didx, idx, 8 * iters * B, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(dXt, Xt, 8 * nt * D, hipMemcpyHostToDevice));

  /* SVGP(variance * (SqExponentialKernel() o ScaleTransform(scale)), LogisticLikelihood(), AnalyticSVI(B), Z; optimiser=false) */
  agp_svgp_desc d;
  memset(&d, 0, sizeof d);
  d.dtype = AGP_F64;
  d.n_latent = 1;
  d.stochastic = 1;
  d.m = m;
  d.D = D;
  d.max_batch = B;
  d.lik.kind = AGP_LIK_LOGISTIC;
  d.lik.n_class = 1;
  d.rm_kappa = 0.51;
  d.rm_tau = 1.0;
  d.elbo_mode = AGP_ELBO_CORRECTED;
  agp_svgp* h = NULL;
  AGP(agp_svgp_create(ctx, &d, &h));
  agp_kernel_desc k;
  memset(&k, 0, sizeof k);
  k.kind = AGP_K_SQEXP;
  k.variance = ks[1];
  k.scale = ks[0];
  k.has_variance = 1;
  k.has_transform = 1;
  AGP(agp_svgp_set_kernel(h, 0, &k));
  AGP(agp_svgp_set_Z(h, 0, dZ, D));
  AGP(agp_svgp_refresh_K(h));
  const double rho = (double)N / (double)B;
  for (int64_t it = 0; it < iters; ++it) {
    AGP(agp_svgp_cavi_step(h, dX, D, dy, didx + it * B, B, rho));
    if (it + 1 < iters) AGP(agp_svgp_prefetch(h, dX, D, didx + (it + 1) * B, B)); /* the look-ahead a train! loop issues */
  }
  AGP(agp_svgp_check_status(h));
  double elbo = 0.0;
  AGP(agp_svgp_elbo(h, dX, D, dy, didx + (iters - 1) * B, B, rho, 0, &elbo));
  AGP(agp_svgp_get_state(h, 0, dmu, dSg, de1, de2));
  AGP(agp_svgp_refresh_K(h)); /* compute_Ks at the end of train! (training.jl:107) */
  AGP(agp_svgp_predict_f(h, dXt, D, nt, dpm, dpv));
  AGP(agp_ctx_sync(ctx));

  double* e1 = malloc(8 * m);
  double* e2 = malloc(8 * m * m);
  double* mu = malloc(8 * m);
  double* Sg = malloc(8 * m * m);
  double* pm = malloc(8 * nt);
  double* pv = malloc(8 * nt);
  HIP_OK(hipMemcpy(e1, de1, 8 * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(e2, de2, 8 * m * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(mu, dmu, 8 * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(Sg, dSg, 8 * m * m, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(pm, dpm, 8 * nt, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(pv, dpv, 8 * nt, hipMemcpyDeviceToHost));

  const double r_e1 = rel_err(e1, e1x, m), r_e2 = rel_err(e2, e2x, m * m), r_mu = rel_err(mu, mux, m),
               r_Sg = rel_err(Sg, Sgx, m * m), r_pm = rel_err(pm, pmx, nt), r_pv = rel_err(pv, pvx, nt),
               r_el = fabs(elbo - elx[0]) / fabs(elx[0]);
  printf("rel err: eta1 %.2e eta2 %.2e mu %.2e Sigma %.2e elbo %.2e predict mean %.2e var %.2e\n", r_e1, r_e2, r_mu, r_Sg, r_el,
         r_pm, r_pv);
  const int ok = r_e1 < 1e-9 && r_e2 < 1e-9 && r_mu < 1e-8 && r_Sg < 1e-8 && r_el < 1e-8 && r_pm < 1e-8 && r_pv < 1e-7;
  AGP(agp_svgp_destroy(h));
  AGP(agp_ctx_destroy(ctx));
  HIP_OK(hipStreamDestroy(stream));
  printf(ok ? "ABI_SMOKE_OK\n" : "ABI_SMOKE_MISMATCH\n");
  return ok ? 0 : 1;
}
