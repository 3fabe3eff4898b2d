/* A C host of libmjx.so: no Python, no torch -- only include/mjx.h.
 *
 *   gcc -std=c99 -I include tests/c/c_caller.c -L mjrl_amd/csrc -lmjx -Wl,-rpath,$PWD/mjrl_amd/csrc -Wl,-rpath-link,/opt/rocm/lib -o c_caller
 *   ./c_caller in.bin out.bin
 *
 * in.bin  (written by tests/test_c_caller.py): int32 {n, m, n_hidden, h[0], h[1], N, d, cg_iters}, float32 {damping, step},
 *         float32 theta[d], obs[N*n], act[N*m], adv[N]   (adv already whitened: what process_paths hands NPG, batch_reinforce.py:185)
 * out.bin: float32 theta_new[d], float64 results[16] (include/mjx.h: mjx_npg_update)
 *
 * One NPG update (mjrl/algos/npg_cg.py:108-142) through mjx_malloc / mjx_memcpy_* / mjx_bind_* / mjx_npg_update, the way a
 * cgo / JNI / plain-C embedding would drive the library. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mjx.h"

#define CHECK(call)                                                                              \
  do {                                                                                           \
    int rc_ = (call);                                                                            \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, mjx_last_error()); return 2; } \
  } while (0)

static int read_exact(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes ? 0 : -1; }

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t hdr[8];
  float sc[2];
  if (read_exact(f, hdr, sizeof hdr) || read_exact(f, sc, sizeof sc)) { fprintf(stderr, "short header\n"); return 1; }
  const int n = hdr[0], m = hdr[1], n_hidden = hdr[2], cg_iters = hdr[7];
  const int hidden[2] = {hdr[3], hdr[4]};
  const int64_t N = hdr[5], d = hdr[6];
  float* theta = (float*)malloc((size_t)d * 4);
  float* obs = (float*)malloc((size_t)N * n * 4);
  float* act = (float*)malloc((size_t)N * m * 4);
  float* adv = (float*)malloc((size_t)N * 4);
  if (read_exact(f, theta, (size_t)d * 4) || read_exact(f, obs, (size_t)N * n * 4) || read_exact(f, act, (size_t)N * m * 4) ||
      read_exact(f, adv, (size_t)N * 4)) { fprintf(stderr, "short payload\n"); return 1; }
  fclose(f);

  if (mjx_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 3; }
  mjx_ctx* ctx = NULL;
  CHECK(mjx_create(&ctx, 0, n, m, hidden, n_hidden));
  if (mjx_num_params(ctx) != d) { fprintf(stderr, "parameter count %lld != %lld\n", (long long)mjx_num_params(ctx), (long long)d); return 1; }

  void *d_theta_new, *d_theta_old, *d_obs, *d_act, *d_adv, *d_grad, *d_x, *d_res;
  CHECK(mjx_malloc(&d_theta_new, d * 4));
  CHECK(mjx_malloc(&d_theta_old, d * 4));
  CHECK(mjx_malloc(&d_obs, N * n * 4));
  CHECK(mjx_malloc(&d_act, N * m * 4));
  CHECK(mjx_malloc(&d_adv, N * 4));
  CHECK(mjx_malloc(&d_grad, d * 4));
  CHECK(mjx_malloc(&d_x, d * 4));
  CHECK(mjx_malloc(&d_res, 64 * 8));
  CHECK(mjx_memcpy_h2d(d_theta_new, theta, d * 4, NULL));
  CHECK(mjx_memcpy_h2d(d_theta_old, theta, d * 4, NULL));
  CHECK(mjx_memcpy_h2d(d_obs, obs, N * n * 4, NULL));
  CHECK(mjx_memcpy_h2d(d_act, act, N * m * 4, NULL));
  CHECK(mjx_memcpy_h2d(d_adv, adv, N * 4, NULL));

  /* transforms NULL = identity (fc_network.py:27-37 defaults); theta_new == theta_old at entry of every update */
  CHECK(mjx_bind_policy(ctx, (const float*)d_theta_new, (const float*)d_theta_old, NULL, NULL, 1));
  CHECK(mjx_bind_batch(ctx, (const float*)d_obs, (const float*)d_act, (const float*)d_adv, N, N));
  CHECK(mjx_npg_update(ctx, cg_iters, sc[0], 1e-10, (double)sc[1], NAN, -3.0f, (float*)d_grad, (float*)d_x, (float*)d_theta_new,
                       (double*)d_res, NULL));
  CHECK(mjx_stream_sync(NULL));

  double res[16];
  CHECK(mjx_memcpy_d2h(theta, d_theta_new, d * 4, NULL));
  CHECK(mjx_memcpy_d2h(res, d_res, sizeof res, NULL));
  CHECK(mjx_stream_sync(NULL));
  f = fopen(argv[2], "wb");
  if (!f) { perror(argv[2]); return 1; }
  fwrite(theta, 4, (size_t)d, f);
  fwrite(res, 8, 16, f);
  fclose(f);
  printf("alpha %.9g  kl %.9g  surr_improvement %.9g\n", res[9], res[1] / (double)N, (res[0] - res[4]) / (double)N);

  mjx_free(d_theta_new); mjx_free(d_theta_old); mjx_free(d_obs); mjx_free(d_act); mjx_free(d_adv);
  mjx_free(d_grad); mjx_free(d_x); mjx_free(d_res);
  mjx_destroy(ctx);
  free(theta); free(obs); free(act); free(adv);
  return 0;
}
