/* A C99 consumer of the C ABI (include/jutul_hip.h), compiled and run by tests/test_abi.py WITHOUT a GPU: the header is valid C, the
 * library links from C, a planning context (jh_context_create_host) hands back the reference-exact tables of the Poisson 3x1 case
 * (test/test_systems/variable_poisson.jl:28-35: cells 1-2-3 in a row), every compute entry point refuses it with a message, and a
 * context on a missing device fails loudly -- there is no CPU fallback behind the boundary. */
#include <stdio.h>
#include <string.h>

#include "jutul_hip.h"

#define CHECK(call)                                                                  \
  do {                                                                               \
    if ((call) != 0) {                                                               \
      char msg[512];                                                                 \
      jh_last_error(msg, sizeof msg);                                                \
      fprintf(stderr, "FAILED %s: %s\n", #call, msg);                                \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

int main(void) {
  char msg[512];
  if (jh_version() < 100) { fprintf(stderr, "jh_version\n"); return 1; }
  jh_context ctx = NULL;
  CHECK(jh_context_create_host(&ctx));
  /* N is 2 x nf, column-major like Julia's Matrix{Int64}: faces (1,2) and (2,3) */
  const int64_t N[4] = {1, 2, 2, 3};
  jh_tpfa disc = NULL;
  CHECK(jh_tpfa_create(ctx, 3, 2, N, 1, 0 /* JH_REORDER_NONE */, NULL, 0, 0, &disc));
  int64_t nc, nf, nhf, nnzb;
  int32_t bn;
  CHECK(jh_tpfa_sizes(disc, &nc, &nf, &nhf, &nnzb, &bn));
  if (nc != 3 || nf != 2 || nhf != 4 || nnzb != 7 || bn != 1) { fprintf(stderr, "sizes %lld %lld %lld %lld %d\n", (long long)nc, (long long)nf, (long long)nhf, (long long)nnzb, bn); return 1; }
  int64_t rowptr[4], colidx[7];
  CHECK(jh_tpfa_get_pattern(disc, rowptr, colidx));
  const int64_t want_rp[4] = {1, 3, 6, 8}, want_ci[7] = {1, 2, 1, 2, 3, 2, 3};   /* 1-based, sorted, diagonal present */
  if (memcmp(rowptr, want_rp, sizeof want_rp) || memcmp(colidx, want_ci, sizeof want_ci)) { fprintf(stderr, "pattern\n"); return 1; }
  int64_t face_pos[4], self[4], other[4], face[4], sign[4];
  CHECK(jh_tpfa_get_conn(disc, face_pos, self, other, face, sign));
  const int64_t want_fp[4] = {1, 2, 4, 5}, want_other[4] = {2, 1, 3, 2}, want_sign[4] = {1, -1, 1, -1};
  if (memcmp(face_pos, want_fp, sizeof want_fp) || memcmp(other, want_other, sizeof want_other) || memcmp(sign, want_sign, sizeof want_sign)) {
    fprintf(stderr, "half-face map\n");
    return 1;
  }
  /* everything that computes refuses a context without a device */
  jh_law law = NULL;
  if (jh_law_create(disc, 0, NULL, &law) == 0) {
    jh_vec r = NULL;
    jh_csr A = NULL;
    CHECK(jh_csr_create(disc, &A));
    CHECK(jh_vec_create(disc, &r));
    if (jh_assemble(law, 1.0, A, r) == 0) { fprintf(stderr, "jh_assemble ran on a planning context\n"); return 1; }
    jh_last_error(msg, sizeof msg);
    if (!strstr(msg, "no device")) { fprintf(stderr, "unexpected message: %s\n", msg); return 1; }
    jh_vec_destroy(r);
    jh_csr_destroy(A);
    jh_law_destroy(law);
  } else {
    jh_last_error(msg, sizeof msg);
    if (!strstr(msg, "no device")) { fprintf(stderr, "unexpected message: %s\n", msg); return 1; }
  }
  CHECK(jh_tpfa_destroy(disc));
  CHECK(jh_context_destroy(ctx));
  /* a device that is not there: an error code and a message, never a silent fallback */
  jh_context bad = NULL;
  if (jh_context_create(4096, &bad) == 0) { fprintf(stderr, "device 4096 exists?\n"); return 1; }
  if (jh_last_error(msg, sizeof msg) <= 0) { fprintf(stderr, "no error message\n"); return 1; }
  printf("C_ABI_CONSUMER_OK\n");
  return 0;
}
