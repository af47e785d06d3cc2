/* The boundary is a C ABI: this file is compiled as C99 (no C++), includes the public header, takes the
 * address of every entry point with its declared prototype, and -- when a GPU is present -- runs one small
 * batch through the host-pointer path.  Built and run by tests/test_abi.py. */
#include <stdio.h>
#include <string.h>
#include "decaf377_amd.h"

int main(int argc, char** argv) {
  /* prototypes as declared: a mismatch between header and library surfaces at link time */
  int (*f1)(d377_ctx*, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint8_t*) = d377_batch_scalar_mul_var;
  int (*f2)(d377_ctx*, int, const uint8_t*, const uint8_t*, size_t, uint8_t*, uint8_t*) = d377_batch_sqrt_ratio_zeta_ex;
  int (*f3)(d377_ctx*, int, void*, int, const void*, const void*, size_t, void*, void*) = d377_batch_sharded_dev;
  int (*f4)(d377_ctx*, const uint64_t*, const uint8_t*, size_t, uint8_t*, uint64_t*) = d377_msm;
  (void)f1; (void)f2; (void)f3; (void)f4;
  printf("%s\n", d377_version());
  if (argc > 1 && strcmp(argv[1], "--run") == 0) {
    d377_ctx* ctx = NULL;
    int dev = 0;
    if (d377_ctx_create(&dev, 1, &ctx) != D377_OK) { printf("ctx: %s\n", d377_last_error()); return 2; }
    uint8_t k[4 * 32], out[4 * 32], gen[32], out2[4 * 32], st[4];
    uint64_t g[16];
    memset(k, 0, sizeof k);
    k[0] = 1; k[32] = 2; k[64] = 3; k[96] = 0;
    if (d377_batch_scalar_mul_base(ctx, k, 4, out) != D377_OK) return 3;
    d377_generator(g);
    if (d377_batch_compress(ctx, g, 1, gen) != D377_OK) return 4;
    if (memcmp(gen, out, 32) != 0 || gen[0] != 8) return 5;            /* 1 * B = B = [8, 0, ...] (tests/encoding.rs:29-52) */
    {
      uint8_t pts[4 * 32];
      int i;
      for (i = 0; i < 4; ++i) memcpy(pts + 32 * i, gen, 32);
      if (d377_batch_scalar_mul_var(ctx, pts, k, 4, out2, st) != D377_OK) return 6;
      if (memcmp(out, out2, sizeof out) != 0 || st[0] || st[3]) return 7;   /* k * B both ways; 0 * B = identity = zeros */
      for (i = 0; i < 32; ++i) if (out2[96 + i]) return 8;
    }
    d377_ctx_destroy(ctx);
    printf("C_ABI_RUN_OK\n");
  }
  printf("C_ABI_OK\n");
  return 0;
}
