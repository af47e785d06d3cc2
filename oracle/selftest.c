/* Sanitizer self-test of the oracle (test infrastructure): built with
 * -fsanitize=address,undefined by `make -C oracle selftest` and run by tests/test_oracle.py.
 * Exercises every batch entry point on seeded inputs and checks a few invariants. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "d377_oracle.c"

static uint64_t sm = 666;
static uint64_t splitmix(void) { uint64_t z = (sm += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }
static void fill(uint8_t *p, size_t n) { for (size_t i = 0; i < n; ++i) p[i] = (uint8_t)splitmix(); }

int main(void) {
    enum { N = 96 };
    uint8_t r0[N * 32], k[N * 32], enc[N * 32], out[N * 32], out2[N * 32], st[N], wide[N * 64];
    uint64_t xyzt[N * 16], xy[N * 8];
    fill(r0, sizeof r0); fill(k, sizeof k); fill(wide, sizeof wide);
    d377o_init();
    d377o_encode_to_curve(r0, N, enc);
    d377o_roundtrip(enc, N, out, st);
    for (int i = 0; i < N; ++i) if (st[i] || memcmp(out + 32 * i, enc + 32 * i, 32)) { printf("roundtrip mismatch %d\n", i); return 1; }
    d377o_decompress(enc, N, xyzt, st);
    d377o_compress(xyzt, N, out);
    if (memcmp(out, enc, sizeof enc)) { printf("compress mismatch\n"); return 1; }
    d377o_scalar_mul_var(enc, k, N, out, st);
    d377o_run_threads(2, enc, k, N, out2, st, 3);
    if (memcmp(out, out2, sizeof out)) { printf("threaded mismatch\n"); return 1; }
    d377o_scalar_mul_base(k, 8, out);
    d377o_sqrt_ratio_zeta(r0, k, N, out, st);
    d377o_hash_to_curve(r0, k, N, out);
    d377o_fq_from_wide_bytes(wide, 64, N, out);
    d377o_fq_from_wide_bytes(wide, 48, N, out);
    d377o_encode_to_curve_wide(wide, 64, 8, out);
    d377o_to_affine(xyzt, N, xy);
    d377o_sqrt_ratio_zeta_min_curve(r0, k, 8, out, st);
    { uint64_t neg[N * 16], fa[N * 4]; uint8_t idn[N];
      d377o_neg_xyzt(xyzt, N, neg); d377o_is_identity(neg, N, idn);
      for (int op = 0; op < 6; ++op) d377o_fq_op(op, xyzt, xyzt + 4 * N, N, fa, st); }
    uint8_t e1[32]; uint64_t x1[16];
    d377o_msm(xyzt, k, 16, e1, x1);
    fill(enc, sizeof enc);                       /* raw strings: mostly invalid encodings */
    d377o_roundtrip(enc, N, out, st);
    d377o_scalar_mul_var(enc, k, N, out, st);
    printf("ORACLE_SELFTEST_OK\n");
    return 0;
}
