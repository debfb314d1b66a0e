#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "fhe_ring.h"
int main(void) {
    uint64_t pr[16];
    int n = fhe_two_adic_primes(60, 16, 16, pr);
    printf("primes %d first %llu prime? %d %d\n", n, (unsigned long long)pr[0], fhe_is_prime(pr[0]), fhe_is_prime(pr[0] + 2));
    const uint64_t qs[] = {1073707009ull, 18014398509404161ull, 1152921504606748673ull, 268369921ull, 12289ull, 3ull, 2ull, 15ull, 0ull};
    for (unsigned i = 0; i < sizeof qs / sizeof qs[0]; ++i) {
        fhe_ctx *c = NULL;
        int rc = fhe_ctx_create(qs[i], -1, &c);
        printf("ctx q=%llu rc=%d\n", (unsigned long long)qs[i], rc);
        if (rc == 0) {
            uint64_t q, g, w; int s;
            fhe_ctx_info(c, &q, &s, &g, &w);
            uint64_t *tw = malloc(1 << 20);
            int r2 = fhe_ctx_twiddles(c, 0, tw, 1 << 17);
            int r3 = fhe_ctx_twiddles(c, 1, tw, 16);
            uint64_t a[8] = {0};
            int r4 = fhe_ntt_fwd(c, a, 8, 1, FHE_MEM_HOST, NULL); /* no device: must be a status, not a crash */
            printf("  s=%d g=%llu tw rc=%d %d ntt rc=%d\n", s, (unsigned long long)g, r2, r3, r4);
            free(tw);
            fhe_ctx_destroy(c);
        }
    }
    fhe_rns_ctx *r = NULL;
    int rc = fhe_rns_ctx_create(pr, 8, pr + 8, 8, -1, &r);
    printf("rns rc=%d\n", rc);
    if (rc == 0) {
        uint64_t buf[64] = {0}, out[64];
        printf("rns extend rc=%d\n", fhe_rns_extend_bases(r, buf, out, 4, 1, FHE_MEM_HOST, NULL));
        fhe_rns_ctx_destroy(r);
    }
    rc = fhe_rns_ctx_create(pr, 2, pr, 2, -1, &r);   /* duplicate moduli */
    printf("rns dup rc=%d\n", rc);
    uint64_t v[4] = {1, 2, 3, 4}, o[16];
    printf("decompose no-device rc=%d\n", fhe_decompose(12289, 3, 4, v, 4, 1, o, FHE_MEM_HOST, NULL));
    printf("lincomb bad rc=%d\n", fhe_lwe_lincomb(12289, 0, NULL, NULL, 0, o, 4, FHE_MEM_HOST, NULL));
    fhe_torus_ctx *t = NULL;
    printf("torus ctx rc=%d\n", fhe_torus_ctx_create(-1, &t));
    if (t) fhe_torus_ctx_destroy(t);
    {   /* rank-k entries: argument checks that need no device */
        fhe_tggswk_key *kk = NULL;
        uint64_t w[8] = {0};
        printf("tggswk prepare rc=%d\n", fhe_tggswk_prepare(NULL, 2, 8, 8, w, 4, 1, FHE_MEM_HOST, &kk));
        printf("tggswk ext rc=%d\n", fhe_tggswk_external_product(NULL, NULL, 0, w, 1, FHE_MEM_HOST, NULL));
        printf("tglwek rotate rc=%d\n", fhe_tglwek_rotate(w, 0, 4, 1, w, 1, FHE_MEM_HOST, NULL));
        printf("tglwek extract rc=%d\n", fhe_tglwek_sample_extract(w, 1, 4, 9, w, w, 1, FHE_MEM_HOST, NULL));
        fhe_tggswk_key_destroy(kk);
    }
    return 0;
}
