/* CPU oracle for the EM-Adapt E-step, plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * Restates the algorithm of the reference's estep.py:33-84 (estep_py) + the argmax of
 * deeplab.py:123, with the class visiting order as an explicit input.  It exists so that
 * parity tests at the large configurations (B=16, 65x65x41) finish in well under a second
 * and so that bench.py has a native scalar E-step to time beside the GPU ("port", 1 core).
 * It mirrors oracle/estep_oracle.py line for line; tests/test_oracle_estep.py pins both
 * against golden vectors produced by the reference's own estep.py.
 *
 * Never linked or loaded by the product path (em-adapt-tensorflow_amd/).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, fp32 stays fp32)
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* numpy's float32 pairwise summation (estep.py:56,80 use np.mean on float32). */
static float np_pairwise_sum_f32(const float *a, long n) {
    if (n < 8) {
        float res = 0.0f;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        long i;
        for (int j = 0; j < 8; j++) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        long n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
    }
}

/* k-th smallest (0-indexed) by quickselect on a scratch copy; exact, value-only. */
static float kth_smallest(float *v, long n, long k) {
    long lo = 0, hi = n - 1;
    while (lo < hi) {
        float p = v[lo + (hi - lo) / 2];
        long i = lo, j = hi;
        while (i <= j) {
            while (v[i] < p) i++;
            while (v[j] > p) j--;
            if (i <= j) {
                float t = v[i]; v[i] = v[j]; v[j] = t;
                i++; j--;
            }
        }
        if (k <= j) hi = j;
        else if (k >= i) lo = i;
        else break;
    }
    return v[k];
}

/* label set of estep.py:35-44: cast to uint8, ignore values >= C.  label_f32 is [B,HW]. */
void estep_oracle_label_sets(const float *label_f32, int B, int HW, int C, uint8_t *multihot) {
    memset(multihot, 0, (size_t)B * C);
    for (int b = 0; b < B; b++)
        for (int i = 0; i < HW; i++) {
            unsigned v = (uint8_t)(int)label_f32[(size_t)b * HW + i];
            if ((int)v < C) multihot[(size_t)b * C + v] = 1;
        }
}

/* logits [B,HW,ldc] (first C of each row used) -> map_out [B,HW,C], labels_out [B,HW].
 * Returns 0, or -1 on allocation failure / bad arguments. */
int estep_oracle_c(const float *logits, int B, int HW, int C, int ldc,
                   const uint8_t *multihot, const int32_t *perms,
                   int suppress, int num_iter, float margin, int k_bg, int k_fg,
                   float *map_out, int64_t *labels_out) {
    if (B < 0 || HW <= 0 || C <= 0 || ldc < C) return -1;
    float *mx = (float *)malloc(sizeof(float) * HW);
    float *d = (float *)malloc(sizeof(float) * HW);
    if (!mx || !d) { free(mx); free(d); return -1; }

    for (int b = 0; b < B; b++)
        for (int i = 0; i < HW; i++)
            memcpy(map_out + ((size_t)b * HW + i) * C, logits + ((size_t)b * HW + i) * ldc,
                   sizeof(float) * C);

    float gmax = -FLT_MAX; /* np.amax over the whole batch, estep.py:50 */
    for (size_t i = 0; i < (size_t)B * HW * C; i++)
        if (map_out[i] > gmax) gmax = map_out[i];

    for (int b = 0; b < B; b++) {
        float *f = map_out + (size_t)b * HW * C;
        const uint8_t *L = multihot + (size_t)b * C;
        if (suppress) { /* estep.py:46-55 */
            for (int i = 0; i < HW; i++) {
                float *p = f + (size_t)i * C;
                float m = FLT_MAX;
                for (int c = 0; c < C; c++) {
                    float t = L[c] ? p[c] : p[c] + gmax;
                    if (t < m) m = t;
                }
                float repl = m - margin;
                for (int c = 0; c < C; c++)
                    if (!L[c] && p[c] > m) p[c] = repl;
            }
        }
        for (int i = 0; i < HW; i++) { /* estep.py:56 */
            const float *p = f + (size_t)i * C;
            float m = p[0];
            for (int c = 1; c < C; c++) if (p[c] > m) m = p[c];
            mx[i] = m;
        }
        float mean0 = np_pairwise_sum_f32(mx, HW) / (float)HW;

        for (int it = 0; it < num_iter; it++) { /* estep.py:62-79 */
            for (int q = 0; q < C; q++) {
                int j = (q == 0) ? 0 : perms[(size_t)it * (C - 1) + (q - 1)];
                if (!L[j]) continue;
                for (int i = 0; i < HW; i++) d[i] = mx[i] - f[(size_t)i * C + j];
                float t = kth_smallest(d, HW, j == 0 ? k_bg : k_fg);
                for (int i = 0; i < HW; i++) {
                    float v = f[(size_t)i * C + j] + t;
                    f[(size_t)i * C + j] = v;
                    if (v > mx[i]) mx[i] = v;
                }
            }
        }
        float mean1 = np_pairwise_sum_f32(mx, HW) / (float)HW;
        float shift = mean0 - mean1; /* estep.py:80-82 */
        for (int i = 0; i < HW; i++) {
            float *p = f + (size_t)i * C;
            int best = 0;
            float bv = 0.f;
            for (int c = 0; c < C; c++) {
                p[c] = p[c] + shift;
                if (c == 0 || p[c] > bv) { bv = p[c]; best = c; } /* first max: deeplab.py:123 */
            }
            labels_out[(size_t)b * HW + i] = best;
        }
    }
    free(mx);
    free(d);
    return 0;
}
