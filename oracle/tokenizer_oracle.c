/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported, linked or executed by the product path
 * (point-sam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 *
 * Plain-C restatement of the point tokenizer's integer/index arithmetic on the Point-SAM hot path:
 *
 *   fps_f32        farthest point sampling.  Reference call site: pc_sam/model/common.py:91
 *                  (`sample_farthest_points(xyz.float(), self.num_groups)`), implemented by the
 *                  third-party torkit3d CUDA extension (git submodule third_party/torkit3d, EMPTY in the
 *                  reference checkout and with no recorded commit => PARITY UNPINNED for FPS).
 *                  Published algorithm (PointNet++ lineage, which torkit3d follows): start at index 0,
 *                  keep a running min squared distance to the selected set, pick the argmax each step.
 *                  Our pinned-down spec: fp32, d = dx*dx + dy*dy + dz*dz evaluated left-to-right with NO
 *                  fused multiply-add (built with -ffp-contract=off), ties -> lowest index.
 *   knn_f32        K nearest points of each center.  Reference: common.py:27-56 (torch.cdist + topk,
 *                  sorted=False).  Spec here ("exact" mode): same d formula as above (squared, no sqrt --
 *                  monotone, so the same set), ascending by (d, index).
 *   three_nn_f32   3 nearest centers of each point + inverse-squared-distance weights.  Reference:
 *                  common.py:238-255 (cdist -> topk(3) -> 1/clamp(dist^2, 1e-8) -> normalise).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float dist2(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float s = dx * dx;
    s = s + dy * dy;
    s = s + dz * dz;
    return s;
}

/* xyz [N,3] -> idx [G] (int64).  mind is caller scratch [N]. */
int fps_f32(const float *xyz, int64_t N, int64_t G, int64_t *idx, float *mind) {
    if (N <= 0 || G <= 0 || G > N) return -1;
    for (int64_t n = 0; n < N; ++n) mind[n] = INFINITY;
    int64_t last = 0;
    idx[0] = 0;
    for (int64_t j = 1; j < G; ++j) {
        const float *c = xyz + 3 * last;
        float best = -1.0f;
        int64_t besti = 0;
        for (int64_t n = 0; n < N; ++n) {
            float d = dist2(xyz + 3 * n, c);
            float m = mind[n];
            if (d < m) { m = d; mind[n] = d; }
            if (m > best) { best = m; besti = n; } /* strict > : lowest index wins ties */
        }
        idx[j] = besti;
        last = besti;
    }
    return 0;
}

typedef struct { float d; int64_t i; } cand_t;

static int cand_less(const cand_t *a, const cand_t *b) {
    return (a->d < b->d) || (a->d == b->d && a->i < b->i);
}

/* centers [G,3], xyz [N,3] -> idx [G,K] ascending by (d, index), optionally d2 [G,K]. */
int knn_f32(const float *centers, const float *xyz, int64_t G, int64_t N, int64_t K, int64_t *idx, float *d2out) {
    if (K <= 0 || K > N) return -1;
    cand_t *heap = (cand_t *)malloc(sizeof(cand_t) * (size_t)K); /* sorted insertion buffer */
    if (!heap) return -2;
    for (int64_t g = 0; g < G; ++g) {
        const float *c = centers + 3 * g;
        int64_t cnt = 0;
        for (int64_t n = 0; n < N; ++n) {
            cand_t x = { dist2(c, xyz + 3 * n), n };
            if (cnt == K && !cand_less(&x, &heap[K - 1])) continue;
            int64_t p = (cnt < K) ? cnt++ : K - 1;
            while (p > 0 && cand_less(&x, &heap[p - 1])) { heap[p] = heap[p - 1]; --p; }
            heap[p] = x;
        }
        for (int64_t k = 0; k < K; ++k) {
            idx[g * K + k] = heap[k].i;
            if (d2out) d2out[g * K + k] = heap[k].d;
        }
    }
    free(heap);
    return 0;
}

/* xyz [N,3], centers [G,3] -> idx [N,3] (int64), w [N,3].
 * dist = sqrtf(d2); w_k = (1/max(dist*dist, eps)) / sum  (common.py:252-254). */
int three_nn_f32(const float *xyz, const float *centers, int64_t N, int64_t G, float eps, int64_t *idx, float *w) {
    if (G < 3) return -1;
    for (int64_t n = 0; n < N; ++n) {
        cand_t b[3] = { { INFINITY, -1 }, { INFINITY, -1 }, { INFINITY, -1 } };
        for (int64_t g = 0; g < G; ++g) {
            cand_t x = { dist2(xyz + 3 * n, centers + 3 * g), g };
            if (!cand_less(&x, &b[2])) continue;
            int p = 2;
            while (p > 0 && cand_less(&x, &b[p - 1])) { b[p] = b[p - 1]; --p; }
            b[p] = x;
        }
        float inv[3], s = 0.0f;
        for (int k = 0; k < 3; ++k) {
            float dist = sqrtf(b[k].d);
            float sq = dist * dist;
            inv[k] = 1.0f / (sq < eps ? eps : sq);
        }
        s = inv[0] + inv[1];
        s = s + inv[2];
        for (int k = 0; k < 3; ++k) {
            idx[n * 3 + k] = b[k].i;
            w[n * 3 + k] = inv[k] / s;
        }
    }
    return 0;
}

/* Farthest-from-border point of a region: among points with region[n]==1, the one whose nearest point with
 * region==0 is farthest (squared distance, fp32, same dist2 as above); first maximum = lowest index on ties.
 * Reference: sample_furthest_points_from_border (pc_sam/model/common.py:443-474), which calls the third-party
 * torkit3d chamfer_distance (absent) for the nearest-background distance and torch.argmax/max on it.
 * Returns 0 and (*idx, *dist) or 1 with idx=-1, dist=-1 when the region or its complement is empty (common.py:458-460). */
int border_farthest_f32(const float *xyz, const unsigned char *region, int64_t N, int64_t *idx, float *dist) {
    int64_t nfg = 0, nbg = 0;
    for (int64_t n = 0; n < N; ++n) { if (region[n]) ++nfg; else ++nbg; }
    *idx = -1; *dist = -1.0f;
    if (nfg == 0 || nbg == 0) return 1;
    float best = -1.0f;
    for (int64_t n = 0; n < N; ++n) {
        if (!region[n]) continue;
        float m = INFINITY;
        for (int64_t k = 0; k < N; ++k) {
            if (region[k]) continue;
            float d = dist2(xyz + 3 * n, xyz + 3 * k);
            if (d < m) m = d;
        }
        if (m > best) { best = m; *idx = n; }
    }
    *dist = best;
    return 0;
}
