/* lgconv_ref.c -- scalar C restatement of ONE LGConv layer.  TEST INFRASTRUCTURE ONLY.
 *
 * Second, independent statement of the arithmetic the torch oracle (lightgcn_oracle.py) uses for
 * the un-vendored PyG operator called at src/lightgcn.py:96: weighted
 * in-degree by target accumulated sequentially in edge order in fp32; deg^-1/2 with +inf -> 0;
 * val = dis[src] * w * dis[dst] left to right; y[dst] += val * x[src] in edge order, the product
 * rounded before the add.  Parity with upstream PyG itself is unpinned (no PyG here, no reference
 * fixture at that boundary); tests/test_oracle_golden.py checks this file against the torch oracle.
 * Only tests/ may load the library built from it.  Build: make -C oracle  (gcc, -ffp-contract=off).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* edge_index: int64 [2, n_edges] (row 0 = source, row 1 = target); w may be NULL (= ones). */
void lgconv_ref_norm(const int64_t *edge_index, const float *w, int64_t n_nodes, int64_t n_edges,
                     float *deg, float *val) {
    const int64_t *src = edge_index, *dst = edge_index + n_edges;
    for (int64_t i = 0; i < n_nodes; ++i) deg[i] = 0.0f;
    for (int64_t e = 0; e < n_edges; ++e) deg[dst[e]] = deg[dst[e]] + (w ? w[e] : 1.0f);
    for (int64_t e = 0; e < n_edges; ++e) {
        float ds = 1.0f / sqrtf(deg[src[e]]), dd = 1.0f / sqrtf(deg[dst[e]]);
        if (isinf(ds)) ds = 0.0f;
        if (isinf(dd)) dd = 0.0f;
        float t = ds * (w ? w[e] : 1.0f);
        val[e] = t * dd;
    }
}

/* y[n_nodes, dim] = sum over edges of val[e] * x[src[e]] scattered to dst[e], in edge order. */
void lgconv_ref_hop(const int64_t *edge_index, const float *val, int64_t n_nodes, int64_t n_edges,
                    const float *x, int64_t dim, float *y) {
    const int64_t *src = edge_index, *dst = edge_index + n_edges;
    memset(y, 0, sizeof(float) * (size_t)(n_nodes * dim));
    for (int64_t e = 0; e < n_edges; ++e) {
        const float *xs = x + src[e] * dim;
        float *yd = y + dst[e] * dim;
        for (int64_t c = 0; c < dim; ++c) {
            float m = val[e] * xs[c];
            yd[c] = yd[c] + m;
        }
    }
}
