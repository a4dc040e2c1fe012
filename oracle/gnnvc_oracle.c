/*
 * gnnvc_oracle.c — CPU restatement of the reference GNN forward (see header).
 * TEST INFRASTRUCTURE ONLY; not linked into the product.
 *
 * Build: gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp (oracle/Makefile).
 * -ffp-contract=off: the only fused operations are the explicit fmaf() calls,
 * every other add / divide is separately rounded exactly as written.
 */
#include "gnnvc_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- parser */

typedef struct {
    const char *p, *end;
} cursor;

static int next_token(cursor *c, char *buf, size_t cap) {
    while (c->p < c->end && isspace((unsigned char)*c->p)) c->p++;
    if (c->p >= c->end) return 0;
    size_t i = 0;
    while (c->p < c->end && !isspace((unsigned char)*c->p)) {
        if (i + 1 < cap) buf[i++] = *c->p;
        c->p++;
    }
    buf[i] = 0;
    return 1;
}

static int next_size(cursor *c, uint32_t *v) {
    char t[64];
    if (!next_token(c, t, sizeof t)) return 0;
    char *e;
    unsigned long x = strtoul(t, &e, 10);
    if (*e) return 0;
    *v = (uint32_t)x;
    return 1;
}

/* matrix operator>> (src/matrix.cpp:97-104): "<h> <w>" then h*w floats. */
static float *read_matrix(cursor *c, uint32_t *h, uint32_t *w) {
    if (!next_size(c, h) || !next_size(c, w)) return NULL;
    size_t cnt = (size_t)*h * *w;
    float *d = (float *)malloc((cnt ? cnt : 1) * sizeof(float));
    char t[64];
    for (size_t i = 0; i < cnt; i++) {
        if (!next_token(c, t, sizeof t)) { free(d); return NULL; }
        d[i] = strtof(t, NULL); /* istream >> float rounds like strtof */
    }
    return d;
}

/* gnn::operator>> (src/gnn_inference.cpp:120-139). */
oracle_model *oracle_model_parse(const char *text, size_t len) {
    cursor c = {text, text + len};
    oracle_model *m = (oracle_model *)calloc(1, sizeof *m);
    char t[64];
    uint32_t n;
    m->weight_scale = 120.0f;
    if (!next_token(&c, m->name, sizeof m->name) || !next_size(&c, &n) ||
        !next_token(&c, t, sizeof t))
        goto fail;
    m->layers = (oracle_layer *)calloc(n ? n : 1, sizeof(oracle_layer));
    for (uint32_t i = 0; i < n; i++) {
        if (!next_token(&c, t, sizeof t)) goto fail;
        oracle_layer *l = &m->layers[m->n_layers];
        if (!strcmp(t, "Linear_Layer")) {
            uint32_t bh, bw;
            l->kind = ORACLE_LAYER_LINEAR;
            if (!next_token(&c, t, sizeof t)) goto fail; /* "Weights:" */
            l->W = read_matrix(&c, &l->k, &l->m);
            if (!l->W) goto fail;
            m->n_layers++;
            if (!next_token(&c, t, sizeof t)) goto fail; /* "Bias:" */
            l->bias = read_matrix(&c, &bh, &bw);
            if (!l->bias || bh != 1 || bw != l->m) goto fail;
        } else if (!strcmp(t, "Graph_Layer")) {
            l->kind = ORACLE_LAYER_GRAPH;
            m->n_layers++;
        } else if (!strcmp(t, "ReLU_Activation")) {
            l->kind = ORACLE_LAYER_RELU;
            m->n_layers++;
        } else if (!strcmp(t, "Sigmoid_Activation")) {
            l->kind = ORACLE_LAYER_SIGMOID;
            m->n_layers++;
        } /* unknown tokens are skipped, as the reference's if-chain does */
    }
    return m;
fail:
    oracle_model_free(m);
    return NULL;
}

void oracle_model_free(oracle_model *m) {
    if (!m) return;
    for (int i = 0; m->layers && i < m->n_layers; i++) {
        free(m->layers[i].W);
        free(m->layers[i].bias);
    }
    free(m->layers);
    free(m);
}

/* model::set_weight_scale (src/gnn_inference.cpp:83-90) */
void oracle_model_set_weight_scale(oracle_model *m, float ws) { m->weight_scale = ws; }

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---------------------------------------------------------------- layers */

/* graph_layer::forward (src/gnn_inference.cpp:27-42).
 * out is zero-filled (:28-29); out[u][0:f] accumulates in[v][0:f] over adj(u)
 * in stored order, one rounded fp32 add per neighbour starting from 0.0f
 * (:33-36); in[u] is copied to out[u][f:2f] (:37); then degree, W/ws, NW/ws
 * are written at columns f+1, f+2, f+3 (:38-40) — for f=16 these overwrite
 * copied columns 17..19 and columns 32..34 stay zero. */
static void graph_row(const oracle_graph *g, float ws, uint32_t f, const float *in,
                      float *out, uint32_t u) {
    const uint32_t wd = 2 * f + 3;
    float *o = out + (size_t)u * wd;
    for (uint32_t j = 0; j < wd; j++) o[j] = 0.0f;
    for (uint64_t e = g->rowptr[u]; e < g->rowptr[u + 1]; e++) {
        const float *r = in + (size_t)g->col[e] * f;
        for (uint32_t j = 0; j < f; j++) o[j] = o[j] + r[j];
    }
    const float *self = in + (size_t)u * f;
    for (uint32_t j = 0; j < f; j++) o[f + j] = self[j];
    o[f + 1] = (float)(uint32_t)(g->rowptr[u + 1] - g->rowptr[u]);
    o[f + 2] = (float)g->w[u] / ws;
    o[f + 3] = (float)g->nw[u] / ws;
}

void oracle_graph_layer(const oracle_graph *g, float ws, uint32_t f, const float *in,
                        float *out, int parallel_rows) {
    if (parallel_rows) {
#pragma omp parallel for schedule(dynamic, 1024)
        for (int64_t u = 0; u < (int64_t)g->n; u++) graph_row(g, ws, f, in, out, (uint32_t)u);
    } else {
        /* as shipped: the reference's "#pragma omp parallel for" is inert
         * (Makefile has no -fopenmp), so the aggregation is serial. */
        for (uint32_t u = 0; u < g->n; u++) graph_row(g, ws, f, in, out, u);
    }
}

/* linear_layer::forward (src/gnn_inference.cpp:20-25): dot(in, W, out) with
 * beta = 0 (src/matrix.cpp:106-122 -> cblas_sgemm row-major, no transposes),
 * then a separately rounded per-row bias add.  SGEMM restated as a
 * sequential-k fmaf chain from +0.0f per output.  Rows are independent, so
 * the row-parallel loop gives the same bits with any thread count (OpenBLAS
 * is the reference's only multi-threaded piece). */
void oracle_linear_layer(uint32_t n, uint32_t k, uint32_t m, const float *in,
                         const float *W, const float *bias, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        const float *a = in + (size_t)i * k;
        float *o = out + (size_t)i * m;
        float acc[64];
        if (m <= 64) {
            for (uint32_t j = 0; j < m; j++) acc[j] = 0.0f;
            for (uint32_t kk = 0; kk < k; kk++) {
                const float av = a[kk];
                const float *wr = W + (size_t)kk * m;
                for (uint32_t j = 0; j < m; j++) acc[j] = fmaf(av, wr[j], acc[j]);
            }
            for (uint32_t j = 0; j < m; j++) o[j] = acc[j] + bias[j];
        } else {
            for (uint32_t j = 0; j < m; j++) {
                float s = 0.0f;
                for (uint32_t kk = 0; kk < k; kk++) s = fmaf(a[kk], W[(size_t)kk * m + j], s);
                o[j] = s + bias[j];
            }
        }
    }
}

/* The reference's own call pattern for timing (bench.py's cpu_baseline): dot() hands the product to
 * cblas_sgemm (src/matrix.cpp:112-121: RowMajor, no transposes, alpha 1, beta 0, lda = k, ldb = ldc = m) —
 * here through a function pointer the caller found by dlopen'ing an OpenBLAS, since the image has no
 * header to link against — and linear_layer::forward then adds the bias row by row, serially
 * (src/gnn_inference.cpp:22-24).  Same bits as oracle_linear_layer (tests/test_openblas_seam.py). */
typedef void (*oracle_cblas_sgemm_fn)(int order, int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                                      const float *B, int ldb, float beta, float *C, int ldc);
static oracle_cblas_sgemm_fn g_cblas_sgemm = NULL;

void oracle_set_cblas_sgemm(void *fn) { g_cblas_sgemm = (oracle_cblas_sgemm_fn)fn; }
int oracle_has_cblas_sgemm(void) { return g_cblas_sgemm != NULL; }

static void linear_layer_as_shipped(uint32_t n, uint32_t k, uint32_t m, const float *in, const float *W, const float *bias,
                                    float *out) {
    if (!g_cblas_sgemm) {
        oracle_linear_layer(n, k, m, in, W, bias, out);
        return;
    }
    g_cblas_sgemm(101 /* CblasRowMajor */, 111 /* CblasNoTrans */, 111, (int)n, (int)m, (int)k, 1.0f, in, (int)k, W, (int)m, 0.0f,
                  out, (int)m);
    for (uint32_t i = 0; i < n; i++) {
        float *o = out + (size_t)i * m;
        for (uint32_t j = 0; j < m; j++) o[j] = o[j] + bias[j];
    }
}

/* dot() in full (src/matrix.cpp:106-122): C = op(A) * op(B) + beta * C, row-major; m, n, k after op.
 * The restatement the engine's gnnvc_sgemm documents (include/gnnvc.h): one sequential-k fmaf chain per output
 * from +0.0f, then — only when beta != 0 — fmaf(beta, C_old, chain).  The inference path always passes
 * beta = 0 (src/gnn_inference.cpp:21).  With beta != 0 OpenBLAS itself rounds in two different ways
 * depending on the problem size (its small-matrix kernels fuse beta * C into the final add, its blocked
 * path scales C first: tests/test_openblas_seam.py records both), so there is no single third-party
 * behaviour to restate there; the fused form is the documented contract. */
void oracle_sgemm(int ta, int tb, uint32_t m, uint32_t n, uint32_t k, const float *A, uint32_t lda, const float *B,
                  uint32_t ldb, float beta, float *C, uint32_t ldc) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)m; i++)
        for (uint32_t j = 0; j < n; j++) {
            float acc = 0.0f;
            for (uint32_t kk = 0; kk < k; kk++) {
                const float a = ta ? A[(size_t)kk * lda + i] : A[(size_t)i * lda + kk];
                const float b = tb ? B[(size_t)j * ldb + kk] : B[(size_t)kk * ldb + j];
                acc = fmaf(a, b, acc);
            }
            float *c = &C[(size_t)i * ldc + j];
            *c = (beta == 0.0f) ? acc : fmaf(beta, *c, acc);
        }
}

static void relu_serial(size_t count, const float *in, float *out) {
    for (size_t i = 0; i < count; i++) {
        float x = in[i];
        out[i] = (x < 0.0f) ? 0.0f : x;
    }
}

/* ReLU::forward (src/gnn_inference.cpp:44-47): std::max(x, 0.0f) — returns x
 * unless x < 0 (so -0.0f and NaN pass through like std::max does). */
void oracle_relu(size_t count, const float *in, float *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)count; i++) {
        float x = in[i];
        out[i] = (x < 0.0f) ? 0.0f : x;
    }
}

/* sigmoid::forward (src/gnn_inference.cpp:49-52): host libm expf. */
void oracle_sigmoid(size_t count, const float *in, float *out) {
    for (size_t i = 0; i < count; i++) out[i] = 1.0f / (1.0f + expf(-in[i]));
}

void oracle_neighbourhood_weights(uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                                  const uint32_t *w, uint32_t *nw) {
    for (uint32_t u = 0; u < n; u++) {
        uint32_t s = 0;
        for (uint64_t e = rowptr[u]; e < rowptr[u + 1]; e++) s += w[col[e]];
        nw[u] = s;
    }
}

/* model::predict (src/gnn_inference.cpp:67-81): copy the input, run the
 * layers in order ping-ponging two buffers. */
int oracle_predict(const oracle_model *m, const oracle_graph *g, uint32_t in_width,
                   const float *in, float *out, uint32_t *out_width, int stop_after,
                   int flags) {
    if (!m || m->n_layers <= 0) return -1;
    const uint32_t n = g->n;
    uint32_t maxw = in_width, wd = in_width;
    for (int i = 0; i < m->n_layers; i++) {
        const oracle_layer *l = &m->layers[i];
        if (l->kind == ORACLE_LAYER_LINEAR) {
            if (l->k != wd) return -2;
            wd = l->m;
        } else if (l->kind == ORACLE_LAYER_GRAPH) {
            wd = 2 * wd + 3;
        }
        if (wd > maxw) maxw = wd;
    }
    size_t cap = (size_t)(n ? n : 1) * maxw;
    float *a = (float *)malloc(cap * sizeof(float));
    float *b = (float *)malloc(cap * sizeof(float));
    if (!a || !b) { free(a); free(b); return -3; }
    memcpy(a, in, (size_t)n * in_width * sizeof(float));
    wd = in_width;
    int last = (stop_after < 0 || stop_after >= m->n_layers) ? m->n_layers - 1 : stop_after;
    for (int i = 0; i <= last; i++) {
        const oracle_layer *l = &m->layers[i];
        switch (l->kind) {
        case ORACLE_LAYER_LINEAR:
            if (flags & 2) linear_layer_as_shipped(n, l->k, l->m, a, l->W, l->bias, b);
            else oracle_linear_layer(n, l->k, l->m, a, l->W, l->bias, b);
            wd = l->m;
            break;
        case ORACLE_LAYER_GRAPH:
            oracle_graph_layer(g, m->weight_scale, wd, a, b, flags & 1);
            wd = 2 * wd + 3;
            break;
        case ORACLE_LAYER_RELU:
            if (flags & 2) relu_serial((size_t)n * wd, a, b);   /* the reference's std::transform is serial */
            else oracle_relu((size_t)n * wd, a, b);
            break;
        case ORACLE_LAYER_SIGMOID:
            oracle_sigmoid((size_t)n * wd, a, b);
            break;
        default:
            free(a); free(b);
            return -4;
        }
        float *t = a; a = b; b = t;
    }
    memcpy(out, a, (size_t)n * wd * sizeof(float));
    if (out_width) *out_width = wd;
    free(a);
    free(b);
    return 0;
}

/* ------------------------------------------------------- reduction predicates (f-2) */

static uint32_t deg_of(const oracle_graph *g, uint32_t u) { return (uint32_t)(g->rowptr[u + 1] - g->rowptr[u]); }

/* reduction_graph::is_dominating (include/reduction_graph.hpp:201-224): N(v) \ {u} subset of N(u),
 * with the reference's exact merge loop over the two sorted lists. */
static int is_dominating(const oracle_graph *g, uint32_t u, uint32_t v) {
    if (deg_of(g, u) < deg_of(g, v) || (uint32_t)(g->w[u] + g->nw[u]) < (uint32_t)(g->w[v] + g->nw[v])) return 0;
    const uint32_t *f1 = g->col + g->rowptr[u], *l1 = g->col + g->rowptr[u + 1];
    const uint32_t *f2 = g->col + g->rowptr[v], *l2 = g->col + g->rowptr[v + 1];
    while (f2 != l2) {
        if (*f2 == u) {
            ++f2;
            if (f2 == l2) break;
        }
        if (f1 == l1 || *f2 < *f1) return 0;
        if (!(*f1 < *f2)) ++f2;
        ++f1;
    }
    return 1;
}

/* reduction_graph::is_twin (include/reduction_graph.hpp:180-186) */
static int is_twin(const oracle_graph *g, uint32_t u, uint32_t v) {
    const uint32_t d = deg_of(g, u);
    if (d != deg_of(g, v) || g->nw[u] != g->nw[v] || u == v) return 0;
    return memcmp(g->col + g->rowptr[u], g->col + g->rowptr[v], (size_t)d * sizeof(uint32_t)) == 0;
}

/* small_mwvc_solver::solve (include/small_solve.hpp:44-74): minimum-weight vertex cover of a graph of k <= 16
 * nodes by enumeration of all 2^k subsets — a subset is a cover when every node is in it or has all its
 * neighbours in it; weights are summed as int32. */
static int64_t small_mwvc(int k, const int32_t *wt, const uint16_t *adj) {
    int32_t best = INT32_MAX;
    for (uint32_t s = 0; s < (1u << k); s++) {
        int ok = 1;
        int32_t cost = 0;
        for (int j = 0; j < k && ok; j++) {
            if (s >> j & 1u) cost += wt[j];
            else if ((s & adj[j]) != adj[j]) ok = 0;
        }
        if (ok && cost < best) best = cost;
    }
    return best;
}

static int has_edge(const oracle_graph *g, uint32_t a, uint32_t b) { /* b in adj(a)?  (ascending lists) */
    uint64_t lo = g->rowptr[a], hi = g->rowptr[a + 1];
    while (lo < hi) {
        const uint64_t mid = (lo + hi) / 2;
        if (g->col[mid] < b) lo = mid + 1;
        else hi = mid;
    }
    return lo < g->rowptr[a + 1] && g->col[lo] == b;
}

/* cover weight of the subgraph induced by `nodes` (the way the rules feed the solver, mwvc_reductions.hpp:212-218,
 * 239-244: add_node in list order, add_edge for every stored neighbour — edges to nodes outside the list are
 * dropped by add_edge, small_solve.hpp:33-42) */
static int64_t induced_mwvc(const oracle_graph *g, int k, const uint32_t *nodes) {
    int32_t wt[16];
    uint16_t adj[16];
    for (int i = 0; i < k; i++) {
        wt[i] = (int32_t)g->w[nodes[i]];
        adj[i] = 0;
    }
    for (int i = 0; i < k; i++)
        for (int j = 0; j < k; j++)
            if (i != j && nodes[i] != nodes[j] && has_edge(g, nodes[i], nodes[j])) adj[i] |= (uint16_t)(1u << j);
    return small_mwvc(k, wt, adj);
}

#define ORACLE_MAX_SMALL 8 /* max_small_solve, mwvc_reductions.hpp:20 */

/* neighborhood_meta_reduction (mwvc_reductions.hpp:234-252): D(u) <= 8 and W(u) >= NW(u) - MWVC(G[N(u)]) */
static int rule_neighborhood_meta(const oracle_graph *g, uint32_t u) {
    const uint32_t d = deg_of(g, u);
    if (d > ORACLE_MAX_SMALL) return 0;
    const uint64_t vc = (uint64_t)induced_mwvc(g, (int)d, g->col + g->rowptr[u]);
    return (uint64_t)g->w[u] >= (uint64_t)g->nw[u] - vc;
}

/* neighborhood_difference(g, a, b, res, cutoff) (mwvc_reductions.hpp:179-202): the elements of adj(a) that are
 * not in adj(b) and are not b itself — with the reference's two peculiarities: it gives up (returns) as soon as
 * cutoff + 1 elements have been written, and once adj(b) is exhausted the rest of adj(a) is copied WITHOUT the
 * "not b itself" test.  Returns the number of elements written (at most cap are stored). */
static uint32_t neighborhood_difference(const oracle_graph *g, uint32_t a, uint32_t b, uint32_t *res, uint32_t cap,
                                        uint32_t cutoff) {
    const uint32_t *f1 = g->col + g->rowptr[a], *l1 = g->col + g->rowptr[a + 1];
    const uint32_t *f2 = g->col + g->rowptr[b], *l2 = g->col + g->rowptr[b + 1];
    uint32_t t = 0;
    while (f1 != l1 && f2 != l2) {
        if (*f1 < *f2) {
            if (*f1 != b) {
                if (t < cap) res[t] = *f1;
                ++t;
                if (t > cutoff) return t;
            }
            ++f1;
        } else if (*f2 < *f1) {
            ++f2;
        } else {
            ++f1;
            ++f2;
        }
    }
    for (; f1 != l1; ++f1) {
        if (t < cap) res[t] = *f1;
        ++t;
    }
    return t;
}

/* neighbor_meta_reduction (mwvc_reductions.hpp:204-232) */
static int rule_neighbor_meta(const oracle_graph *g, uint32_t u) {
    const uint32_t du = deg_of(g, u);
    for (uint64_t e = g->rowptr[u]; e < g->rowptr[u + 1]; e++) {
        const uint32_t v = g->col[e];
        const uint32_t dv = deg_of(g, v);
        if (g->w[v] <= g->w[u] || (dv > du && dv - du > ORACLE_MAX_SMALL)) continue;
        uint32_t tmp[ORACLE_MAX_SMALL + 1];
        const uint32_t k = neighborhood_difference(g, v, u, tmp, ORACLE_MAX_SMALL + 1, ORACLE_MAX_SMALL);
        if (k > ORACLE_MAX_SMALL) continue;
        uint32_t c = 0; /* Tw arithmetic */
        for (uint32_t i = 0; i < k; i++) c += g->w[tmp[i]];
        const uint32_t vc = (uint32_t)induced_mwvc(g, (int)k, tmp);
        if ((uint32_t)(c - vc + g->w[u]) <= g->w[v]) return 1;
    }
    return 0;
}

/* reference src/GNN_VC.cpp:196 (`min(out(a, 0), 1.0f - out(a, 0))`, std::min returns its first
 * argument unless the second is smaller) and :213/:220 (`out(nodes[i], 0) > 0.5f`) */
void oracle_score_keys(size_t n, const float *scores, float *keys, uint8_t *above_half) {
    for (size_t i = 0; i < n; i++) {
        const float s = scores[i], t = 1.0f - s;
        keys[i] = t < s ? t : s;
        above_half[i] = s > 0.5f;
    }
}

void oracle_reduction_flags(const oracle_graph *g, uint32_t max_degree, uint8_t *flags) {
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t uu = 0; uu < (int64_t)g->n; uu++) {
        const uint32_t u = (uint32_t)uu;
        const uint32_t d = deg_of(g, u);
        uint8_t f = 0;
        if (d <= max_degree) {
            const uint32_t *adj = g->col + g->rowptr[u];
            if (rule_neighbor_meta(g, u)) f |= 1u << 5;
            if (rule_neighborhood_meta(g, u)) f |= 1u << 6;
            if (g->nw[u] <= g->w[u]) f |= 1u << 0;
            if (d > 0) {
                const uint32_t last = adj[d - 1]; /* "first_neighbor = *(end(g[u]) - 1)" (mwvc_reductions.hpp:144) */
                for (uint64_t e = g->rowptr[last]; e < g->rowptr[last + 1]; e++) {
                    const uint32_t v = g->col[e];
                    if (v != u && is_twin(g, u, v)) { f |= 1u << 1; break; }
                }
                int all = 1;
                uint32_t wmin = g->w[adj[0]];
                for (uint32_t i = 0; i < d; i++) {
                    const uint32_t v = adj[i];
                    if (!(f & 4u)) {
                        if ((g->w[v] >= g->w[u] && is_dominating(g, u, v)) ||
                            (g->w[v] <= g->w[u] && is_dominating(g, v, u)))
                            f |= 1u << 2;
                    }
                    if (all && !is_dominating(g, v, u)) all = 0;
                    if (g->w[v] < wmin) wmin = g->w[v];
                }
                if (all) f |= 1u << 3;
                if (g->w[u] >= (uint32_t)(g->nw[u] - wmin)) f |= 1u << 4;
            } else {
                f |= 1u << 3; /* std::all_of over no neighbours is true (reduction_graph.hpp:189-199) */
            }
        }
        flags[u] = f;
    }
}
