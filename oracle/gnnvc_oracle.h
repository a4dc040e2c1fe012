/*
 * gnnvc_oracle.h — CPU restatement of the reference GNN forward.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.  The
 * shipped engine (gnn-mwvc_amd/csrc) never links or calls it and has no CPU
 * fallback.
 *
 * What it restates (all citations relative to /root/reference):
 *   gnn::model::predict        src/gnn_inference.cpp:67-81
 *   gnn::graph_layer::forward  src/gnn_inference.cpp:27-42  (incl. the F+1..F+3 column layout)
 *   gnn::linear_layer::forward src/gnn_inference.cpp:20-25  -> dot() src/matrix.cpp:106-122
 *   gnn::ReLU::forward         src/gnn_inference.cpp:44-47
 *   gnn::sigmoid::forward      src/gnn_inference.cpp:49-52
 *   model text parser          src/gnn_inference.cpp:120-139, src/matrix.cpp:97-104
 *
 * Third-party arithmetic: dot() calls cblas_sgemm of OpenBLAS (un-vendored,
 * version unpinned: README.md:24-27, Makefile:32).  It is restated here as
 * the published row-major SGEMM definition C[i][j] = sum_k A[i][k]*B[k][j]
 * evaluated as one sequential-k fmaf chain per output starting from +0.0f;
 * tests/test_openblas_seam.py checks that restatement bit-for-bit against the
 * genuine OpenBLAS 0.3.28 that SciPy bundles in this image at all nine model
 * shapes.  Parity pin: the *.scores.f32 files under tests/golden are outputs of the unmodified
 * reference (linked to that OpenBLAS) recorded during the survey; see
 * tests/golden/README.md.
 */
#ifndef GNNVC_ORACLE_H
#define GNNVC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    ORACLE_LAYER_LINEAR = 0,
    ORACLE_LAYER_GRAPH = 1,
    ORACLE_LAYER_RELU = 2,
    ORACLE_LAYER_SIGMOID = 3
};

typedef struct oracle_layer {
    int kind;
    /* linear only */
    uint32_t k, m;  /* W is k x m row-major, bias has m entries */
    float *W;
    float *bias;
} oracle_layer;

typedef struct oracle_model {
    char name[64];
    int n_layers;
    oracle_layer *layers;
    float weight_scale; /* graph_layer::WEIGHT_SCALE, default 120 (include/gnn_inference.hpp:25) */
} oracle_model;

/* A packed view of what the forward reads from reduction_graph
 * (include/reduction_graph.hpp:141-158,693-704): vertices 0..N-1, adjacency of u
 * = col[rowptr[u] .. rowptr[u+1]) in the stored (CSR) order, W(u), NW(u). */
typedef struct oracle_graph {
    uint32_t n;
    const uint64_t *rowptr; /* n+1 */
    const uint32_t *col;    /* rowptr[n] */
    const uint32_t *w;      /* n */
    const uint32_t *nw;     /* n */
} oracle_graph;

/* Parse the reference's model text.  Returns NULL on malformed input. */
oracle_model *oracle_model_parse(const char *text, size_t len);
void oracle_model_free(oracle_model *m);
void oracle_model_set_weight_scale(oracle_model *m, float ws);

/* Per-layer functions, out-of-place like the reference's forward()s. */
void oracle_graph_layer(const oracle_graph *g, float ws, uint32_t f,
                        const float *in /* n x f */, float *out /* n x (2f+3) */,
                        int parallel_rows);
void oracle_linear_layer(uint32_t n, uint32_t k, uint32_t m, const float *in,
                         const float *W, const float *bias, float *out);
void oracle_relu(size_t count, const float *in, float *out);
void oracle_sigmoid(size_t count, const float *in, float *out);

/* NW(u) = sum of W over adj(u) in uint32 wrap-around arithmetic
 * (include/reduction_graph.hpp:104-128). */
void oracle_neighbourhood_weights(uint32_t n, const uint64_t *rowptr,
                                  const uint32_t *col, const uint32_t *w,
                                  uint32_t *nw);

/* Whole forward: in is n x in_width (1 for the shipped model).
 * stop_after < 0 runs every layer; otherwise stops after layer index
 * stop_after (0-based) and returns that layer's output.
 * *out_width receives the width of the returned matrix; out must hold
 * n * 35 floats in the worst case (or n * max layer width).
 * flags bit0: parallelise the aggregation over rows (the variant the
 * reference's inert pragma intended); 0 = serial as shipped.
 * flags bit1: the reference's threading for timing — the products through the cblas_sgemm given to
 * oracle_set_cblas_sgemm (its own thread pool), bias adds and ReLU serial; same bits.
 * Returns 0 on success. */
int oracle_predict(const oracle_model *m, const oracle_graph *g,
                   uint32_t in_width, const float *in, float *out,
                   uint32_t *out_width, int stop_after, int flags);

/* bench.py's cpu_baseline: a cblas_sgemm found at run time (dlopen of an OpenBLAS), used by
 * oracle_predict(flags bit1) the way src/matrix.cpp:112-121 calls it.  NULL = the internal fmaf loops. */
void oracle_set_cblas_sgemm(void *fn);
int oracle_has_cblas_sgemm(void);

/* dot() with transposes and beta (src/matrix.cpp:106-122) as include/gnnvc.h documents gnnvc_sgemm. */
void oracle_sgemm(int ta, int tb, uint32_t m, uint32_t n, uint32_t k, const float *A, uint32_t lda, const float *B,
                  uint32_t ldb, float beta, float *C, uint32_t ldc);

/* Reduction-rule predicates on the unmutated graph, one byte per vertex:
 * bit r = "rule r of reduce_graph's switch would fire on u right now", for all seven local rules —
 * the two that run the small exact solver (include/small_solve.hpp) included: their covers are
 * enumerated the way the solver does.  Vertices with
 * D(u) > max_degree get 0, like reduce_graph skips them (include/mwvc_reductions.hpp:344).
 *   bit 0 neighborhood_reduction   NW(u) <= W(u)                          (:131-139)
 *   bit 1 twin_fold                a twin among the neighbours of u's LAST neighbour (:141-160)
 *   bit 2 domination_reduction     some neighbour dominates u / is dominated (:162-177)
 *   bit 3 isolated_fold            every neighbour dominates u            (:270-284, reduction_graph.hpp:189-199)
 *   bit 4 independent_fold         W(u) >= NW(u) - min neighbour weight   (:246-268)
 *   bit 5 neighbor_meta_reduction      some heavier neighbour v: MWIS(G[N(v) \\ N[u]]) + W(u) <= W(v)  (:204-232, :179-202)
 *   bit 6 neighborhood_meta_reduction  D(u) <= 8 and W(u) >= NW(u) - MWVC(G[N(u)])                  (:234-252)
 * Adjacency lists must be ascending (they are in the reference's graphs). */
void oracle_reduction_flags(const oracle_graph *g, uint32_t max_degree, uint8_t *flags);

/* What the driver's sort reads from the scores (reference src/GNN_VC.cpp:194-206, 213, 220):
 * keys[u] = min(s, 1.0f - s) — the confidence key of the comparator, std::min semantics —
 * and above_half[u] = s > 0.5f, the class the selection loop acts on. */
void oracle_score_keys(size_t n, const float *scores, float *keys, uint8_t *above_half);

/* Number of worker threads the linear layers use (OpenMP), 1 if built without. */
int oracle_num_threads(void);
void oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
