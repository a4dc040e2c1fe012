/*
 * gnnvc_abi_oracle.c — TEST DOUBLE of the C ABI (include/gnnvc.h) served by the
 * CPU oracle.  Lives under tests/ and is linked ONLY into the link-level
 * drop-in checks of oracle/_ref (CPU container, no GPU): it lets the reference's
 * own driver run against this repo's host mirror so the boundary can be checked
 * end to end.  The product library (libgnnvc_hip.so) contains none of this and
 * has no CPU path.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gnnvc.h"
#include "gnnvc_oracle.h"

struct gnnvc_engine {
    oracle_model *m;
    uint32_t n;
    uint64_t *rowptr;
    uint32_t *col, *w, *nw;
    int in_w, out_w, sig;
    unsigned long forwards;
};

static void widths(gnnvc_engine *e) {
    int wd = 1, k0 = -1, graphs = 0;
    for (int i = 0; i < e->m->n_layers && k0 < 0; i++) {
        if (e->m->layers[i].kind == ORACLE_LAYER_LINEAR) k0 = (int)e->m->layers[i].k;
        else if (e->m->layers[i].kind == ORACLE_LAYER_GRAPH) graphs++;
    }
    if (k0 >= 0) { wd = k0; while (graphs--) wd = (wd - 3) / 2; }
    e->in_w = wd;
    for (int i = 0; i < e->m->n_layers; i++) {
        if (e->m->layers[i].kind == ORACLE_LAYER_LINEAR) wd = (int)e->m->layers[i].m;
        else if (e->m->layers[i].kind == ORACLE_LAYER_GRAPH) wd = 2 * wd + 3;
    }
    e->out_w = wd;
    e->sig = e->m->n_layers > 0 && e->m->layers[e->m->n_layers - 1].kind == ORACLE_LAYER_SIGMOID;
}

int gnnvc_abi_version(void) { return GNNVC_ABI_VERSION; }
const char *gnnvc_strerror(int code) { return code == 0 ? "ok" : "error (oracle-backed test double)"; }
const char *gnnvc_last_error(const gnnvc_engine *e) { (void)e; return ""; }

int gnnvc_create(gnnvc_engine **out, const char *text, size_t len, int device) {
    (void)device;
    gnnvc_engine *e = (gnnvc_engine *)calloc(1, sizeof *e);
    e->m = oracle_model_parse(text, len);
    if (!e->m) { free(e); return GNNVC_ERR_INVALID; }
    widths(e);
    *out = e;
    return GNNVC_OK;
}

/* (the double has no devices: a multi-device handle is the same engine) */
int gnnvc_create_multi(gnnvc_engine **out, const char *text, size_t len, const int *devices, int n_devices) {
    if (!devices || n_devices < 1) return GNNVC_ERR_INVALID;
    return gnnvc_create(out, text, len, devices[0]);
}

/* (the double has no plans to switch: every option is accepted and changes nothing) */
int gnnvc_set_option(gnnvc_engine *e, const char *key, long value) {
    (void)value;
    return (e && key) ? GNNVC_OK : GNNVC_ERR_INVALID;
}

static void drop_graph(gnnvc_engine *e) {
    free(e->rowptr); free(e->col); free(e->w); free(e->nw);
    e->rowptr = NULL; e->col = e->w = e->nw = NULL;
}

void gnnvc_destroy(gnnvc_engine *e) {
    if (!e) return;
    if (getenv("GNNVC_TEST_DOUBLE_TRACE")) fprintf(stderr, "gnnvc test double: %lu forwards\n", e->forwards);
    drop_graph(e);
    oracle_model_free(e->m);
    free(e);
}

int gnnvc_set_weight_scale(gnnvc_engine *e, float ws) { oracle_model_set_weight_scale(e->m, ws); return GNNVC_OK; }
int gnnvc_last_forward_ms(gnnvc_engine *e, float *total_ms, float *stage_ms, int max_stages) {
    (void)e; (void)stage_ms; (void)max_stages;
    if (total_ms) *total_ms = 0.0f;   /* no device behind the double */
    return GNNVC_OK;
}
int gnnvc_get_info(const gnnvc_engine *e, const char *key, long *value) {
    (void)e; (void)key;
    if (value) *value = 0;
    return GNNVC_OK;
}
int gnnvc_in_width(const gnnvc_engine *e) { return e->in_w; }
int gnnvc_out_width(const gnnvc_engine *e) { return e->out_w; }

static void *dup(const void *p, size_t bytes) {
    void *q = malloc(bytes ? bytes : 1);
    if (bytes) memcpy(q, p, bytes);
    return q;
}

int gnnvc_upload_graph(gnnvc_engine *e, uint32_t n, const uint64_t *rowptr, const uint32_t *col,
                       const uint32_t *w, const uint32_t *nw) {
    drop_graph(e);
    e->n = n;
    uint64_t zero = 0;
    e->rowptr = (uint64_t *)dup(n ? rowptr : &zero, ((size_t)n + 1) * 8);
    e->col = (uint32_t *)dup(col, (size_t)(n ? rowptr[n] : 0) * 4);
    e->w = (uint32_t *)dup(w, (size_t)n * 4);
    e->nw = (uint32_t *)dup(nw, (size_t)n * 4);
    return GNNVC_OK;
}

/* staged hand-off: plain heap staging, committed through the same path as gnnvc_upload_graph */
static uint32_t *st_rowptr, *st_col, *st_w, *st_nw;
static size_t st_ncap, st_zcap;
static uint32_t st_n;
static uint64_t st_nnz, st_sent;

int gnnvc_graph_staging(gnnvc_engine *e, uint32_t n, uint64_t nnz, uint32_t **rowptr, uint32_t **col,
                        uint32_t **w, uint32_t **nw) {
    (void)e;
    if ((size_t)n + 1 > st_ncap) {
        st_ncap = (size_t)n + 1;
        st_rowptr = (uint32_t *)realloc(st_rowptr, st_ncap * 4);
        st_w = (uint32_t *)realloc(st_w, st_ncap * 4);
        st_nw = (uint32_t *)realloc(st_nw, st_ncap * 4);
    }
    if (nnz + 1 > st_zcap) {
        st_zcap = nnz + 1;
        st_col = (uint32_t *)realloc(st_col, st_zcap * 4);
    }
    st_n = n; st_nnz = nnz; st_sent = 0;
    if (rowptr) *rowptr = st_rowptr;
    if (col) *col = st_col;
    if (w) *w = st_w;
    if (nw) *nw = st_nw;
    return GNNVC_OK;
}
int gnnvc_staged_columns_ready(gnnvc_engine *e, uint64_t first, uint64_t count) {
    (void)e;
    if (first != st_sent || first + count > st_nnz) return GNNVC_ERR_INVALID;
    st_sent = first + count;
    return GNNVC_OK;
}
int gnnvc_commit_staged_graph(gnnvc_engine *e) {
    uint64_t *rp = (uint64_t *)malloc(((size_t)st_n + 1) * 8);
    for (size_t i = 0; i <= st_n; i++) rp[i] = st_n ? st_rowptr[i] : 0;
    if (st_n && rp[st_n] != st_nnz) { free(rp); return GNNVC_ERR_INVALID; }
    int rc = gnnvc_upload_graph(e, st_n, rp, st_col, st_w, st_nw);
    free(rp);
    return rc;
}

static oracle_graph view(const gnnvc_engine *e) {
    oracle_graph g = {e->n, e->rowptr, e->col, e->w, e->nw};
    return g;
}

int gnnvc_forward(gnnvc_engine *e, const float *x, float *scores, float *logits) {
    if (!e->rowptr) return GNNVC_ERR_STATE;
    if (e->n == 0) return GNNVC_OK;
    if (getenv("GNNVC_TEST_DOUBLE_TRACE")) fprintf(stderr, "gnnvc test double: forward N=%u nnz=%llu\n", e->n, (unsigned long long)e->rowptr[e->n]);
    e->forwards++;
    oracle_graph g = view(e);
    uint32_t wd = 0;
    float *tmp = (float *)malloc((size_t)e->n * 35 * sizeof(float) + 64);
    int rc = oracle_predict(e->m, &g, (uint32_t)e->in_w, x, tmp, &wd, -1, 0);
    if (rc == 0) memcpy(scores, tmp, (size_t)e->n * wd * sizeof(float));
    if (rc == 0 && logits && e->sig) {
        rc = oracle_predict(e->m, &g, (uint32_t)e->in_w, x, tmp, &wd, e->m->n_layers - 2, 0);
        if (rc == 0) memcpy(logits, tmp, (size_t)e->n * wd * sizeof(float));
    }
    free(tmp);
    return rc == 0 ? GNNVC_OK : GNNVC_ERR_INVALID;
}

int gnnvc_graph_layer_forward(gnnvc_engine *e, uint32_t f, const float *in, float *out) {
    if (!e->rowptr) return GNNVC_ERR_STATE;
    oracle_graph g = view(e);
    oracle_graph_layer(&g, e->m->weight_scale, f, in, out, 0);
    return GNNVC_OK;
}

int gnnvc_linear_forward(gnnvc_engine *e, uint32_t n, uint32_t k, uint32_t m, const float *in,
                         const float *W, const float *bias, float *out) {
    (void)e;
    oracle_linear_layer(n, k, m, in, W, bias, out);
    return GNNVC_OK;
}

int gnnvc_relu_forward(gnnvc_engine *e, size_t count, const float *in, float *out) {
    (void)e;
    oracle_relu(count, in, out);
    return GNNVC_OK;
}

int gnnvc_sigmoid_forward(gnnvc_engine *e, size_t count, const float *in, float *out) {
    (void)e;
    oracle_sigmoid(count, in, out);
    return GNNVC_OK;
}

/* tools/make_golden_layers.py only: with a genuine cblas_sgemm installed here (dlopen'ed OpenBLAS), the products the
 * reference's layer code asks dot() for go to THAT library exactly as src/matrix.cpp:112-121 would send them
 * (row-major, no transposes, alpha 1, beta 0) — so the fixtures it writes do not depend on this repo's restatement
 * of the product. */
typedef void (*test_cblas_sgemm_fn)(int order, int ta, int tb, int M, int N, int K, float alpha, const float *A, int lda,
                                    const float *B, int ldb, float beta, float *C, int ldc);
static test_cblas_sgemm_fn g_test_cblas = 0;
static unsigned long g_test_cblas_calls = 0;
void gnnvc_test_set_cblas_sgemm(void *fn) { g_test_cblas = (test_cblas_sgemm_fn)fn; g_test_cblas_calls = 0; }
unsigned long gnnvc_test_cblas_calls(void) { return g_test_cblas_calls; }

int gnnvc_sgemm(gnnvc_engine *e, int ta, int tb, uint32_t m, uint32_t n, uint32_t k, const float *A,
                uint32_t lda, const float *B, uint32_t ldb, float beta, float *C, uint32_t ldc) {
    (void)e;
    if (g_test_cblas) {
        g_test_cblas(101, ta ? 112 : 111, tb ? 112 : 111, (int)m, (int)n, (int)k, 1.0f, A, (int)lda, B, (int)ldb, beta, C, (int)ldc);
        ++g_test_cblas_calls;
        return GNNVC_OK;
    }
    for (uint32_t i = 0; i < m; i++)
        for (uint32_t j = 0; j < n; j++) {
            float acc = 0.0f;
            for (uint32_t kk = 0; kk < k; kk++) {
                float a = ta ? A[(size_t)kk * lda + i] : A[(size_t)i * lda + kk];
                float b = tb ? B[(size_t)j * ldb + kk] : B[(size_t)kk * ldb + j];
                acc = fmaf(a, b, acc);
            }
            float *c = &C[(size_t)i * ldc + j];
            *c = (beta == 0.0f) ? acc : fmaf(beta, *c, acc);
        }
    return GNNVC_OK;
}

/* the device-side derivation of the next graph has no counterpart in this CPU double: the host wrapper falls back to
 * its full hand-off when these refuse */
int gnnvc_derive_graph_begin(gnnvc_engine *e, uint32_t n_new, const uint32_t *old_row, const uint32_t *rowptr_new, uint32_t *tail) {
    (void)e; (void)n_new; (void)old_row; (void)rowptr_new; (void)tail;
    return GNNVC_ERR_UNSUPPORTED;
}
int gnnvc_derive_graph_commit(gnnvc_engine *e, const uint32_t *tail_cols, uint64_t n_tail, const uint32_t *w, const uint32_t *nw) {
    (void)e; (void)tail_cols; (void)n_tail; (void)w; (void)nw;
    return GNNVC_ERR_UNSUPPORTED;
}
int gnnvc_graph_row_hashes(gnnvc_engine *e, uint64_t *hashes) {
    (void)e; (void)hashes;
    return GNNVC_ERR_UNSUPPORTED;
}
