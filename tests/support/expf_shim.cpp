// Host build of gnn-mwvc_amd/csrc/expf_glibc.h for tests/test_expf_restatement.py.
#include <math.h>
#include <stddef.h>

#include "../../gnn-mwvc_amd/csrc/expf_glibc.h"

extern "C" void expf_restated(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = gnnvc::expf_glibc(in[i]);
}
extern "C" void expf_libm(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = expf(in[i]);
}
// 1.0f / (1.0f + exp(-x)) both ways
extern "C" void sigmoid_restated(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = 1.0f / (1.0f + gnnvc::expf_glibc(-in[i]));
}
extern "C" void sigmoid_libm(const float *in, float *out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = 1.0f / (1.0f + expf(-in[i]));
}
