// host_glue.cpp — the gnnvc_host:: helpers (normally defined in
// gnn-mwvc_amd/host/gnn_inference.cpp) for the _ref builds that link the
// REFERENCE's gnn_inference.cpp instead of ours but still use our matrix TU.
#include <cstdio>
#include <cstdlib>

#include "gnnvc.h"
#include "gnnvc_host.hpp"

namespace gnnvc_host {
int device_ordinal() { return 0; }
void check(int rc, const char *what, const gnnvc_engine *) {
    if (rc == GNNVC_OK) return;
    std::fprintf(stderr, "gnnvc: %s failed: %d\n", what, rc);
    std::abort();
}
gnnvc_engine *ops_engine() {
    static gnnvc_engine *eng = [] {
        gnnvc_engine *e = nullptr;
        static const char kEmpty[] = "ops 0 Layers\n";
        check(gnnvc_create(&e, kEmpty, sizeof kEmpty - 1, 0), "gnnvc_create(ops)");
        return e;
    }();
    return eng;
}
}  // namespace gnnvc_host
