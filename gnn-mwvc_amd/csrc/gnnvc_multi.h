// gnnvc_multi.h — several devices behind one handle (gnnvc_create_multi, include/gnnvc.h).  Internal.
//
// The front handle is an ordinary engine on devices[0]; its `multi` member points to the state below, which drives one
// ordinary engine per device — each from a host thread of its own — through the PUBLIC ABI (gnnvc_attach_graph_slice /
// gnnvc_stage_forward_device / gnnvc_pack_rows / gnnvc_push_piece / gnnvc_unpack_pieces): each holds the CSR slice of its rows
// and full-size replicated feature buffers, and after the first and second stage every device pushes the rows it computed,
// packed to their live columns and piece by piece, into every peer's receive buffer (one xGMI link per peer, no ring).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>

namespace gnnvc {

struct MultiState;

int multi_create(MultiState **out, const char *model_text, size_t len, const int *devices, int n_devices, std::string &err);
void multi_destroy(MultiState *m);
int multi_devices(const MultiState *m);
uint32_t multi_vertices(const MultiState *m);
bool multi_has_graph(const MultiState *m);
int multi_set_weight_scale(MultiState *m, float ws);
int multi_set_option(MultiState *m, const char *key, long value);
// exactly one of rowptr64 / rowptr32 is non-null (host arrays: the plain upload, or the staged hand-off's page-locked arrays)
int multi_upload(MultiState *m, uint32_t n, const uint64_t *rowptr64, const uint32_t *rowptr32, const uint32_t *col,
                 const uint32_t *w, const uint32_t *nw, std::string &err);
// d_x / d_scores / d_logits live on devices[0]; complete when it returns (every device's stream is drained)
int multi_forward_device(MultiState *m, const float *d_x, float *d_scores, float *d_logits, std::string &err);
int multi_synchronize(MultiState *m);
// rows / entries of part r, wall time of the last forward and of its exchanges as seen by the host
int multi_part_info(const MultiState *m, int part, uint32_t *row_lo, uint32_t *row_hi, uint64_t *entries);
double multi_last_forward_ms(const MultiState *m);
// gnnvc_get_info keys "multi_*" (pieces, packing, bytes shipped per peer and exchange, a part's span of the last forward)
bool multi_get_info(MultiState *m, const char *key, long *value);

}  // namespace gnnvc
