// gnnvc_host.hpp — glue shared by the host mirror's translation units: the
// process-wide engine used by the layer-level entry points and the loud-failure
// policy (the reference signals no errors, src/GNN_VC.cpp prints and returns; a
// missing GPU library must not silently produce numbers, so we abort).
#pragma once
#include <vector>

#include "gnnvc.h"

namespace gnnvc_host {

// Engine without a model, for dot() and the per-layer forward()s.  Device from
// GNNVC_DEVICE (default 0).
gnnvc_engine *ops_engine();

// Aborts with a message when rc != GNNVC_OK.
void check(int rc, const char *what, const gnnvc_engine *e = nullptr);

int device_ordinal();

// GNNVC_DEVICES: "4" = ordinals 0 .. 3, "0,0,1" = that list; empty when unset (one device: device_ordinal()).
std::vector<int> device_list();

}  // namespace gnnvc_host
