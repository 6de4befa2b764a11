// api.hip — export-level C-ABI (host pointers) of libmxgpu: one entry point per
// Rcpp export of the reference's hot path, marshalling host vectors to the
// device, running the HIP kernels and bringing the result back.  There is no
// CPU fallback: without a usable GPU every call fails with an error.
#include "mx_common.h"
#include "tile_geometry.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>
#include <csignal>
#include <execinfo.h>
#include <unistd.h>

// One translation unit, five files along its layers (everything below `namespace mx` shares file-local state —
// the CSR cache, the per-thread lanes, the small-call stages — so the parts are included, not linked):
#include "api_core.inc"           // helpers, device buffers, CSR cache, lanes, registration, row partition
#include "api_small.inc"          // the small-call path
#include "api_spmm_export.inc"    // pipelined + sharded CSR x dense exports
#include "api_spmv_export.inc"    // options, the SpMV export; closes namespace mx
#include "api_exports.inc"        // extern "C": the export level, the offload gate
