// SCAN family, feature dims after time (I > 1): V-trace on the map-forming transposing kernel (gae_interleaved.h).
#include "gae_interleaved.h"

LaunchResult launch_vtrace_turn(hipStream_t s, int vdt, int cdt, const VtraceArgs &a, bool aligned) {
    return dispatch_turn<VtraceTurn>(s, vdt, cdt, a, aligned, nullptr, "vtrace_scan_interleaved");
}
