// SCAN family, feature dims after time (I > 1): GAE on the map-forming transposing kernel (gae_interleaved.h).
#include "gae_interleaved.h"

LaunchResult launch_gae_turn(hipStream_t s, int vdt, int cdt, const GaeArgs &a, bool aligned) {
    return dispatch_turn<GaeTurn>(s, vdt, cdt, a, aligned, a.partials, "gae_scan_interleaved");
}

// outer rows of more than a team's chunk that the 1024-thread kernel takes (GAE, TD(lambda), TD(1); capi.hip AUTO)
bool scan_turn_big_ok(int vdt, i64 T, i64 I) {
    Plan p;
    return turn_big_plan(1 << 20, T, I, vdt == 0 ? 4 : 2, true, p);
}
