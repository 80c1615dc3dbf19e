// SCAN family, feature dims after time (I > 1): TD(1) with rolling gamma on the map-forming transposing kernel.
#include "gae_interleaved.h"

LaunchResult launch_td1_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned) {
    return dispatch_turn<Td1Turn>(s, vdt, cdt, a, aligned, nullptr, "td1_scan_interleaved");
}
