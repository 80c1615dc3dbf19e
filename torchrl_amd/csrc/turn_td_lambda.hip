// SCAN family, feature dims after time (I > 1): TD(lambda) with rolling gamma on the map-forming transposing kernel.
#include "gae_interleaved.h"

LaunchResult launch_td_lambda_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned) {
    return dispatch_turn<TdLambdaTurn>(s, vdt, cdt, a, aligned, nullptr, "td_lambda_scan_interleaved");
}
