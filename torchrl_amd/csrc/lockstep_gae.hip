// Lock-step kernels, gae instantiations (lockstep_impl.h).
#include "lockstep_impl.h"

LaunchResult launch_gae_lockstep(hipStream_t s, int vdt, int cdt, const GaeArgs &a) {
    const bool al = al16(a.reward) && al16(a.value) && al16(a.next_value) && al16(a.done) && al16(a.terminated) &&
                    al16(a.gamma_t) && al16(a.lg_t) && al16(a.advantage) && al16(a.value_target);
    return dispatch_ls<GaeOp>(s, vdt, cdt, a, al, a.partials, "gae_scan_lockstep");
}
