// Lock-step kernels, vtrace instantiations (lockstep_impl.h).
#include "lockstep_impl.h"

LaunchResult launch_vtrace_lockstep(hipStream_t s, int vdt, int cdt, const VtraceArgs &a) {
    const bool al = al16(a.log_pi) && al16(a.log_mu) && al16(a.value) && al16(a.next_value) && al16(a.reward) &&
                    al16(a.done) && al16(a.terminated) && al16(a.gamma_t) && al16(a.advantage) && al16(a.vs);
    return dispatch_ls<VtraceOp>(s, vdt, cdt, a, al, nullptr, "vtrace_scan_lockstep");
}
