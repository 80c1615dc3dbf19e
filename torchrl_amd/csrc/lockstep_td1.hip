// Lock-step kernels, td1 instantiations (lockstep_impl.h).
#include "lockstep_impl.h"

LaunchResult launch_td1_lockstep(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a) {
    const bool al = al16(a.reward) && al16(a.next_value) && al16(a.done) && al16(a.terminated) && al16(a.gamma_t) &&
                    al16(a.returns);
    return dispatch_ls<Td1Op>(s, vdt, cdt, a, al, nullptr, "td1_scan_lockstep");
}
