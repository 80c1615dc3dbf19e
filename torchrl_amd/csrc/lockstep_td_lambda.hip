// Lock-step kernels, td_lambda instantiations (lockstep_impl.h).
#include "lockstep_impl.h"

LaunchResult launch_td_lambda_lockstep(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a) {
    const bool al = al16(a.reward) && al16(a.next_value) && al16(a.done) && al16(a.terminated) && al16(a.gamma_t) &&
                    al16(a.lmbda_t) && al16(a.returns);
    return dispatch_ls<TdLambdaOp>(s, vdt, cdt, a, al, nullptr, "td_lambda_scan_lockstep");
}
