// SCAN family, forward estimators: instantiations of scan_impl.h behind the launch_*_scan entry points.
#include "scan_impl.h"
#include "gae_interleaved.h"


size_t scan_lookback_bytes(i64 O, i64 T, i64 I) {
    // worst case over dtypes: chunk = 512 steps (fp64)
    if (I != 1 || T <= 512) return 0;
    const i64 nch = (T + 511) / 512;
    if (O >= 4096 || O * nch > (1LL << 22)) return 0;
    return sizeof(LbHeader) + (size_t)(O * nch) * sizeof(LbRecord);
}

i64 scan_max_partials(i64 O, i64 T, i64 I) {
    (void)O; (void)T; (void)I;
    return 4096;  // >= every grid this file launches (LONG <= 2048, SHORT <= 512, interleaved <= 4 * #CUs)
}

LaunchResult launch_gae_scan(hipStream_t s, int vdt, int cdt, const GaeArgs &a, void *lb_ws, size_t lb_bytes) {
    // shifted layout (next_value == nullptr): the value buffer is read with scalar loads whatever its alignment
    const bool al = aligned16(a.reward) && (!a.next_value || aligned16(a.value)) && aligned16(a.next_value) && aligned16(a.done) &&
                    aligned16(a.terminated) && aligned16(a.gamma_t) && aligned16(a.lg_t) && aligned16(a.advantage) &&
                    aligned16(a.value_target);
    // I > 1: GAE's own transposing kernel (delta / disc formed before the LDS turn; gae_interleaved.h)
    if (a.I > 1 && a.next_value) {
        const LaunchResult r = launch_gae_turn(s, vdt, cdt, a, al);
        if (r.status != 2) return r;
    }
    auto go = [&](auto op) {
        typedef decltype(op) Op;
        // I == 1, rows of 64 < T <= 128 steps that do not tile a wave, sent through the interleaved kernel as well (an LDS turn
        // buys the row phase its own lane mapping) were measured: short-row kernel / that, us: [20480x100] 9.35 / 8.78,
        // [65536x100] 32.8 / 29.7, [20480x80] 8.0 / 7.5, fp64 [20480x80] 17.9 / 14.7 -- 5 - 10 %, not taken: the un-normalised
        // scan would stop being the very scan of the fused kernel (bit-equal value_target on both routes).
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus(), lb_ws ? lb_bytes : 0, a.partials != nullptr);
        if (!p.ok || p.interleaved) return LaunchResult{2, 0, nullptr};  // I > 1 here: the shifted layout (I == 1 kernels only)
        return launch_scan<Op>(s, a, p, a.O, a.T, a.I, a.partials,
                               p.lookback ? "gae_scan_lookback" : p.long_rows ? "gae_scan_long" : "gae_scan_short", lb_ws);
    };
    return a.next_value ? dispatch_scan<GaeOp, GaeArgs>(vdt, cdt, go) : dispatch_scan<GaeShiftedOp, GaeArgs>(vdt, cdt, go);
}

LaunchResult launch_td_lambda_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, void *lb_ws, size_t lb_bytes) {
    const bool al = aligned16(a.reward) && aligned16(a.next_value) && aligned16(a.done) && aligned16(a.terminated) &&
                    aligned16(a.gamma_t) && aligned16(a.lmbda_t) && aligned16(a.returns);
    if (a.I > 1) {  // the map-forming transposing kernel (end of this file)
        const LaunchResult r = launch_td_lambda_turn(s, vdt, cdt, a, al);
        if (r.status != 2) return r;
    }
    return dispatch_scan<TdLambdaOp, TdLambdaArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus(), lb_ws ? lb_bytes : 0, false);
        if (!p.ok || p.interleaved) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, a.I, nullptr,
                               p.lookback ? "td_lambda_scan_lookback" : p.long_rows ? "td_lambda_scan_long" : "td_lambda_scan_short", lb_ws);
    });
}

LaunchResult launch_td1_scan(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, void *lb_ws, size_t lb_bytes) {
    const bool al = aligned16(a.reward) && aligned16(a.next_value) && aligned16(a.done) && aligned16(a.terminated) &&
                    aligned16(a.gamma_t) && aligned16(a.returns);
    if (a.I > 1) {  // the map-forming transposing kernel (end of this file)
        const LaunchResult r = launch_td1_turn(s, vdt, cdt, a, al);
        if (r.status != 2) return r;
    }
    return dispatch_scan<Td1Op, TdLambdaArgs>(vdt, cdt, [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus(), lb_ws ? lb_bytes : 0, false);
        if (!p.ok || p.interleaved) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, a.I, nullptr,
                               p.lookback ? "td1_scan_lookback" : p.long_rows ? "td1_scan_long" : "td1_scan_short", lb_ws);
    });
}

LaunchResult launch_vtrace_scan(hipStream_t s, int vdt, int cdt, const VtraceArgs &a, void *lb_ws, size_t lb_bytes) {
    if (a.rho_t || a.c_t) return LaunchResult{2, 0, nullptr};  // per-element thresholds: SEQ family
    const bool al = aligned16(a.log_pi) && aligned16(a.log_mu) && (!a.next_value || aligned16(a.value)) && aligned16(a.next_value) &&
                    aligned16(a.reward) && aligned16(a.done) && aligned16(a.terminated) && aligned16(a.gamma_t) &&
                    aligned16(a.advantage) && aligned16(a.vs);
    if (a.I > 1 && a.next_value) {  // the map-forming transposing kernel (end of this file)
        const LaunchResult r = launch_vtrace_turn(s, vdt, cdt, a, al);
        if (r.status != 2) return r;
    }
    auto go = [&](auto op) {
        typedef decltype(op) Op;
        Plan p = make_plan(a.O, a.T, a.I, Op::V, al, device_cus(), lb_ws ? lb_bytes : 0, false);
        if (!p.ok) return LaunchResult{2, 0, nullptr};
        return launch_scan<Op>(s, a, p, a.O, a.T, a.I, nullptr,
                               p.interleaved ? "vtrace_scan_interleaved" : p.lookback ? "vtrace_scan_lookback" : p.long_rows ? "vtrace_scan_long" : "vtrace_scan_short", lb_ws);
    };
    return a.next_value ? dispatch_scan<VtraceOp, VtraceArgs>(vdt, cdt, go) : dispatch_scan<VtraceShiftedOp, VtraceArgs>(vdt, cdt, go);
}

// ---- feature dims after time (I > 1): the map-forming transposing kernels of gae_interleaved.h, one policy per estimator --------
// They are instantiated HERE, beside the I == 1 kernels, on purpose: compiled in translation units of their own (tried for build
// time: they are most of this file's 5 minutes) the very same kernels come out 19 % longer (1842 instead of 1552 instructions for
// the C3 instantiation: the next chunk's mask loads sink below the stores, scalar tails lose a range fact) and 9 % slower --
// C3 8.7 -> 9.9 us, TD(lambda) 8.8 -> 9.5 us; rocprofv3 kernel averages 9.27 vs 10.12 us.  Force-inlining the kernel's lambdas
// does not change that; what in this file's context the optimiser feeds on was not found.
LaunchResult launch_gae_turn(hipStream_t s, int vdt, int cdt, const GaeArgs &a, bool aligned) {
    return dispatch_turn<GaeTurn>(s, vdt, cdt, a, aligned, a.partials, "gae_scan_interleaved");
}
LaunchResult launch_td_lambda_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned) {
    return dispatch_turn<TdLambdaTurn>(s, vdt, cdt, a, aligned, nullptr, "td_lambda_scan_interleaved");
}
LaunchResult launch_td1_turn(hipStream_t s, int vdt, int cdt, const TdLambdaArgs &a, bool aligned) {
    return dispatch_turn<Td1Turn>(s, vdt, cdt, a, aligned, nullptr, "td1_scan_interleaved");
}
LaunchResult launch_vtrace_turn(hipStream_t s, int vdt, int cdt, const VtraceArgs &a, bool aligned) {
    return dispatch_turn<VtraceTurn>(s, vdt, cdt, a, aligned, nullptr, "vtrace_scan_interleaved");
}

// outer rows of more than a team's chunk that the 1024-thread kernel takes (capi.hip AUTO)
bool scan_turn_big_ok(int vdt, i64 T, i64 I) {
    Plan p;
    return turn_big_plan(1 << 20, T, I, vdt == 0 ? 4 : 2, true, p);
}
