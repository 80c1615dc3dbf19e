// Wavefront-scan kernels ("SCAN" family): kernel templates, plan and launcher, shared by scan_kernels.hip (forward
// estimators) and bwd_kernels.hip (their adjoints).  Time-contiguous rows (I == 1) unless noted.
//
// The recurrence x_t = b_t + a_t * x_{t+1} (GAE: a = lmbda*gamma*(1-done), b = delta;
// TD(lambda): a = gamma*lmbda*(1-done), b = r + gamma*v'*(1 - lmbda*(1-done)); V-trace:
// a = gamma*(1-done)*c, b = rho*delta) is a reverse scan of affine maps (a, b).  The flat
// (O*T) array is cut into chunks of kWaves*64*V consecutive elements:
//   1. every lane loads V consecutive time steps with one 16-byte load per tensor
//      (fully coalesced: a wave reads 1 KiB per instruction), fuses the delta / coefficient
//      computation and composes its V maps in registers;
//   2. the 64 lane aggregates are suffix-scanned with wave shuffles (6 steps);
//   3. the wave aggregates of a chunk are exchanged through LDS (one barrier per chunk);
//   4. every lane re-runs its V steps from its carry-in with the reference's exact
//      per-step expression and writes its outputs with 16-byte stores.
// Row ends are *hard* segment boundaries (the carry is discarded by a select, never
// multiplied), `done` boundaries are soft zeros exactly as in the reference
// (x*0, functional.py:174-178), so at a done step the result is delta_t bit-for-bit.
//
// Two geometries, no inter-block communication in either:
//   SHORT rows (T <= chunk): a chunk is R whole rows (R*T <= chunk) and needs no carry.  The
//                            kernel is persistent: ~one block of 4 teams (1024 threads) per CU,
//                            each team walks chunks grid-stride with the next chunk's loads
//                            issued before the current one is scanned, so reads and writes of
//                            neighbouring chunks overlap in time.
//   LONG rows  (T >  chunk): a block (one team) owns whole rows and walks each row's chunks from
//                            the end, carrying x (and V-trace's vs) in registers, again with the
//                            next chunk prefetched.
//
// HBM traffic = algorithmic bytes: each input element is read once, each output written
// once (GAE fp32 22 B / transition); per-block {sum, sumsq} partials add 16 B per block.
#pragma once
#include <type_traits>

#include "scan_core.h"
#include "moment_gather.h"

namespace {

using namespace scan;
constexpr int kTeams = 4;                       // SHORT geometry: teams per block
constexpr int kShortThreads = kTeams * kTeam;   // 1024
constexpr int kLongThreads = kTeam;             // 256

struct ScanGeom {
    i64 O, T;
    int rows_per_chunk;  // SHORT geometry
    int scan_lanes;      // bound on the lanes one row spans (shuffle steps)
    i64 nchunks;         // SHORT geometry
};

// An op whose inputs do not fit the register file twice (V-trace backward: 10 arrays of V) can say kPrefetchShort /
// kPrefetchLong = false: that chunk loop then loads at the top of an iteration instead of holding the next chunk's inputs
// while this one is scanned.
template <class Op, class = void>
struct OpPrefetchShort { static constexpr bool value = true; };
template <class Op>
struct OpPrefetchShort<Op, std::void_t<decltype(Op::kPrefetchShort)>> { static constexpr bool value = Op::kPrefetchShort; };
template <class Op, class = void>
struct OpPrefetchLong { static constexpr bool value = true; };
template <class Op>
struct OpPrefetchLong<Op, std::void_t<decltype(Op::kPrefetchLong)>> { static constexpr bool value = Op::kPrefetchLong; };

__device__ __forceinline__ int clamp_nvalid(int count, int lane_off, int V) {
    const int nv = count - lane_off;
    return nv < 0 ? 0 : (nv > V ? V : nv);
}

// ---- SHORT rows: persistent, 4 teams per block -------------------------------------------------
template <class Op, bool VEC, bool WAVE_INDEP>
__global__ __launch_bounds__(kShortThreads) void scan_short_kernel(typename Op::Args args, ScanGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int NW = kShortThreads / kWave;
    __shared__ TeamShared<Op> sh[kTeams];
    __shared__ double red[2 * NW];
    const int team = threadIdx.x / kTeam;
    const int lane_off = (threadIdx.x & (kTeam - 1)) * V;
    const unsigned T = (unsigned)geo.T;
    const i64 stride = (i64)gridDim.x * kTeams;
    const i64 niter = (geo.nchunks + stride - 1) / stride;  // uniform over the grid (barriers inside scan_chunk)

    auto geom_of = [&](i64 it, i64 &start, int &count) {
        const i64 c = it * stride + (i64)blockIdx.x * kTeams + team;
        const i64 row0 = c * geo.rows_per_chunk;
        i64 nrows = c < geo.nchunks ? geo.O - row0 : 0;
        if (nrows > geo.rows_per_chunk) nrows = geo.rows_per_chunk;
        if (nrows < 0) nrows = 0;
        start = row0 * geo.T;
        count = (int)(nrows * geo.T);
    };

    constexpr bool PF = OpPrefetchShort<Op>::value;
    double s = 0.0, ss = 0.0;
    typename Op::Raw cur, nxt;
    i64 start;
    int count;
    geom_of(0, start, count);
    if (PF) load_inputs<Op, VEC>(args, start + lane_off, clamp_nvalid(count, lane_off, V), cur);
    for (i64 it = 0; it < niter; ++it) {
        if (PF && it + 1 < niter) {
            i64 nstart;
            int ncount;
            geom_of(it + 1, nstart, ncount);
            load_inputs<Op, VEC>(args, nstart + lane_off, clamp_nvalid(ncount, lane_off, V), nxt);
        }
        geom_of(it, start, count);
        if (!PF) load_inputs<Op, VEC>(args, start + lane_off, clamp_nvalid(count, lane_off, V), cur);
        const int nvalid = clamp_nvalid(count, lane_off, V);
        const i64 flat = start + lane_off;
        Op op;
        op.prepare(args, cur);
        TC Xc = (TC)0;
        TS Xv = (TS)0;
        scan_chunk<Op, WAVE_INDEP>(op, sh[team], (int)(it & 1), nvalid, (unsigned)lane_off % T, T, geo.scan_lanes, Xc, Xv,
                                   [&](const TC(&xs)[V], const bool(&row_end)[V], TC x_in, TS v_next) {
                                       emit_outputs<Op, VEC>(op, args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
                                   });
        if (PF) cur = nxt;
    }
    if (partials) {
        block_sum2<kShortThreads>(s, ss, red);
        if (threadIdx.x == 0) {
            partials[2 * (i64)blockIdx.x] = s;
            partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}


// ---- SHORT rows with feature dims (I > 1, T*I <= chunk): gae_interleaved.h ---------------------------------
// (the generic transposing kernel that lived here -- every input array staged through LDS, the Op's own scan_chunk on the
// gathered rows -- served TD(lambda) / TD(1) / V-trace until round 3; all four estimators now run on the map-forming kernel of
// gae_interleaved.h, which moves 2 arrays instead of 5 - 7 across LDS.  make_plan below still lays the chunks out for it.)

// ---- LONG rows: one team per block walks (row, chunk) pairs from each row's end -----------------
// DUAL: interior chunks (no row end inside) take the scan without the hard-reset machinery.  Chosen by the launcher for rows
// of 4+ chunks (a second copy of the scan code costs a two-chunk row more in instruction fetch than it saves: C1 5.1 -> 7.8 us)
// and for ops that keep their registers with two copies inlined (not V-trace backward: 116 -> 159 us at C5).
template <class Op, class = void>
struct OpDualScan { static constexpr bool value = true; };
template <class Op>
struct OpDualScan<Op, std::void_t<decltype(Op::kDualScan)>> { static constexpr bool value = Op::kDualScan; };

template <class Op, bool VEC, bool DUAL>
__global__ __launch_bounds__(kLongThreads) void scan_long_kernel(typename Op::Args args, ScanGeom geo, double *partials) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int kChunk = kLongThreads * V;
    __shared__ TeamShared<Op> sh;
    __shared__ double red[2 * kWaves];
    double s = 0.0, ss = 0.0;
    const unsigned T = (unsigned)geo.T;
    const int lane_off = (int)threadIdx.x * V;
    const int nchunks = (int)((geo.T + kChunk - 1) / kChunk);
    auto chunk_count = [&](int cc) {
        i64 cnt = geo.T - (i64)cc * kChunk;
        return (int)(cnt > kChunk ? kChunk : cnt);
    };
    int buf = 0;
    i64 row = blockIdx.x;
    int c = nchunks - 1;
    constexpr bool PF = OpPrefetchLong<Op>::value;
    typename Op::Raw cur, nxt;
    if (PF && row < geo.O)
        load_inputs<Op, VEC>(args, row * geo.T + (i64)c * kChunk + lane_off, clamp_nvalid(chunk_count(c), lane_off, V), cur);
    TC Xc = (TC)0;
    TS Xv = (TS)0;
    while (row < geo.O) {
        i64 nrow = row;
        int nc = c - 1;
        if (nc < 0) { nrow = row + gridDim.x; nc = nchunks - 1; }
        if (PF && nrow < geo.O)
            load_inputs<Op, VEC>(args, nrow * geo.T + (i64)nc * kChunk + lane_off,
                                   clamp_nvalid(chunk_count(nc), lane_off, V), nxt);
        const i64 start = row * geo.T + (i64)c * kChunk;
        const int nvalid = clamp_nvalid(chunk_count(c), lane_off, V);
        const i64 flat = start + lane_off;
        if (!PF) load_inputs<Op, VEC>(args, flat, nvalid, cur);
        Op op;
        op.prepare(args, cur);
        auto emit = [&](const TC(&xs)[V], const bool(&row_end)[V], TC x_in, TS v_next) {
            emit_outputs<Op, VEC>(op, args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
        };
        auto no_carry = [](TeamShared<Op> &, int, TC &, TS &) {};
        // only a row's last chunk holds its end: every other chunk scans without the hard-reset machinery (scan_core.h)
        if (!DUAL || c == nchunks - 1) scan_chunk_t<Op, false, kScanGeneral>(op, sh, buf, nvalid, (unsigned)((i64)c * kChunk + lane_off), T, kWave, Xc, Xv, emit, no_carry);
        else scan_chunk_t<Op, false, kScanNoRowEnd>(op, sh, buf, nvalid, (unsigned)((i64)c * kChunk + lane_off), T, kWave, Xc, Xv, emit, no_carry);
        buf ^= 1;
        if (PF) cur = nxt;
        row = nrow;
        c = nc;
    }
    if (partials) {
        block_sum2<kLongThreads>(s, ss, red);
        if (threadIdx.x == 0) {
            partials[2 * (i64)blockIdx.x] = s;
            partials[2 * (i64)blockIdx.x + 1] = ss;
        }
    }
}


// ---- LONG rows, few of them: one chunk (4 teams) per block, decoupled look-back between chunks -------
// When there are too few rows to fill the chip (O < ~2 * #CUs) a row's chunks are scanned in PARALLEL:
// every block takes one chunk, publishes the chunk's aggregate map, then resolves its carry by
// walking the records of the chunks that follow it in time ("look-back", here toward later time
// because the scan runs from T-1 to 0) until it meets one whose inclusive value is known, and
// publishes its own inclusive value.  Chunks are handed out through an atomic ticket in reverse
// time order, so every chunk a block may wait on belongs to a block that has already started and
// publishes its aggregate without waiting on anyone: progress is guaranteed for any dispatch order
// or XCD placement.  Records are {flag | payload}: the payload is written with agent-scope
// (write-through) stores, drained (s_waitcnt vmcnt(0)), then the flag is stored; the consumer polls
// the flag with agent-scope loads and reads the payload with agent-scope loads
// (MI355X_MICROARCH.md, valid hand-off forms).  Ticket and records are zeroed by a memset node on the
// same stream before every launch.  Spins are bounded (1 s): on timeout NaNs are written.
struct LbRecord {             // one 64-byte line per chunk, zeroed before every launch; every 8-byte word carries its own state in
                              // its low 2 bits (0 = not written yet) and is never rewritten once set
    unsigned long long a, b;  // aggregate map of the chunk: double bits | 1, one 16-byte store
    unsigned long long x;     // inclusive value (x of the chunk's first step): double bits | 2
    unsigned long long v0;    // V-trace: state_value of the chunk's first step: double bits | 1
    unsigned long long pad[4];
};
static_assert(sizeof(LbRecord) == 64, "record size");
constexpr int kLbMaxRows = 512;
struct LbHeader {
    unsigned error;
    unsigned pad[63];
    struct { unsigned next; unsigned pad[15]; } row[kLbMaxRows];  // one ticket counter per row, each on its own 64-byte line
};

template <typename T>
__device__ __forceinline__ unsigned long long lb_bits(T x) { return (unsigned long long)__double_as_longlong((double)x); }
template <typename T>
__device__ __forceinline__ T lb_val(unsigned long long b) { return (T)__longlong_as_double((long long)b); }
// The state rides in the two low mantissa bits of the double: nothing is lost for fp32 carries (29 zero bits there), 2^-51
// relative for fp64 ones.  One poll = ONE round trip: both 16-byte halves {a, b} and {x, v0} of a record are loaded together.
__device__ __forceinline__ unsigned long long lb_tag(unsigned long long bits, unsigned state) { return (bits & ~3ull) | state; }
template <typename T>
__device__ __forceinline__ T lb_untag(unsigned long long b) { return lb_val<T>(b & ~3ull); }
__device__ __forceinline__ void lb_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void lb_store2(unsigned long long *p, unsigned long long v0, unsigned long long v1) {
    gather::store_granule(p, v0, v1);
}
__device__ __forceinline__ void lb_load_record(const LbRecord *r, gather::u64x2 &ab, gather::u64x2 &xv) {
    asm volatile(
        "global_load_dwordx4 %0, %2, off sc1\n\t"
        "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(ab), "=&v"(xv)
        : "v"(r)
        : "memory");
}

constexpr int kLbTeams = 4;  // teams per look-back block: one ticket / one record per 4 * 256 * V steps

// waves per SIMD an op asks the look-back kernel to be compiled for (Op::kLbWavesPerSimd; default: no constraint): a look-back
// block spends most of its life waiting -- for its loads, then for its successors' records -- so a second resident
// 1024-thread block per CU (8 waves per SIMD, <= 64 VGPRs) overlaps those waits
template <class Op, class = void>
struct LbWavesPerSimd { static constexpr int value = 1; };
template <class Op>
struct LbWavesPerSimd<Op, std::void_t<decltype(Op::kLbWavesPerSimd)>> { static constexpr int value = Op::kLbWavesPerSimd; };

template <class Op, bool VEC>
__global__ __launch_bounds__(kLbTeams * kTeam, LbWavesPerSimd<Op>::value) void scan_lookback_kernel(typename Op::Args args, ScanGeom geo, LbHeader *hdr,
                                                                        LbRecord *rec) {
    typedef typename Op::TC TC;
    typedef typename Op::TS TS;
    constexpr int V = Op::V;
    constexpr int kChunk = kTeam * V;            // steps per team
    constexpr int kBlk = kLbTeams * kChunk;      // steps per block = per ticket = per record
    __shared__ TeamShared<Op> sh[kLbTeams];
    __shared__ unsigned s_ticket;
    __shared__ TC s_xc[kLbTeams];
    __shared__ TS s_xv[kLbTeams];
    const unsigned T = (unsigned)geo.T;
    const int team = threadIdx.x / kTeam;
    const int lane_off = (int)(threadIdx.x & (kTeam - 1)) * V;
    const int nchunks = (int)((geo.T + kBlk - 1) / kBlk);

    // One ticket per block, drawn from its ROW's counter (blocks are dealt to rows round-robin, so the rows'
    // counters are hit in parallel): atomics on one address serialise (~13 ns each, measured), so they -- not
    // HBM -- bound the kernel unless a ticket covers enough bytes.
    const i64 row = blockIdx.x % geo.O;
    if (threadIdx.x == 0) s_ticket = __hip_atomic_fetch_add(&hdr->row[row].next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int c = nchunks - 1 - (int)s_ticket;  // later chunks of a row draw earlier tickets
    if (c < 0) return;
    const i64 t_team = (i64)c * kBlk + (i64)team * kChunk;  // time index of the team's first step
    i64 cnt = geo.T - t_team;
    if (cnt > kChunk) cnt = kChunk;
    if (cnt < 0) cnt = 0;
    const int nvalid = clamp_nvalid((int)cnt, lane_off, V);
    const i64 flat = row * geo.T + t_team + lane_off;
    LbRecord *myrec = rec + row * nchunks + c;

    typename Op::Raw raw;
    load_inputs<Op, VEC>(args, flat, nvalid, raw);
    Op op;
    op.prepare(args, raw);
    TC Xc = (TC)0;
    TS Xv = (TS)0;
    double s = 0.0, ss = 0.0;
    scan_chunk<Op, false>(
        op, sh[team], 0, nvalid, (unsigned)(t_team + lane_off), T, kWave, Xc, Xv,
        [&](const TC(&xs)[V], const bool(&row_end)[V], TC x_in, TS v_next) {
            emit_outputs<Op, VEC>(op, args, flat, nvalid, xs, row_end, x_in, v_next, s, ss);
        },
        [&](TeamShared<Op> &, int buf, TC &xc, TS &xv) {
            // every team's wave aggregates are in LDS (scan_chunk's barrier is block-wide)
            if (threadIdx.x < kWave) {  // wave 0 resolves the carries: 64 records per look-back round
                const int lane = threadIdx.x;
                Aff<TC> tagg[kLbTeams];  // team aggregates: x_first(team) = G0(G1(G2(G3(x_in))))
#pragma unroll
                for (int tm = 0; tm < kLbTeams; ++tm) {
                    Aff<TC> g{sh[tm].a[buf][kWaves - 1], sh[tm].b[buf][kWaves - 1], sh[tm].h[buf][kWaves - 1]};
#pragma unroll
                    for (int w = kWaves - 2; w >= 0; --w) g = compose(Aff<TC>{sh[tm].a[buf][w], sh[tm].b[buf][w], sh[tm].h[buf][w]}, g);
                    tagg[tm] = g;
                }
                Aff<TC> agg = tagg[kLbTeams - 1];
#pragma unroll
                for (int tm = kLbTeams - 2; tm >= 0; --tm) agg = compose(tagg[tm], agg);
                const TS my_v0 = Op::kNeedsNext ? sh[0].v0[buf][0] : (TS)0;
                TC carry = (TC)0;
                TS carry_v = (TS)0;
                bool ok = true;
                if (agg.h) {
                    // the block holds its row's end: inclusive value known at once
                    if (lane == 0) lb_store2(&myrec->x, lb_tag(lb_bits(agg.b), 2u), lb_tag(lb_bits(my_v0), 1u));
                } else {
                    if (lane == 0) {
                        lb_store2(&myrec->a, lb_tag(lb_bits(agg.a), 1u), lb_tag(lb_bits(agg.b), 1u));
                        if (Op::kNeedsNext) lb_store(&myrec->v0, lb_tag(lb_bits(my_v0), 1u));
                    }
                    // look back: lane l reads the record of chunk base + l; the records up to the first inclusive
                    // one are composed in chunk order by a wave reduction; no inclusive among the 64 -> next 64
                    Aff<TC> acc{(TC)1, (TC)0, 0};
                    const unsigned long long t_start = wall_clock64();
                    for (int base = c + 1; ok; base += kWave) {
                        const int idx = base + lane;
                        const bool valid = idx < nchunks;
                        const LbRecord *r = rec + row * nchunks + (valid ? idx : nchunks - 1);
                        const bool want_v0 = Op::kNeedsNext && base == c + 1 && lane == 0;
                        gather::u64x2 ab = {0ull, 0ull}, xv = {0ull, 0ull};
                        bool have_agg = false, have_incl = false, ready = false;
                        int first = kWave;
                        for (;;) {
                            if (valid && !ready) {  // only the records still missing are re-polled
                                lb_load_record(r, ab, xv);
                                have_agg = (ab.x & 3ull) == 1ull && (ab.y & 3ull) == 1ull;
                                have_incl = (xv.x & 3ull) == 2ull;
                                ready = (have_agg || have_incl) && (!want_v0 || (xv.y & 3ull) == 1ull);
                            }
                            const unsigned long long incl = __ballot(valid && ready && have_incl);
                            first = incl ? (int)__ffsll((long long)incl) - 1 : kWave;
                            if (__all(!(valid && lane <= first) || ready)) break;
                            if (wall_clock64() - t_start > 100000000ull) { ok = false; break; }
                        }
                        if (!ok) break;
                        if (first == kWave && !__all(valid)) { ok = false; break; }  // ran past the row's end: cannot happen
                        Aff<TC> el{(TC)1, (TC)0, 0};
                        if (valid && lane < first) el = Aff<TC>{lb_untag<TC>(ab.x), lb_untag<TC>(ab.y), 0};
                        else if (valid && lane == first) el = Aff<TC>{(TC)0, lb_untag<TC>(xv.x), 1};  // constant map
                        if (want_v0) carry_v = lb_untag<TS>(xv.y);
#pragma unroll
                        for (int off = 1; off < kWave; off <<= 1) {
                            const Aff<TC> p = shfl_down_aff(el, off);
                            if (lane + off < kWave) el = compose(el, p);
                        }
                        acc = compose(acc, Aff<TC>{__shfl(el.a, 0, kWave), __shfl(el.b, 0, kWave), __shfl(el.h, 0, kWave)});
                        if (first < kWave) break;
                    }
                    if (ok) carry = apply(acc, (TC)0);
                    else {
                        carry = (TC)__longlong_as_double(0x7ff8000000000000LL);
                        if (lane == 0) hdr->error = 1u;
                    }
                    if (lane == 0) lb_store(&myrec->x, lb_tag(lb_bits(apply(agg, carry)), 2u));
                }
                if (lane == 0) {
                    // carries into the teams, last team first: what follows team tm is team tm+1 ... then `carry`
                    TC x = carry;
#pragma unroll
                    for (int tm = kLbTeams - 1; tm >= 0; --tm) {
                        s_xc[tm] = x;
                        s_xv[tm] = (tm == kLbTeams - 1) ? carry_v : (Op::kNeedsNext ? sh[tm + 1 < kLbTeams ? tm + 1 : tm].v0[buf][0] : (TS)0);
                        x = apply(tagg[tm], x);
                    }
                }
            }
            __syncthreads();
            xc = s_xc[team];
            xv = s_xv[team];
        });
}

// ---- host side ----------------------------------------------------------------------------------
struct Plan {
    bool ok;
    bool long_rows, wave_indep, vec, interleaved, lookback;
    int rows_per_chunk;
    i64 nchunks;
    i64 nblocks;
};

constexpr i64 kMaxLongGrid = 256 * 8;  // LONG geometry: blocks grid-stride over rows
constexpr i64 kMaxShortGrid = 512;     // SHORT geometry: persistent blocks (<= 2 per CU)

i64 gcd_i64(i64 a, i64 b) { while (b) { i64 t = a % b; a = b; b = t; } return a; }

int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

// V: elements per lane; ptr_align_ok: every tensor base pointer is 16-byte aligned.
Plan make_plan(i64 O, i64 T, i64 I, int V, bool ptr_align_ok, int cus, size_t lb_bytes = 0, bool want_partials = false) {
    Plan p{};
    const i64 chunk = (i64)kTeam * V;
    if (I > 1) {
        // interleaved feature dims: whole outer rows must fit a chunk
        if (T * I > chunk || I > 1024) return p;
        i64 R = chunk / (T * I);
        const i64 q = V / gcd_i64(T * I, V);
        if (R >= q) R -= R % q;
        if (R > O) R = O;
        p.ok = true;
        p.interleaved = true;
        p.rows_per_chunk = (int)R;
        p.nchunks = (O + R - 1) / R;
        i64 nb = p.nchunks;
        if (nb > (i64)cus * 8) nb = (i64)cus * 8;
        if (nb > 4096) nb = 4096;  // up to 8 resident teams per CU (15.0 vs 16.8 us at C3 with 4), chunks walked grid-stride
        p.nblocks = nb;
        p.vec = ptr_align_ok && (p.nchunks == 1 || (R * T * I) % V == 0);
        p.wave_indep = ((i64)kWave * V) % T == 0;
        return p;
    }
    if (T >= (1LL << 31) - 4096) return p;  // not this family's shape
    p.ok = true;
    if (T <= chunk) {
        // R whole rows per chunk; prefer R*T multiple of V so every chunk start stays aligned
        i64 R = chunk / T;
        const i64 q = V / gcd_i64(T, V);
        if (R >= q) R -= R % q;
        if (R > O) R = O;
        p.rows_per_chunk = (int)R;
        p.nchunks = (O + R - 1) / R;
        i64 nb = (p.nchunks + kTeams - 1) / kTeams;
        if (nb > cus) nb = cus;  // one persistent block per CU; chunks beyond are walked grid-stride
        if (nb > kMaxShortGrid) nb = kMaxShortGrid;
        p.nblocks = nb;
        p.vec = ptr_align_ok && (p.nchunks == 1 || (R * T) % V == 0);
        p.wave_indep = ((i64)kWave * V) % T == 0;
        p.long_rows = false;
    } else {
        p.long_rows = true;
        p.nblocks = O < kMaxLongGrid ? O : kMaxLongGrid;
        p.vec = ptr_align_ok && (T % V == 0);
        p.wave_indep = false;
        // too few rows to fill the chip (measured crossover: ~0.9 * #CUs rows; the in-block chunk loop streams
        // ~13 GB/s per row): scan each row's chunks in parallel with decoupled look-back
        const i64 lb_chunk = chunk * kLbTeams;  // steps per look-back block
        const i64 nch = (T + lb_chunk - 1) / lb_chunk;
        if (!want_partials && 8 * O < 7 * (i64)cus && O <= kLbMaxRows && T >= 8 * chunk && O * nch <= (1LL << 22) &&
            lb_bytes >= sizeof(LbHeader) + (size_t)(O * nch) * sizeof(LbRecord)) {
            p.lookback = true;
            p.nchunks = nch;
            p.nblocks = O * nch;
        }
    }
    return p;
}

template <class Op>
LaunchResult launch_scan(hipStream_t s, const typename Op::Args &a, const Plan &p, i64 O, i64 T, i64 I, double *partials,
                         const char *name, void *lb_ws = nullptr) {
    ScanGeom geo{O, T, p.rows_per_chunk, scan_lanes_for(T, Op::V, p.wave_indep), p.nchunks};
    const dim3 grid((unsigned)p.nblocks);
    // I > 1 (whole outer rows inside one chunk): the forward estimators run on the kernels of gae_interleaved.h; the
    // shifted layout and the adjoints have I == 1 kernels only
    if (p.interleaved) return LaunchResult{2, 0, nullptr};
    if (p.lookback) {
        LbHeader *hdr = (LbHeader *)lb_ws;
        LbRecord *rec = (LbRecord *)((char *)lb_ws + sizeof(LbHeader));
        if (hipMemsetAsync(lb_ws, 0, sizeof(LbHeader) + (size_t)p.nblocks * sizeof(LbRecord), s) != hipSuccess)
            return LaunchResult{4, 0, nullptr};
        if (p.vec) hipLaunchKernelGGL((scan_lookback_kernel<Op, true>), grid, dim3(kLbTeams * kTeam), 0, s, a, geo, hdr, rec);
        else hipLaunchKernelGGL((scan_lookback_kernel<Op, false>), grid, dim3(kLbTeams * kTeam), 0, s, a, geo, hdr, rec);
    } else if (p.long_rows) {
        const bool dual = OpDualScan<Op>::value && T >= 4 * (i64)kLongThreads * Op::V;
        if constexpr (OpDualScan<Op>::value) {
            if (dual) {
                if (p.vec) hipLaunchKernelGGL((scan_long_kernel<Op, true, true>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
                else hipLaunchKernelGGL((scan_long_kernel<Op, false, true>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
            }
        }
        if (!dual) {
            if (p.vec) hipLaunchKernelGGL((scan_long_kernel<Op, true, false>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
            else hipLaunchKernelGGL((scan_long_kernel<Op, false, false>), grid, dim3(kLongThreads), 0, s, a, geo, partials);
        }
    } else {
#define TRL_LAUNCH(VEC, WI) hipLaunchKernelGGL((scan_short_kernel<Op, VEC, WI>), grid, dim3(kShortThreads), 0, s, a, geo, partials)
        if (p.wave_indep) { if (p.vec) TRL_LAUNCH(true, true); else TRL_LAUNCH(false, true); }
        else              { if (p.vec) TRL_LAUNCH(true, false); else TRL_LAUNCH(false, false); }
#undef TRL_LAUNCH
    }
    return LaunchResult{0, p.nblocks, name};
}

inline bool aligned16(const void *p) { return p == nullptr || (((uintptr_t)p) & 15) == 0; }

template <template <typename, typename, int> class OpT, typename Args, typename F>
LaunchResult dispatch_scan(int vdt, int cdt, F &&f) {
    // V = 16 bytes of the storage dtype per lane
    if (vdt == 0 && cdt == 0) return f(OpT<float, float, 4>{});
    // (fp64 with V = 4, two 16-byte loads per tensor and lane, was measured at C5: 0 - 5 % slower than V = 2 on every kernel)
    if (vdt == 1 && cdt == 1) return f(OpT<double, double, 2>{});
    if (vdt == 1 && cdt == 0) return f(OpT<double, float, 2>{});
    if (vdt == 0 && cdt == 1) return f(OpT<float, double, 4>{});
    return LaunchResult{1, 0, nullptr};
}

}  // namespace
