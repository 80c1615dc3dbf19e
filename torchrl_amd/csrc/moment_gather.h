// In-launch all-gather of per-block fp64 moments {sum, sumsq} across a persistent, co-resident grid -- the step that lets
// a kernel normalise values it still holds in registers (fused_kernels.hip: GAE + normalise; norm_kernels.hip:
// standardize in one launch).  No fences, no atomics, no counters:
//   * a block publishes its pair as ONE 16-byte granule {sum | tag, sumsq | tag} (8-bit epoch tag in the low mantissa
//     bits of each half: 45 mantissa bits remain, relative 6e-14 per block partial) with one write-through 16-byte store
//     per replica into a PACKED array (16 bytes per block);
//   * one wave per block sweeps its replica -- 4 KB for 256 blocks, four coalesced 16-byte sc1 loads per lane and round
//     -- until every tag matches, and sums the granules in a fixed order (bitwise reproducible);
//   * every spin is bounded (1 s wall clock); a wave that gives up re-tags what it published with the PREVIOUS tag so
//     that nothing of it can match the next launch's tag, and reports failure to its caller.
// Measured at C4 (tools/kbench/kfused3.hip, kxchg.hip): 1 / 4 / 8 / 16 / 32 replicas 7.81 / 7.48 / 7.45 / 7.49 / 7.58 us per
// launch; one 128-byte line per block polled with 8-byte loads (round 2) 7.95; line strides of 256 B - 16 KB no better;
// two-level gathers (16 x 16, 8 x 32 groups) 7.5 - 7.7; two poll rounds in flight: slower.
// Rule for the callers: EVERY slot of the workspace is rewritten by EVERY launch (by the block that owns it, or -- grids
// smaller than the layout's stride -- by refresh_absent_slots): a stale tag is then at most one launch old and never matches.
#pragma once
#include "common.h"

namespace gather {

constexpr int kMaxBlocks = 512;  // granules per replica array a caller may lay out
constexpr int kHeaderReplicas = 8;  // replicas of the header shared by the FUSED GAE kernel and the ungrouped standardize kernel

// {epoch, error} header shared by the kernels that gather; the slot arrays follow at byte 256
struct Header {
    unsigned epoch;
    unsigned error;
    unsigned pad[62];
    unsigned long long slots[1];
};

typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

// one write-through 16-byte store (the s_nop: the compiler does not know the asm is a > 64-bit store and would otherwise
// reuse the data VGPRs inside the store-data hazard window)
__device__ __forceinline__ void store_granule(unsigned long long *p, unsigned long long a, unsigned long long b) {
    u64x2 v = {a, b};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// four agent-scope 16-byte loads issued together, one wait (sc1: served from the memory side, never a stale L2 line)
__device__ __forceinline__ void load_granules4(const unsigned long long *p0, const unsigned long long *p1, const unsigned long long *p2,
                                               const unsigned long long *p3, u64x2 &v0, u64x2 &v1, u64x2 &v2, u64x2 &v3) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
__device__ __forceinline__ bool granule_ok(const u64x2 &v, unsigned tag) {
    return ((unsigned)(v.x & 0xffull) == tag) && ((unsigned)(v.y & 0xffull) == tag);
}
__device__ __forceinline__ unsigned long long tag_bits(double x, unsigned tag) {
    return ((unsigned long long)__double_as_longlong(x) & ~0xffull) | (unsigned long long)tag;
}
__device__ __forceinline__ double untag(unsigned long long b) { return __longlong_as_double((long long)(b & ~0xffull)); }

// Called by ONE whole wave of every block of the grid.  `slots`: nrep packed arrays of `stride` granules each (nrep <= 64),
// of which this launch uses the first nblocks (stride = the largest grid any launch on this workspace uses: the layout
// does not depend on the grid).  Publishes {bs, bss} as block `block`'s granule in every replica, sweeps replica
// block % nrep until all nblocks tags match and returns the sums over the blocks (wave-uniform, fixed order); false if it
// gave up after 1 s.
__device__ __forceinline__ bool publish_and_gather(unsigned long long *slots, int nrep, int stride, int nblocks, int block, unsigned tag,
                                                   int lane, double bs, double bss, double &tot, double &tot2) {
    __builtin_amdgcn_s_setprio(3);  // the publish and the polls go ahead of the other waves' traffic (8.25 -> 8.18 us at C4)
    if (lane < nrep) store_granule(slots + ((size_t)lane * stride + block) * 2, tag_bits(bs, tag), tag_bits(bss, tag));
    // lane l gathers blocks l, l + 64, ... of its replica: every round re-reads the whole array (re-polling only the missing
    // granules saves nothing on coalesced loads)
    const unsigned long long *rep = slots + (size_t)(block % nrep) * stride * 2;
    const int i0 = lane, i1 = lane + 64, i2 = lane + 128, i3 = lane + 192;
    auto at = [&](int i) { return rep + (size_t)(i < nblocks ? i : nblocks - 1) * 2; };  // lanes past the grid re-read the last granule
    u64x2 v[8] = {};
    const unsigned long long t_start = wall_clock64();
    bool ok;
    for (;;) {
        load_granules4(at(i0), at(i1), at(i2), at(i3), v[0], v[1], v[2], v[3]);
        ok = granule_ok(v[0], tag) && granule_ok(v[1], tag) && granule_ok(v[2], tag) && granule_ok(v[3], tag);
        if (nblocks > 4 * kWave) {  // grids of 257 - 512 blocks (kMaxBlocks)
            load_granules4(at(i0 + 256), at(i1 + 256), at(i2 + 256), at(i3 + 256), v[4], v[5], v[6], v[7]);
            ok = ok && granule_ok(v[4], tag) && granule_ok(v[5], tag) && granule_ok(v[6], tag) && granule_ok(v[7], tag);
        }
        if (__all(ok)) break;
        if (wall_clock64() - t_start > 100000000ull) break;  // 1 s at 100 MHz: give up loudly
        // no sleep between rounds (sleep 1 / 4: +0.03 / +0.1 us); a second round in flight: +0.1 us (kxchg.hip)
    }
    __builtin_amdgcn_s_setprio(0);
    const bool all_ok = __all(ok);
    if (!all_ok && lane < nrep)  // nothing this block published may match the NEXT launch's tag
        store_granule(slots + ((size_t)lane * stride + block) * 2, tag_bits(bs, (tag - 1u) & 0xffu), tag_bits(bss, (tag - 1u) & 0xffu));
    double acc = 0.0, acq = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (lane + kWave * j < nblocks) { acc += untag(v[j].x); acq += untag(v[j].y); }
    tot = wave_sum(acc);
    tot2 = wave_sum(acq);
    return all_ok;
}

// A launch whose grid is smaller than `stride` keeps the slots of the absent blocks [nblocks, stride) one launch fresh (the
// rule above) -- with the PREVIOUS tag: nobody reads them in this launch and they can never match the next one, whatever tag
// this block computed.  Called by one whole wave of every block that is not the gathering wave (off the critical path).
__device__ __forceinline__ void refresh_absent_slots(unsigned long long *slots, int nrep, int stride, int nblocks, int block, unsigned tag,
                                                     int lane) {
    const int absent = (stride - nblocks) * nrep;                  // (replica, block) pairs nobody writes in this launch
    const int per = (absent + nblocks - 1) / nblocks;              // ... dealt evenly to the blocks that exist
    const unsigned old = (tag - 1u) & 0xffu;
    for (int k = lane; k < per; k += kWave) {
        const int idx = block * per + k;
        if (idx < absent) {
            const int rep = idx / (stride - nblocks), b = nblocks + idx % (stride - nblocks);
            store_granule(slots + ((size_t)rep * stride + b) * 2, tag_bits(0.0, old), tag_bits(0.0, old));
        }
    }
}

}  // namespace gather
