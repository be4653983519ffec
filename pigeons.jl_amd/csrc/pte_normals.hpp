// pte_normals.hpp -- one wave fills a replica's state with d i.i.d. normals / sd in the REFERENCE'S sequential draw order
// (sample_iid! of the scaled-precision MVN path, src/targets/toy_mvn_target.jl:15-21; create_replicas' initialization,
// :10-11; ToyExplorer.jl:7-12) and returns the fixed-tree sum of squares.  Bit-identical to wave_randn_block + x / sd
// (pte_device.hpp), organised for HBM-rate output (round 3).
//
// The stream is counter based (draw k = mix64(seed + k gamma)), but output i does not sit at stream position i: 1.2 % of the
// ziggurat draws leave the fast path and consume extra draws (wedge test: one uniform, then either the same value or a fresh
// attempt; tail: a loop of pairs).  What the old kernel did per 64 draws -- resolve the first such event sequentially, re-draw
// the lanes behind it, repeat (54 % of the blocks) -- is amortised here over a CHUNK of 512 outputs:
//   1. every lane evaluates 9 stream positions (576 = 512 + 64 slack): value as if the fast path applied (the wedge's accepted
//      value is the same expression), stored to LDS by POSITION; the non-fast positions are compacted into an event list;
//   2. ONE divergent pass: event lane e re-derives its draw and the next one and runs the wedge test (exp); tails are rare
//      (3e-4 of the draws) and resolved by the exact sequential code;
//   3. a scalar walk over the (sorted) events yields the map output -> position: position = output + shift, the shift grows by
//      1 at an accepted wedge (its uniform), by 2 at a rejected one (draw + uniform, the output restarts), by the tail's draws;
//   4. outputs are gathered from LDS at output + shift, FOUR CONSECUTIVE OUTPUTS PER LANE (32-byte stores), divided by sd,
//      squared and summed: the first two levels of the fixed tree are in-lane adds, four DPP steps finish four 64-leaf blocks
//      at once (the old layout, one leaf per lane, paid six DPP steps per block: 4.2 SIMD cycles per v_mov_b32_dpp).
// x / sd is q = x r, q' = fma(fma(-q, sd, x), r, q) with r = RN(1 / sd): correctly rounded (Markstein 1990, Thm: y = RN(1/b),
// q = RN(a y), then RN(q + RN(a - b q) y) = RN(a / b) unless b's significand is all ones -- that case takes the IEEE division;
// 4e8 random (x, sd) pairs against x / sd on the host: 0 mismatches; tests/test_gpu_normals.py holds the device to x / sd).
#pragma once
#include "pte_device.hpp"

namespace pte {

constexpr int NRM_CO = 512;                 // outputs per chunk (two groups of 256)
constexpr int NRM_CP = NRM_CO + 64;         // stream positions evaluated per chunk
constexpr int NRM_SLOTS = NRM_CP / 64;      // positions per lane
constexpr int NRM_MAX_EV = 64;              // events resolved lane-parallel per chunk (more: the chunk is cut short)

#ifndef NRM_PAD
#define NRM_PAD 0
#endif
#ifndef NRM_UNROLL
#define NRM_UNROLL 9
#endif
// val[] is read back four consecutive outputs per lane (a stride of 4 doubles across the lanes: 4-way bank conflicts on a dense
// array).  NRM_PAD = 1 puts element i at i + (i >> 5), which spreads the strided reads over all banks -- measured: no gain (the
// kernel is not bound by the LDS), so the dense layout stays.
__device__ __forceinline__ constexpr int nrm_pad(int i) { return NRM_PAD ? i + (i >> 5) : i; }
struct NormalsLds {                         // one per wave (= per workgroup)
    double wi[256];
    unsigned long long ki[256];             // (directly behind wi: one ds_read2st64_b64 fetches both)
    double val[NRM_CP + NRM_CP / 32 + 2];
    int ev[NRM_MAX_EV];
    double bs[64];
};

__device__ __forceinline__ void normals_lds_init(NormalsLds &L, int lane) {
    for (int i = lane; i < 256; i += 64) { L.wi[i] = ZIG_WI[i]; L.ki[i] = ZIG_KI[i]; }
    __syncthreads();
}

__device__ __forceinline__ double dpp_row_step(double v, int which) {
    // the four in-row levels of wave_sum_dpp (pte_device.hpp): lanes l ^ 1, l ^ 2, the other group of 4, the other group of 8
    switch (which) {
    case 0: return dpp_add_step<0xB1, 0xF>(v);
    case 1: return dpp_add_step<0x4E, 0xF>(v);
    case 2: return dpp_add_step<0x141, 0xF>(v);
    default: return dpp_add_step<0x140, 0xF>(v);
    }
}

// Fills xrow[0 .. d) and returns lane b = sum of squares of block b (64 leaves), to be fed to upper_tree_root<NLU>.
// `r` is advanced exactly as d sequential randn(rng) calls would advance it.  One wave per workgroup.
__device__ __forceinline__ double normals_row(NormalsLds &L, SeqRng &r, double *xrow, int64_t d, double sd, int lane) {
    const double rinv = 1.0 / sd;
    const bool markstein = (__double_as_longlong(sd) & 0x000fffffffffffffLL) != 0x000fffffffffffffLL;      // uniform
    const uint64_t gamma = r.gamma;
    const uint64_t g64 = gamma << 6;
    int64_t done = 0;                                       // outputs written so far (a multiple of 256 until the last group)
    while (done < d) {
        // ---- 1. positions base + 1 .. base + NRM_CP of the stream
        const uint64_t base = r.seed;
        uint64_t zc = base + (uint64_t)(lane + 1) * gamma;
        int n_ev = 0;                                       // uniform
        __builtin_amdgcn_wave_barrier();
#pragma unroll NRM_UNROLL
        for (int j = 0; j < NRM_SLOTS; ++j) {
            const uint64_t raw = mix64(zc) & MASK52;
            zc += g64;
            const uint64_t rabs = raw >> 1;
            const int idx = (int)(rabs & 0xFF);
                        const double w = L.wi[idx];
            const unsigned long long k = L.ki[idx];
            // (double)(u & 1 ? -rabs : rabs) * wi[idx]: rabs < 2^51 goes exactly into the significand of 2^52 + rabs; the sign is
            // applied to the product (round-to-nearest is symmetric).  rabs = 0 with the sign bit set would give -0.0 where the
            // reference has +0.0: the division step below returns +0.0 for it (fma(+0.0, r, -0.0) = +0.0).
            const double mag = __longlong_as_double((long long)(rabs | 0x4330000000000000ULL)) - 4503599627370496.0;
            const double prod = mag * w;
            const double v = __longlong_as_double(__double_as_longlong(prod) ^ (long long)(raw << 63));
            L.val[nrm_pad(64 * j) + nrm_pad(lane)] = v;            // = nrm_pad(64 j + lane)
            const bool slow = !(rabs < k);
            const uint64_t m = ballot64(slow);
            if (m) {                                        // uniform branch, 54 % of the slots
                const int at = n_ev + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                if (slow && at < NRM_MAX_EV) L.ev[at] = 64 * j + lane;
                n_ev += __popcll(m);
            }
        }
        int pos_limit = NRM_CP;                             // positions below this one are resolved (events beyond the list are not)
        if (n_ev > NRM_MAX_EV) { n_ev = NRM_MAX_EV; }
        __builtin_amdgcn_wave_barrier();
        if (n_ev == NRM_MAX_EV) pos_limit = L.ev[NRM_MAX_EV - 1] + 1;  // (essentially never: 7 events expected; later positions may hide unlisted events)
        // ---- 2. one divergent pass over the events: lane e < n_ev resolves event e
        int e_pos = 0x7fffffff, e_kind = 0;                 // kind: 1 wedge accepted, 2 wedge rejected, 3 tail
        if (lane < n_ev) {
            e_pos = L.ev[lane];
            const uint64_t z = base + (uint64_t)(e_pos + 1) * gamma;
            const uint64_t raw = mix64(z) & MASK52;
            const int64_t rabs = (int64_t)(raw >> 1);
            const int idx = (int)(rabs & 0xFF);
            if (idx == 0) {
                e_kind = 3;
            } else {
                                const double x = (double)((raw & 1) ? -rabs : rabs) * ZIG_WI[idx];
                const double u1 = u52_to_unit(mix64(z + gamma));
                                const double f1 = ZIG_FI[idx - 1], f0 = ZIG_FI[idx];
                e_kind = ((f1 - f0) * u1 + f0 < exp(-0.5 * x * x)) ? 1 : 2;
            }
        }
        // ---- 3 + 4. groups of 256 outputs; the scalar walk over the events runs alongside
        int shift = 0, ev = 0, consumed_end = -1;           // uniform: positions <= consumed_end are extra draws of an earlier event
        int emitted = 0, shift_emitted = 0;
        const int want = (int)min((int64_t)NRM_CO, d - done);
        for (int g0 = 0; g0 < want; g0 += 256) {
            const int gend = min(g0 + 256, want);
            int k0 = g0 + 4 * lane;                         // this lane's outputs k0 .. k0 + 3 (relative to the chunk)
            int s0 = shift, s1 = shift, s2 = shift, s3 = shift;
            bool cut = false;
            while (ev < n_ev) {
                const int p = __builtin_amdgcn_readlane(e_pos, ev);
                if (p <= consumed_end) { ev += 1; continue; }                   // not the start of an attempt
                const int kout = p - shift;                 // the output this attempt belongs to
                if (kout >= gend) break;
                int kind = __builtin_amdgcn_readlane(e_kind, ev);
                int delta, thr;
                if (kind == 3) {
                    // tail of the normal ziggurat (Random/src/normal.jl randn_unlikely, idx == 0): exact sequential code at this position
                    SeqRng t{base + (uint64_t)(p + 1) * gamma, gamma};
                    const uint64_t raw = mix64(t.seed) & MASK52;
                    const int64_t rabs = (int64_t)(raw >> 1);
                    int pairs = 0;
                    double xx, yy;
                    do {
                        xx = ZIG_NOR_INV_R * zig_tail_neglog(t.rand());
                        yy = zig_tail_neglog(t.rand());
                        pairs += 1;
                    } while (!(yy + yy > xx * xx) && pairs < 4096);
                    const double tv = ((rabs >> 8) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
                    if (p + 2 * pairs >= NRM_CP) { cut = true; break; }          // (its draws leave the chunk: redo from here next chunk)
                    if (lane == 0) L.val[nrm_pad(p)] = tv;
                    __builtin_amdgcn_wave_barrier();
                    delta = 2 * pairs; thr = kout + 1;
                } else if (kind == 1) { delta = 1; thr = kout + 1; }
                else { delta = 2; thr = kout; }
                consumed_end = p + ((kind == 2) ? 1 : delta);
                s0 += (k0 >= thr) ? delta : 0;
                s1 += (k0 + 1 >= thr) ? delta : 0;
                s2 += (k0 + 2 >= thr) ? delta : 0;
                s3 += (k0 + 3 >= thr) ? delta : 0;
                shift += delta;
                ev += 1;
            }
            // every position this group reads must be resolved and inside the chunk
            if (cut || gend - 1 + shift >= pos_limit) break;
            const double a0 = L.val[nrm_pad(k0 + s0)], a1 = L.val[nrm_pad(k0 + 1 + s1)], a2 = L.val[nrm_pad(k0 + 2 + s2)], a3 = L.val[nrm_pad(k0 + 3 + s3)];
            double q0, q1, q2, q3;
            if (markstein) {
                q0 = a0 * rinv; q1 = a1 * rinv; q2 = a2 * rinv; q3 = a3 * rinv;
                q0 = __builtin_fma(__builtin_fma(-q0, sd, a0), rinv, q0);
                q1 = __builtin_fma(__builtin_fma(-q1, sd, a1), rinv, q1);
                q2 = __builtin_fma(__builtin_fma(-q2, sd, a2), rinv, q2);
                q3 = __builtin_fma(__builtin_fma(-q3, sd, a3), rinv, q3);
            } else {
                q0 = a0 / sd + 0.0; q1 = a1 / sd + 0.0; q2 = a2 / sd + 0.0; q3 = a3 / sd + 0.0;
            }
            const int64_t o = done + k0;
            const int nv = gend - k0;                       // valid outputs of this lane: >= 4 except in the last group
#ifdef NRM_NO_STORE                          // measurement builds only: what the arithmetic alone costs
            if (q0 == 1.2345e300) xrow[o] = q1 + q2 + q3;
#else
            if (nv >= 4) {
                double2 *dst = reinterpret_cast<double2 *>(xrow + o);
                dst[0] = make_double2(q0, q1); dst[1] = make_double2(q2, q3);
            } else {
                if (nv > 0) xrow[o] = q0; else q0 = 0.0;
                if (nv > 1) xrow[o + 1] = q1; else q1 = 0.0;
                if (nv > 2) xrow[o + 2] = q2; else q2 = 0.0;
                q3 = 0.0;
            }
#endif
            // fixed tree over 64-leaf blocks: two levels in the lane, four across the 16 lanes of a row (= one block)
            double s = (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
            s = dpp_row_step(s, 0); s = dpp_row_step(s, 1); s = dpp_row_step(s, 2); s = dpp_row_step(s, 3);
            if ((lane & 15) == 0) L.bs[(int)((done + g0) >> 6) + (lane >> 4)] = s;
            emitted = gend; shift_emitted = shift;          // (every event of an output below gend has been walked, none of a later one)
        }
        if (emitted == 0) {
            // Unreachable for the slack chosen unless a tail's draws leave the chunk right at its start or > 64 events crowd the first
            // group: produce this group with the plain block-by-block procedure (wave_randn_block resolves events one at a time).
            const int gl = (int)min((int64_t)256, d - done);
            for (int b0 = 0; b0 < gl; b0 += 64) {
                const int nl = min(64, gl - b0);
                double v = wave_randn_block(r, lane, nl) / sd + 0.0;
                if (lane < nl) xrow[done + b0 + lane] = v; else v = 0.0;
                const double sq = wave_sum_dpp(v * v);
                if (lane == 0) L.bs[(int)((done + b0) >> 6)] = sq;
            }
            done += gl;
            continue;
        }
        r.seed = base + (uint64_t)(emitted + shift_emitted) * gamma;     // stream position of the next output's first draw
        done += emitted;
    }
    __builtin_amdgcn_wave_barrier();
    const int B = (int)((d + 63) >> 6);
    return (lane < B) ? L.bs[lane] : 0.0;
}

}  // namespace pte
