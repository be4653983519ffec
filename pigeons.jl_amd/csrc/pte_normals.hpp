// pte_normals.hpp -- one wave fills a replica's state with d i.i.d. normals / sd in the REFERENCE'S sequential draw order
// (sample_iid! of the scaled-precision MVN path, src/targets/toy_mvn_target.jl:15-21; create_replicas' initialization,
// :10-11; ToyExplorer.jl:7-12) and returns the fixed-tree sum of squares.  Bit-identical to wave_randn_block + x / sd
// (pte_device.hpp), organised for HBM-rate output (round 3).
//
// The stream is counter based (draw k = mix64(seed + k gamma)), but output i does not sit at stream position i: 1.5 % of the
// ziggurat draws leave the fast path and consume extra draws (wedge test: one uniform, then either the same value or a fresh
// attempt; tail: a loop of pairs).  What the old kernel did per 64 draws -- resolve the first such event sequentially, re-draw
// the lanes behind it, repeat (54 % of the blocks) -- becomes a STREAM COMPACTION over a chunk of 512 outputs:
//   1. every lane evaluates 9 stream positions (576 = 512 + 64 slack) and writes each value to LDS at its position (NRM_POS_LDS; the
//      first version kept them in registers: 143 VGPRs, one wave per SIMD less): the value as if the
//      fast path applied (the wedge's accepted value is the same expression); the non-fast positions go to an event list;
//   2. ONE divergent pass: event lane e re-derives its draw and the next one and runs the wedge test (exp), or the tail's loop;
//   3. lane-parallel over the (sorted) events: which of them start an attempt (an event whose position was consumed by the live
//      event before it does not), a prefix sum of the consumed draws, and a bitmap of the CONSUMED positions (an accepted wedge
//      consumes the uniform behind it; a rejected one its own draw and the uniform; a tail its trials);
//   4. every position that was not consumed is an output: its index is a running count (ballot prefix, v_mbcnt) -- the values
//      are moved to OUTPUT order IN PLACE (output k never lies above position k; a slot's 64 values are read before its outputs are written); a tail writes its value over the slot of its position;
//   5. outputs are read back FOUR CONSECUTIVE PER LANE (two 16-byte LDS reads, 32-byte stores), divided by sd, squared and
//      summed: the first two levels of the fixed tree are in-lane adds, four DPP steps finish four 64-leaf blocks at once (the
//      old layout, one leaf per lane, paid six DPP steps per block at 4.2 SIMD cycles per v_mov_b32_dpp).
// No loop over events anywhere, no scalar hop per event: a lone wave went from 9,900 cycles per chunk (scalar walk over the
// events: 4,900 of them) to the issue time of its ~650 VALU instructions.
// x / sd is q = x r, q' = fma(fma(-q, sd, x), r, q) with r = RN(1 / sd): correctly rounded (Markstein 1990, Thm: y = RN(1/b),
// q = RN(a y), then RN(q + RN(a - b q) y) = RN(a / b) unless b's significand is all ones -- that case takes the IEEE division;
// 4e8 random (x, sd) pairs against x / sd on the host: 0 mismatches; tests/test_gpu_normals.py holds the device to x / sd).
#pragma once
#include "pte_device.hpp"

namespace pte {

#ifndef NRM_CHUNK
#define NRM_CHUNK 512                       // measured at 1024 (round 3): LDS then allows 3 waves per SIMD -- the same within 2 %
#endif
constexpr int NRM_CO = NRM_CHUNK;           // outputs per chunk (groups of 256)
#ifndef NRM_SLACK
#define NRM_SLACK 64                        // (measurement builds: 0 with -DNRM_MEASURE_IGNORE_LIMIT = what a chunk without the slack slot would cost; wrong samples)
#endif
constexpr int NRM_CP = NRM_CO + NRM_SLACK;  // stream positions evaluated per chunk
constexpr int NRM_SLOTS = NRM_CP / 64;      // positions per lane
#ifndef NRM_MAX_EV_
#define NRM_MAX_EV_ 32                       // (test builds force the cut-short / fallback paths with a small value)
#endif
constexpr int NRM_MAX_EV = NRM_MAX_EV_;              // events resolved lane-parallel per chunk, two lanes each (more: the chunk is cut short)
constexpr int NRM_EV_CAP = 128;             // entries of the event list (the list itself never overflows: the index is clamped)

#ifndef NRM_OCC
#define NRM_OCC 5                           // > 0: waves per SIMD the register allocation must allow (4 / 5 waves = 2.83 / 2.95 TB/s at N = 32768 in round 3; 6 does not fit the LDS)
#endif
#if NRM_OCC > 0
#define NRM_ATTR __attribute__((amdgpu_waves_per_eu(NRM_OCC, NRM_OCC)))
#else
#define NRM_ATTR
#endif
#ifndef NRM_PRIO
#define NRM_PRIO 4                          // > 0: wave priority by remaining work, in NRM_PRIO steps over the row (4: quarters; 0: off).  Measured, bit-identical
                                            // (tools/ubench/normals_dev.hip): N = 8192, d = 4096 83.4 -> 81.0 us; N = 4096 50.2 -> 46.4; N = 32768 283 -> 288 (8 steps: 81.9 / 48.0)
#endif
// Round 5: the kernel is bound by a wave's DEPENDENCY CHAIN as much as by issue -- one row alone on its SIMD takes 32 us, each further wave of the
// SIMD adds 5 (tools/ubench/normals_dev.hip, N = 1024 ... 5120 at d = 4096), and five waves per SIMD is all the LDS allows -- so latency taken out of
// a chunk shows at every occupancy.  Both switches are bit-identical (checksum of the harness) and measured together with NRM_PRIO:
// lone wave 31.7 -> 29.2 us per row, N = 8192 83.4 -> 79.4 us (3.21 -> 3.38 TB/s); per chunk of a lone wave: positions 2,365 -> 1,912 cycles,
// events' bookkeeping + compaction 1,810 -> 1,416 (tools/ubench/normals_prof.hip).
#ifndef NRM_COMPACT_ILP
#define NRM_COMPACT_ILP 1                   // the compaction requests all the slots' values in one LDS round trip
#endif
#ifndef NRM_PIPE_POS
#define NRM_PIPE_POS 1                      // slot j + 1's SplitMix64 evaluation and table read in flight while slot j is converted, stored and listed
#endif
#ifndef NRM_WPB
#define NRM_WPB 4                           // waves (replicas) per workgroup; they share the ziggurat tables (10 KB)
#endif
// Table entry of the fast path, indexed by the low NINE bits of the draw (bit 0 = sign, bits 1..8 = ziggurat layer), one ds_read_b128:
//   w = +-wi[layer] / 2  -- the draw's 52 bits with bit 0 cleared are 2 rabs, and (2 rabs)(w / 2) is the same product as rabs w bit for
//                           bit (scaling by 2 is exact); the sign rides on the table instead of being xor-ed into the product;
//   k = 2 ki[layer] | 0x433 << 52  -- "rabs < ki" as ONE unsigned compare of the bit pattern of 2^52 + 2 rabs, which the conversion
//                           builds anyway (same register pair: no copy).
// Per 64 positions this drops a 64-bit shift, the 52-bit mask, a register move and the two sign instructions (29 -> 25 VALU).
static_assert(NRM_EV_CAP > NRM_MAX_EV, "the clamped index must lie behind the events the pass reads");
struct alignas(16) NrmWK { double w; unsigned long long k; };
struct NormalsLds {                         // one per workgroup
    NrmWK wk[512];
    double fi[256];                         // wedge test of the event pass
    alignas(16) double out[NRM_WPB][NRM_CP];   // the chunk's values by stream position, then compacted in place to OUTPUT order
    unsigned short ev[NRM_WPB][NRM_EV_CAP]; // positions that left the fast path, in stream order
    double bs[NRM_WPB][64];
#ifdef NRM_PROF                             // tools/ubench/normals_prof.hip only: cycles per phase
    unsigned long long prof[8];
#endif
};

__device__ __forceinline__ void normals_lds_init(NormalsLds &L, int lane) {
#ifdef NRM_MEASURE_NO_INIT          // measurement builds only (wrong samples): what the table build costs a launch
    if (L.bs[0][0] != 1.2345) return;
#endif
    for (int i = NRM_WPB > 1 ? (int)threadIdx.x : lane; i < 512; i += 64 * NRM_WPB) {
        const double w = 0.5 * ZIG_WI[i >> 1];
        L.wk[i].w = (i & 1) ? -w : w;
        L.wk[i].k = (ZIG_KI[i >> 1] << 1) | 0x4330000000000000ull;
        if (i < 256) L.fi[i] = ZIG_FI[i];
    }
    __syncthreads();
}

// inclusive prefix sum over the 64 lanes (Kogge-Stone inside the rows of 16 by row_shr 1, 2, 4, 8; then the row totals)
__device__ __forceinline__ int wave_iscan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);     // row_bcast15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);     // row_bcast31 -> rows 2, 3
    return v;
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
// `r` is advanced exactly as d sequential randn(rng) calls would advance it.  One wave per replica.
__device__ __forceinline__ double normals_row(NormalsLds &L, SeqRng &r, double *xrow, int64_t d, double sd, int lane) {
    const int wv = NRM_WPB > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // wave-uniform (kept in an SGPR)
    const double rinv = 1.0 / sd;
    const bool markstein = (__double_as_longlong(sd) & 0x000fffffffffffffLL) != 0x000fffffffffffffLL;      // uniform
    const uint64_t gamma = r.gamma;
    const uint64_t g64 = gamma << 6;
    const double NRM_DEAD = __longlong_as_double(0x7ff8dead00000000LL);     // a consumed stream position (no fast-path value is a NaN)
    const bool tail_log1p = (g_rng_policy & PTE_RNG_TAIL_LOG1P) != 0;       // (read once: zig_tail_neglog reloads the policy word at every call)
    double *const out = L.out[wv];
    int64_t done = 0;                                       // outputs written so far (a multiple of 256 until the last group)
    while (done < d) {
#if NRM_PRIO
        // The issue arbiter favours the oldest wave of a SIMD, so the 4-5 waves that share one finish one after the other and the last of
        // them runs its final chunks ALONE, at a lone wave's efficiency (stamps of tools/ubench/normals_dev.hip -DNRM_STAMP at N = 4096,
        // d = 4096, every wave started within 1 us: first exit 24 us, median 33, last 45 -- for equal work).  Priority by REMAINING work
        // (s_setprio 3 in the first quarter of the row ... 0 in the last) lets the laggards catch up: the waves of a SIMD end together
        // and the SIMD stays saturated to the end.
        { const int lvl = (int)((done * NRM_PRIO) / d);
          switch (lvl) { case 0: __builtin_amdgcn_s_setprio(3); break; case 1: __builtin_amdgcn_s_setprio(2); break; case 2: __builtin_amdgcn_s_setprio(1); break; default: __builtin_amdgcn_s_setprio(0); break; } }
#endif
        // ---- 1. positions base + 1 .. base + NRM_CP of the stream
#ifdef NRM_PROF
        const unsigned long long pt0 = __builtin_readcyclecounter();
#endif
        const uint64_t base = r.seed;
#ifdef NRM_MEASURE_EXTRA_SALU      // measurement builds only: what N more scalar / vector instructions per chunk cost at full occupancy
        { int t_ = (int)base; for (int i_ = 0; i_ < NRM_MEASURE_EXTRA_SALU; ++i_) asm volatile("s_add_u32 %0, %0, 1" : "+s"(t_)); asm volatile("" :: "s"(t_)); }
#endif
#ifdef NRM_MEASURE_EXTRA_VALU
        { int t_ = lane; for (int i_ = 0; i_ < NRM_MEASURE_EXTRA_VALU; ++i_) asm volatile("v_add_u32 %0, 1, %0" : "+v"(t_)); asm volatile("" :: "v"(t_)); }
#endif
        uint64_t zc = base + (uint64_t)(lane + 1) * gamma;
        int n_ev = 0;                                       // uniform
#if NRM_PIPE_POS
        // software pipeline: the SplitMix64 evaluation and the table read of slot j + 1 are in flight while slot j is converted, stored and
        // listed -- the uniform branch around the event list is a scheduling barrier, and behind it a slot used to wait for its own table
        // read (~120 cycles of LDS latency x 9 slots: a lone wave spent 2,365 cycles per chunk here for 1,030 cycles of issue)
        uint64_t raw_n = mix64(zc);
        NrmWK t_n = L.wk[(uint32_t)raw_n & 0x1FFu];
#pragma unroll
        for (int j = 0; j < NRM_SLOTS; ++j) {
            const uint64_t raw = raw_n;
            const NrmWK t = t_n;
            if (j + 1 < NRM_SLOTS) {
                zc += g64;
                raw_n = mix64(zc);
                t_n = L.wk[(uint32_t)raw_n & 0x1FFu];
            }
            const uint32_t lo = (uint32_t)raw, hi = (uint32_t)(raw >> 32);
            const uint64_t mb = ((uint64_t)((hi & 0x000FFFFFu) | 0x43300000u) << 32) | (lo & ~1u);
            out[64 * j + lane] = (__longlong_as_double((long long)mb) - 4503599627370496.0) * t.w;
            const bool slow = !(mb < t.k);
            const uint64_t m = ballot64(slow);
            if (m) {
                const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)n_ev));
                if (slow) L.ev[wv][min(at, NRM_EV_CAP - 1)] = (unsigned short)(64 * j + lane);
                n_ev += __popcll(m);
            }
        }
#else
#pragma unroll
        for (int j = 0; j < NRM_SLOTS; ++j) {              // (three slots issued together, no branch between them: the same, measured)
#ifdef NRM_MEASURE_NO_MIX          // measurement builds only (wrong samples): what the SplitMix64 finaliser costs
            const uint64_t raw = zc ^ (zc >> 29);
#else
            const uint64_t raw = mix64(zc);
#endif
            zc += g64;
            const uint32_t lo = (uint32_t)raw, hi = (uint32_t)(raw >> 32);
            const NrmWK t = L.wk[lo & 0x1FFu];
            // the bit pattern of 2^52 + 2 rabs; minus 2^52: 2 rabs, exactly      (v_bfi_b32 for the and + or: the same pipe time, measured)
            const uint64_t mb = ((uint64_t)((hi & 0x000FFFFFu) | 0x43300000u) << 32) | (lo & ~1u);
            // rabs = 0 with the sign bit set gives -0.0 where the reference has +0.0: the division step below returns +0.0 for it
            // (fma(+0.0, r, -0.0) = +0.0).
            out[64 * j + lane] = (__longlong_as_double((long long)mb) - 4503599627370496.0) * t.w;
            const bool slow = !(mb < t.k);                  // 2 rabs < 2 ki, on the bit patterns
            const uint64_t m = ballot64(slow);
            if (m) {                                        // uniform branch, 60 % of the slots
                // (the index is clamped: a list that stops growing at NRM_MAX_EV instead -- one scalar compare more in the branch condition, no
                // v_min -- measured 1.8 % SLOWER at N = 8192)
                const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)n_ev));
                if (slow) L.ev[wv][min(at, NRM_EV_CAP - 1)] = (unsigned short)(64 * j + lane);
                n_ev += __popcll(m);
            }
        }
#endif
        int pos_limit = NRM_CP;                             // positions below this one are resolved (events beyond the list are not)
        __builtin_amdgcn_wave_barrier();
        if (n_ev >= NRM_MAX_EV) { n_ev = NRM_MAX_EV; pos_limit = L.ev[wv][NRM_MAX_EV - 1] + 1; }  // (essentially never: 9 events expected; later positions may hide unlisted events)
#ifdef NRM_PROF
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long pt1 = __builtin_readcyclecounter();
#endif
        // ---- 2. one divergent pass over the events: lanes 2e and 2e + 1 resolve event e < n_ev together -- the wave pays per instruction
        // ISSUED, not per lane, so the event's own draw and the uniform behind it are ONE SplitMix64 evaluation (exchanged inside the
        // pair by DPP), and a tail's two logarithms per trial are one (a fifth of the kernel's time went here with one lane per event)
        //      kind 1: wedge accepted (the output takes the value at its position, one uniform consumed)
        //      kind 2: wedge rejected (draw + uniform consumed, the output restarts behind them)
        //      kind 3: tail (the output is +-(ZIG_NOR_R + xx), two draws per trial consumed)
        const int role = lane & 1;
        const bool in_pass = (lane >> 1) < n_ev;
        const bool is_ev = in_pass && role == 0;            // the even lane of a pair carries the event from here on
        int e_pos = 0x3fffffff, e_kind = 0, e_delta = 0;    // e_delta: extra stream positions the event consumes
        double e_tail = 0.0;
#ifdef NRM_MEASURE_NO_EVENTS       // measurement builds only (wrong samples): every event taken as an accepted wedge, no exp
        if (in_pass) { e_pos = L.ev[wv][lane >> 1]; e_kind = 1; e_delta = 1; }
        if (false) {
#elif defined(NRM_MEASURE_SHARED_EVENTS)   // measurement builds only (wrong samples): the upper bound of "one event pass per workgroup" -- wave 0 runs
        // the pass (for its own events: a real handler would hold all four waves'), the others take theirs as accepted wedges, two barriers per chunk
        __syncthreads();
        if (wv != 0 && in_pass) { e_pos = L.ev[wv][lane >> 1]; e_kind = 1; e_delta = 1; }
        if (wv == 0 && in_pass) {
#else
        if (in_pass) {
#endif
            e_pos = L.ev[wv][lane >> 1];
            const double x = out[e_pos];                    // the value the fast path would have returned: (+-rabs) wi[idx], the wedge's candidate
            const uint64_t zr = base + (uint64_t)(e_pos + 1 + role) * gamma;      // even lane: the event's own draw, odd lane: the draw behind it
            const uint64_t mine = mix64(zr);
            const uint64_t theirs = ((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(mine >> 32), 0xB1, 0xF, 0xF, true) << 32)
                                    | (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)mine, 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
            const uint64_t raw = role ? theirs : mine, nxt = role ? mine : theirs;     // both lanes of the pair hold both draws
            const int idx = (int)((raw >> 1) & 0xFF);
            if (idx == 0) {
                // tail of the normal ziggurat (Random/src/normal.jl randn_unlikely, idx == 0), exactly, at this stream position: trial k
                // takes the draws 2k - 1 (-> xx) and 2k (-> yy) behind the event; the even lane evaluates the first, the odd lane the second
                uint64_t zt = zr + gamma;
                int pairs = 0;
                double xx, yy;
                do {
                    const double ut = u52_to_unit(mix64(zt));
                    const double v = tail_log1p ? -log1p(-ut) : -log(ut);
                    zt += gamma + gamma;
                    const double pv = __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), 0xB1, 0xF, 0xF, true),
                                                       __builtin_amdgcn_update_dpp(0, __double2loint(v), 0xB1, 0xF, 0xF, true));
                    xx = ZIG_NOR_INV_R * (role ? pv : v);
                    yy = role ? v : pv;
                    pairs += 1;
                } while (!(yy + yy > xx * xx) && pairs < NRM_CP);         // (a tail that runs past the chunk is cut by pos_limit and redone by the next chunk)
                e_kind = 3; e_delta = 2 * pairs;
                e_tail = ((raw >> 9) & 1) ? (-ZIG_NOR_R - xx) : (ZIG_NOR_R + xx);
            } else {
                const double u1 = u52_to_unit(nxt);
                const double f1 = L.fi[idx - 1], f0 = L.fi[idx];
                // wedge test  y < exp(-x^2 / 2)  (Random/src/normal.jl randn_unlikely).  The double-precision exp is a chain of ~60
                // dependent FP64 instructions that a handful of lanes execute while the rest of the wave waits -- a fifth of the whole
                // kernel's time when it ran for every chunk.  Decide with the hardware's single-precision exp2 instead and keep the
                // exact evaluation for the band the approximation cannot decide: relative error of (float)t -> t * log2(e) -> v_exp_f32
                // < 1e-6 for t in [-6.7, 0] (three roundings of <= 6e-8 relative on an exponent below 9.7, one ulp of the exp itself);
                // the band is 8e-6 wide on either side, so the decision is the exact one outside it and the exact code runs inside
                // it (probability 1.6e-5 per event).
                const double t = -0.5 * x * x;
                const double y = (f1 - f0) * u1 + f0;
                const double ea = (double)__builtin_amdgcn_exp2f((float)t * 1.44269504088896341f);
                e_kind = (y < ea * (1.0 - 8e-6)) ? 1 : ((y > ea * (1.0 + 8e-6)) ? 2 : 0);
                if (e_kind == 0) e_kind = (y < exp(t)) ? 1 : 2;
                e_delta = e_kind;
            }
        }
#ifdef NRM_PROF
        asm volatile("" :: "v"(e_kind), "v"(e_pos));
        const unsigned long long pt2 = __builtin_readcyclecounter();
#endif
#ifdef NRM_MEASURE_SHARED_EVENTS
        __syncthreads();
#endif
        // ---- 3. which events start an attempt (the events are sorted by position): an event does NOT iff the live event before it
        // consumed its position -- a wedge consumes the position right behind it, a tail the 2 x trials behind it
        const int e_cons = (e_kind == 3) ? e_delta : 1;
        const int prev_end = __shfl_up(e_pos + e_cons, 2, 64);       // (the event before this one sits two lanes down)
        const bool covered = is_ev && lane > 1 && e_pos <= prev_end;
        bool live = is_ev;
        if (ballot64(covered) != 0ull) {
            // runs of adjacent events (13 % of the chunks have a pair): alternate along the run.  Three Jacobi steps settle runs of up
            // to four; anything longer -- or a tail covering more than its neighbour -- is settled sequentially
            for (int it = 0; it < 3; ++it) { const bool pl = __shfl_up((int)live, 2, 64) != 0; live = is_ev && !(covered && pl); }
            const bool pl = __shfl_up((int)live, 2, 64) != 0;
            const int prev2_end = __shfl_up(e_pos + e_cons, 4, 64);
            const bool bad = (live != (is_ev && !(covered && pl))) || (is_ev && lane > 3 && e_pos <= prev2_end);
            if (ballot64(bad) != 0ull) {
                uint64_t lm = 0ull; int cend = -1;
                for (int j = 0; j < 2 * n_ev; j += 2) {
                    const int p = __builtin_amdgcn_readlane(e_pos, j);
                    if (p <= cend) continue;
                    lm |= 1ull << j;
                    cend = p + __builtin_amdgcn_readlane(e_cons, j);
                }
                live = __builtin_amdgcn_inverse_ballot_w64(lm);
            }
        }
        {   // a live tail whose trials leave the chunk: nothing at or behind its position is emitted
            const uint64_t cm = ballot64(live && e_kind == 3 && e_pos + e_delta >= NRM_CP);
            if (cm != 0ull) pos_limit = min(pos_limit, __builtin_amdgcn_readlane(e_pos, (int)__builtin_ctzll(cm)));
        }
        const int e_d = live ? e_delta : 0;
        const int e_incl = wave_iscan_i32(e_d);             // draws consumed by the events up to and including this one
        const int e_kout = e_pos - (e_incl - e_d);          // the output the attempt belongs to
        if (live) {
            // consumed positions: accepted wedge p + 1; rejected wedge p, p + 1; tail p + 1 .. p + e_delta (its own slot takes the tail's value)
            const int first = e_pos + ((e_kind == 2) ? 0 : 1), last = min(e_pos + ((e_kind == 3) ? e_delta : 1), NRM_CP - 1);
            for (int q = first; q <= last; ++q) out[q] = NRM_DEAD;
            if (e_kind == 3) out[e_pos] = e_tail;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- 4. compaction in place: the positions that were not consumed, in order, are the outputs.  A slot's keep mask is the
        // compare of its own values (ordered <=> not NRM_DEAD) -- straight into a scalar pair, no bitmap to build, clear or read back.
        // Output k never lies above position k and a slot's 64 values are read before its outputs are written.
        // (A vector-only form -- v_bcnt against the lane's own lanes-below mask, consumed positions stored into dump slots, no EXEC
        // mask -- makes a lone wave 16 % faster per chunk and the kernel at full occupancy 17 % SLOWER: measured in round 3 and dropped.)
        {
            int cum = 0;                                    // uniform: outputs before this slot
#if NRM_COMPACT_ILP
            // all the slots' values are requested in ONE LDS round trip (a slot's outputs land below the next slot's positions, so reading
            // every slot first is the same function; the compiler cannot know and would keep read j + 1 behind write j): the kernel is bound
            // by a wave's dependency chain as much as by issue (a lone wave's row takes 32 us, each further wave of the SIMD adds 5)
            double vall[NRM_SLOTS];
#pragma unroll
            for (int j = 0; j < NRM_SLOTS; ++j) vall[j] = out[64 * j + lane];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int j = 0; j < NRM_SLOTS; ++j) {
                const double vj = vall[j];
                const uint64_t keep = ballot64(vj == vj);
                const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, (unsigned)cum));
                if (vj == vj) out[at] = vj;
                cum += __popcll(keep);
            }
#else
#pragma unroll
            for (int j = 0; j < NRM_SLOTS; ++j) {
                const double vj = out[64 * j + lane];
                const uint64_t keep = ballot64(vj == vj);
                const int at = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(keep >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)keep, (unsigned)cum));
                if (vj == vj) out[at] = vj;
                cum += __popcll(keep);
            }
#endif
        }
        __builtin_amdgcn_wave_barrier();
#ifdef NRM_PROF
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long walk_cyc = __builtin_readcyclecounter() - pt2;
#endif
        // ---- 5. groups of 256 outputs: divide, store, tree
        int emitted = 0, shift_emitted = 0;
        const int want = (int)min((int64_t)NRM_CO, d - done);
        // draws consumed by the attempts of the outputs below `gend` (uniform): the live events are sorted by output, so it is the
        // running total at the last of them below gend
        auto consumed_below = [&](int gend) -> int {
            const uint64_t below = ballot64(live && e_kout < gend);
            return below ? __builtin_amdgcn_readlane(e_incl, 63 - (int)__builtin_clzll(below)) : 0;
        };
        const int shift_all = consumed_below(NRM_CO);
#ifdef NRM_MEASURE_IGNORE_LIMIT     // measurement builds only (wrong samples): the straight-line path whatever the chunk resolved
        if (want == NRM_CO && markstein) {
#else
        if (want == NRM_CO && markstein && NRM_CO - 1 + shift_all < pos_limit) {
#endif
            // the whole chunk (all but the last chunk of a row, and every position it takes is resolved): straight-line code, the
            // groups' dependent chains (Markstein steps, DPP tree levels) issued level by level
            constexpr int NG = NRM_CO / 256;
            double q[NG][4];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const double2 a01 = *reinterpret_cast<const double2 *>(&out[256 * g + 4 * lane]), a23 = *reinterpret_cast<const double2 *>(&out[256 * g + 4 * lane + 2]);
                q[g][0] = a01.x; q[g][1] = a01.y; q[g][2] = a23.x; q[g][3] = a23.y;
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) { const double a = q[g][i], t = a * rinv; q[g][i] = __builtin_fma(__builtin_fma(-t, sd, a), rinv, t); }
            double sq[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
#ifndef NRM_NO_STORE
                double2 *dst = reinterpret_cast<double2 *>(xrow + done + 256 * g + 4 * lane);
                dst[0] = make_double2(q[g][0], q[g][1]); dst[1] = make_double2(q[g][2], q[g][3]);
#endif
                sq[g] = (q[g][0] * q[g][0] + q[g][1] * q[g][1]) + (q[g][2] * q[g][2] + q[g][3] * q[g][3]);
            }
#ifndef NRM_PLAIN_TREE
            if constexpr (NG == 2) {
                // the same additions with the two groups' partial sums sharing a register from level 2 on (pte_device.hpp, wave_sum_pairs):
                // even lanes carry group 0, odd lanes group 1; row_shr:4 / row_shr:8 keep a lane's parity and leave a row's (= one
                // block's) two sums in its lanes 12 and 13 -- 17 instead of 26 vector instructions per chunk
                sq[0] = dpp_add_step<0xB1, 0xF>(sq[0]); sq[1] = dpp_add_step<0xB1, 0xF>(sq[1]);
                double m = select_lanes_f64(0xAAAAAAAAAAAAAAAAull, sq[0], sq[1]);
                m = dpp_add_step<0x4E, 0xF>(m);
                m = dpp_add_step<0x114, 0xF>(m);
                m = dpp_add_step<0x118, 0xF>(m);
                if ((lane & 14) == 12) L.bs[wv][(int)(done >> 6) + 4 * (lane & 1) + (lane >> 4)] = m;      // lane 16 r + 12 + g: block 4 g + r of the chunk
            } else
#endif
            {
#pragma unroll
            for (int lv = 0; lv < 4; ++lv)
#pragma unroll
                for (int g = 0; g < NG; ++g) sq[g] = dpp_row_step(sq[g], lv);
            // every lane of a row holds the row's (= one block's) sum: lane 16 r + g writes block 4 g + r of the chunk
            double sel = sq[0];
#pragma unroll
            for (int g = 1; g < NG; ++g) sel = ((lane & 15) == g) ? sq[g] : sel;
            if ((lane & 15) < NG) L.bs[wv][(int)(done >> 6) + 4 * (lane & 15) + (lane >> 4)] = sel;
            }
            emitted = NRM_CO; shift_emitted = shift_all;
        } else
        for (int g0 = 0; g0 < want; g0 += 256) {
            const int gend = min(g0 + 256, want);
            const int k0 = g0 + 4 * lane;                   // this lane's outputs k0 .. k0 + 3 (relative to the chunk)
            const int shift = consumed_below(gend);
            if (gend - 1 + shift >= pos_limit) break;        // every position the group takes must be resolved
            const double2 a01 = *reinterpret_cast<const double2 *>(&out[k0]), a23 = *reinterpret_cast<const double2 *>(&out[k0 + 2]);
            const double a0 = a01.x, a1 = a01.y, a2 = a23.x, a3 = a23.y;
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
            double sq = (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
            sq = dpp_row_step(sq, 0); sq = dpp_row_step(sq, 1); sq = dpp_row_step(sq, 2); sq = dpp_row_step(sq, 3);
            if ((lane & 15) == 0) L.bs[wv][(int)((done + g0) >> 6) + (lane >> 4)] = sq;
            emitted = gend; shift_emitted = shift;          // (the attempts of every output below gend, of none at or above it)
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
                if (lane == 0) L.bs[wv][(int)((done + b0) >> 6)] = sq;
            }
            done += gl;
            continue;
        }
        r.seed = base + (uint64_t)(emitted + shift_emitted) * gamma;     // stream position of the next output's first draw
#ifdef NRM_PROF
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        { const unsigned long long pt3 = __builtin_readcyclecounter();
          if (lane == 0) { L.prof[0] += pt1 - pt0; L.prof[1] += pt2 - pt1; L.prof[2] += walk_cyc; L.prof[3] += pt3 - pt2 - walk_cyc; L.prof[4] += 1; L.prof[5] += n_ev; } }
#endif
        done += emitted;
    }
    __builtin_amdgcn_wave_barrier();
    const int B = (int)((d + 63) >> 6);
    return (lane < B) ? L.bs[wv][lane] : 0.0;
}

}  // namespace pte
