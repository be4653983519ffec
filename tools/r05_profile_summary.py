"""Round 5: per-scan summary of the rocprofv3 passes of bench.py (tools/prof_round5.sh), where pte_run_scans is ONE launch of k_scans_slice8
holding S scans.  Prints every dispatch of the fused kernel (the first one of a --kernel-trace run carries the profiler's own start-up), the
per-scan duration, the SQ counters per wave and scan, and the HBM traffic per scan with the corrections of profiles/traffic.json (FETCH_SIZE
counts KB / 2 for this engine's 8-byte lanes, WRITE_SIZE counts KB).  Usage: python tools/r05_profile_summary.py gpurun_out/prof_<tag> S [N d]"""
import json, os, sqlite3, sys

def main():
    O, S = sys.argv[1], int(sys.argv[2])
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    d = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
    def q(db, sql):
        con = sqlite3.connect(os.path.join(O, db)); rows = con.execute(sql).fetchall(); con.close(); return rows
    print("# rocprofv3 passes of `python3 bench.py --no-cpu-baseline --no-extra --round-trip-rounds 0 --steps %d --warmup %d` (tools/prof_round5.sh):" % (S, S))
    print("# warm-up, timed and event-free passes are three launches of the fused scan loop, %d scans each (the two shorter dispatches before them: bench.py's untimed preparation, 2 x 64 scans); N = %d chains, d = %d" % (S, N, d))
    for db in ("stats_results.db", "sq_results.db"):
        rows = q(db, "select name, (end-start)/1e6 from kernels where name like '%k_scans%' order by start")
        print("%-18s k_scans dispatches, ms: %s  -> per scan, us: %s" % (db, ["%.2f" % r[1] for r in rows], ["%.1f" % (r[1] * 1e3 / S) for r in rows]))
    rows = q("stats_results.db", "select name, total_calls, total_duration, average, percentage from top_kernels")
    print("\n== --kernel-trace --stats (top kernels)\n%-76s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:6]:
        print("%-76s %6d %12.0f %12.0f %7.2f" % (r[0][:76], r[1], r[2], r[3], r[4]))
    # round 6: bench.py also launches the scan loop for its untimed preparation (2 x 64 scans) -- the counters are those of the LAST THREE dispatches
    # (warm-up, timed, event-free: S scans each), chosen by dispatch id, averaged per dispatch
    def last3(db):
        ids = [r[0] for r in q(db, "select dispatch_id, min(start) from counters_collection where kernel_name like '%k_scans%' group by dispatch_id order by min(start)")][-3:]
        return ",".join(str(i) for i in ids)
    c = {r[0]: r[1] / 3.0 for r in q("sq_results.db", "select counter_name, sum(value) from counters_collection where kernel_name like '%%k_scans%%' and dispatch_id in (%s) group by counter_name" % last3("sq_results.db"))}
    W = c["SQ_WAVES"]
    per = lambda k: c[k] / W / S
    ins = per("SQ_INSTS_VALU") + per("SQ_INSTS_SALU") + per("SQ_INSTS_BRANCH")
    cyc = per("SQ_WAVE_CYCLES") * 4.0                 # SQ_WAVE_CYCLES counts quad-cycles
    print("\n== SQ counters per wave and SCAN (mean of the three launches / %d waves / %d scans; *_CYCLES and WAIT / ACTIVE count quad-cycles)" % (int(W), S))
    print("VALU %.0f  SALU %.0f  branch %.0f  -> %.1f k instructions per wave-scan = %.1f per coordinate update (3 d = %d)" % (per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_BRANCH"), ins / 1e3, ins / (3 * d), 3 * d))
    print("wave cycles %.3f M per scan (incl. the hand-shake's polls and s_sleep) = %.2f cycles per instruction; SQ_WAIT_ANY %.3f, SQ_WAIT_INST_ANY %.3f, SQ_ACTIVE_INST_ANY %.3f of the wave cycles"
          % (cyc / 1e6, cyc / ins, c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
    print("issue roofline of a lone wave: instructions x 4.44 / wave cycles = %.3f" % (ins * 4.44 / cyc))
    f = q("fetch_results.db", "select sum(value) / 3.0 from counters_collection where kernel_name like '%%k_scans%%' and counter_name = 'FETCH_SIZE' and dispatch_id in (%s)" % last3("fetch_results.db"))[0][0]
    w = q("write_results.db", "select sum(value) / 3.0 from counters_collection where kernel_name like '%%k_scans%%' and counter_name = 'WRITE_SIZE' and dispatch_id in (%s)" % last3("write_results.db"))[0][0]
    fb, wb = f * 1024 * 2 / S, w * 1024 / S
    alg = (16 * d + 32 + 96) * N
    print("\n== HBM traffic per SCAN: FETCH_SIZE %.1f KB per launch -> %.2f MB fetched, WRITE_SIZE %.1f KB per launch -> %.2f MB written; algorithmic (16 d + 32 + 96) N = %.2f MB"
          % (f, fb / 1e6, w, wb / 1e6, alg / 1e6))
    print("   (fetch: the %d MB of state stay in the XCDs' L2s from scan to scan inside ONE launch -- the per-scan launches of rounds 1-4 fetched 9.5 MB per scan;" % (8 * d * N // 2 ** 20))
    print("    write: every wave's agent-scope release writes back ALL dirty lines of its XCD's L2, so a row dirtied in pass 1, 2 and 3 can leave three times: %.2fx the 8 d N written algorithmically)" % (wb / (8.0 * d * N)))
    print(json.dumps({"k_scans_slice8": {"config": "toy_mvn_target(%d), n_chains=%d" % (d, N), "per_scan": True, "fetch_bytes": round(fb), "write_bytes": round(wb),
                                          "issue": {"valu_per_wave": round(per("SQ_INSTS_VALU")), "salu_per_wave": round(per("SQ_INSTS_SALU")), "branch_per_wave": round(per("SQ_INSTS_BRANCH")),
                                                    "instructions_per_wave": round(ins), "wave_cycles": round(cyc), "cycles_per_instruction": round(cyc / ins, 2),
                                                    "instructions_per_coordinate": round(ins / (3 * d), 1), "sq_wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                                                    "sq_wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3), "sq_active_inst_any_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}}}))

if __name__ == "__main__":
    main()
