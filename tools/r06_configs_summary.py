"""Round 6: per-config summary of the rocprofv3 passes of tools/prof_configs6.py (tools/prof_configs6.sh) -> profiles/r06_configs_pmc_summary.txt
and the static side of bench.py's extra_configs[].roofline, profiles/r06_configs.json.

Per config the TIMED dispatches are the last `timed_dispatches` dispatches of its kernel (fused: REPS launches of `scans` scans; otherwise
REPS x scans explore launches and as many k_swap launches, told apart from other configs' by the grid size).  Everything is per SCAN.
Units and corrections: SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE counts KB at half the
bytes of this engine's 8-byte-lane reads (x 2: profiles/traffic.json, r01_traffic_calibration.txt), WRITE_SIZE counts KB.
VALU issue: a wave64 VALU instruction occupies its SIMD's vector ALU for 4 cycles (16 lanes per cycle; FP64 at the same rate: 128 flop / clk / CU
= the 78.6 TF datasheet peak) -> valu_issue_frac = SQ_INSTS_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE), the share of all VALU issue slots
of the chip the kernel fills while it runs.  FP64 flops EXECUTED = (ADD_F64 + MUL_F64 + TRANS_F64 + 2 FMA_F64) wave-instructions x 64 lanes
(EXEC masks not applied: an upper bound where lanes are masked off -- the speculative kernels run most lanes on hypotheses that are discarded,
and count them: this is what the hardware executed, not what the algorithm needed).
Usage: python tools/r06_configs_summary.py gpurun_out/prof_configs6_<tag>"""
import json, os, sqlite3, sys

O = sys.argv[1]
KERNEL_OF = {"C1": "k_scans_slice8<0, 9>", "C2": "k_scans_slice8<4, 9>", "C3": "k_scans_automala_wg<2, 2, true>", "C4_shard": "k_explore_slice8<6, 9>",
             "C4_one_gpu": "k_explore_slice8_lds10k<6, 9>", "C5_shard": "k_explore_ising_spec<false>"}
LIMITED = {"C1": "hand-shake + launch latency (10 waves on the chip)", "C2": "instruction issue of one wave per replica (768 of 1024 SIMDs idle)",
           "C3": "FP64 VALU + exp / log + DPP latency, one wave per SIMD", "C4_shard": "instruction issue of one wave per replica",
           "C4_one_gpu": "instruction issue, four waves per SIMD", "C5_shard": "integer VALU issue of one wave per replica (half the SIMDs idle)"}


def q(db, sql, args=()):
    con = sqlite3.connect(os.path.join(O, db)); rows = con.execute(sql, args).fetchall(); con.close(); return rows


def cfg_lines():
    out = []
    for ln in open(os.path.join(O, "stats.log")):
        if ln.startswith("PC6 "):
            out.append(json.loads(ln[4:]))
    return out


def last_dispatches(db, like, n, grid=None):
    """ids (in this db) of the last n dispatches of the kernel"""
    if db == "stats_results.db":
        rows = q(db, "select id, (end - start), grid_x from kernels where name like ? order by start", ("%" + like + "%",))
    else:
        rows = q(db, "select dispatch_id, min(end - start), max(grid_size) from counters_collection where kernel_name like ? group by dispatch_id order by min(start)", ("%" + like + "%",))
    if grid is not None:
        rows = [r for r in rows if r[2] == grid]
    return rows[-n:]


def counters(db, like, n, grid=None):
    ids = [r[0] for r in last_dispatches(db, like, n, grid)]
    if not ids:
        return {}
    rows = q(db, "select counter_name, sum(value) from counters_collection where kernel_name like ? and dispatch_id in (%s) group by counter_name" % ",".join(str(i) for i in ids), ("%" + like + "%",))
    return {r[0]: r[1] for r in rows}


def main():
    cfgs = cfg_lines()
    static = {}
    print("# rocprofv3 passes of `python3 tools/prof_configs6.py` (tools/prof_configs6.sh; library = HEAD of round 6): every config of bench.py's extra_configs,")
    print("# prepared as bench.py prepares it (rounds 1..r of the algorithm), then %d pte_run_scans calls of the config's timed scan count; per SCAN below." % cfgs[0]["reps"])
    print("# passes: --kernel-trace --stats | SQ (waves, cycles, VALU / SALU / branch, waits) | FP64 (ADD / MUL / FMA / TRANS_F64, LDS, INT32/64) | GRBM_GUI_ACTIVE | FETCH_SIZE | WRITE_SIZE")
    rows = q("stats_results.db", "select name, total_calls, total_duration, average, percentage from top_kernels")
    print("\n== --kernel-trace --stats (top kernels of the whole run, preparation included)\n%-84s %6s %12s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in rows[:14]:
        print("%-84s %6d %12.0f %12.1f %7.2f" % (r[0][:84], r[1], r[2], r[3], r[4]))
    for c in cfgs:
        key, like, S, reps, N = c["key"], KERNEL_OF[c["key"]], c["scans"], c["reps"], c["n_chains"]
        fused = bool(c["scan_loop"])
        nd = c["timed_dispatches"]
        scans_total = reps * S
        print("\n" + "=" * 150)
        print("== %s   [%s]" % (c["config"], like))
        print("   scan loop: %s; timed: %d calls x %d scans = %d dispatches of the kernel; wall clock %.4f ms per scan (unprofiled pass of this program: stats.log)"
              % (c["scan_loop"] or "two launches per scan", reps, S, nd, c["wall_ms_per_scan"]))
        tr = last_dispatches("stats_results.db", like, nd)
        k_us = sum(r[1] for r in tr) / 1e3 / scans_total
        sw_us = 0.0
        if not fused:
            grid = ((N + 255) // 256) * 256
            ts = last_dispatches("stats_results.db", "k_swap(", nd, grid)
            sw_us = sum(r[1] for r in ts) / 1e3 / scans_total if ts else 0.0
        print("   kernel trace: %s %.1f us per scan (dispatches, us: %s)%s" % ("scan loop" if fused else "explore", k_us,
              " ".join("%.0f" % (r[1] / 1e3) for r in tr[:8]) + (" ..." if len(tr) > 8 else ""), "" if fused else "; k_swap %.1f us per scan" % sw_us))
        sq = counters("sq_results.db", like, nd)
        fp = counters("fp_results.db", like, nd)
        gr = counters("grbm_results.db", like, nd)
        fe = counters("fetch_results.db", like, nd)
        wr = counters("write_results.db", like, nd)
        ent = {"config": c["config"], "kernel": like, "source": "profiles/r06_configs_pmc_summary.txt (rocprofv3 PMC passes of tools/prof_configs6.py)", "limited_by": LIMITED[key],
               "kernel_us_per_scan_in_trace": k_us + sw_us}
        if sq.get("SQ_WAVES"):
            W = sq["SQ_WAVES"] / (reps if fused else nd)            # waves per dispatch
            per = lambda k: sq.get(k, 0.0) / W / scans_total * (1 if fused else 1)      # per wave and scan (fused: a wave lives `scans` scans)
            if not fused:
                per = lambda k: sq.get(k, 0.0) / W / nd
            ins = per("SQ_INSTS_VALU") + per("SQ_INSTS_SALU") + per("SQ_INSTS_BRANCH")
            cyc = per("SQ_WAVE_CYCLES") * 4.0
            print("   SQ per wave and scan (%d waves): VALU %.0f  SALU %.0f  branch %.0f = %.1f k instructions; wave cycles %.3f M = %.2f cycles per instruction; "
                  "WAIT_ANY %.3f  WAIT_INST_ANY %.3f  ACTIVE_INST_ANY %.3f of the wave cycles"
                  % (W, per("SQ_INSTS_VALU"), per("SQ_INSTS_SALU"), per("SQ_INSTS_BRANCH"), ins / 1e3, cyc / 1e6, cyc / max(ins, 1),
                     sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"]))
            ent.update({"waves": W, "valu_per_wave_scan": per("SQ_INSTS_VALU"), "salu_per_wave_scan": per("SQ_INSTS_SALU"), "branch_per_wave_scan": per("SQ_INSTS_BRANCH"),
                        "cycles_per_instruction": round(cyc / max(ins, 1), 2), "sq_wait_any_frac": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 3),
                        "sq_active_inst_any_frac": round(sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 3)})
        if gr.get("GRBM_GUI_ACTIVE") and gr.get("SQ_INSTS_VALU"):
            # GRBM_GUI_ACTIVE: cycles the GPU was busy over the timed dispatches (per XCD instance summed? -> normalise by the kernel-trace time below instead if absurd)
            gui = gr["GRBM_GUI_ACTIVE"]
            dur_s = sum(r[1] for r in last_dispatches("grbm_results.db", like, nd)) / 1e9
            mhz = gui / dur_s / 1e6 if dur_s > 0 else 0.0
            n_inst = 1
            while mhz / n_inst > 3000.0:           # the counter is summed over its instances (one per XCD): 8 x 2.4 GHz shows up as 19 GHz
                n_inst *= 2
            clk = gui / n_inst                      # shader-clock cycles of the timed dispatches
            frac = gr["SQ_INSTS_VALU"] * 4.0 / (1024.0 * clk)
            print("   VALU issue: SQ_INSTS_VALU %.3g x 4 cycles / (1024 SIMDs x %.3g cycles [GRBM_GUI_ACTIVE / %d instances = %.0f MHz effective]) = %.4f of the chip's VALU issue slots"
                  % (gr["SQ_INSTS_VALU"], clk, n_inst, mhz / n_inst, frac))
            ent.update({"valu_issue_frac": frac, "effective_clock_MHz": mhz / n_inst})
        if fp:
            f64 = fp.get("SQ_INSTS_VALU_ADD_F64", 0) + fp.get("SQ_INSTS_VALU_MUL_F64", 0) + fp.get("SQ_INSTS_VALU_TRANS_F64", 0) + 2 * fp.get("SQ_INSTS_VALU_FMA_F64", 0)
            flops = f64 * 64.0 / scans_total
            tf = flops / ((k_us) * 1e-6) / 1e12 if k_us > 0 else 0.0
            print("   FP64 wave-instructions per scan: ADD %.3g  MUL %.3g  FMA %.3g  TRANS %.3g  (INT32 %.3g, INT64 %.3g, LDS %.3g) -> %.4g flops EXECUTED per scan (x 64 lanes, FMA = 2) = %.3f TFLOP/s = %.4f of the 78.6 TF FP64 vector peak"
                  % (tuple(fp.get(k, 0) / scans_total for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_LDS")) + (flops, tf, tf / 78.6)))
            ent.update({"fp64_flops_per_scan": flops, "fp64_TFLOPs_in_trace": tf, "int32_valu_per_scan": fp.get("SQ_INSTS_VALU_INT32", 0) / scans_total})
        alg = c["bytes_per_replica_scan"] * N
        if fe.get("FETCH_SIZE") is not None and wr.get("WRITE_SIZE") is not None:
            fb = fe["FETCH_SIZE"] * 1024 * 2 / scans_total; wb = wr["WRITE_SIZE"] * 1024 / scans_total
            print("   HBM per scan: fetched %.3f MB (FETCH_SIZE x 2), written %.3f MB; algorithmic %d B x %d = %.3f MB -> traffic / algorithmic = %.2f; algorithmic bytes / kernel time = %.1f GB/s = %.5f of 8 TB/s"
                  % (fb / 1e6, wb / 1e6, c["bytes_per_replica_scan"], N, alg / 1e6, (fb + wb) / alg, alg / ((k_us + sw_us) * 1e-6) / 1e9, alg / ((k_us + sw_us) * 1e-6) / 8e12))
            ent.update({"fetch_bytes_per_scan": fb, "write_bytes_per_scan": wb, "traffic_bytes_per_scan": fb + wb, "algorithmic_bytes_per_scan": alg})
        static[key] = ent
    json.dump(static, open(os.path.join(O, "configs.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
