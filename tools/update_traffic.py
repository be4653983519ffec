"""profiles/traffic.json entry of the dominant kernel from a tools/prof_round.sh summary (rocprofv3 PMC passes of bench.py).
usage: python tools/update_traffic.py gpurun_out/prof_r03/summary.txt profiles/r03_slice8_summary.txt "<note>"
FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B; calibrated on this engine's 8-B-per-lane pattern, see the _comment),
WRITE_SIZE taken as is; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles."""
import json, os, re, shutil, sys
src, dst, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shutil.copyfile(src, os.path.join(ROOT, dst))
c = {}
for ln in open(src):
    m = re.match(r"void pte::(k_explore_slice8\w*)<.*?\s(SQ_\w+|FETCH_SIZE|WRITE_SIZE|GRBM_\w+)\s+([\d.]+)\s+(\d+)", ln)
    if m:
        c[m.group(2)] = float(m.group(3)); kern = m.group(1)
waves = c["SQ_WAVES"]
inst = (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_BRANCH"]) / waves
wc = c["SQ_WAVE_CYCLES"] * 4 / waves
d, n_passes = 1024, 3
entry = {"config": "toy_mvn_target(1024), n_chains=1024", "fetch_bytes": int(round(c["FETCH_SIZE"] * 1024 * 2)), "write_bytes": int(round(c["WRITE_SIZE"] * 1024)),
         "source": dst + (" (%s)" % note if note else ""), "fetch_counter_kb": c["FETCH_SIZE"],
         "issue": {"_comment": "SQ_* PMC pass of the same command (%s); SQ_WAVE_CYCLES counts quad-cycles" % dst,
                   "valu_per_wave": int(c["SQ_INSTS_VALU"] / waves), "salu_per_wave": int(c["SQ_INSTS_SALU"] / waves), "branch_per_wave": int(c["SQ_INSTS_BRANCH"] / waves),
                   "instructions_per_wave": int(inst), "wave_cycles": int(wc), "cycles_per_instruction": round(wc / inst, 2),
                   "instructions_per_coordinate": round(inst / (d * n_passes), 1),
                   "sq_wait_any_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3), "sq_wait_inst_any_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3),
                   "sq_active_inst_any_frac": round(c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3)}}
p = os.path.join(ROOT, "profiles", "traffic.json")
j = json.load(open(p))
j[kern] = entry
json.dump(j, open(p, "w"), indent=2)
print(json.dumps(entry, indent=1))
