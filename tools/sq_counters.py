#!/usr/bin/env python3
"""gpurun_out/sq/pass*.txt (tools/sq_profile.sh) -> profiles/<tag>_sq_counters.txt: per kernel and dispatch (= one tick of 256
pictures) the SQ counters that say what bounds it.  usage: sq_counters.py <tag>"""
import collections, glob, json, os, re, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "sq")
vals = collections.defaultdict(dict)
cmds = []
for f in sorted(glob.glob(os.path.join(src, "pass*.txt"))):
    for l in open(f):
        if l.startswith("#"):
            cmds.append(l.rstrip()); continue
        m = re.match(r"(?:void )?h264k::(\w+)(<\w+>)?\(.*?\s+(\w+)\s+total=(\S+) per_dispatch=(\S+) n=(\d+)", l)
        if m:
            k = m.group(1)
            d = vals[k].setdefault(m.group(3), [0.0, 0])
            d[0] += float(m.group(4)); d[1] = max(d[1], int(m.group(6)))     # the two instantiations of a template add up
bl = json.loads(open(os.path.join(src, "bench_line.json")).read())
ms = bl["roofline"]["device_ms_per_step"]
ticks = 73
GHZ = 2.4
VALU_CYC = 4.3            # cycles a SIMD spends per wave64 VOP3 / VOP3P instruction (profiles/r04_valu_rate_probe.txt; rounds 1-3 assumed 4)
order = ["k_copy", "k_recon_inter", "k_dbk", "k_frame_intra", "k_frame_dbk"]
out = ["# SQ / TCC counters of the lock-step bench (256 x 1080p streams), " + tag + "; one rocprofv3 --pmc pass per group, no runtime traces:"]
out += ["#   " + c[2:] for c in cmds]
out += ["# per dispatch (= one tick of 256 pictures; values of the two instantiations of a templated kernel added).  SQ_WAVE_CYCLES / SQ_WAIT_* /",
        "# SQ_ACTIVE_* count quad-cycles summed over wavefronts (MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.",
        f"# 'valu pipe' = SQ_INSTS_VALU x {VALU_CYC} cycles (measured: tools/probes/valu_rate_probe.hip) / (1024 SIMDs x kernel time per tick from the un-profiled bench line of the same build x {GHZ} GHz).",
        f"# bench line of this build: {bl['value'] / 1e6:.1f} M MB/s, device ms per step " + json.dumps({k: round(v, 1) for k, v in ms.items() if isinstance(v, (int, float))}),
        "# 'lanes' = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = active lanes per wave64 VALU instruction (calibrated: tools/probes/lane_util_probe.hip gives 64.0 / 32.0 / 16.0 / 8.0 / 1.0 for 64 / 32 / 16 / 8 / 1 active lanes: no further factor).",
        "kernel           dispatches  VALU instr  SALU instr   LDS instr  VMEM rd/wr   parked  issue-stall  issuing  LDS-stall  bank-confl  L2 hit  valu pipe  lanes"]
for k in order:
    v = {n: (t / c if c else 0.0) for n, (t, c) in vals.get(k, {}).items()}
    if not v:
        continue
    wc = v.get("SQ_WAVE_CYCLES", 0) or 1
    t_us = ms[k] * 1e3 / ticks
    pipe = v.get("SQ_INSTS_VALU", 0) * VALU_CYC / (1024 * t_us * 1e-6 * GHZ * 1e9)
    hit = v.get("TCC_HIT_sum", 0) / max(1.0, v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0))
    n = max(c for _, c in vals[k].values())
    out.append(f"{k:16s} {n:10d}  {v.get('SQ_INSTS_VALU', 0):10.3e}  {v.get('SQ_INSTS_SALU', 0):10.3e}  {v.get('SQ_INSTS_LDS', 0):10.3e}  "
               f"{v.get('SQ_INSTS_VMEM_RD', 0):.2e}/{v.get('SQ_INSTS_VMEM_WR', 0):.2e}  {v.get('SQ_WAIT_ANY', 0) / wc:5.0%}  {v.get('SQ_WAIT_INST_ANY', 0) / wc:10.0%}  "
               f"{v.get('SQ_ACTIVE_INST_ANY', 0) / wc:6.0%}  {v.get('SQ_WAIT_INST_LDS', 0) / wc:8.1%}  {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, v.get('SQ_LDS_IDX_ACTIVE', 0)):9.1%}  {hit:5.0%}  {pipe:8.0%}  {(v.get('SQ_THREAD_CYCLES_VALU', 0) / v['SQ_INSTS_VALU']) if v.get('SQ_INSTS_VALU') else 0:5.1f}")
out.append("")
out.append("# raw per-dispatch values")
for k in order:
    for n in sorted(vals.get(k, {})):
        t, c = vals[k][n]
        out.append(f"{k:16s} {n:24s} {t / c:.4g}")
open(os.path.join(root, "profiles", f"{tag}_sq_counters.txt"), "w").write("\n".join(out) + "\n")
# the table bench.py reads for roofline.valu: wave-level instruction counts per launch (= per tick of 256 pictures), tied to the
# kernel sources like the traffic table
sys.path.insert(0, root)
from h264bsd_amd.srchash import kernel_source_sha256
table = {"source": f"profiles/{tag}_sq_counters.txt (separate rocprofv3 --pmc passes of the lock-step bench, 256 x 1080p streams)",
         "kernel_source_sha256": kernel_source_sha256(root), "cycles_per_wave_instruction": VALU_CYC,
         "cycles_source": "tools/probes/valu_rate_probe.hip: 4.2-4.5 cycles per wave64 VOP3 / VOP3P instruction on one SIMD (v_pk_*_i16, v_perm_b32, v_mad_*, v_dot4)",
         "streams": 256, "kernels": {}}
for k in order:
    v = {n: (t / c if c else 0.0) for n, (t, c) in vals.get(k, {}).items()}
    if v:
        table["kernels"][k] = {"valu_wave_instr_per_launch": v.get("SQ_INSTS_VALU", 0.0), "salu_wave_instr_per_launch": v.get("SQ_INSTS_SALU", 0.0),
                               "lds_wave_instr_per_launch": v.get("SQ_INSTS_LDS", 0.0), "active_lanes_per_valu_instr": (v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_INSTS_VALU"]) if v.get("SQ_INSTS_VALU") else None}
json.dump(table, open(os.path.join(root, "profiles", f"{tag}_sq_counters.json"), "w"), indent=1)
print("\n".join(out[:24]))

# the committed bench line was printed before the counter passes of this session existed: fill its roofline.valu from them, with
# the launch durations of that very line (the same arithmetic as bench.py, which reports it live once this table matches the sources)
blp = os.path.join(root, "profiles", f"{tag}_bench_line.json")
if os.path.exists(blp):
    line = json.loads(open(blp).read())
    rf = line.get("roofline", {})
    v0 = rf.get("valu")
    if (v0 is None or "per_kernel" not in v0 or any("lane_util" not in p_ for p_ in v0["per_kernel"].values())) and rf.get("device_ms_per_step") and rf.get("launches_per_step"):
        peak = 1024 * 2.4e9 / VALU_CYC
        n_mbs_tick = float(rf.get("mbs_per_launch", 256 * 8160))
        per = {}
        for k, kv in table["kernels"].items():
            ms, n = rf["device_ms_per_step"].get(k), rf["launches_per_step"].get(k)
            if not ms or not n:
                continue
            n_i, t_s = kv["valu_wave_instr_per_launch"], ms * 1e-3 / n
            per[k] = {"wave_instr_per_launch": n_i, "avg_launch_us": t_s * 1e6, "achieved": n_i / t_s, "frac": n_i / t_s / peak,
                      "wave_instr_per_macroblock": n_i / n_mbs_tick, "lane_util": (kv.get("active_lanes_per_valu_instr") or 0.0) / 64.0}
        tot_i = sum(p["wave_instr_per_launch"] for p in per.values())
        ticks = max(rf["launches_per_step"].values())
        total_s = rf["device_ms_per_step"]["total"] * 1e-3
        rf["valu"] = {"peak_wave_instr_per_s": peak, "cycles_per_wave_instruction": VALU_CYC, "source": f"profiles/{tag}_sq_counters.json",
                      "filled_in_by": "tools/sq_counters.py: counters collected in separate rocprofv3 --pmc passes right after this line on the same box, launch durations of this line",
                      "per_kernel": per, "whole_path": {"wave_instr_per_tick": tot_i, "achieved": tot_i * ticks / total_s, "frac": tot_i * ticks / total_s / peak,
                                                        "note": "every kernel runs once per tick; k_dbk runs next to the others, its time is not in the total"}}
        open(blp, "w").write(json.dumps(line) + "\n")
        print("filled roofline.valu of", os.path.relpath(blp, root), {k: round(p["frac"], 3) for k, p in per.items()}, "whole path", round(rf["valu"]["whole_path"]["frac"], 3))
