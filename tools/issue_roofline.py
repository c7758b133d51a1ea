#!/usr/bin/env python3
"""Issue-side roofline of the VALU-bound kernels, from TRACKED files only (no GPU needed):

  profiles/<tag>_counters.json    rocprofv3 --pmc: SQ_INSTS_VALU, SQ_WAVES, kernel durations, effective clock (tools/collect_counters.py)
  profiles/<tag>_isa_counts.json  static instruction mix of every kernel and the issue cycles of its mix (tools/isa_counts.py)

For a kernel:  cycles per VALU instruction of its mix  c = issue_cycles_per_wave / valu  (static, from the ISA),
               dynamic VALU instructions per wave      I = SQ_INSTS_VALU / SQ_WAVES     (measured; equals the static count for the unrolled kernels),
               issue floor = waves x I x c / (1024 SIMDs x effective clock)             -- the time the launch needs if every SIMD issues
                                                                                           VALU back to back and nothing ever waits,
               issue_frac  = issue floor / measured duration.
An HBM-bound kernel shows a small issue_frac and a large HBM figure; a kernel near issue_frac 1 can only get faster with fewer or
cheaper instructions.  Both hashes must agree (same kernel sources).  usage: python tools/issue_roofline.py [tag] > profiles/<tag>_issue_roofline.txt"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
cnt = json.load(open(os.path.join(ROOT, "profiles", tag + "_counters.json")))
isa = json.load(open(os.path.join(ROOT, "profiles", tag + "_isa_counts.json")))
print("counters: kernel_source_hash %s   isa counts: kernel_source_hash %s%s" % (cnt["kernel_source_hash"], isa["kernel_source_hash"],
      "" if cnt["kernel_source_hash"] == isa["kernel_source_hash"] else "   (DIFFERENT SOURCES: kernels whose static VALU count differs from the measured one are marked *)"))
print("cycles per wave64 instruction on one SIMD:", isa["cycles_per_wave_instruction"])
print("%-40s %9s %9s %8s %8s %7s %9s %9s %7s %9s" % ("kernel", "waves", "insts/wv", "static", "cyc/inst", "clk_GHz", "floor_us", "meas_us", "issue", "HBM_GB/s"))


def find(name):
    key = re.sub(r"\s+", "", name)
    for k, v in isa["kernels"].items():
        if re.sub(r"\s+", "", k) == key:
            return v
    return None


for w, rec in cnt["workloads"].items():
    print("== %s: %s" % (w, rec["command"]))
    for k, m in rec["kernels"].items():
        st = find(k)
        d = m.get("derived", {})
        waves, insts = m.get("SQ_WAVES"), m.get("SQ_INSTS_VALU")
        dur = m.get("duration_us_in_this_pass_passA")
        clk = d.get("effective_clock_ghz")
        if not (st and waves and insts and dur and clk and st["valu"]):
            continue
        per_wave = insts / waves
        c = st["issue_cycles_per_wave"] / st["valu"]
        floor_us = waves * per_wave * c / (1024 * clk * 1e3)
        hbm = (d.get("fetch_bytes_x2_guide_correction", 0) + d.get("write_bytes", 0)) / (dur * 1e-6) / 1e9
        mark = "" if abs(per_wave - st["valu"]) / st["valu"] < 0.02 or st["has_branch"] and per_wave > st["valu"] else "*"
        print("%-40s %9.0f %9.0f %8d %8.2f %7.2f %9.1f %9.1f %7.2f %9.0f %s" % (k[:40], waves, per_wave, st["valu"], c, clk, floor_us, dur, floor_us / dur, hbm, mark))
