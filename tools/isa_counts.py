#!/usr/bin/env python3
"""Instruction counts of every kernel of libfhe_hip.so from the gfx950 ISA hipcc emits (runs anywhere hipcc does -- no GPU).

For each kernel: the straight-line instruction mix (the kernels are fully unrolled; the few that loop over ciphertext
terms are marked `has_loop`, their counts are per trip of the unrolled body) and the ISSUE CYCLES one wave needs on its
SIMD if nothing ever stalls -- the ceiling the VALU-issue-bound kernels are measured against (tools/issue_roofline.py):

  class      instructions                                        cycles per wave64 instruction on one SIMD
  mad64      v_mad_u64_u32, v_mad_i64_i32                        4.87   (52.5 lanes/clk/CU measured, tools/ubench.hip)
  mulhi      v_mul_hi_u32, v_mul_hi_i32                          7.76   (33.0)
  mullo      v_mul_lo_u32                                        8.87   (28.9)
  fp64       v_fma_f64 v_mul_f64 v_add_f64 v_rndne_f64 ...       4.35   (57.7 - 60.4)
  alu64      v_lshl_add_u64 v_lshlrev_b64 v_lshrrev_b64 v_cmp_*_u64 v_ashrrev_i64    4.0   (half rate)
  alu32      every other VALU instruction                        2.0   (128 lanes/clk/CU: the FP32 / INT32 vector rate)
  s_nop N    hazard padding                                      N + 1
  lds / vmem / salu / smem                                        counted, not priced (they issue beside the VALU)

Writes profiles/<tag>_isa_counts.json with the hash of the kernel sources.  usage: python tools/isa_counts.py [tag]"""
import collections, hashlib, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fully-homomorphic-image-processing_amd", "csrc")
CYCLES = {"mad64": 4.87, "mulhi": 7.76, "mullo": 8.87, "fp64": 4.35, "alu64": 4.0, "alu32": 2.0}


def source_hash():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()[:16]


def classify(op):
    if op in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return "mad64"
    if op.startswith("v_mul_hi_"):
        return "mulhi"
    if op.startswith("v_mul_lo_u32"):
        return "mullo"
    if op.endswith("_f64") or op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_cvt_f64", "v_fract_f64", "v_trunc_f64", "v_floor_f64", "v_ceil_f64", "v_ldexp_f64", "v_max_f64", "v_min_f64")):
        return "fp64"
    if re.match(r"v_(lshl_add_u64|lshlrev_b64|lshrrev_b64|ashrrev_i64|cmp_\w+_[ui]64|cmpx_\w+_[ui]64)", op):
        return "alu64"
    if op.startswith("v_"):
        return "alu32"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "isa"
    out = {"kernel_source_hash": source_hash(), "cycles_per_wave_instruction": CYCLES, "tool": "tools/isa_counts.py", "kernels": {}}
    filt = "/usr/bin/c++filt"
    for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
        asm = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", "-", os.path.join(CSRC, src)],
                             capture_output=True, text=True, check=True).stdout
        # kernels: symbols that have an .amdhsa_kernel descriptor
        names = re.findall(r"\.amdhsa_kernel\s+(\S+)", asm)
        for sym in names:
            m = re.search(r"^%s:[^\n]*\n(.*?)\n\.Lfunc_end" % re.escape(sym), asm, re.S | re.M)
            if not m:
                continue
            cnt, nop_cycles, ops = collections.Counter(), 0, collections.Counter()
            has_loop = False
            for line in m.group(1).split("\n"):
                t = line.strip()
                if not t or t.startswith((";", ".")) or t.endswith(":"):
                    continue
                op = t.split()[0]
                c = classify(op)
                cnt[c] += 1
                if c == "s_nop":
                    nop_cycles += int(t.split()[1], 0) + 1
                if op.startswith(("s_cbranch", "s_branch")):
                    has_loop = True
                if c in ("mad64", "mulhi", "mullo", "fp64"):
                    ops[op] += 1
            issue = sum(cnt[k] * v for k, v in CYCLES.items()) + nop_cycles
            dem = subprocess.run([filt, sym], capture_output=True, text=True).stdout.strip() if os.path.exists(filt) else sym
            short = re.sub(r"^void ", "", dem)
            short = re.sub(r"\(anonymous namespace\)::", "", short)
            short = short[:short.index("(")] if "(" in short else short
            vg = re.search(r"\.amdhsa_kernel\s+%s\n(.*?)\.end_amdhsa_kernel" % re.escape(sym), asm, re.S)
            vgpr = None
            if vg:
                mm = re.search(r"\.amdhsa_next_free_vgpr\s+(\d+)", vg.group(1))
                vgpr = int(mm.group(1)) if mm else None
            out["kernels"][short] = {"file": src, "valu": sum(cnt[k] for k in CYCLES), "classes": dict(cnt), "multiplier_ops": dict(ops), "nop_cycles": nop_cycles,
                                     "issue_cycles_per_wave": round(issue, 1), "has_branch": has_loop, "next_free_vgpr": vgpr}
    path = os.path.join(ROOT, "profiles", "%s_isa_counts.json" % tag)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in sorted(out["kernels"].items()):
        print("%-60s valu %6d  mad64 %5d mulhi %4d mullo %4d fp64 %5d  issue %8.0f cyc/wave%s" % (k[:60], v["valu"], v["classes"].get("mad64", 0), v["classes"].get("mulhi", 0),
              v["classes"].get("mullo", 0), v["classes"].get("fp64", 0), v["issue_cycles_per_wave"], "  (branches)" if v["has_branch"] else ""))
    print("wrote", path)


if __name__ == "__main__":
    main()
