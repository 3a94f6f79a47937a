"""Instruction census of one kernel's innermost loop(s) from hipcc's assembly (-S --cuda-device-only).

  python scripts/isa_census.py <file.s> <mangled-kernel-name-substring> [out.json]

Finds the kernel's body, takes every backward branch as a loop [target label .. branch], and counts the instructions
of each loop by category.  For the rollout the time loop is unrolled twice (two named register sets): one loop
iteration = two steps."""
import collections
import json
import re
import sys

CATS = [
    ("fp64_fma", r"^v_fma_f64|^v_fmac_f64"), ("fp64_mul", r"^v_mul_f64"), ("fp64_add", r"^v_add_f64"),
    ("fp64_minmax", r"^v_(min|max)_f64"), ("fp64_rcp_rndne_etc", r"^v_(rcp|rndne|div_|trig|frexp|ldexp|rsq|sqrt)\w*_f64|^v_cvt_\w*f64|^v_cvt_f64"),
    ("fp64_cmp", r"^v_cmp\w*_f64"), ("valu_cmp_other", r"^v_cmp"), ("valu_select_mov", r"^v_cndmask|^v_mov|^v_accvgpr|^v_readlane|^v_readfirstlane|^v_writelane"),
    ("valu_int_logic", r"^v_(and|or|xor|lshl|lshr|ashr|add_u|add_co|addc|sub|bfe|not|mad_u|mul_lo|mul_hi|add3|lshl_add|and_or|bfi|perm|alignbit)"),
    ("valu_other", r"^v_"), ("vmem_load", r"^(global|buffer|flat)_load"), ("vmem_store", r"^(global|buffer|flat)_store"),
    ("lds", r"^ds_"), ("smem", r"^s_load|^s_buffer_load"), ("waitcnt", r"^s_waitcnt"), ("branch", r"^s_cbranch|^s_branch"),
    ("salu", r"^s_"),
]


def classify(op):
    for name, pat in CATS:
        if re.match(pat, op):
            return name
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(key) or
                 (l.startswith("_Z") and key in l.split(":")[0]))
    # the function's end, not its first s_endpgm: a kernel with an early exit has one in the middle of its body
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {}
    insts = []
    for l in body:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\w+):", s)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        insts.append(s)
    loops = []
    for i, s in enumerate(insts):
        m = re.match(r"^s_cbranch\w*\s+(\.LBB\w+)|^s_branch\s+(\.LBB\w+)", s)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, bool(m.group(1))))
    out = {"kernel": key, "total_instructions": len(insts), "loops": []}
    for a, b, cond in loops:
        c = collections.Counter(classify(s.split()[0]) for s in insts[a:b + 1])
        # conditional: closed by an s_cbranch (a loop); an unconditional backward s_branch is an out-of-line block
        # (a slow path placed behind the loop) jumping back into it
        d = {"first": a, "last": b, "conditional": cond, "instructions": b - a + 1, "by_category": dict(sorted(c.items(), key=lambda kv: -kv[1]))}
        d["valu"] = sum(v for k, v in c.items() if k.startswith(("fp64", "valu")))
        d["fp64_flops_per_lane"] = 2 * c.get("fp64_fma", 0) + c.get("fp64_mul", 0) + c.get("fp64_add", 0)
        out["loops"].append(d)
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
