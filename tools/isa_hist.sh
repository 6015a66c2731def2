#!/bin/bash
# Developer tool: static instruction mix of the routing kernels as built for gfx950 (no GPU needed).
#     tools/isa_hist.sh > profiles/rNN_isa_mix.txt
# Compiles the device side of trmc.hip to assembly with the flags of the shipped build and, for every kernel named
# below, counts the instructions between its label and its s_endpgm by opcode class.
set -e
cd "$(dirname "$0")/../t-route_amd/csrc"
T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off --offload-device-only -S trmc.hip -o $T/trmc.s
python3 - "$T/trmc.s" <<'EOF'
import collections
import re
import subprocess
import sys

want = ("k_mc_ctile", "k_mc_tile", "k_mc_step", "k_mc_flow_lean", "k_mc_flow", "k_emit")
lines = open(sys.argv[1]).read().splitlines()


def demangle(s):
    return subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()


def cls(op):
    if op.startswith(("v_cmp", "v_cmpx")):
        return "vector compare"
    if op.startswith("v_cndmask"):
        return "vector select"
    if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_permlane", "v_swap")):
        return "vector move"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup", "v_rcp", "v_rsq", "v_sqrt", "v_frexp", "v_ldexp")):
        return "division / root helpers"
    if op.startswith(("v_min", "v_max", "v_med3")):
        return "vector min / max"
    if op.startswith("v_") and ("_f64" in op):
        return "fp64 arithmetic"
    if op.startswith("v_") and ("_f32" in op):
        return "fp32 arithmetic"
    if op.startswith("v_cvt"):
        return "conversions"
    if op.startswith("v_"):
        return "vector integer / logic"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "memory: global"
    if op.startswith("ds_"):
        return "memory: LDS"
    if op.startswith(("s_load", "s_buffer_load")):
        return "memory: scalar loads"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branches"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier", "s_setprio")):
        return "waits / nops / barriers"
    if op.startswith("s_"):
        return "scalar ALU"
    return "other"


i = 0
print("static instruction mix, gfx950, flags of the shipped build (-O3 -ffp-contract=off); per kernel instance")
while i < len(lines):
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
    if not m:
        i += 1
        continue
    name = demangle(m.group(1))
    if not any(re.search(r"\b" + w + r"\b", name) for w in want):
        i += 1
        continue
    hist = collections.Counter()
    ops = collections.Counter()
    blocks = 0
    j = i + 1
    while j < len(lines):
        s = lines[j].strip()
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks += 1
        mm = re.match(r"^([a-z][a-z0-9_]+)\b", s)
        if mm and not s.startswith("."):
            op = mm.group(1)
            hist[cls(op)] += 1
            ops[op] += 1
            if op == "s_endpgm":
                break
        j += 1
    total = sum(hist.values())
    regs = {}
    for k in range(j, min(len(lines), j + 400)):
        for key in ("NumVgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
            mk = re.search(r";\s*" + key + r":\s*(\d+)", lines[k])
            if mk and key not in regs:
                regs[key] = int(mk.group(1))
    name = name.replace("(anonymous namespace)::", "")
    print(f"\n== {name}\n   {total} instructions in {blocks + 1} basic blocks; " + ", ".join(f"{k} {v}" for k, v in regs.items()))
    for k, v in hist.most_common():
        print(f"   {v:6d}  {100.0 * v / total:5.1f} %  {k}")
    print("   most frequent opcodes: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(14)))
    i = j + 1
EOF
rm -rf $T
