#!/usr/bin/env python3
"""Instruction histogram of one kernel of a hipcc -S listing, per basic block (compile only, no GPU needed).
  usage: tools/isa_hist.py listing.s 'substring of the demangled kernel name' [min_block_size]
Prints every basic block with at least min_block_size instructions: VALU split into full-rate (f32 add/mul/fma, mov,
add_u32) and the rest (the ~4.5-cycle classes of profiles/r02_microbench_oprate.txt), DPP, LDS, VMEM, SALU."""
import re, subprocess, sys, collections

FULL = re.compile(r"^v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|mac_f32|mov_b32|add_u32|sub_u32|subrev_u32|fmaak_f32|fmamk_f32|madak_f32|madmk_f32|mul_legacy_f32)(_e32|_e64)?$")

def main():
    path, want = sys.argv[1], sys.argv[2]
    minb = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    syms = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(s for _, s in syms), capture_output=True, text=True).stdout.split("\n")
    start = end = None
    for k, ((i, s), d) in enumerate(zip(syms, dem)):
        if want in d:
            start = i; end = syms[k + 1][0] if k + 1 < len(syms) else len(lines)
            print("kernel:", d[:160]); break
    if start is None:
        sys.exit("kernel not found")
    blocks, cur, name = [], collections.Counter(), "entry"
    tot = collections.Counter()
    def cls(op):
        if op.startswith("v_"):
            if "dpp" in op: return "valu_dpp"
            return "valu_full" if FULL.match(op) else "valu_other"
        if op.startswith("ds_"): return "lds"
        if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
        if op.startswith("s_waitcnt"): return "wait"
        if op.startswith("s_"): return "salu"
        return "other"
    ops_in = collections.defaultdict(collections.Counter)
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            blocks.append((name, cur)); cur, name = collections.Counter(), m.group(1); continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
        if not m or l.strip().startswith((";", ".")): continue
        op = m.group(1)
        if " dpp" in l or "row_" in l or "wave_sh" in l: op += "_dpp"
        c = cls(op); cur[c] += 1; tot[c] += 1; ops_in[name][op] += 1
    blocks.append((name, cur))
    for n, c in blocks:
        s = sum(c.values())
        if s >= minb:
            print("%-14s n=%4d  full %4d other %4d dpp %3d | lds %3d vmem %3d salu %3d wait %3d" % (n, s, c["valu_full"], c["valu_other"], c["valu_dpp"], c["lds"], c["vmem"], c["salu"], c["wait"]))
            if len(sys.argv) > 4:
                print("    ", ", ".join("%s %d" % kv for kv in ops_in[n].most_common(40)))
    print("total", dict(tot))

main()
