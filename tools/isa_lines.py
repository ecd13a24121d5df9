#!/usr/bin/env python3
"""Per-source-line instruction counts of one kernel (compile with -gline-tables-only -S; no GPU needed).
  usage: tools/isa_lines.py listing.s 'substring of the demangled kernel name' [first_label last_label] [top N]
Every instruction is charged to the innermost source line of its `.loc` (inlined helpers are charged to THEIR lines: the table
says what kind of work the instructions do, not who asked for it). Classes as in isa_hist.py; `cyc` prices the VALU classes with
the measured issue costs of profiles/r02_microbench_oprate.txt (2.6 / 4.5 cycles; transcendental 8.5)."""
import re, subprocess, sys, collections
FULL = re.compile(r"^v_(add_f32|sub_f32|subrev_f32|mul_f32|fma_f32|fmac_f32|mac_f32|mov_b32|add_u32|sub_u32|subrev_u32|fmaak_f32|fmamk_f32|madak_f32|madmk_f32|mul_legacy_f32|ashrrev_i32)(_e32|_e64)?$")
TRANS = re.compile(r"^v_(rcp|rsq|sqrt|exp|log|sin|cos)_")
def main():
    path, want = sys.argv[1], sys.argv[2]
    lo = sys.argv[3] if len(sys.argv) > 4 else None
    hi = sys.argv[4] if len(sys.argv) > 4 else None
    top = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    lines = open(path).read().split("\n")
    files = {}
    for l in lines:
        m = re.match(r'^\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
        if m: files[int(m.group(1))] = m.group(2)
    syms = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    dem = subprocess.run(["c++filt"], input="\n".join(s for _, s in syms), capture_output=True, text=True).stdout.split("\n")
    start = end = None
    for k, ((i, s), d) in enumerate(zip(syms, dem)):
        if want in d:
            start = i; end = syms[k + 1][0] if k + 1 < len(syms) else len(lines); print("kernel:", d[:150]); break
    if start is None: sys.exit("kernel not found")
    cnt = collections.defaultdict(collections.Counter)
    cur = ("?", 0); on = lo is None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\w+):", l)
        if m and lo is not None:
            if m.group(1) == lo: on = True
            elif m.group(1) == hi: on = False
        m = re.match(r"^\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m: cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2))); continue
        if not on: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s", l + " ")
        if not m or l.strip().startswith((";", ".")): continue
        op = m.group(1)
        if op.startswith("v_"):
            c = "dpp" if (" dpp" in l or "row_" in l or "wave_sh" in l) else ("full" if FULL.match(op) else ("trans" if TRANS.match(op) else "other"))
        elif op.startswith("ds_"): c = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c = "wait"
        elif op.startswith("s_"): c = "salu"
        else: c = "misc"
        cnt[cur][c] += 1
    def cyc(c): return 2.6 * c["full"] + 4.5 * (c["other"] + c["dpp"]) + 8.5 * c["trans"]
    tot = collections.Counter()
    for c in cnt.values(): tot.update(c)
    print("total: valu %d (full %d other %d dpp %d trans %d) lds %d vmem %d salu %d  ~%.0f VALU issue cycles" % (
        tot["full"] + tot["other"] + tot["dpp"] + tot["trans"], tot["full"], tot["other"], tot["dpp"], tot["trans"], tot["lds"], tot["vmem"], tot["salu"], cyc(tot)))
    src = {}
    for (f, ln), c in sorted(cnt.items(), key=lambda kv: -cyc(kv[1]))[:top]:
        if f not in src:
            try: src[f] = open("differender_amd/csrc/" + f).read().split("\n")
            except OSError: src[f] = []
        text = src[f][ln - 1].strip()[:90] if 0 < ln <= len(src[f]) else ""
        print("%-20s %5d  full %3d other %3d dpp %3d tr %2d lds %3d salu %3d  cyc %5.0f | %s" % (f[-20:], ln, c["full"], c["other"], c["dpp"], c["trans"], c["lds"], c["salu"], cyc(c), text))
main()
