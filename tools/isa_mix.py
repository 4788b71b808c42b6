"""Instruction mix of the MFMA kernels from hipcc's device assembly: per kernel, vector-ALU / MFMA / LDS / buffer /
scalar instruction counts inside the basic blocks that hold MFMAs (the main loop) and outside them (prologue +
epilogue), plus the register budget.  usage: hipcc ... -S --cuda-device-only mfma_conv.hip -o x.s; python
tools/isa_mix.py x.s [name-filter]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else "igemm"
i = 0
while i < len(lines):
    m = re.match(r"^(_Z\S+):\s", lines[i])
    if not m or flt not in m.group(1):
        i += 1; continue
    name = m.group(1)
    blocks, cur = [], []
    i += 1
    while i < len(lines) and not lines[i].startswith("\t.end_amdhsa_kernel") and not re.match(r"^_Z\S+:\s", lines[i]):
        ln = lines[i].strip()
        if re.match(r"^\.LBB\S+:", ln):
            blocks.append(cur); cur = []
        elif ln and not ln.startswith((".", ";")):
            cur.append(ln.split()[0])
        i += 1
    blocks.append(cur)
    meta = {}
    j = i
    while j < len(lines) and j < i + 400:
        for key in ("vgpr_count", "sgpr_count", "accum_offset", "NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "LDSByteSize"):
            mm = re.search(r"[;.]\s*%s:?\s+(\d+)" % key, lines[j])
            if mm and key not in meta: meta[key] = int(mm.group(1))
        j += 1
    def cls(op):
        if op.startswith("v_mfma"): return "mfma"
        if op.startswith("v_"): return "valu"
        if op.startswith("ds_"): return "lds"
        if op.startswith(("buffer_", "global_", "flat_", "scratch_")): return "vmem"
        if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): return "wait"
        if op.startswith("s_"): return "salu"
        return "other"
    inside, outside = collections.Counter(), collections.Counter()
    for b in blocks:
        c = collections.Counter(cls(o) for o in b)
        (inside if c["mfma"] >= 8 else outside).update(c)
    short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)[:70]
    print("%-70s loop: mfma %4d valu %4d lds %4d vmem %3d salu %4d | outside: valu %4d lds %3d vmem %3d salu %4d mfma %3d | %s" % (
        short, inside["mfma"], inside["valu"], inside["lds"], inside["vmem"], inside["salu"],
        outside["valu"], outside["lds"], outside["vmem"], outside["salu"], outside["mfma"],
        " ".join("%s=%s" % kv for kv in sorted(meta.items()))))
