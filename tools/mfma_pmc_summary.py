"""rocprofv3 --pmc counter_collection.csv of `python tools/conv_bench.py 512 o` -> per-kernel means:
MFMA instructions, other VALU instructions per MFMA, matrix-pipe busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)).
usage: mfma_pmc_summary.py counter_collection.csv > summary.json"""
import collections
import csv
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if not any(k in name for k in ("igemm_kernel", "igemm_occ_kernel", "igemm_split_kernel", "igemm_u8_kernel", "wgrad_fast_kernel",
                                        "wgrad_split_kernel", "wgrad_u8_kernel", "bwd_pair_kernel")):
            continue
        name = name[:name.index("(")]           # (template arguments kept: they tell the tile shapes and routes apart)
        acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (name, grid), c in acc.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    mfma, valu = m.get("SQ_INSTS_MFMA", 0.), m.get("SQ_INSTS_VALU", 0.)
    gui = m.get("GRBM_GUI_ACTIVE", 0.) / 8.
    rows.append(dict(kernel=name, grid=grid, launches=len(c["GRBM_GUI_ACTIVE"]), mfma_insts=mfma, valu_insts=valu,
                     valu_per_mfma=round((valu - mfma) / mfma, 2) if mfma else None,
                     mfma_busy_cycles=m.get("SQ_VALU_MFMA_BUSY_CYCLES"), gui_active_per_xcd=round(gui),
                     mfma_pipe_util=round(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.) / (1024. * gui), 3) if gui else None))
rows.sort(key=lambda r: -r["mfma_insts"])
print(json.dumps(dict(
    note="rocprofv3 --pmc (one pass, 8 SQ/GRBM counters) over `python tools/conv_bench.py 512 o`: means per kernel and "
         "grid; mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8 XCDs), i.e. the fraction of "
         "SIMD-cycles (at the counter's clock) the matrix pipe was busy while the kernel ran",
    kernels=rows), indent=1))
