"""Derived per-kernel figures from the tools/counters_report.sh passes (MI355X: 256 CUs, 1024 SIMDs, 8 XCDs).
GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count in units of 4 clocks.
A SIMD issues one wave64 VALU instruction every 2 clocks at best, so VALU utilisation = VALU instructions x 2 / SIMD clocks.
TA busy (the address unit in front of the vector L1) is the direct measure of how loaded the gather path is."""
import collections, csv, glob, re, sys
root = sys.argv[1]
acc = collections.defaultdict(float)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    run = f.split("/")[-3] if f.split("/")[-2].startswith("vm") or True else ""
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel\d*(?:<[\w, ]+>)?)", r["Kernel_Name"])
        if not m:
            continue
        acc[(m.group(1), r["Counter_Name"])] += float(r["Counter_Value"])
kernels = sorted({k for k, _ in acc})
for k in kernels:
    g = lambda c: acc.get((k, c), 0.0)
    gui = g("GRBM_GUI_ACTIVE") / 4.0 / 8.0  # four passes collected it; per XCD
    if gui <= 0 or g("SQ_WAVES") <= 0:
        continue
    simd_clk = gui * 1024.0
    print(f"== {k} ==")
    print(f"  busy clocks per XCD {gui:14.0f}   waves {g('SQ_WAVES'):12.0f}")
    print(f"  mean waves per SIMD (wave clocks / SIMD clocks)        {g('SQ_WAVE_CYCLES') * 4 / simd_clk:6.2f}")
    print(f"  VALU utilisation (VALU instr x 2 clk / SIMD clk)        {g('SQ_INSTS_VALU') * 2 / simd_clk:6.3f}")
    print(f"  wave time: waiting {g('SQ_WAIT_ANY') / max(g('SQ_WAVE_CYCLES'), 1):5.3f}  issue-stalled {g('SQ_WAIT_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):5.3f}"
          f"  issuing {g('SQ_ACTIVE_INST_ANY') / max(g('SQ_WAVE_CYCLES'), 1):5.3f}")
    print(f"  per SIMD clock: VALU pipe active {g('SQ_ACTIVE_INST_VALU') * 4 / simd_clk:5.3f}  scalar unit active {g('SQ_ACTIVE_INST_SCA') * 4 / simd_clk / 2:5.3f}"
          f"  LDS issue active {g('SQ_ACTIVE_INST_LDS') * 4 / simd_clk / 2:5.3f}  (the second and third were collected in another pass)")
    print(f"  wave lifetime (clocks) {g('SQ_WAVE_CYCLES') * 4 / g('SQ_WAVES'):12.0f}   VALU instructions x 4 clk / wave lifetime {g('SQ_INSTS_VALU') * 4 / max(g('SQ_WAVE_CYCLES') * 4, 1):5.3f}")
    print(f"  per wave: VALU {g('SQ_INSTS_VALU') / g('SQ_WAVES'):9.0f}  SALU {g('SQ_INSTS_SALU') / g('SQ_WAVES'):9.0f}  VMEM rd {g('SQ_INSTS_VMEM_RD') / g('SQ_WAVES'):8.0f}"
          f"  VMEM wr {g('SQ_INSTS_VMEM_WR') / g('SQ_WAVES'):8.0f}  LDS {g('SQ_INSTS_LDS') / g('SQ_WAVES'):8.0f}")
    if g("TCC_REQ_sum"):
        print(f"  L2: hit rate {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):5.3f}  requests {g('TCC_REQ_sum'):14.0f}  fabric reads {g('TCC_EA0_RDREQ_sum'):14.0f}")
    if g("TCP_TOTAL_ACCESSES_sum"):
        print(f"  L1: TA busy {g('TA_TA_BUSY_sum') / 256 / gui:5.3f}  L1 clock-enabled {g('TCP_GATE_EN1_sum') / 256 / gui:5.3f}  accesses {g('TCP_TOTAL_ACCESSES_sum'):14.0f}"
              f"  L1->L2 read requests {g('TCP_TCC_READ_REQ_sum'):14.0f}")
