"""Probe helper: from a rocprofv3 kernel-trace csv, busy time and gaps of the sweep launches in windows of 2000."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
sw = [r for r in rows if "sweep_kernel" in r[2]]
print("sweep launches:", len(sw))
W = 2000
for i in range(0, len(sw) - W + 1, W):
    w = sw[i:i + W]
    busy = sum(e - s for s, e, _ in w)
    span = w[-1][1] - w[0][0]
    gaps = sorted((w[j + 1][0] - w[j][1] for j in range(W - 1)), reverse=True)
    nred = sum(1 for r in w if "true, true>" in r[2] or ", true>(" in r[2])
    print(f"window {i // W:3d}: span {span / W / 1e3:7.2f} us/launch, busy {busy / W / 1e3:7.2f}, idle {(span - busy) / W / 1e3:7.2f}; "
          f"largest gaps us {[round(g / 1e3) for g in gaps[:5]]}; gaps > 20 us: {sum(1 for g in gaps if g > 20000)}")
