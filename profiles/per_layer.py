"""Per-launch breakdown of one recognizer step from a rocprofv3 --kernel-trace CSV.
usage: python profiles/per_layer.py <kernel_trace.csv> [lines=1024] [width=256]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rocclr" not in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
w = int(sys.argv[3]) if len(sys.argv) > 3 else 256
T, w2, w4 = w // 4 - 1, w // 2, w // 4
step = rows[-15:]
flops = [2 * n * 32 * w * 64 * 9, 2 * n * 16 * w2 * 128 * 576, 2 * n * 8 * w4 * 256 * 1152, 2 * n * 8 * w4 * 256 * 2304,
         2 * n * 4 * w4 * 512 * 2304, 2 * n * 4 * w4 * 512 * 4608, 2 * n * T * 512 * 2048,
         2 * n * T * 2048 * 512, 2 * n * T * 2 * 1024 * 256 * 2, 2 * n * T * 256 * 512, 2 * n * T * 2048 * 256,
         2 * n * T * 2 * 1024 * 256 * 2, 2 * n * T * 256 * 512, 2 * n * T * 95 * 256, None]
names = ["conv_first", "L1 64->128 p2x2", "L2 128->256", "L3 256->256 p2x1", "L4 256->512", "L5 512->512 p2x1",
         "L6 2x2 512->512", "xproj1 512->2048", "lstm1", "lin1 512->256", "xproj2 256->2048", "lstm2", "lin2",
         "pred 256->95", "ctc"]
tot = 0.0
for r, nm, f in zip(step, names, flops):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{nm:20s} {d:9.1f} us  grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:<4s} vgpr={r['VGPR_Count']:>3s}+{r['Accum_VGPR_Count']:<3s} "
          + (f"{f / d / 1e6:7.0f} TFLOP/s" if f else ""))
print(f"sum of kernels {tot:.1f} us; wall first->last "
      f"{(int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3:.1f} us")
