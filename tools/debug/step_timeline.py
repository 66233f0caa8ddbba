# Analyse a rocprofv3 kernel trace of `bench.py --mode train [--train-encoder]`: one steady-state step's timeline by kernel
# class, the GPU-busy union and the largest idle gaps.   usage: step_timeline.py <dir with *kernel_trace.csv>
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "repack_kernel" in r["Kernel_Name"]]
a, b = marks[-3], marks[-2]
step = rows[a:b]
T0 = int(step[0]["Start_Timestamp"]); T1 = int(rows[b]["Start_Timestamp"])
def cls(n):
    if any(k in n for k in ("trunk_", "bn_", "conv_dw", "maxpool_bwd", "upsample_bwd", "add2", "transpose")): return "trunk"
    if "conv_mfma" in n or "maxpool_kernel" in n or "upsample_concat" in n or "image_to" in n: return "trunk"
    if "latent_grad" in n: return "latgrad"
    if "pny_mlp_h2_kernel" in n or "pny_mlp_kernel" in n: return "mlp fwd"
    if "mlp_bwd" in n: return "mlp chain"
    if "dw_gemm" in n or "dw_reduce" in n: return "mlp dW"
    if "pny::" in n: return "render misc"
    return "aten"
print("step span %.2f ms, %d kernels" % ((T1 - T0) / 1e6, len(step)))
busy = collections.defaultdict(float); first = {}; last = {}; cnt = collections.Counter()
for r in step:
    c = cls(r["Kernel_Name"]); s_, e_ = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy[c] += (e_ - s_) / 1e6; cnt[c] += 1
    first.setdefault(c, (s_ - T0) / 1e6); last[c] = (e_ - T0) / 1e6
for c in sorted(busy, key=lambda c: first[c]):
    print("  %-12s n=%3d  kernel-ms %6.2f   first %.2f  last %.2f" % (c, cnt[c], busy[c], first[c], last[c]))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in step)
u = 0; cs, ce = iv[0]; gaps = []
for s_, e_ in iv[1:]:
    if s_ > ce:
        gaps.append((s_ - ce, (ce - T0) / 1e6)); u += ce - cs; cs, ce = s_, e_
    else: ce = max(ce, e_)
u += ce - cs
print("GPU busy (union) %.2f ms; idle gaps > 50 us:" % (u / 1e6), ", ".join("%.0f us @%.2f" % (g / 1e3, t) for g, t in sorted(gaps, reverse=True)[:12] if g > 5e4))
