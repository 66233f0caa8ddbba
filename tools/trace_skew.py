"""Barrier-arrival skew per wave from a PNYOLO_TRACE_FILE (diagnostic build): for every block entry of
workgroup 0, how long each wave waited at the first barrier, and the duration of its fc_0 GEMM."""
import sys
import numpy as np

txt = open(sys.argv[1]).read().split('#\n')
for bi, blk in enumerate(txt[:2]):
    rows = [l for l in blk.strip().split('\n') if l.strip()]
    if not rows:
        continue
    T = np.array([[int(x) for x in l.split()] for l in rows], dtype=np.int64)
    n_ev = (T > 0).sum(1).min() // 4 * 4
    A, B, C, D = (T[:, k:n_ev:4] for k in range(4))     # arrive sync1, past sync1, past sync2, end fc_0 GEMM
    wait = (B - A)                                       # per wave, per block entry
    gemm = (D - C)
    np.set_printoptions(linewidth=200)
    print("launch %d: %d waves, %d block entries" % (bi, T.shape[0], A.shape[1]))
    print("  mean wait at sync1 per wave:", wait.mean(1).astype(int))
    print("  mean fc_0 GEMM ticks per wave:", gemm.mean(1).astype(int))
    late = (A == A.max(0, keepdims=True))
    print("  how often each wave is the last to arrive:", late.sum(1))
    print("  mean spread (last - first arrival):", int((A.max(0) - A.min(0)).mean()), " mean GEMM:", int(gemm.mean()))
