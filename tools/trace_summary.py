"""Summarises a PNYOLO_TRACE_FILE written by the diagnostic build (tools/stamp_build.sh)."""
import sys
import numpy as np

txt = open(sys.argv[1]).read().split('#\n')
blk = txt[0].strip().split('\n')
T = np.array([[int(x) for x in l.split()] for l in blk], dtype=np.int64)
T = T - T[:, 0].min()
d = np.array([T[:, k + 3] - T[:, k + 2] for k in range(0, 40, 4)])
print("fc0 GEMM duration per wave (mean over 10 blocks):", d.mean(0).astype(int), " slowest-wave mean:",
      int(d.max(1).mean()), "(ideal 131072)")
C = T[0, 2::4]
print("cycles per tile:", C[11] - C[0], "(ideal 4096000)")
