#!/usr/bin/env python
"""Stall attribution of the fp32 LDS-DMA GEMM from the per-wave cycle stamps tools/probe/gemm_timeline writes.

    python tools/gemm_timeline.py trace.csv [mfma_per_ktile]

For every wave: prologue (entry -> first barrier), each k tile (barrier to barrier), epilogue.  Waves are grouped by the SIMD
they ran on (XCC, SE, CU, SIMD from HW_ID): a SIMD's MFMA pipe needs `mfma_per_ktile x 64` cycles per k tile of each of its
waves (v_mfma_f32_32x32x2_f32 = 16 passes x 4 cycles), so `needed / span` is the pipe utilisation the timeline implies, and
the k-tile durations at each residency level say where the rest goes."""
import sys
from collections import defaultdict

import numpy as np


def main():
    path = sys.argv[1]
    head = open(path).readline()
    meta = dict(kv.split("=") for kv in head[1:].split())
    nk = int(meta["nk"])
    cfg = int(meta["cfg"])
    wave_tiles = {8: 4, 9: 2, 10: 2, 11: 1}.get(cfg, 4)          # 32x32 accumulator tiles per wave
    bk = 16
    mfma_per_ktile = int(sys.argv[2]) if len(sys.argv) > 2 else wave_tiles * bk // 2
    rows = np.loadtxt(path, delimiter=",", comments="#", dtype=np.int64)
    wg, wave, xcc, se, cu, simd, rt0, rt1 = (rows[:, i] for i in range(8))
    c = rows[:, 8:8 + nk + 3]
    t_in, t_pro, t_loop_end, t_out = c[:, 0], c[:, 1], c[:, nk + 1], c[:, nk + 2]
    span_cyc = t_out.max() - t_in.min()
    span_rt = (rt1.max() - rt0.min()) / 100.0          # us (100 MHz)
    print(f"{meta}  waves={len(rows)}")
    print(f"kernel span: {span_rt:.1f} us, {span_cyc} shader cycles -> {span_cyc / span_rt:.0f} MHz average clock")
    kt = np.diff(c[:, 1:nk + 2], axis=1)                 # [waves, nk]
    pro, epi, loop = t_pro - t_in, t_out - t_loop_end, t_loop_end - t_pro
    q = lambda a: "p10 %d  p50 %d  p90 %d  max %d" % tuple(np.percentile(a, [10, 50, 90, 100]))  # noqa: E731
    print(f"per wave   prologue: {q(pro)}\n           k tile:   {q(kt.ravel())}   (pipe time of ONE wave's k tile: {mfma_per_ktile * 64})\n"
          f"           main loop: {q(loop)}\n           epilogue: {q(epi)}")
    tot = (pro + loop + epi).astype(float)
    print(f"share of a wave's life: prologue {pro.sum() / tot.sum():.3f}, main loop {loop.sum() / tot.sum():.3f}, epilogue {epi.sum() / tot.sum():.3f}")
    # per SIMD
    groups = defaultdict(list)
    for i in range(len(rows)):
        groups[(xcc[i], se[i], cu[i], simd[i])].append(i)
    util, nw = [], []
    res_dur = defaultdict(list)                          # residency level -> k-tile durations
    for key, idx in groups.items():
        idx = np.array(idx)
        need = len(idx) * nk * mfma_per_ktile * 64
        span = t_out[idx].max() - t_in[idx].min()
        util.append(need / span)
        nw.append(len(idx))
        # residency of the SIMD (waves between entry and exit) at the midpoint of every k tile of every wave
        starts, ends = np.sort(t_in[idx]), np.sort(t_out[idx])
        for i in idx:
            mid = (c[i, 1:nk + 1] + c[i, 2:nk + 2]) // 2
            res = np.searchsorted(starts, mid, side="right") - np.searchsorted(ends, mid, side="right")
            for r, d in zip(res, kt[i]):
                res_dur[int(r)].append(d)
    util = np.array(util)
    print(f"SIMDs seen: {len(groups)} (waves per SIMD: min {min(nw)} median {int(np.median(nw))} max {max(nw)})")
    print(f"MFMA pipe utilisation implied per SIMD over ITS OWN span: mean {util.mean():.3f}  p10 {np.percentile(util, 10):.3f}  p90 {np.percentile(util, 90):.3f}")
    total_need = len(rows) * nk * mfma_per_ktile * 64
    print(f"over the KERNEL span (all {len(groups)} SIMDs): {total_need / (len(groups) * span_cyc):.3f}")
    print("k-tile duration by SIMD residency (waves resident on the SIMD), against the pipe-bound time r x one wave's pipe time:")
    for r in sorted(res_dur):
        d = np.array(res_dur[r])
        print(f"  r={r}: n={len(d):7d}  median {int(np.median(d)):6d}  mean {d.mean():8.0f}  p90 {int(np.percentile(d, 90)):6d}   pipe-bound {r * mfma_per_ktile * 64:6d}"
              f"   -> efficiency {r * mfma_per_ktile * 64 / d.mean():.3f}  (share of k tiles {len(d) / kt.size:.3f})")
    # start skew: when do workgroups start
    first = t_in.min()
    st = (t_in[wave == 0] - first)
    print(f"workgroup start times (cycles after the first): p50 {int(np.percentile(st, 50))}  p90 {int(np.percentile(st, 90))}  max {st.max()}")
    en = t_out.max() - t_out[wave == 0]
    print(f"workgroup end times (cycles before the last):   p50 {int(np.percentile(en, 50))}  p10 {int(np.percentile(en, 10))}")


if __name__ == "__main__":
    main()
