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
    phases = int(meta.get("phases", 1))
    if phases == 3:      # three stamps per k tile: MFMAs issued | own DMA landed | barrier passed
        full = rows[:, 8:8 + 3 * nk + 3]
        body = full[:, 2:2 + 3 * nk].reshape(len(rows), nk, 3)
        prev = np.concatenate([full[:, 1:2], body[:, :-1, 2]], axis=1)            # start of each k tile
        comp, dma, bar = body[:, :, 0] - prev, body[:, :, 1] - body[:, :, 0], body[:, :, 2] - body[:, :, 1]
        tot = (comp + dma + bar).sum()
        pq = lambda a: "p50 %d  p90 %d  mean %.0f" % (np.percentile(a, 50), np.percentile(a, 90), a.mean())  # noqa: E731
        print(f"phases of a k tile (cycles): issue DMA + reads + MFMAs  {pq(comp)}  ({comp.sum() / tot:.3f} of the loop)\n"
              f"                             wait for own DMA (vmcnt 0)  {pq(dma)}  ({dma.sum() / tot:.3f})\n"
              f"                             wait at the barrier         {pq(bar)}  ({bar.sum() / tot:.3f})")
        age = np.argsort(np.argsort(rows[:, 6]))      # nothing: placeholder for symmetry
        rows = np.concatenate([rows[:, :8], full[:, :2], body[:, :, 2], full[:, -1:]], axis=1)
    c = rows[:, 8:8 + nk + 3].copy()
    # The shader cycle counter (s_memtime) is NOT one chip-wide clock: different CUs report unrelated offsets (measured: spans of
    # 5.6M .. 17.8M "cycles" inside one XCD for a 0.41 ms kernel).  Durations inside a wave and comparisons between waves of ONE
    # CU are meaningful; for the chip-wide picture every CU's stamps are shifted so that its first wave entry sits at that
    # wave's realtime stamp (100 MHz, chip-wide) expressed in cycles of the measured average clock.
    life_cyc = (c[:, nk + 2] - c[:, 0]).sum()
    life_ticks = (rt1 - rt0).sum()
    clk_per_tick = life_cyc / life_ticks
    cu_key = (xcc * 64 + se * 16 + cu)
    for key in np.unique(cu_key):
        m = np.nonzero(cu_key == key)[0]
        first = m[np.argmin(c[m, 0])]
        c[m] += int((rt0[first] - rt0.min()) * clk_per_tick) - c[first, 0]
    t_in, t_pro, t_loop_end, t_out = c[:, 0], c[:, 1], c[:, nk + 1], c[:, nk + 2]
    span_cyc = t_out.max() - t_in.min()
    span_rt = (rt1.max() - rt0.min()) / 100.0          # us (100 MHz)
    print(f"{meta}  waves={len(rows)}")
    print(f"kernel span: {span_rt:.1f} us; average shader clock over the waves' lives {clk_per_tick * 100:.0f} MHz -> {span_rt * clk_per_tick * 100:.0f} cycles")
    span_cyc = int(span_rt * clk_per_tick * 100)
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
    # per SIMD: how long were 0, 1, 2, ... of its waves INSIDE the main loop (between the prologue barrier and the last k tile's
    # barrier)?  Time at level 0 inside the SIMD's span is pipe time nothing can use (all resident waves in prologue / epilogue,
    # or the slot is waiting for its next workgroup); the MFMA work done at each level gives the pipe efficiency at that level.
    lvl_time = defaultdict(float)
    lvl_work = defaultdict(float)
    span_sum = 0.0
    for key, idx in groups.items():
        idx = np.array(idx)
        ev = sorted([(t_pro[i], 1) for i in idx] + [(t_loop_end[i], -1) for i in idx])
        lo, hi = t_in[idx].min(), t_out[idx].max()
        span_sum += hi - lo
        cur, last = 0, lo
        for t, d in ev:
            lvl_time[cur] += t - last
            cur, last = cur + d, t
        lvl_time[0] += hi - last
        # MFMA work attributed to a level: every k tile of every wave, at the level found at its midpoint
        ins, outs = np.sort(t_pro[idx]), np.sort(t_loop_end[idx])
        for i in idx:
            mid = (c[i, 1:nk + 1] + c[i, 2:nk + 2]) // 2
            lv = np.searchsorted(ins, mid, side="right") - np.searchsorted(outs, mid, side="right")
            for l in lv:
                lvl_work[int(l)] += mfma_per_ktile * 64
    print("SIMD time by number of its waves inside the main loop (share of all SIMD spans; pipe efficiency at that level):")
    for l in sorted(lvl_time):
        eff = lvl_work.get(l, 0.0) / lvl_time[l] if lvl_time[l] > 0 else 0.0
        print(f"  {l} waves in loop: {lvl_time[l] / span_sum:.3f} of the time, pipe efficiency {eff:.3f} -> contributes {lvl_work.get(l, 0.0) / span_sum:.3f}")
    # start skew: when do workgroups start
    first = t_in.min()
    st = (t_in[wave == 0] - first)
    print(f"workgroup start times (cycles after the first): p50 {int(np.percentile(st, 50))}  p90 {int(np.percentile(st, 90))}  max {st.max()}")
    en = t_out.max() - t_out[wave == 0]
    print(f"workgroup end times (cycles before the last):   p50 {int(np.percentile(en, 50))}  p10 {int(np.percentile(en, 10))}")


if __name__ == "__main__":
    main()
