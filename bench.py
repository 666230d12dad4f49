#!/usr/bin/env python3
"""bench.py -- GCUPS of the DP matrix-fill + traceback path on N MI355X GPUs of one node.

A "step" is one pass of the hot path (fill kernel + exact re-fills + traceback kernel [+ RCCL gather of the 48-byte
summaries when N > 1]) over one batch of synthetic pairs that is already resident in HBM.

Workload (BASELINE.json configs[4], "C5"): 100 000 protein pairs, both lengths iid uniform in [200, 2000],
core local semantics (SimpleLocalAligner), BLOSUM62, del 11 / ext 2.  The batch is fixed; with N > 1 the pairs are
sharded over the ranks by balanced cells (LPT), so scaling is STRONG (total work fixed), as north_star asks
("scaling at 8 GPUs on a 100k-pair batch").  `--pairs` shrinks the batch for quick runs.
At N = 1 the line also carries "end_to_end" (the same batch through aln_align_batch: host buffers in, host buffers out),
the single-pair configuration (configs[3], 10k x 10k) as "single_pair", configs[1] and [2] as "configs", and the CPU oracle
timed on a bounded sample ("cpu_baseline", all cores and one core; its results also check the GPU's on that sample).

Launch: `python bench.py` (N=1) or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

# the library asks the HIP runtime for 8 hardware queues when it is loaded before the runtime initialises (aln_host.hip); here
# torch comes first, so the same request is made explicitly (never overriding the user's value)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured-achievable)
ALG_BYTES_PER_CELL = 0.25      # one 2-bit direction per cell is the only per-cell datum that must leave the chip


def cpu_baseline(batch, S, gpu_results, target_seconds=12.0):
    """Times the CPU oracle (the reference's algorithm and memory behaviour, oracle/aligner_oracle.c) on a bounded
    sample of the SAME workload: all host cores with static pair partitioning (statistics/mod.rs:255-286 style) and ONE core
    (the reference is single-threaded per pair).  The sample's oracle results also check the GPU's summaries of those pairs."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = max(1, min(cores, 64))
    # grow the sample until it is ~target_seconds of CPU work (bounded: at most three tries, at most the whole batch)
    n = min(len(batch), 16 * threads)
    for _ in range(3):
        sample = batch.select(range(n))
        t0 = time.perf_counter()
        ref, _, _ = oracle.align_batch(oracle.CORE_LOCAL, sample.seqs, sample.q_off, sample.q_len, sample.t_off, sample.t_len, 11,
                                       2, S, threads)
        dt = time.perf_counter() - t0
        if dt >= 0.6 * target_seconds or n == len(batch):
            break
        n = int(min(len(batch), max(n + 1, n * target_seconds / max(dt, 1e-3))))
    mismatches = 0
    for i in range(n):
        r, g = ref[i], gpu_results[i]
        mismatches += (g["status"], g["score"], g["end_y"], g["end_x"], g["start_y"], g["start_x"], g["aln_len"]) != \
                      (r.status, r.score, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len)
    n1 = max(1, min(n, 24))
    one = batch.select(range(n1))
    t0 = time.perf_counter()
    oracle.align_batch(oracle.CORE_LOCAL, one.seqs, one.q_off, one.q_len, one.t_off, one.t_len, 11, 2, S, 1)
    dt1 = time.perf_counter() - t0
    return {"value": round(sample.cells / dt / 1e9, 4), "unit": "GCUPS", "cores": threads, "kind": "port",
            "sample": "first %d of the batch's pairs (%.3g cells), fill+argmax+traceback, %.1f s wall" % (
                n, sample.cells, dt),
            "one_core": {"value": round(one.cells / dt1 / 1e9, 5), "unit": "GCUPS", "cores": 1,
                         "sample": "first %d pairs (%.3g cells), %.1f s" % (n1, one.cells, dt1)},
            "parity_vs_gpu": {"pairs_compared": n, "mismatches": int(mismatches)}}


def end_to_end(batch, S, local_rank, calls=3):
    """The same job through aln_align_batch: HOST buffers in (residue codes, offsets), HOST buffers out (summaries and both
    aligned strings of every pair) -- what a caller of the C ABI sees.  The pool is warm after the first call."""
    from aligner_amd import _ffi, runtime
    from aligner_amd.batch import RESULT_DTYPE
    lib = _ffi.load()
    p, keep = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
    res = np.zeros(len(batch), dtype=RESULT_DTYPE)
    tb_off, total = batch.tb_layout()
    tb = np.zeros(max(total, 1), dtype=np.uint8)
    ctx = runtime.context(local_rank)
    times = []
    for i in range(calls + 1):
        t0 = time.perf_counter()
        st = lib.aln_align_batch(ctx, C.byref(p), batch.seqs.ctypes.data, batch.q_off.ctypes.data, batch.q_len.ctypes.data,
                                 batch.t_off.ctypes.data, batch.t_len.ctypes.data, len(batch), res.ctypes.data, tb.ctypes.data,
                                 tb_off.ctypes.data)
        times.append(time.perf_counter() - t0)
        runtime.raise_for_status(st, "aln_align_batch")
    warm = sorted(times[1:])
    med = warm[len(warm) // 2]
    return {"what": "aln_align_batch: host buffers in, summaries + both aligned strings of every pair out (chunked pipeline)",
            "ms": round(med * 1e3, 3), "gcups": round(batch.cells / med / 1e9, 2), "calls": calls,
            "first_call_ms_cold_pool": round(times[0] * 1e3, 2), "host_bytes_in": int(len(batch.seqs)), "host_bytes_out": int(total),
            "pairs_ok": int((res["status"] == 0).sum())}, res


def small_configs(S, local_rank, stream, torch):
    """BASELINE.json configs[1] (C2) and configs[2] (C3), plus the PWM window batch of SURVEY 8f-1: recorded beside the
    headline so that every quoted figure has a line in the driver's record."""
    from aligner_amd import _ffi, runtime, workloads
    from aligner_amd.batch import PairBatch, StagedBatch
    from aligner_amd.matrices import nucleotide_matrix
    out = {}
    outs = _ffi.OUT_SCORE | _ffi.OUT_TRACEBACK

    def staged(b, sem, d, e, M, reps):
        sb = StagedBatch(b, sem, d, e, M, device=local_rank, outputs=outs)
        with torch.cuda.stream(stream):
            sb.run(stream.cuda_stream)
        torch.cuda.synchronize()
        sb.enable_timing(True)
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            for _ in range(reps):
                sb.run(stream.cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tm = sb.timing()
        r = sb.fetch(want_traceback=False).results
        dirs = sb.direction_bytes
        sb.close()
        return dt, tm, r, dirs

    def pair_walls(sem, q, t, d, e, M, reps=30, **kw):
        runtime.align_pair(sem, q, t, d, e, M, device=local_rank, **kw)
        walls = []
        for _ in range(reps):
            t0 = time.perf_counter()
            runtime.align_pair(sem, q, t, d, e, M, device=local_rank, **kw)
            walls.append(time.perf_counter() - t0)
        walls.sort()
        return walls

    # C2: one 1k x 1k protein pair, core local 11/2 -- the uniform pair (one fill pass) and the homolog variant (its row-1
    # advice changes, so the strip pipeline runs twice)
    for name, homolog in (("C2", False), ("C2_homolog", True)):
        q, t = workloads.c2_pair(homolog=homolog)
        one = PairBatch.from_pairs([(q, t)])
        dt, tm, r, _ = staged(one, _ffi.CORE_LOCAL, 11, 2, S, 50)
        walls = pair_walls(_ffi.CORE_LOCAL, q, t, 11, 2, S)
        out[name] = {"workload": "one 1000 x %d protein pair (%s), core local, BLOSUM62, 11/2" % (
                         len(t), "10 %% substitutions + 2 %% indels" if homolog else "uniform random"),
                     "device_ms": round(tm["fill_ms"] + tm["traceback_ms"], 4), "fill_ms": round(tm["fill_ms"], 4),
                     "traceback_ms": round(tm["traceback_ms"], 4), "fill_passes": int(r[0]["passes"] & 0x7f),
                     "staged_run_ms": round(dt * 1e3, 4), "gcups_staged": round(one.cells / dt / 1e9, 3),
                     "aln_align_pair_wall_ms_median": round(walls[len(walls) // 2] * 1e3, 4),
                     "aln_align_pair_wall_ms_min": round(walls[0] * 1e3, 4), "score": float(r[0]["score"]),
                     "status": int(r[0]["status"])}
    # the same uniform pair with a real-valued matrix (what every HeuristicAligner iteration runs, heuristic/mod.rs:58-77): f64
    # kernels; and with the full AlignmentResult (direction matrix from the fast kernels; H only from the generic ones)
    q, t = workloads.c2_pair(homolog=False)
    cells = len(q) * len(t)
    w = pair_walls(_ffi.CORE_LOCAL, q, t, 11.5, 2.25, S * 0.5, reps=10, force_f64=True)
    out["f64_pair"] = {"workload": "the C2 pair, real-valued matrix (BLOSUM62 x 0.5, del 11.5 / ext 2.25), f64 kernels forced (a dyadic scheme: by itself "
                                   "the library runs it on the integer kernels, see dyadic_batch): one workgroup per pair",
                       "aln_align_pair_wall_ms_median": round(w[len(w) // 2] * 1e3, 3), "gcups": round(cells / w[len(w) // 2] / 1e9, 3)}
    # (that pair needs a second advice pass; most do not: eight more random 1000 x 1000 pairs)
    rng8 = np.random.default_rng(8)
    med, one_pass = [], 0
    for _ in range(8):
        q8, t8 = rng8.integers(0, 20, 1000).astype(np.uint8), rng8.integers(0, 20, 1000).astype(np.uint8)
        w8 = pair_walls(_ffi.CORE_LOCAL, q8, t8, 11.5, 2.25, S * 0.5, reps=6, force_f64=True)
        med.append(w8[len(w8) // 2])
        one_pass += int((runtime.align_pair(_ffi.CORE_LOCAL, q8, t8, 11.5, 2.25, S * 0.5, device=local_rank, force_f64=True)[0].passes & 0x7f) == 1)
    out["f64_pair"]["other_pairs"] = {"what": "8 uniform-random 1000 x 1000 pairs, same scoring", "wall_ms_mean": round(float(np.mean(med)) * 1e3, 3),
                                      "wall_ms_min": round(min(med) * 1e3, 3), "wall_ms_max": round(max(med) * 1e3, 3), "one_pass": one_pass}
    w = pair_walls(_ffi.CORE_LOCAL, q, t, 11, 2, S, reps=10, want_directions=True)
    out["pair_with_direction_matrix"] = {"workload": "the C2 pair + the (M+1) x (N+1) Direction bytes (fast kernels + unpack)",
                                         "aln_align_pair_wall_ms_median": round(w[len(w) // 2] * 1e3, 3)}
    w = pair_walls(_ffi.CORE_LOCAL, q, t, 11, 2, S, reps=5, want_directions=True, want_h=True)
    out["pair_with_h_and_direction_matrix"] = {"workload": "the C2 pair + direction bytes + the f64 H matrix (generic kernels, 9 B per cell over PCIe)",
                                               "aln_align_pair_wall_ms_median": round(w[len(w) // 2] * 1e3, 3)}
    # PWM windows (the inner loop of latent-repeat-search, engine/calc.rs:107-136): 100 000 windows of 330 nt, one every
    # 30 nt of a chromosome, against a 4 x 300 PWM -- windows are (start, length) into the one chromosome array
    from aligner_amd.pwm import align_window_offsets
    rng = np.random.default_rng(300)
    pwm = rng.integers(-3, 4, (4, 300)).astype(np.float64)
    chrom = rng.integers(0, 4, 100000 * 30 + 400).astype(np.uint8)
    starts, lens = np.arange(100000, dtype=np.uint64) * np.uint64(30), np.full(100000, 330, dtype=np.uint64)
    keep = {}
    for name, tbk in (("pwm_windows_score_only", False), ("pwm_windows_with_alignments", True)):
        align_window_offsets(chrom, starts, lens, 3, 1, pwm, device=local_rank, want_traceback=tbk, want_alignments=False, reuse=keep)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            resw, _ = align_window_offsets(chrom, starts, lens, 3, 1, pwm, device=local_rank, want_traceback=tbk, want_alignments=False,
                                           reuse=keep)
            ts.append(time.perf_counter() - t0)
        dtw = sorted(ts)[1]
        out[name] = {"workload": "100000 windows of 330 nt (every 30 nt of a 3 Mnt chromosome) x a 4 x 300 PWM, del 3 / ext 1, "
                                 "host buffers in and out" + (", numbered + residue strings of every window fetched" if tbk else ""),
                     "ms": round(dtw * 1e3, 2), "gcups": round(100000 * 330 * 300 / dtw / 1e9, 2),
                     "windows_ok": int((resw["status"] == 0).sum())}
    # the same windows against a REAL-VALUED PWM (latent-repeat-search re-estimates its weights: reals, engine/calc.rs:107-136): the f64
    # kernels (the lean f64 strip, r03); every 997th window against the oracle
    import oracle
    pwm_r = np.round(rng.normal(0, 1, (4, 300)), 3)
    for _ in range(2):
        resw, _ = align_window_offsets(chrom, starts, lens, 3.5, 1.25, pwm_r, device=local_rank, want_traceback=False, want_alignments=False, reuse=keep)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        resw, _ = align_window_offsets(chrom, starts, lens, 3.5, 1.25, pwm_r, device=local_rank, want_traceback=False, want_alignments=False, reuse=keep)
        ts.append(time.perf_counter() - t0)
    dtw = sorted(ts)[1]
    samp = range(0, 100000, 997)
    same = sum(1 for i in samp if float(resw["f"][i]) == oracle.align_pwm(chrom[i * 30:i * 30 + 330], 3.5, 1.25, pwm_r)["f"])
    out["pwm_windows_real_valued"] = {"workload": "the same 100000 windows x a real-valued 4 x 300 PWM (normal weights, 3 decimals), del 3.5 / ext 1.25, "
                                                  "score only, host buffers in and out: f64 kernels",
                                      "ms": round(dtw * 1e3, 2), "gcups": round(100000 * 330 * 300 / dtw / 1e9, 2),
                                      "integer_kernels": bool((resw["flags"] & 1).all()), "windows_ok": int((resw["status"] == 0).sum()),
                                      "windows_filled_twice": int(((resw["passes"] & 0xff) >= 2).sum()),
                                      "sample_windows_equal_oracle": "%d of %d" % (same, len(samp))}
    # C3: 10 000 nucleotide read pairs 150 x 150, core global, +5/-4, 10/1
    b3 = workloads.c3_batch(10000)
    dt, tm, r, dirs = staged(b3, _ffi.CORE_GLOBAL, 10, 1, nucleotide_matrix(), 50)
    alg = 0.25 * b3.cells + float((b3.q_len + b3.t_len).sum()) + 48.0 * len(b3)
    out["C3"] = {"workload": "10000 nucleotide read pairs 150 x 150, core global, +5/-4, del 10 / ext 1",
                 "ms": round(dt * 1e3, 4), "gcups": round(b3.cells / dt / 1e9, 2), "fill_ms": round(tm["fill_ms"], 4),
                 "traceback_ms": round(tm["traceback_ms"], 4), "fill_only_gcups": round(b3.cells / tm["fill_ms"] / 1e6, 2),
                 "alg_bytes_per_launch": alg, "alg_hbm_gbs": round(alg / (tm["fill_ms"] / 1e3) / 1e9, 2),
                 "direction_bytes_stored": dirs, "pairs_ok": int((r["status"] == 0).sum())}
    # the reference's one batch driver: calculate_p_value (statistics/mod.rs:240-307) -- one 350-aa query against 4 999 shuffled
    # copies of a 350-aa target, core local, score only (only alignment.f is kept, :273-279); host buffers in and out, the CPU
    # oracle on the same pairs beside it (all host cores, as the reference's ten threads)
    import oracle
    from aligner_amd.statistics import shuffled_scores
    rngp = np.random.default_rng(350)
    qp, tp = rngp.integers(0, 20, 350).astype(np.uint8), rngp.integers(0, 20, 350).astype(np.uint8)
    _, _, pb = shuffled_scores(qp, tp, 0.0, 11, 2, S, rng=np.random.default_rng(1), device=local_rank)
    pp, keep_p = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=_ffi.OUT_SCORE)
    from aligner_amd.batch import RESULT_DTYPE
    lib = _ffi.load()
    ctxp = runtime.context(local_rank)
    resp = np.zeros(len(pb), dtype=RESULT_DTYPE)
    tsp = []
    for _ in range(6):
        t0 = time.perf_counter()
        st = lib.aln_align_batch(ctxp, C.byref(pp), pb.seqs.ctypes.data, pb.q_off.ctypes.data, pb.q_len.ctypes.data, pb.t_off.ctypes.data,
                                 pb.t_len.ctypes.data, len(pb), resp.ctypes.data, None, None)
        tsp.append(time.perf_counter() - t0)
        runtime.raise_for_status(st, "aln_align_batch")
    dtp = sorted(tsp[1:])[len(tsp[1:]) // 2]
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t0 = time.perf_counter()
    refp, _, _ = oracle.align_batch(oracle.CORE_LOCAL, pb.seqs, pb.q_off, pb.q_len, pb.t_off, pb.t_len, 11, 2, S, max(1, min(cores, 64)))
    dto = time.perf_counter() - t0
    out["p_value_batch"] = {"workload": "calculate_p_value's batch: one 350-aa query x 4999 shuffled 344..350-aa targets, core local, BLOSUM62 11/2, "
                                        "score only, host buffers in and out (one aln_align_batch call)",
                            "ms": round(dtp * 1e3, 3), "gcups": round(pb.cells / dtp / 1e9, 2),
                            "cpu_oracle_ms": round(dto * 1e3, 1), "cpu_oracle_threads": max(1, min(cores, 64)),
                            "scores_equal_oracle": bool(all(float(resp[i]["f"]) == refp[i].f for i in range(len(pb))))}
    # batches of LARGE pairs (r02: every pair of >= 2^24 cells took the single-pair route, one after the other): 256 pairs of
    # 4200 x 4200 through aln_align_batch, host buffers in and out -- the batch kernel shares the strips of a pair between waves
    pbig, keep_b = runtime.make_params(_ffi.CORE_LOCAL, 11, 2, S, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
    for key, npairs in (("large_pairs_batch", 256), ("large_pairs_batch_1024", 1024)):
        bb = workloads.c5_batch(n_pairs=npairs, lo=4200, hi=4200)
        resb = np.zeros(len(bb), dtype=RESULT_DTYPE)
        tbo, tbt = bb.tb_layout()
        tbb = np.zeros(max(tbt, 1), dtype=np.uint8)
        tsb = []
        for _ in range(4):
            t0 = time.perf_counter()
            st = lib.aln_align_batch(ctxp, C.byref(pbig), bb.seqs.ctypes.data, bb.q_off.ctypes.data, bb.q_len.ctypes.data, bb.t_off.ctypes.data,
                                     bb.t_len.ctypes.data, len(bb), resb.ctypes.data, tbb.ctypes.data, tbo.ctypes.data)
            tsb.append(time.perf_counter() - t0)
            runtime.raise_for_status(st, "aln_align_batch")
        dtb = min(tsb[1:])
        out[key] = {"workload": "%d protein pairs of 4200 x 4200 (1.8e7 cells each), core local, BLOSUM62 11/2, summaries + both strings, "
                                "host buffers in and out" % npairs,
                    "ms": round(dtb * 1e3, 3), "gcups": round(bb.cells / dtb / 1e9, 2),
                    "pairs_on_single_pair_route": int(((resb["flags"] & 2) != 0).sum()), "pairs_ok": int((resb["status"] == 0).sum())}
    # batches with a real-valued matrix (f64 kernels): the first 20000 C5 pairs, host to host.  Two schemes: BLOSUM62 x 0.5 with 11.5 /
    # 2.25 is DYADIC (every score a multiple of 0.25: exact zeros everywhere in the bottom rows, a third of the pairs fill twice -- the
    # worst case of the row-1 hazard, and what r02 measured); BLOSUM62 x 0.37 with 11.3 / 2.1 is what a re-estimated matrix
    # (heuristic/mod.rs:58-77) looks like: 1 % of the pairs fill twice.  200 pairs of each against the oracle.
    from aligner_amd.batch import align_batch
    bf = workloads.c5_batch(20000)
    sample = bf.select(range(200))
    # dyadic_batch: the dyadic scheme as the library runs it by itself -- on the integer kernels, scores scaled back (exact: DESIGN 4.4)
    for key, scale, de_, ex_, f64 in (("f64_batch", 0.5, 11.5, 2.25, True), ("f64_batch_nondyadic", 0.37, 11.3, 2.1, False), ("dyadic_batch", 0.5, 11.5, 2.25, False)):
        tsf, rf = [], None
        for _ in range(4):
            t0 = time.perf_counter()
            rf = align_batch(bf, _ffi.CORE_LOCAL, de_, ex_, S * scale, device=local_rank, want_traceback=True, out=rf, force_f64=f64)   # the caller keeps its buffers
            tsf.append(time.perf_counter() - t0)
        refs, _, _ = oracle.align_batch(oracle.CORE_LOCAL, sample.seqs, sample.q_off, sample.q_len, sample.t_off, sample.t_len, de_, ex_, S * scale,
                                        max(1, min(cores, 64)))
        same = sum(1 for i in range(len(sample)) if (float(rf.results[i]["score"]), int(rf.results[i]["end_y"]), int(rf.results[i]["end_x"]),
                                                     int(rf.results[i]["aln_len"])) == (refs[i].score, refs[i].end_y, refs[i].end_x, refs[i].aln_len))
        out[key] = {"workload": "the first 20000 C5 pairs, real-valued matrix (BLOSUM62 x %g, del %g / ext %g): %s, summaries + strings, host to host" % (
                        scale, de_, ex_, "integer kernels (the scheme times 4), scores scaled back" if key == "dyadic_batch" else "f64 kernels" + (" forced" if f64 else "")),
                    "integer_kernels": bool((rf.results["flags"] & 1).all()),
                    "ms": round(min(tsf[1:]) * 1e3, 2), "gcups": round(bf.cells / min(tsf[1:]) / 1e9, 2), "pairs_ok": int((rf.results["status"] == 0).sum()),
                    "pairs_filled_twice": int(((rf.results["passes"] & 0xff) >= 2).sum()), "sample_pairs_equal_oracle": "%d of %d" % (same, len(sample))}
    return out


def cold_start():
    """What a one-shot caller sees (aligner-cli aligns ONE pair per process, aligner-cli/main.rs:41-53): context creation, the
    first pair and the first batch of a fresh process -- measured in a child process without torch."""
    import subprocess
    code = r"""
import ctypes as C, json, os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, %r)
t0 = time.perf_counter()
from aligner_amd import _ffi, runtime, workloads
from aligner_amd.batch import align_batch
from aligner_amd.matrices import get_blosum62
S = get_blosum62()
t_import = time.perf_counter() - t0
t0 = time.perf_counter(); ctx = runtime.context(0); t_create = time.perf_counter() - t0
q, t = workloads.c2_pair(homolog=False)
t0 = time.perf_counter(); runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S); t_pair1 = time.perf_counter() - t0
t0 = time.perf_counter(); runtime.align_pair(_ffi.CORE_LOCAL, q, t, 11, 2, S); t_pair2 = time.perf_counter() - t0
b = workloads.c5_batch(100000)
t0 = time.perf_counter(); r1 = align_batch(b, _ffi.CORE_LOCAL, 11, 2, S); t_b1 = time.perf_counter() - t0
t0 = time.perf_counter(); align_batch(b, _ffi.CORE_LOCAL, 11, 2, S, out=r1); t_b2 = time.perf_counter() - t0
print("COLD " + json.dumps({"aln_create_ms": round(t_create * 1e3, 2), "first_1k_pair_ms": round(t_pair1 * 1e3, 2), "second_1k_pair_ms": round(t_pair2 * 1e3, 3),
                            "first_c5_batch_call_ms": round(t_b1 * 1e3, 1), "second_c5_batch_call_ms": round(t_b2 * 1e3, 1)}))
""" % ROOT
    try:
        outp = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300).stdout
        for ln in outp.splitlines():
            if ln.startswith("COLD "):
                d = json.loads(ln[5:])
                d["what"] = "a fresh process (no torch): aln_create, first and second aln_align_pair of the C2 pair, first and second aln_align_batch of C5 (Python wrapper included; the first call also faults in 0.44 GB of fresh output pages, the second writes the same arrays)"
                return d
    except Exception as e:                               # noqa: BLE001
        return {"error": str(e)[:200]}
    return {"error": "no output"}


def kernel_sources_sha16():
    import hashlib
    h = hashlib.sha256()
    for f in ("aln_kernels.hip", "aln_fast.h", "aln_device.h"):
        h.update(open(os.path.join(ROOT, "aligner_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000, help="size of the C5 batch (default: the full 100 000)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-pair", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-small-configs", action="store_true")
    ap.add_argument("--rehearse-collective", action="store_true",
                    help="world size 1 only: run the RCCL process group and the summary gather of the N > 1 path anyway")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "hbm_traffic.json"))
    args = ap.parse_args()

    # stdout carries ONE JSON line.  Libraries print there too (RCCL writes a five-line version banner to fd 1 when the
    # process group comes up), so fd 1 is pointed at stderr for the whole run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from aligner_amd import _ffi, runtime, workloads
    from aligner_amd.batch import RESULT_DTYPE, PairBatch, StagedBatch
    from aligner_amd.distributed import SummaryGather, device_bytes_as_tensor, lpt_shards
    from aligner_amd.matrices import get_blosum62

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    n_gpus = world
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.rehearse_collective
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    S = get_blosum62()
    # ---- the batch and this rank's shard (every rank derives the same partition from the lengths alone)
    qlen, tlen = workloads.c5_lengths(args.pairs)
    cells_all = qlen * tlen
    shards = lpt_shards(cells_all, world)
    mine = shards[rank]
    batch = workloads.c5_batch(args.pairs, indices=mine)
    total_cells = int(cells_all.sum())

    outs = _ffi.OUT_SCORE | _ffi.OUT_TRACEBACK
    torch.cuda.synchronize()
    t_stage = time.perf_counter()
    sb = StagedBatch(batch, _ffi.CORE_LOCAL, 11, 2, S, device=local_rank, outputs=outs)    # device allocs + H2D of the codes
    torch.cuda.synchronize()
    t_stage = time.perf_counter() - t_stage
    stream = torch.cuda.Stream()
    rec = RESULT_DTYPE.itemsize
    if use_dist:
        gather = SummaryGather([len(s) for s in shards], rank, "cuda")
        view = device_bytes_as_tensor(sb.results_device_ptr, len(batch) * rec)

    def step():
        with torch.cuda.stream(stream):
            sb.run(stream.cuda_stream)
            if use_dist:
                gather(view)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    sb.enable_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    timing = sb.timing()
    sb.enable_timing(False)

    # ---- light integrity check outside the timed region: every pair finished OK, gathered records are the shard's
    t_fetch = time.perf_counter()
    fetched = sb.fetch(want_traceback=True)              # D2H of the 48-byte summaries and both aligned strings of every pair
    t_fetch = time.perf_counter() - t_fetch
    res = fetched.results
    del fetched
    ok = int((res["status"] == 0).sum())
    refills = int((res["passes"] > 1).sum())
    if use_dist:
        everyone = gather.unpack(shards, args.pairs)
        assert (everyone[mine] == res).all() and int((everyone["status"] == 0).sum()) >= ok

    line = None
    if rank == 0:
        gcups = total_cells * args.steps / elapsed / 1e9
        fill_s = timing["fill_ms"] / 1e3
        alg_bytes = ALG_BYTES_PER_CELL * batch.cells + float((batch.q_len + batch.t_len).sum()) + 48.0 * len(batch)
        achieved = alg_bytes / fill_s / 1e9
        traffic = valu_insts = lds_ratio = None
        pmc_fresh = False
        try:
            with open(args.traffic_json) as f:
                tj = json.load(f)
            key = "c5_%d_n%d" % (args.pairs, world)
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            valu_insts = tj.get(key, {}).get("valu_wave_insts_per_launch")
            lds_ratio = tj.get(key, {}).get("lds_bank_conflict_ratio")
            pmc_fresh = tj.get(key, {}).get("kernel_src_sha16") == kernel_sources_sha16()
        except Exception:
            traffic = valu_insts = lds_ratio = None
        # The binding roof of this integer DP is VALU issue, not HBM (HBM runs at ~8 % of its peak).  With the PMC count of
        # the kernel's wave instructions per launch at hand the line is priced against that roof: achieved = instructions per launch
        # / the kernel time measured live in this run, peak = the issue rate of the kernel's own instruction mix -- 10 of a cell's 11
        # VALU instructions are in gfx950's 4-cycle class (profiles/r02_valu_rate.txt): 1024 SIMDs x 2.4 GHz / 4 = 614.4 G/s.
        # Without the count (no PMC record for this batch size) the line falls back to the HBM form.
        hbm = {"achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6),
               "alg_bytes_per_launch": alg_bytes, "traffic_over_alg": (round(traffic / alg_bytes, 3) if traffic else None)}
        if valu_insts:
            roof = {"bound": "valu", "achieved": round(valu_insts / fill_s / 1e9, 3), "peak": 614.4, "unit": "G wave-instructions/s",
                    "frac": round(valu_insts / fill_s / 614.4e9, 4), "valu_insts_per_cell": round(valu_insts * 64.0 / batch.cells, 3),
                    "valu_fullrate_frac": round(valu_insts / fill_s / 1228.8e9, 4)}
        else:
            roof = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"]}
        roof.update({"kernel": "aln_fill_fast_kernel<CORE_LOCAL>", "traffic": traffic, "hbm": hbm,
                     # the PMC figures (traffic, instruction count, LDS ratio) come from profiles/hbm_traffic.json; stale = the kernel
                     # sources have changed since they were measured
                     "pmc_figures_match_current_sources": pmc_fresh,
                     "kernel_ms": round(timing["fill_ms"], 4), "traceback_ms": round(timing["traceback_ms"], 4),
                     "fill_only_gcups_rank0": round(batch.cells / fill_s / 1e9, 3), "direction_bytes_stored": sb.direction_bytes,
                     "lds_bank_conflict_ratio": lds_ratio,     # SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (PMC pass)
                     # SURVEY's accounting: 16 lane-ops per cell against 78.6 T lane-ops/s (every instruction at the 2-cycle rate)
                     "valu_frac_survey_accounting": round(batch.cells / fill_s * 16 / 78.6e12, 5)})
        line = {
            "metric": "GCUPS (DP cell updates/s), fill + traceback, inputs resident in HBM",
            "value": round(gcups, 3), "unit": "GCUPS", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "C5: %d protein pairs, lengths iid U[200,2000], core local (SimpleLocalAligner), "
                                   "BLOSUM62, del 11 / ext 2, seed 0xA11C0005" % args.pairs,
                       "pairs": args.pairs, "cells": total_cells, "sharding": "LPT by cells over %d rank(s)" % world,
                       "pairs_ok_rank0": ok, "pairs_repaired_or_refilled_rank0": refills},
            "roofline": roof,
            # never `value`: the staged API's own hand-over costs (aln_batch_create = allocations + H2D; aln_batch_fetch = D2H of
            # every summary and string), serial around one step.  The pipelined host-buffer call is "end_to_end" below.
            "pcie_inclusive": {"stage_ms": round(t_stage * 1e3, 2), "fetch_ms": round(t_fetch * 1e3, 2),
                               "gcups_rank0": round(batch.cells / (t_stage + elapsed / args.steps + t_fetch) / 1e9, 2)},
        }

    # ---- N = 1 extras: the host-buffer call, the single-pair configuration, the small configurations, the CPU baseline
    sb.close()
    if rank == 0 and world == 1:
        e2e_res = None
        if not args.no_end_to_end:
            line["end_to_end"], e2e_res = end_to_end(batch, S, local_rank)
            assert (e2e_res == res).all(), "the pipelined call and the staged batch disagree"
            # the same call from a process of its own (tools/bench_e2e.py as a child: no torch, no other stream sharing the GPU's
            # hardware queues) -- what a standalone caller of the C ABI sees
            try:
                import subprocess
                env = dict(os.environ, E2E_JSON="1")
                outp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_e2e.py"), str(args.pairs), "3"], env=env,
                                      capture_output=True, text=True, timeout=180).stdout
                for ln in outp.splitlines():
                    if ln.startswith("E2E_JSON "):
                        line["end_to_end"]["standalone_process"] = json.loads(ln[len("E2E_JSON "):])
            except Exception as e:                       # noqa: BLE001 -- the in-process figure stands on its own
                line["end_to_end"]["standalone_process"] = {"error": str(e)[:200]}
        if not args.no_small_configs:
            line["configs"] = small_configs(S, local_rank, stream, torch)
            line["cold_start"] = cold_start()
        if not args.no_single_pair:
            q, t = workloads.c4_pair(homolog=False)
            one = PairBatch.from_pairs([(q, t)])
            sp = StagedBatch(one, _ffi.CORE_LOCAL, 11, 2, S, device=local_rank, outputs=outs)
            with torch.cuda.stream(stream):
                sp.run(stream.cuda_stream)
            torch.cuda.synchronize()
            sp.enable_timing(True)
            t0 = time.perf_counter()
            reps = 3
            with torch.cuda.stream(stream):
                for _ in range(reps):
                    sp.run(stream.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            tm = sp.timing()
            r1 = sp.fetch(want_traceback=False).results[0]
            line["single_pair"] = {"workload": "C4: one 10000 x 10000 protein pair, core local, BLOSUM62, 11/2",
                                   "gcups": round(one.cells / dt / 1e9, 3), "ms": round(dt * 1e3, 3),
                                   "fill_ms": round(tm["fill_ms"], 3), "traceback_ms": round(tm["traceback_ms"], 3),
                                   "fill_only_gcups": round(one.cells / tm["fill_ms"] / 1e6, 3),
                                   "score": float(r1["score"]), "status": int(r1["status"])}
            sp.close()
            # the same shape over other inputs: about half of all large pairs need their row-1 advice corrected in a few leading
            # columns, which the strip pipeline repairs locally (r01: a second run of the whole pipeline, ~37 GCUPS)
            rates, repaired, two_pass = [], 0, 0
            for kind in ("uniform", "homolog"):
                for seed in range(4):
                    qq = workloads.random_codes(workloads.SEED_C4 + 31 * seed, 10000, 20)
                    tt = workloads.mutate(qq, seed + 5, 20, 0.10, 0.02) if kind == "homolog" else workloads.random_codes(77 + seed, 10000, 20)
                    ob = PairBatch.from_pairs([(qq, tt)])
                    sv = StagedBatch(ob, _ffi.CORE_LOCAL, 11, 2, S, device=local_rank, outputs=outs)
                    sv.run(); sv.sync()
                    sv.enable_timing(True)
                    for _ in range(3):
                        sv.run()
                    sv.sync()
                    tv = sv.timing()
                    rv = sv.fetch(want_traceback=False).results[0]
                    sv.close()
                    rates.append(ob.cells / (tv["fill_ms"] + tv["traceback_ms"]) / 1e6)
                    repaired += bool(int(rv["passes"]) & 0x100)
                    two_pass += (int(rv["passes"]) & 0x7f) > 1
            line["single_pair"]["other_inputs"] = {
                "what": "4 uniform-random + 4 homolog (10 % substitutions, 2 % indels) 10000 x ~10000 pairs, device fill + traceback",
                "gcups_min": round(min(rates), 2), "gcups_median": round(sorted(rates)[len(rates) // 2], 2),
                "locally_repaired": repaired, "second_full_pass": two_pass}
            # pairs beyond the old ~29 400-column limit of the strip-pipelined route, through the blocking host call (upload, all
            # kernels, download of the summary and both strings included); tests/test_gpu_parity.py checks them against the oracle
            from aligner_amd import runtime
            longp = {}
            for N_, M_ in ((40000, 10000), (10000, 40000)):
                qq = workloads.random_codes(4040 + N_, N_, 20)
                tt = workloads.mutate(qq[:M_] if M_ <= N_ else np.concatenate([qq, workloads.random_codes(9 + N_, M_ - N_, 20)]), 1234 + N_, 20, 0.10, 0.02, out_len=M_)
                runtime.align_pair(_ffi.CORE_LOCAL, qq, tt, 11, 2, S)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    rr = runtime.align_pair(_ffi.CORE_LOCAL, qq, tt, 11, 2, S)
                    best = min(best, time.perf_counter() - t0)
                longp["%dx%d" % (N_, M_)] = {"aln_align_pair_wall_ms": round(best * 1e3, 3), "gcups": round(N_ * M_ / best / 1e9, 2),
                                            "strip_pipelined_route": bool(rr[0].flags & 2)}
            line["single_pair"]["long_pairs"] = longp
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(batch, S, res)
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
