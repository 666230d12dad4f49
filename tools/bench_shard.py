#!/usr/bin/env python3
"""Rehearsal of the strong-scaling runs on ONE GPU: times the C5 shard a rank would get at world size W (no gather), to see
what the fixed per-step costs do to the 8-GPU efficiency before the driver's 8-GPU node measures it.
usage: python tools/bench_shard.py [--pairs 100000] [--worlds 1,2,4,8] [--steps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=100000)
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    from aligner_amd import _ffi, workloads
    from aligner_amd.batch import StagedBatch
    from aligner_amd.distributed import lpt_shards
    from aligner_amd.matrices import get_blosum62
    S = get_blosum62()
    qlen, tlen = workloads.c5_lengths(args.pairs)
    cells_all = qlen * tlen
    total = int(cells_all.sum())
    stream = torch.cuda.Stream()
    base = None
    for w in [int(x) for x in args.worlds.split(",")]:
        shards = lpt_shards(cells_all, w)
        worst = 0.0
        for r in sorted({0, w - 1}):
            batch = workloads.c5_batch(args.pairs, indices=shards[r])
            sb = StagedBatch(batch, _ffi.CORE_LOCAL, 11, 2, S, device=0, outputs=_ffi.OUT_SCORE | _ffi.OUT_TRACEBACK)
            with torch.cuda.stream(stream):
                sb.run(stream.cuda_stream)
            torch.cuda.synchronize()
            sb.enable_timing(True)
            t0 = time.perf_counter()
            with torch.cuda.stream(stream):
                for _ in range(args.steps):
                    sb.run(stream.cuda_stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.steps
            tm = sb.timing()
            print("world %d rank %d: %6d pairs %.3e cells  step %.3f ms (fill %.3f, traceback %.3f)" % (
                w, r, len(batch), batch.cells, dt * 1e3, tm["fill_ms"], tm["traceback_ms"]), flush=True)
            worst = max(worst, dt)
            del sb
        if base is None:
            base = worst
        print("world %d: %.1f GCUPS whole job (no gather), speed-up %.2f of %d" % (w, total / worst / 1e9, base / worst, w), flush=True)


if __name__ == "__main__":
    main()
