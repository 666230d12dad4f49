"""The N > 1 path on CPU: LPT sharding and the summary gather, world_size 2 over gloo (no GPU).
The records gathered here are produced by the CPU oracle standing in for each rank's GPU results -- what is under test
is the sharding / padding / all_gather / scatter-back logic of aligner_amd/distributed.py that bench.py uses with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_shards_partition_and_balance():
    from aligner_amd import workloads
    from aligner_amd.distributed import lpt_shards
    ql, tl = workloads.c5_lengths(5000)
    cells = ql * tl
    for world in (1, 2, 4, 8):
        shards = lpt_shards(cells, world)
        allidx = np.sort(np.concatenate(shards))
        assert (allidx == np.arange(len(cells))).all()                     # a partition
        loads = np.array([cells[s].sum() for s in shards], dtype=np.float64)
        assert loads.max() / loads.mean() < 1.01                           # balanced to 1 %
        again = lpt_shards(cells, world)
        assert all((a == b).all() for a, b in zip(shards, again))          # deterministic on every rank


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from aligner_amd import workloads
    from aligner_amd.batch import RESULT_DTYPE
    from aligner_amd.distributed import SummaryGather, lpt_shards
    from aligner_amd.matrices import get_blosum62
    ql, tl = workloads.c5_lengths(n_pairs, 20, 90)
    shards = lpt_shards(ql * tl, world)
    mine = shards[rank]
    b = workloads.c5_batch(n_pairs, 20, 90, indices=mine)
    ref, _, _ = oracle.align_batch(oracle.CORE_LOCAL, b.seqs, b.q_off, b.q_len, b.t_off, b.t_len, 11, 2,
                                   get_blosum62(), 1, want_traceback=False)
    rec = np.zeros(len(b), dtype=RESULT_DTYPE)
    for i in range(len(b)):
        r = ref[i]
        rec[i] = (r.f, r.score, r.end_y, r.end_x, r.start_y, r.start_x, r.aln_len, r.status, 1, 1)
    g = SummaryGather([len(s) for s in shards], rank, "cpu")
    local = torch.from_numpy(rec.view(np.uint8).copy())
    g(local)
    everyone = g.unpack(shards, n_pairs)
    assert (everyone[mine] == rec).all()
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), everyone)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world_size_2_gloo(tmp_path, orc, blosum62):
    from aligner_amd import workloads
    n_pairs, world = 37, 2                      # odd count: the shards differ in size, exercising the padding
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_pairs, str(tmp_path)), nprocs=world, join=True)
    a = np.load(os.path.join(str(tmp_path), "rank0.npy"))
    b = np.load(os.path.join(str(tmp_path), "rank1.npy"))
    assert (a == b).all()                       # every rank ends with the same global table
    full = workloads.c5_batch(n_pairs, 20, 90)
    for i in range(n_pairs):                    # and it is the single-process answer, in global pair order
        one = orc.align(orc.CORE_LOCAL, full.query(i), full.target(i), 11, 2, blosum62)
        assert a[i]["status"] == one["status"]
        if one["status"] == 0:
            assert (a[i]["score"], a[i]["end_y"], a[i]["end_x"], a[i]["aln_len"]) == \
                   (one["score"], one["end"][0], one["end"][1], len(one["qa"]))
