"""CPU, world_size 2 over gloo: the N>1 path of bench.py/run.py — contiguous utterance shards, no data-path
collective, max-over-ranks timing, host-side gather of token ids.  The per-rank "engine" here is the oracle's
greedy decoder on a toy model (the HIP engine needs a GPU); the test checks that sharded == unsharded."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import sharding, synthetic
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = synthetic.get_config("toy-short")
    W = cpu_ref.to_torch(synthetic.make_weights(cfg, 21))
    begin, end = sharding.utterance_shard(total, world, rank)
    rows = []
    with torch.no_grad():
        for b0, b1 in sharding.batches(begin, end, 2):
            mel = np.concatenate([synthetic.make_mel(cfg, index=i, batch=1) for i in range(b0, b1)])
            rows += cpu_ref.transcribe(W, cfg, torch.from_numpy(mel)).tolist()
    slow = sharding.max_over_ranks(1.0 + rank, dist)
    allrows = sharding.gather_ids(rows, dist)
    if rank == 0:
        q.put((slow, allrows))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cpu_ref
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import synthetic
    total, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    slow, allrows = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert slow == 2.0                                     # max over ranks
    cfg = synthetic.get_config("toy-short")
    W = cpu_ref.to_torch(synthetic.make_weights(cfg, 21))
    with torch.no_grad():
        want = [cpu_ref.transcribe(W, cfg, torch.from_numpy(synthetic.make_mel(cfg, index=i, batch=1)))[0].tolist() for i in range(total)]
    assert allrows == want
