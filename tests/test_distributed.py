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


def _plan_worker(rank, world, port, totals, q):
    sys.path.insert(0, ROOT)
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import sharding, synthetic
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    report = {}
    for total in totals:
        row = lambda i: [i, (7 * i) % 13, total]              # what "decoding utterance i" yields, whichever rank does it
        # (1) contiguous shards + rank-order gather: the dataset-order path of run.py / cal_wer.py / bench.py
        begin, end = sharding.utterance_shard(total, world, rank)
        mine = [row(i) for b0, b1 in sharding.batches(begin, end, 8) for i in range(b0, b1)]
        contiguous = sharding.gather_ids(mine, dist)
        # (2) length-aware plan: sorted batches dealt round-robin, (index, row) pairs gathered and re-ordered on the host
        dur, steps = synthetic.librispeech_like_lengths(total, seed=3)
        groups = sharding.length_sorted_batches(dur, 8, world, rank)
        pairs = sorted(sharding.gather_objects([(i, row(i)) for g in groups for i in g], dist))
        sizes = sharding.gather_objects([end - begin], dist)
        n_groups = sharding.gather_objects([len(groups)], dist)
        report[total] = (contiguous, pairs, sizes, n_groups)
    slow = sharding.max_over_ranks(0.5 + rank, dist)
    if rank == 0:
        q.put((slow, report))
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_plans_cover_every_utterance_once():
    """world_size 8 over gloo, the shapes of BASELINE config 5: 64 utterances -> 8 x 8; LibriSpeech test-clean's 2620 -> 328 / 327 per
    rank; and fewer utterances than ranks.  Both batching plans of run.py / cal_wer.py return every utterance exactly once, in dataset
    order after the host-side gather; no rank exchanges anything but these small Python lists."""
    totals, world = [64, 2620, 5], 8
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, totals, q)) for r in range(world)]
    for p in procs:
        p.start()
    slow, report = q.get()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert slow == 7.5
    for total in totals:
        contiguous, pairs, sizes, n_groups = report[total]
        want = [[i, (7 * i) % 13, total] for i in range(total)]
        assert contiguous == want
        assert [i for i, _ in pairs] == list(range(total)) and [r for _, r in pairs] == want
        assert sum(sizes) == total and max(sizes) - min(sizes) <= 1
        assert sum(n_groups) == (total + 7) // 8 and max(n_groups) - min(n_groups) <= 1
    assert report[64][2] == [8] * 8 and report[2620][2] == [328] * 4 + [327] * 4


def test_length_sorted_batches_raise_slot_utilisation():
    """Host logic of the variable-length workload: sorted batches never lose an utterance, keep ties in dataset order, and spend
    fewer (row, step) slots than dataset-order batches on a LibriSpeech-like length mix."""
    sys.path.insert(0, ROOT)
    import whisper_trtllm_amd  # noqa: F401
    from whisper_trtllm_amd import audio, sharding, synthetic
    dur, steps = synthetic.librispeech_like_lengths(200, seed=1)
    assert 6.0 < float(np.mean(dur)) < 9.0 and min(dur) >= 1.3 and max(dur) <= 30.0 and all(2 <= s <= 446 for s in steps)
    in_order = [list(range(a, b)) for a, b in sharding.batches(0, 200, 8)]
    by_len = sharding.length_sorted_batches(dur, 8)
    assert sorted(i for g in by_len for i in g) == list(range(200))
    rows = [s + 1 for s in steps]
    u0, u1 = sharding.slot_utilisation(rows, in_order), sharding.slot_utilisation(rows, by_len)
    assert u0 < 0.6 and u1 > 0.9
    assert sharding.length_sorted_batches([5, 9, 5, 9], 2) == [[1, 3], [0, 2]]
    assert sharding.slot_utilisation([4, 4], [[0, 1]]) == 1.0 and sharding.slot_utilisation([], []) == 1.0
    # the length proxy: frames of real audio recovered from the trailing padding of a log-mel
    cfg = synthetic.get_config("whisper-tiny.en")
    mel = torch.from_numpy(np.stack([synthetic.make_mel_padded(cfg, 0, 2.5), synthetic.make_mel_padded(cfg, 1, 30.0),
                                     synthetic.make_mel_padded(cfg, 2, 0.01)]))
    assert audio.valid_frames(mel) == [250, 3000, 1]
    assert audio.valid_frames(torch.from_numpy(synthetic.make_mel(cfg, 0, 1))) == [3000]
