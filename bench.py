#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of whisper-medium.en fp32 greedy decoding on MI355X.

One "step" = one pass of the hot path over one batch per GPU: encoder (8 x 30 s of 80x3000 synthetic log-mel)
+ cross-KV projection + greedy decode to max_length=448 (447 decoder steps; random-init weights never emit
EOS, so this is the natural, no-work-skipped length).  Inputs are resident in HBM before the timed region.
Utterance batches shard embarrassingly over ranks (no data-path collective); torch.distributed is used only
for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0 (see README/DESIGN for the fields `roofline` and `cpu_baseline`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3   # v_mfma_f32_32x32x2_f32 dense peak


def usable_cores() -> int:
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def under_rocprof() -> bool:
    """rocprofv3's tool library is loaded into this process (whisper_trtllm_amd.runtime.under_rocprof: WhisperPipeline clamps itself to one
    worker then -- a 4-worker run under `rocprofv3 --kernel-trace` aborted in round 3, DESIGN.md "The four-worker abort under rocprofv3"),
    so the multi-worker legs would only repeat the single-worker ones: they are skipped."""
    import whisper_trtllm_amd as w
    return w.runtime.under_rocprof()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="whisper-medium.en")
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--max-length", type=int, default=448)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-batch16", action="store_true", help="skip the secondary batch-16 measurement")
    ap.add_argument("--no-varlen", action="store_true", help="skip the variable-length (LibriSpeech-like) workload")
    ap.add_argument("--no-fp16-decoder", action="store_true", help="skip the fp16-engine (encoder + decoder) batch-16 measurement")
    ap.add_argument("--varlen-utterances", type=int, default=64, help="utterances per GPU in the variable-length workload")
    ap.add_argument("--varlen-long", type=int, default=256, help="utterances per GPU of the LONGER variable-length run (continuous mode and "
                    "length-sorted batches only): a 64-utterance workload is mostly ramp-down -- the last utterances decode beside empty slots")
    ap.add_argument("--no-two-workers", action="store_true", help="skip the two-workers-per-GPU figures (WhisperPipeline)")
    ap.add_argument("--workers", type=int, default=1, help="engine pairs per GPU for the TIMED steps (runtime.WhisperPipeline).  Default 1 = one batch in "
                    "flight per GPU, the configuration the metric is quoted on; N > 1 runs the K steps N at a time (N x batch utterances in flight) "
                    "and says so in config.workers_per_gpu")
    ap.add_argument("--cpu-decode-steps", type=int, default=128, help="decoder steps timed on the CPU (about 10 s of CPU work in total)")
    ap.add_argument("--encoder-precision", default="float32", choices=["float32", "float16"],
                    help="float16 = BASELINE config 4 (fp16 encoder + fp32 decoder); the headline metric is float32")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: map every rank onto the visible GPUs round-robin")
    ap.add_argument("--all-legs", action="store_true", help="with --gpus N > 1: also run the secondary legs (default there: the timed headline leg only)")
    args = ap.parse_args()
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not args.all_legs:
        # A multi-GPU run measures the scaling of the HEADLINE: the secondary legs (variable-length workload, batch 16, fp16 engines, up to
        # 4 worker threads per rank, per-kernel rooflines, CPU baseline) only lengthen an 8-GPU lease for numbers that are read at N = 1.
        args.no_varlen = args.no_batch16 = args.no_fp16_decoder = args.no_two_workers = args.no_roofline = args.no_cpu_baseline = True
    if os.environ.get("WT_SAVE_MAPS"):   # tools/profile_round.sh: the module map of every profiled process, so that a crash trace is symbolisable
        try:
            import shutil
            shutil.copyfile("/proc/self/maps", os.environ["WT_SAVE_MAPS"])
        except OSError:
            pass
    if under_rocprof():
        if args.workers > 1:
            raise SystemExit("bench.py --workers N > 1 cannot run under rocprofv3 (see under_rocprof)")
        if not args.no_two_workers:
            log("[bench] rocprofv3 detected: skipping the multi-worker legs")
        args.no_two_workers = True

    import numpy as np
    import torch
    import whisper_trtllm_amd as w

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: there is no CPU execution path for the engine")
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    cfg = w.synthetic.get_config(args.model)
    cfg["max_length"] = args.max_length
    B, S, d, L, H, V = args.batch, cfg["max_source_positions"], cfg["d_model"], cfg["decoder_layers"], cfg["decoder_attention_heads"], cfg["vocab_size"]
    t0 = time.time()
    weights = w.synthetic.make_weights(cfg, args.seed)
    enc_blob = w.convert.build_encoder_engine(cfg, weights, precision=args.encoder_precision)
    dec_blob = w.convert.build_decoder_engine(cfg, weights)
    enc = w.WhisperEncoderEngine(enc_blob)
    dec = w.WhisperDecoderEngine(dec_blob, cfg)
    mel = torch.from_numpy(w.synthetic.make_mel(cfg, index=rank * B, batch=B)).cuda()
    if rank == 0:
        log(f"[bench] engines built in {time.time() - t0:.1f}s ({args.model}, B={B}/GPU, world {world})")

    def one_pass(force_eos_step=None):
        hidden = enc(mel)
        return dec.generate(hidden, force_eos_step=force_eos_step)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, **kw):
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            ids = one_pass(**kw)
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        el = w.sharding.max_over_ranks(el, dist)
        barrier()
        return el, ids

    if args.workers > 1:   # opt-in: the K timed steps handed to N workers (each step is still one pass over one batch of B utterances)
        head_pipe = w.WhisperPipeline(enc_blob, dec_blob, cfg, workers=args.workers)
        head_pipe.transcribe([mel] * max(args.warmup, args.workers))
        barrier()
        t = time.perf_counter()
        ids = head_pipe.transcribe([mel] * args.steps)[-1]
        torch.cuda.synchronize()
        elapsed = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
        barrier()
        del head_pipe
    else:
        for _ in range(args.warmup):
            ids = one_pass()
        elapsed, ids = timed(args.steps)
    assert ids.shape == (B, args.max_length), ids.shape
    audio_s = 30.0 * B * world * args.steps
    value = audio_s / elapsed
    ms_per_step = 1e3 * elapsed / args.steps

    # secondary figure: LibriSpeech-like transcript length (EOS forced at decoder step 32)
    one_pass(force_eos_step=32)
    n32_passes = max(1, min(args.steps, 5))
    el32, ids32 = timed(n32_passes, force_eos_step=32)
    value_n32 = 30.0 * B * world * n32_passes / el32

    # secondary figure: 16 utterances per GPU in one engine batch (the decode step is launch-latency bound at B = 8, so
    # throughput still grows with the batch; BASELINE config 4 uses B = 16)
    value_b16 = None
    if B == 8 and not args.no_batch16:
        mel16 = torch.cat([mel, torch.from_numpy(w.synthetic.make_mel(cfg, index=(world + rank) * B, batch=B)).cuda()])
        pass16 = lambda: dec.generate(enc(mel16))
        pass16()
        barrier()
        t16 = time.perf_counter()
        pass16()
        torch.cuda.synchronize()
        el16 = w.sharding.max_over_ranks(time.perf_counter() - t16, dist)
        barrier()
        value_b16 = 30.0 * 16 * world / el16

    # secondary figure: the realistic-transcript regime.  A seeded LibriSpeech-like set of utterances per GPU (durations and
    # transcript lengths modelled on test-clean, synthetic.librispeech_like_lengths; each log-mel is padded behind its audio like a
    # real clip, row i is made to emit EOS at its own step), decoded in batches of B: once in dataset order (what cal_wer.py did
    # until round 3) and once with length-aware batching (sharding.length_sorted_batches over the duration recovered from the mel's
    # trailing padding).  Slot utilisation = sum of row lengths / sum over batches of B x longest row.
    varlen = None
    if not args.no_varlen:
        n_utt = args.varlen_utterances
        dur, eos_steps = w.synthetic.librispeech_like_lengths(n_utt, seed=rank, max_length=args.max_length)
        vmel = torch.from_numpy(np.stack([w.synthetic.make_mel_padded(cfg, rank * n_utt + i, dur[i]) for i in range(n_utt)])).cuda()

        def run_groups(groups):
            for g in groups:
                dec.generate(enc(vmel[g]), force_eos_steps=[eos_steps[i] for i in g])

        def time_groups(groups):
            run_groups(groups[:1])
            barrier()
            t = time.perf_counter()
            run_groups(groups)
            torch.cuda.synchronize()
            el = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
            barrier()
            return el

        in_order = [list(range(a, b)) for a, b in w.sharding.batches(0, n_utt, B)]
        by_length = w.sharding.length_sorted_batches(w.audio.valid_frames(vmel), B)
        row_steps = [s_ + 1 for s_ in eos_steps]
        el_order, el_sorted = time_groups(in_order), time_groups(by_length)
        # ... and in ARRIVAL order through the continuous mode (wt_decoder_stream_*): B slots stay busy, every utterance stops at its own
        # EOS and the kernel that sees it refills the slot from the waiting queue in the same step (runtime.transcribe_continuous)
        cstats = {}
        w.transcribe_continuous(enc, dec, vmel[:2 * B], slots=B, chunk=B, force_eos_steps=eos_steps[:2 * B])
        barrier()
        t = time.perf_counter()
        cont_ids = w.transcribe_continuous(enc, dec, vmel, slots=B, chunk=B, force_eos_steps=eos_steps, stats=cstats)
        torch.cuda.synchronize()
        el_cont = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
        barrier()
        assert [len(r) - 1 for r in cont_ids] == row_steps, "continuous mode: an utterance did not stop at its own EOS"
        long_run = None
        if args.varlen_long > n_utt:   # the same two plans over a workload long enough that the ramp-down is a small part of it
            n_long = args.varlen_long
            dur_l, eos_l = w.synthetic.librispeech_like_lengths(n_long, seed=1000 + rank, max_length=args.max_length)
            lmel = torch.from_numpy(np.stack([w.synthetic.make_mel_padded(cfg, 7000 + rank * n_long + i, dur_l[i]) for i in range(n_long)])).cuda()
            lstats = {}
            barrier()
            t = time.perf_counter()
            ids_l = w.transcribe_continuous(enc, dec, lmel, slots=B, chunk=B, force_eos_steps=eos_l, stats=lstats)
            torch.cuda.synchronize()
            el_lc = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
            barrier()
            assert [len(r) - 2 for r in ids_l] == eos_l
            groups_l = w.sharding.length_sorted_batches(w.audio.valid_frames(lmel), B)
            barrier()
            t = time.perf_counter()
            for g in groups_l:
                dec.generate(enc(lmel[g]), force_eos_steps=[eos_l[i] for i in g])
            torch.cuda.synchronize()
            el_ls = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
            barrier()
            # the same arrival-order run with 16 decode slots (and encoder chunks of 16): a decode step costs about the same for 16 rows as
            # for 8 (it is launch-latency bound), so this is what a deployment that is free to choose its slot count gets; NOT the batch the
            # metric is quoted on
            s16stats = {}
            w.transcribe_continuous(enc, dec, lmel[:32], slots=16, chunk=16, force_eos_steps=eos_l[:32])   # replans workspace and graphs
            barrier()
            t = time.perf_counter()
            ids_16 = w.transcribe_continuous(enc, dec, lmel, slots=16, chunk=16, force_eos_steps=eos_l, stats=s16stats)
            torch.cuda.synchronize()
            el_l16 = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
            barrier()
            assert all(len(a) == len(b_) and (a == b_).all() for a, b_ in zip(ids_16, ids_l)), "16 slots: ids differ from 8 slots"
            dec.generate(enc(lmel[:B]), force_eos_steps=eos_l[:B])   # back to the batch-B plan for the legs below
            long_run = {"utterances_per_gpu": n_long, "mean_decoder_steps": round(float(np.mean(eos_l)) + 1, 1),
                        "dataset_order_continuous_16_slots": {"value": round(30.0 * n_long * world / el_l16, 2), "slot_utilisation": round(s16stats["slot_utilisation"], 4),
                                                              "decoder_steps": s16stats["steps"]},
                        "dataset_order_continuous": {"value": round(30.0 * n_long * world / el_lc, 2), "slot_utilisation": round(lstats["slot_utilisation"], 4),
                                                     "decoder_steps": lstats["steps"]},
                        "length_sorted": {"value": round(30.0 * n_long * world / el_ls, 2),
                                          "slot_utilisation": round(w.sharding.slot_utilisation([s_ + 1 for s_ in eos_l], groups_l), 4)}}
            del lmel
        varlen = {"utterances_per_gpu": n_utt, "mean_duration_s": round(float(np.mean(dur)), 2), "mean_decoder_steps": round(float(np.mean(row_steps)), 1),
                  "max_decoder_steps": int(max(row_steps)),
                  "dataset_order": {"value": round(30.0 * n_utt * world / el_order, 2), "real_audio_s_per_s": round(float(np.sum(dur)) * world / el_order, 2),
                                    "slot_utilisation": round(w.sharding.slot_utilisation(row_steps, in_order), 4)},
                  "length_sorted": {"value": round(30.0 * n_utt * world / el_sorted, 2), "real_audio_s_per_s": round(float(np.sum(dur)) * world / el_sorted, 2),
                                    "slot_utilisation": round(w.sharding.slot_utilisation(row_steps, by_length), 4)},
                  "dataset_order_continuous": {"value": round(30.0 * n_utt * world / el_cont, 2), "real_audio_s_per_s": round(float(np.sum(dur)) * world / el_cont, 2),
                                               "slot_utilisation": round(cstats["slot_utilisation"], 4), "decoder_steps": cstats["steps"]},
                  "long_run": long_run,
                  "note": "value = 30 s windows per second as in the headline; lengths modelled on LibriSpeech test-clean (no dataset on the box), "
                          "EOS forced per row; rank r uses seed r; dataset_order_continuous = arrival order through the continuous mode "
                          "(slots refilled on the device the step an utterance stops)"}
        if not args.no_two_workers:
            # the same length-sorted batches handed to 2 and 4 WORKERS per GPU (runtime.WhisperPipeline: N engine pairs, N host threads,
            # N streams): one worker's MFMA-bound encoder and launch-latency-bound decode fill the gaps of the others' decodes.
            # ... and the headline's own passes (8 x 30 s, 447 decoder steps each) N at a time.  NOT the headline: `value` keeps ONE batch
            # of 8 in flight per GPU, as the metric is quoted; this is N x 8 in flight (compare value_batch16_per_gpu).
            mb = [vmel[g] for g in by_length]
            kw = [{"force_eos_steps": [eos_steps[i] for i in g]} for g in by_length]
            varlen["length_sorted_workers"], varlen["headline_passes_workers"] = {}, {}
            for nw in (2, 4):
                pipe = w.WhisperPipeline(enc_blob, dec_blob, cfg, workers=nw)
                pipe.transcribe(mb[:nw], kw[:nw])
                barrier()
                t = time.perf_counter()
                pipe.transcribe(mb, kw)
                torch.cuda.synchronize()
                el2 = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
                barrier()
                varlen["length_sorted_workers"][str(nw)] = round(30.0 * n_utt * world / el2, 2)
                # ... and the arrival-order workload through one continuous stream per worker (blocks of 32 utterances each)
                pipe.transcribe_continuous(vmel[:2 * B], slots=B, chunk=B, block=32, force_eos_steps=eos_steps[:2 * B])
                barrier()
                t = time.perf_counter()
                pipe.transcribe_continuous(vmel, slots=B, chunk=B, block=32, force_eos_steps=eos_steps)
                torch.cuda.synchronize()
                el2c = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
                barrier()
                varlen.setdefault("continuous_workers", {})[str(nw)] = round(30.0 * n_utt * world / el2c, 2)
                n2 = nw * max(1, min(args.steps, 3))   # bounded: the secondary legs must not scale with a large --steps
                pipe.transcribe([mel] * nw)
                barrier()
                t = time.perf_counter()
                pipe.transcribe([mel] * n2)
                torch.cuda.synchronize()
                el2h = w.sharding.max_over_ranks(time.perf_counter() - t, dist)
                barrier()
                varlen["headline_passes_workers"][str(nw)] = round(30.0 * B * world * n2 / el2h, 2)
                del pipe
            del mb
        del vmel

    # secondary figure: fp16 ENGINES (build_encoder.py / build_decoder.py --engine_precision float16): half GEMM / GEMV operands and
    # half resident K/V caches, fp32 accumulate / softmax / LayerNorm / residual; batch 16.  Never the headline (which is fp32).
    fp16 = None
    if B == 8 and not args.no_fp16_decoder and not args.no_batch16:
        enc_h = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights, precision="float16"))
        dec_h = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights, precision="float16"), cfg)
        pass_h = lambda: dec_h.generate(enc_h(mel16))
        pass_h()
        barrier()
        th = time.perf_counter()
        pass_h()
        torch.cuda.synchronize()
        elh = w.sharding.max_over_ranks(time.perf_counter() - th, dist)
        barrier()
        fp16 = {"value": round(30.0 * 16 * world / elh, 2), "ms_per_pass": round(elh * 1e3, 2)}
        if rank == 0 and not args.no_roofline:
            dec_h.begin(enc_h(mel16))
            dec_h.steps(args.max_length - 1)
            dec_h.poll()
            Fd_ = cfg["decoder_ffn_dim"]
            kinds_h = {"qkv": 3 * d * d * 2, "self_attn": 16 * H * (args.max_length - 1) * 64 * 2 * 2, "pair": 3 * d * d * 2, "cross_attn": 16 * H * S * 64 * 2 * 2,
                       "cross_out": d * d * 2, "fc1": d * Fd_ * 2, "fc2": d * Fd_ * 2}
            fp16["roofline_skinny"] = {}
            for kind, nbytes in kinds_h.items():
                us = dec_h.time_kernel(kind, iters=20)
                fp16["roofline_skinny"][kind] = {"bytes_per_launch": int(nbytes), "avg_launch_us": round(us, 2), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        del enc_h, dec_h

    out = {
        "metric": f"audio-sec/s, {args.model} fp32 greedy", "value": round(value, 2), "unit": "audio-seconds/second",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.encoder_precision == "float32" else "f16 encoder GEMM operands (f32 accumulate) + f32 decoder", "data": "synthetic",
        "config": {"workload": f"{args.model} {'fp32' if args.encoder_precision == 'float32' else 'fp16-encoder/fp32-decoder'} greedy, batch {B} per GPU x 30 s / 80x3000 synthetic log-mel, "
                               f"encoder + {args.max_length - 1} decoder steps (max_length {args.max_length}), random-init weights",
                   "encoder_gemm": ("fp32 operands exactly split into three bf16 planes (x = b1 + b2 + b3 to 2^-27), six bf16 MFMAs per block, fp32 accumulate: "
                                    "as accurate as the fp32 MFMA instruction (tests/test_gpu_kernels.py::test_gemm_x3_is_an_fp32_gemm)"
                                    if enc.session._lib.wt_engine_gemm_mode(enc.session.handle) == 1 else "native MFMA of the engine's precision"),
                   "batch_per_gpu": B, "workers_per_gpu": args.workers, "decode_steps": args.max_length - 1, "sharding": f"utterance-parallel x{world}, no collective",
                   "value_n32_decode_steps": round(value_n32, 2),
                   "value_batch16_per_gpu": round(value_b16, 2) if value_b16 else None,
                   "value_varlen": varlen["length_sorted"]["value"] if varlen else None,
                   "value_varlen_continuous": varlen["dataset_order_continuous"]["value"] if varlen else None,
                   "value_varlen_continuous_16_slots": varlen["long_run"]["dataset_order_continuous_16_slots"]["value"] if varlen and varlen.get("long_run") else None,
                   "value_varlen_4_workers": varlen["length_sorted_workers"]["4"] if varlen and "length_sorted_workers" in varlen else None,
                   "value_2_workers_per_gpu": varlen["headline_passes_workers"]["2"] if varlen and "headline_passes_workers" in varlen else None,
                   "value_4_workers_per_gpu": varlen["headline_passes_workers"]["4"] if varlen and "headline_passes_workers" in varlen else None,
                   "value_fp16_decoder_b16": fp16["value"] if fp16 else None, "wer": None},
    }
    if varlen:
        out["varlen"] = varlen
    if fp16:
        out["fp16_engines_b16"] = fp16

    if rank == 0 and not args.no_roofline:
        # per-kernel durations: an instrumented eager pass with hipEvents around every launch of the timed kernels
        dec.set_profiling(True)
        enc_lib = enc.session._lib
        w._lib.check(enc_lib.wt_engine_set_profiling(enc.session.handle, 1), "set_profiling")
        one_pass()
        torch.cuda.synchronize()
        us_cross = dec.time_cross_attention(iters=40)   # graph-replayed launches of the kernel alone, hipEvents on the launch stream
        ms_vocab, n_vocab = dec.timer("vocab_proj")
        import ctypes
        kt = w._lib.KernelTimer()
        w._lib.check(enc_lib.wt_engine_get_timer(enc.session.handle, b"gemm_f32", ctypes.byref(kt)), "get_timer")
        ms_gemm, n_gemm = kt.ms_total, kt.launches
        w._lib.check(enc_lib.wt_engine_get_timer(enc.session.handle, b"enc_attn", ctypes.byref(kt)), "get_timer")
        ms_eattn, n_eattn = kt.ms_total, kt.launches
        dec.set_profiling(False)
        w._lib.check(enc_lib.wt_engine_set_profiling(enc.session.handle, 0), "set_profiling")
        # dominant kernel by time: decoder cross-attention (streams the utterances' resident K/V once per step)
        bytes_cross = B * H * S * 64 * 4 * 2          # cross-KV bytes/step/utt / L, x B utterances per launch
        avg_cross = us_cross * 1e-6
        # HBM bytes per launch from the committed PMC passes (bench.py cannot run rocprofv3 on itself; tools/profile_round.sh pmc writes the
        # file).  Reported only while the kernel source it was measured on is the one in the tree: a stale file yields traffic = null.
        traffic, traffic_from = None, None
        try:
            import hashlib
            t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_dominant_kernel.json")))
            sha = hashlib.sha256(open(os.path.join(ROOT, "whisper-trtllm_amd", "csrc", "kernels_decoder.hip"), "rb").read()).hexdigest()[:16]
            if B == 8 and args.model == "whisper-medium.en":
                if t.get("kernels_decoder_sha16") == sha:
                    traffic = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
                    traffic_from = f"profiles/pmc_traffic_dominant_kernel.json ({t.get('source', '')}; kernels_decoder.hip {sha})"
                else:
                    traffic_from = f"stale: profiles/pmc_traffic_dominant_kernel.json was measured on kernels_decoder.hip {t.get('kernels_decoder_sha16')}, the tree holds {sha}"
        except (OSError, KeyError, ValueError):
            pass
        ach = bytes_cross / avg_cross / 1e9
        out["roofline"] = {"bound": "hbm", "kernel": "dec_attn_kernel (cross-attention, S=1500)", "achieved": round(ach, 1),
                           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_from": traffic_from,
                           "bytes_per_launch": bytes_cross, "avg_launch_us": round(avg_cross * 1e6, 2), "launches": 40 * L,
                           "timing": "hipGraph replay of the L per-layer launches over the resident caches, hipEvents on the launch stream"}
        F_, C = cfg["encoder_ffn_dim"], cfg["num_mel_bins"]
        enc_gemm_flop = B * (2 * 2 * S * d * 3 * C + 2 * S * d * 3 * d + cfg["encoder_layers"] * (8 * S * d * d + 4 * S * d * F_))
        gemm_tf = enc_gemm_flop / (ms_gemm * 1e-3) / 1e12 if ms_gemm > 0 else 0.0
        attn_flop = B * cfg["encoder_layers"] * 4 * H * S * S * 64
        half = args.encoder_precision == "float16"
        enc_peak = 2500.0 if half else MFMA_F32_PEAK_TF
        x3 = (not half) and enc_lib.wt_engine_gemm_mode(enc.session.handle) == 1
        if x3:
            # fp32 products formed on the bf16 matrix cores: every operand exactly split into three bf16 planes, six v_mfma_f32_32x32x16_bf16
            # per block (DESIGN.md section 5 "fp32 GEMM on the bf16 matrix cores").  The roofline is the bf16 MFMA peak against the ISSUED bf16
            # flops (6 x the useful fp32 ones); the useful rate is reported beside it, with the fp32 MFMA peak it replaces.
            out["roofline_encoder"] = {"bound": "mfma", "kernel": "gemm_x3_kernel (v_mfma_f32_32x32x16_bf16 x 6 per block: fp32 operands exactly split into three bf16 planes, fp32 accumulate)",
                                       "achieved": round(6 * gemm_tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(6 * gemm_tf / 2500.0, 4),
                                       "useful_fp32_tflops": round(gemm_tf, 2), "fp32_mfma_peak": MFMA_F32_PEAK_TF, "useful_vs_fp32_mfma_peak": round(gemm_tf / MFMA_F32_PEAK_TF, 4),
                                       "launches": int(n_gemm), "total_ms": round(ms_gemm, 3),
                                       "enc_attn_tflops": round(attn_flop / (ms_eattn * 1e-3) / 1e12, 2) if ms_eattn > 0 else None,
                                       "enc_attn_total_ms": round(ms_eattn, 3),
                                       "enc_attn_kernel": "enc_attn_kernel (v_mfma_f32_32x32x2_f32)" if (os.environ.get("WT_TUNING") == "1" and os.environ.get("WT_ATTN_X3") == "0")
                                                          else "enc_attn_x3_kernel (both products as 6 bf16 MFMAs of exactly split operands, fp32 scores / softmax / accumulators); enc_attn_tflops = useful fp32 flops"}
            # what the bf16 matrix pipe of THIS device sustains on random operands with no memory traffic (a bare MFMA loop run as a child
            # process for ~2 s: tools/probes/mfma_power --quick): the x3 kernels run against the chip's power management, so the data-sheet
            # peak is not reachable on real data by any schedule
            probe = os.path.join(ROOT, "tools", "probes", "mfma_power", "mfma_power")
            if os.path.exists(probe) and not under_rocprof():
                try:
                    import subprocess
                    torch.cuda.synchronize()
                    line = subprocess.run([probe, "--quick"], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[-1]
                    sustained = json.loads(line)
                    out["roofline_encoder"]["bare_mfma_loop_random_operands"] = sustained
                    out["roofline_encoder"]["frac_of_bare_mfma_loop"] = round(6 * gemm_tf / sustained["tflops"], 4)
                except Exception as exc:   # the probe is context, never a reason to lose the bench line
                    log(f"mfma_power probe failed: {exc!r}")
        else:
            out["roofline_encoder"] = {"bound": "mfma", "kernel": "gemm_f16_dma4_kernel / gemm_f16_dma3_kernel / gemm_f16_dma_kernel (v_mfma_f32_16x16x32_f16)" if half else "gemm_f32_dma_kernel (v_mfma_f32_32x32x2_f32)",
                                       "achieved": round(gemm_tf, 2), "peak": enc_peak, "unit": "TFLOP/s", "frac": round(gemm_tf / enc_peak, 4),
                                       "launches": int(n_gemm), "total_ms": round(ms_gemm, 3),
                                       "enc_attn_tflops": round(attn_flop / (ms_eattn * 1e-3) / 1e12, 2) if ms_eattn > 0 else None,
                                       "enc_attn_total_ms": round(ms_eattn, 3)}
        # whole decode phase against the HBM roofline: algorithmic bytes of step t for a batch of B =
        #   4*P_step + B*[ 4*L*2*H*S*64 (cross KV) + 4*L*2*H*(t+1)*64 (self KV read) + 4*L*2*H*64 (append) + 4*V (logits) ]
        n_steps = args.max_length - 1
        hidden = enc(mel)
        dec.begin(hidden)
        torch.cuda.synchronize()
        t_dec0 = time.perf_counter()
        dec.steps(n_steps)
        torch.cuda.synchronize()
        t_dec = time.perf_counter() - t_dec0
        p_step = L * (6 * d * d + 2 * d * cfg["decoder_ffn_dim"]) + V * d
        per_utt_fixed = 4 * L * 2 * H * S * 64 + 4 * L * 2 * H * 64 + 4 * V
        bytes_total = n_steps * (4 * p_step + B * per_utt_fixed) + B * 4 * L * 2 * H * 64 * (n_steps * (n_steps + 1) // 2)
        out["roofline_decode"] = {"bound": "hbm", "achieved": round(bytes_total / t_dec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(bytes_total / t_dec / 1e9 / HBM_PEAK_GBS, 4), "steps": n_steps,
                                  "ms_per_step": round(t_dec / n_steps * 1e3, 4), "bytes_per_step_avg": int(bytes_total / n_steps),
                                  "launches_per_step": 7 * L + 2,
                                  "note": "all decoder steps of one batch, wall clock over the replayed step graphs"}
        if not args.no_two_workers and not args.no_varlen:
            # the decode phase alone with FOUR chains in flight (runtime.WhisperPipeline's engines, one host thread each): four times the
            # bytes of one chain over the wall clock of all four -- what the launch-latency-bound chain leaves on the table
            import threading
            pipe4 = w.WhisperPipeline(enc_blob, dec_blob, cfg, workers=4)
            for k, (enc_k, dec_k) in enumerate(pipe4.engines):
                with torch.cuda.stream(pipe4.streams[k]):
                    dec_k.begin(enc_k(mel))

            def chain(k):
                with torch.cuda.stream(pipe4.streams[k]):
                    pipe4.engines[k][1].run()
                    pipe4.streams[k].synchronize()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            th = [threading.Thread(target=chain, args=(k,)) for k in range(4)]
            for x in th:
                x.start()
            for x in th:
                x.join()
            t4 = time.perf_counter() - t4
            out["roofline_decode_4_workers"] = {"bound": "hbm", "achieved": round(4 * bytes_total / t4 / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                "frac": round(4 * bytes_total / t4 / 1e9 / HBM_PEAK_GBS, 4), "chains": 4, "ms_all_chains": round(t4 * 1e3, 2),
                                                "note": "four independent decode chains of batch 8 (all 447 steps each) running concurrently; NOT the headline configuration"}
            del pipe4
        vocab_bytes = V * d * 4
        out["roofline_vocab_proj"] = {"bound": "hbm", "achieved": round(vocab_bytes / (ms_vocab / max(1, n_vocab) * 1e-3) / 1e9, 1),
                                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_us": round(ms_vocab / max(1, n_vocab) * 1e3, 2)}

        # every launch kind of a decode step against the HBM roofline: algorithmic bytes of one launch / its graph-replayed average
        # duration over the L layers' own weights (wt_decoder_time_kernel).  self_attn is timed at the current cache length.
        Fd = cfg["decoder_ffn_dim"]
        kinds = {"qkv": 3 * d * d * 4, "self_attn": B * H * n_steps * 64 * 4 * 2, "pair": 3 * d * d * 4, "cross_attn": bytes_cross,
                 "cross_out": d * d * 4, "fc1": d * Fd * 4, "fc2": d * Fd * 4}
        sk = {}
        for kind, nbytes in kinds.items():
            us = dec.time_kernel(kind, iters=20)
            sk[kind] = {"bytes_per_launch": int(nbytes), "avg_launch_us": round(us, 2), "achieved": round(nbytes / us / 1e3, 1),
                        "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBS, 4)}
        sk["launch_time_sum_us_per_layer"] = round(sum(v["avg_launch_us"] for v in sk.values()), 2)
        out["roofline_skinny"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "per_launch_kind": sk,
                                  "note": "the 7 dependent launches of a decoder layer, each kind graph-replayed alone over all L layers "
                                          f"(self_attn at cache length {n_steps}); a step = L x these + vocabulary projection + finish"}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # reported at N = 1 only (the other ranks would idle behind it)
        # CPU baseline: the oracle (torch-CPU fp32 port of the reference's bundled HF path) on this box's host cores,
        # bounded sample: 1 utterance, full encoder + a few decoder steps; per-step time extrapolated to max_length-1.
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import cpu_ref
        cores = min(usable_cores(), 64)
        torch.set_num_threads(cores)
        Wt = cpu_ref.to_torch(weights)
        x = mel[:1].cpu()
        with torch.no_grad():
            t = time.perf_counter()
            h = cpu_ref.encoder_forward(Wt, cfg, x)
            t_enc = time.perf_counter() - t
            idt = torch.full((1, 1), cfg["decoder_start_token_id"], dtype=torch.long)
            past, t_steps = None, []
            for i in range(args.cpu_decode_steps):
                t = time.perf_counter()
                lg, past = cpu_ref.decoder_forward(Wt, cfg, idt, h, past)
                idt = lg[:, -1].argmax(-1, keepdim=True)
                t_steps.append(time.perf_counter() - t)
        step_avg = float(np.mean(t_steps[1:])) if len(t_steps) > 1 else t_steps[0]
        # later steps attend over a longer self cache: time a few steps at two more cache lengths (self K/V of that length fabricated --
        # timing only) and integrate the per-step time piecewise-linearly over all max_length-1 steps
        n_steps_total = args.max_length - 1
        anchors = [(float(np.mean(range(1, max(2, len(t_steps))))), step_avg)]
        with torch.no_grad():
            for t_len in (n_steps_total // 2, n_steps_total - 8):
                if past is None or t_len <= len(t_steps) + 8:
                    continue
                fake = tuple((torch.randn(sk.shape[0], sk.shape[1], t_len, sk.shape[3]) * 0.1, torch.randn(sv.shape[0], sv.shape[1], t_len, sv.shape[3]) * 0.1, ck, cv)
                             for (sk, sv, ck, cv) in past)
                ts = []
                for _ in range(4):
                    t = time.perf_counter()
                    lg, fake = cpu_ref.decoder_forward(Wt, cfg, idt, h, fake)
                    ts.append(time.perf_counter() - t)
                anchors.append((float(t_len + 2), float(np.mean(ts[1:]))))
                del fake
        xs_, ys_ = [a for a, _ in anchors], [b for _, b in anchors]
        per_step = np.interp(np.arange(1, n_steps_total), xs_, ys_)          # steps 1 .. n-1 (step 0 is timed on its own: it includes cross-KV)
        total = t_enc + t_steps[0] + float(per_step.sum())
        out["cpu_baseline"] = {"value": round(30.0 / total, 3), "unit": "audio-seconds/second", "cores": cores, "kind": "port",
                               "sample": f"1 utterance: full encoder ({t_enc:.2f}s) + {args.cpu_decode_steps} decoder steps "
                                         f"(first {t_steps[0]:.3f}s incl. cross-KV, then {step_avg * 1e3:.1f} ms/step) + 3 steps each at self-cache lengths "
                                         + ", ".join(f"{int(a)}: {b * 1e3:.1f} ms" for a, b in anchors[1:]) +
                                         f"; per-step time interpolated over all {n_steps_total} steps, batch 1 as in run.py:296-315"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
