#!/usr/bin/env python3
"""Does partitioning the chip's CUs between the encoder and the decode pay in the short-transcript regime?

The encoder (MFMA / power bound, workgroups that hold a CU for 150-600 us) and the greedy decode (a chain of 170 dependent 5-18 us
launches per token) leave each other's resources idle, but two unconstrained streams overlap badly: a decode launch waits for a CU that a
GEMM workgroup holds.  This probe runs an encoder loop on a stream created with hipExtStreamCreateWithCUMask (N of the 256 CUs) beside a
decode loop on an ordinary stream and prints both rates alone and together.

usage: cu_mask_probe.py [model] [seconds per arm]"""
import ctypes, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import whisper_trtllm_amd as w

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium.en"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
B, STEPS = 8, 32
hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (mask bit numbers) the stream may use"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


cfg = w.synthetic.get_config(model)
weights = w.synthetic.make_weights(cfg, 0)
enc = w.WhisperEncoderEngine(w.convert.build_encoder_engine(cfg, weights))
dec = w.WhisperDecoderEngine(w.convert.build_decoder_engine(cfg, weights), cfg)
mel = torch.from_numpy(w.synthetic.make_mel(cfg, index=0, batch=B)).cuda()
hidden = enc(mel)
dec.generate(hidden, max_length=STEPS + 1, force_eos_step=STEPS + 8)
torch.cuda.synchronize()


def enc_loop(stream, stop, out):
    torch.cuda.set_device(0)
    n = 0
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        while not stop.is_set():
            enc(mel)
            stream.synchronize()
            n += 1
        out["enc"] = n / (time.perf_counter() - t0)


def dec_loop(stream, stop, out):
    torch.cuda.set_device(0)
    n = 0
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        while not stop.is_set():
            dec.generate(hidden, max_length=STEPS + 1, force_eos_step=STEPS + 8)
            n += STEPS
        stream.synchronize()
        out["dec"] = n / (time.perf_counter() - t0)


def arm(name, enc_stream, dec_stream, which=("enc", "dec")):
    stop, out = threading.Event(), {}
    ths = []
    if "enc" in which:
        ths.append(threading.Thread(target=enc_loop, args=(enc_stream, stop, out)))
    if "dec" in which:
        ths.append(threading.Thread(target=dec_loop, args=(dec_stream, stop, out)))
    for t in ths:
        t.start()
    time.sleep(secs)
    stop.set()
    for t in ths:
        t.join()
    e, d = out.get("enc"), out.get("dec")
    # utterances/s the pair sustains when every utterance needs one encoder row and `STEPS` decode row-steps
    both = min(e * B, d * B / STEPS) if e and d else None
    print(f"{name:<58} encoder {'%7.2f passes/s' % e if e else '      -        '}   decode {'%8.1f steps/s (%.3f ms/step)' % (d, 1e3 / d) if d else '    -'}"
          + (f"   => {both:6.1f} utt/s = {both * 30:7.0f} audio-s/s at {STEPS} steps/utterance" if both else ""), flush=True)


plain_e, plain_d = torch.cuda.Stream(), torch.cuda.Stream()
arm("encoder alone, plain stream", plain_e, None, ("enc",))
arm("decode alone, plain stream", None, plain_d, ("dec",))
arm("both, two plain streams", plain_e, plain_d)
hi = torch.cuda.Stream(priority=-1)
arm("both, decode on a high-priority stream", plain_e, hi)
for n_cu, bits, label in (
        (192, range(192), "mask bits 0..191"),
        (192, [b for b in range(256) if b % 4 != 3], "mask bits b % 4 != 3"),
        (192, [b for b in range(256) if (b // 8) % 4 != 3], "mask bits (b / 8) % 4 != 3"),
        (224, range(224), "mask bits 0..223"),
        (128, range(128), "mask bits 0..127")):
    ms = masked_stream(bits)
    arm(f"encoder alone on {n_cu} CUs ({label})", ms, None, ("enc",))
    arm(f"both, encoder on {n_cu} CUs ({label})", ms, plain_d)
    arm(f"both, encoder on {n_cu} CUs, decode high priority", ms, hi)
