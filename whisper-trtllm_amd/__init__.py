"""whisper-trtllm_amd: MI355X-native Whisper encoder-decoder greedy ASR engine (placeholder init)."""
