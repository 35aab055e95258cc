"""`WhisperEncoder` / `WhisperDecoder` model definitions with the reference's constructor arguments and
parameter tree (tensorrt_llm/models/whisper/model.py:36-111, :153-516).  The forward pass itself is not a
TensorRT graph: calling the model inside `net_guard` registers it with the Network, and the compute is the
hand-written HIP path behind the C-ABI (csrc/engine.hip), which follows model.py's dataflow.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

from . import _dtypes as trt
from .layers import Attention, ColumnLinear, Conv2d, Embedding, LayerNorm, RowLinear
from .module import Module, ModuleList
from .network import default_net


class Tensor:
    """Named engine input (functional.py:38): name, dtype, shape (-1 = dynamic), dim_range."""

    def __init__(self, name, dtype, shape, dim_range=None):
        self.name, self.dtype, self.shape, self.dim_range = name, dtype, tuple(shape), dim_range

    def __repr__(self):
        return f"Tensor({self.name!r}, {self.dtype}, {self.shape})"


class RaggedTensor:
    """(data, row_lengths) pair (functional.py:351).  `length` is threaded through and never consumed."""

    def __init__(self, data, row_lengths, max_row_length=None):
        self.data, self.row_lengths, self.max_row_length = data, row_lengths, max_row_length

    @staticmethod
    def from_row_lengths(data, row_lengths, max_row_length=None):
        return RaggedTensor(data, row_lengths, max_row_length)


def _register(model):
    net = default_net()
    if net is None:
        raise RuntimeError("the model must be called inside `with net_guard(network):` (build_encoder.py:99-102)")
    net._register_model(model)


class WhisperEncoderLayer(Module):
    def __init__(self, d_model=512, encoder_attention_heads=8, activation_function="gelu", encoder_ffn_dim=2048):
        super().__init__()
        self.embed_dim = d_model
        self.self_attn = Attention(d_model, encoder_attention_heads, 1)
        self.self_attn_layer_norm = LayerNorm(d_model)
        self.fc1 = ColumnLinear(d_model, encoder_ffn_dim)
        self.fc2 = ColumnLinear(encoder_ffn_dim, d_model)
        self.final_layer_norm = LayerNorm(d_model)


class WhisperEncoder(Module):
    """model.py:68-124.  Inputs `data` f32 [B,80,3000] (+ unused `length`), output `hidden_states` f32 [B,1500,d]."""

    def __init__(self, d_model=512, num_mel_bins=80, max_source_positions=1500, encoder_layers=6,
                 encoder_attention_heads=8, activation_function="gelu", encoder_ffn_dim=2048):
        super().__init__()
        if activation_function != "gelu":
            raise ValueError("Whisper uses activation_function='gelu'")
        self.d_model, self.num_mel_bins, self.max_source_positions = d_model, num_mel_bins, max_source_positions
        self.encoder_attention_heads, self.encoder_ffn_dim = encoder_attention_heads, encoder_ffn_dim
        self.conv1 = Conv2d(num_mel_bins, d_model, kernel_size=(1, 3), padding=(0, 1))
        self.conv2 = Conv2d(d_model, d_model, kernel_size=(1, 3), stride=(1, 2), padding=(0, 1))
        self.embed_positions_weight = np.zeros((1, max_source_positions, d_model), dtype=np.float32)  # plain ndarray
        self.layers = ModuleList([WhisperEncoderLayer(d_model, encoder_attention_heads, activation_function,
                                                      encoder_ffn_dim) for _ in range(encoder_layers)])
        self.layer_norm = LayerNorm(d_model)

    def prepare_inputs(self):
        frames = 2 * self.max_source_positions
        data = Tensor("data", trt.float32, [1, self.num_mel_bins, frames])
        length = Tensor("length", trt.float32, [1])
        return RaggedTensor.from_row_lengths(data, length)

    def forward(self, input_features: RaggedTensor):
        _register(self)
        return Tensor("hidden_states", trt.float32, [1, self.max_source_positions, self.d_model])


class WhisperDecoderAttention(Module):
    """model.py:153-304: separate q/k/v projections (k without bias) + dense."""

    def __init__(self, hidden_size=512, num_attention_heads=8):
        super().__init__()
        self.hidden_size, self.num_attention_heads = hidden_size, num_attention_heads
        self.attention_head_size = hidden_size // num_attention_heads
        self.norm_factor = math.sqrt(self.attention_head_size)
        self.q_proj = ColumnLinear(hidden_size, hidden_size, bias=True)
        self.k_proj = ColumnLinear(hidden_size, hidden_size, bias=False)
        self.v_proj = ColumnLinear(hidden_size, hidden_size, bias=True)
        self.dense = RowLinear(hidden_size, hidden_size, bias=True)


class WhisperDecoderLayer(Module):
    def __init__(self, d_model=512, decoder_attention_heads=8, activation_function="gelu", decoder_ffn_dim=2048):
        super().__init__()
        self.embed_dim = d_model
        self.self_attn = WhisperDecoderAttention(d_model, decoder_attention_heads)
        self.self_attn_layer_norm = LayerNorm(d_model)
        self.encoder_attn = WhisperDecoderAttention(d_model, decoder_attention_heads)
        self.encoder_attn_layer_norm = LayerNorm(d_model)
        self.fc1 = ColumnLinear(d_model, decoder_ffn_dim)
        self.fc2 = ColumnLinear(decoder_ffn_dim, d_model)
        self.final_layer_norm = LayerNorm(d_model)


class WhisperDecoder(Module):
    """model.py:371-516.  One token per call; caches by value on the Session surface (App. B of SURVEY.md)."""

    def __init__(self, pad_token_id=50256, max_target_positions=448, max_source_positions=1500, d_model=512,
                 scale_embedding=False, vocab_size=51864, decoder_layers=6, decoder_attention_heads=8,
                 activation_function="gelu", decoder_ffn_dim=2048):
        super().__init__()
        if activation_function != "gelu":
            raise ValueError("Whisper uses activation_function='gelu'")
        self.padding_idx = pad_token_id
        self.max_target_positions, self.max_source_positions = max_target_positions, max_source_positions
        self.d_model, self.vocab_size = d_model, vocab_size
        self.embed_scale = math.sqrt(d_model) if scale_embedding else 1.0  # computed, never applied (model.py:389)
        self.decoder_layers, self.decoder_attention_heads = decoder_layers, decoder_attention_heads
        self.decoder_ffn_dim = decoder_ffn_dim
        self.d_head = d_model // decoder_attention_heads
        self.embed_tokens = Embedding(vocab_size, d_model)
        self.embed_positions = Embedding(max_target_positions, d_model)
        self.layers = ModuleList([WhisperDecoderLayer(d_model, decoder_attention_heads, activation_function,
                                                      decoder_ffn_dim) for _ in range(decoder_layers)])
        self.layer_norm = LayerNorm(d_model)
        self.proj_out = ColumnLinear(d_model, vocab_size, bias=False)

    def prepare_inputs(self):
        L, H, S, T, dh = (self.decoder_layers, self.decoder_attention_heads, self.max_source_positions,
                          self.max_target_positions, self.d_head)
        data = Tensor("data", trt.int32, [1, 1], OrderedDict(batch_size=[1], id_len=[1]))
        length = Tensor("length", trt.int32, [1], OrderedDict(batch_size=[1]))
        enc = Tensor("encoder_hidden_states", trt.float32, [1, S, self.d_model])
        kv_range = OrderedDict(num_layers=[L], num_head=[H], kv_seq_len=[[1, 1, T + 1]], embed_per_head=[dh])
        spk = Tensor("self_past_key", trt.float32, [L, H, -1, dh], kv_range)
        spv = Tensor("self_past_value", trt.float32, [L, H, -1, dh], kv_range)
        cpk = Tensor("cross_past_key", trt.float32, [L, H, S, dh])
        cpv = Tensor("cross_past_value", trt.float32, [L, H, S, dh])
        msk_s = Tensor("past_self_cache_mask", trt.float32, [-1], OrderedDict(past_self_cache_length=[[1, 1, T + 1]]))
        msk_c = Tensor("past_cross_cache_mask", trt.float32, [-1], OrderedDict(past_cross_cache_length=[[1, 1, S + 1]]))
        return (RaggedTensor.from_row_lengths(data, length), enc, spk, spv, cpk, cpv, msk_s, msk_c)

    def forward(self, input_ids, encoder_hidden_states, past_self_keys, past_self_values, past_cross_keys,
                past_cross_values, past_self_cache_mask, past_cross_cache_mask):
        _register(self)
        return Tensor("hidden_states", trt.float32, [1, 1, self.vocab_size])
