"""Parameter holders named like the reference's layers (tensorrt_llm/layers/{linear,normalization,conv,
embedding,attention}.py).  Only shapes and names matter here: the arithmetic lives in csrc/*.hip."""
from __future__ import annotations

from .module import Module, Parameter


class Linear(Module):
    """weight [out, in], applied as x @ W^T (layers/linear.py:69-75)."""

    def __init__(self, in_features, out_features, bias=True, dtype=None, tp_group=None, tp_size=1, gather_output=True):
        super().__init__()
        assert tp_size == 1, "Whisper builds with tensor_parallel=1 (build_encoder.py:64)"
        self.in_features, self.out_features = in_features, out_features
        self.weight = Parameter(shape=(out_features, in_features))
        if bias:
            self.bias = Parameter(shape=(out_features,))
        else:
            self.register_parameter("bias", None)


ColumnLinear = Linear
RowLinear = Linear


class LayerNorm(Module):
    def __init__(self, normalized_shape, eps=1e-05, elementwise_affine=True, dtype=None):
        super().__init__()
        self.normalized_shape, self.eps = (normalized_shape,), eps
        self.weight = Parameter(shape=(normalized_shape,))
        self.bias = Parameter(shape=(normalized_shape,))


class Conv2d(Module):
    """The reference expresses Whisper's Conv1d as Conv2d with a (1,3) kernel (model.py:77-79)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=(1, 1), padding=(0, 0), bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = tuple(kernel_size), tuple(stride), tuple(padding)
        self.weight = Parameter(shape=(out_channels, in_channels, *self.kernel_size))
        if bias:
            self.bias = Parameter(shape=(out_channels,))
        else:
            self.register_parameter("bias", None)


class Embedding(Module):
    def __init__(self, num_embeddings, embedding_dim, dtype=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.weight = Parameter(shape=(num_embeddings, embedding_dim))


class Attention(Module):
    """Encoder self-attention with a fused qkv projection (layers/attention.py:128-152)."""

    def __init__(self, hidden_size, num_attention_heads, num_layers=1):
        super().__init__()
        self.hidden_size, self.num_attention_heads = hidden_size, num_attention_heads
        self.qkv = Linear(hidden_size, 3 * hidden_size)
        self.dense = Linear(hidden_size, hidden_size)
