"""Builder: parameter tree -> serialized engine bytes (tensorrt_llm/builder.py:53-267 surface).

`build_engine` does what `trt.Builder.build_serialized_network` does for the reference (builder.py:228):
it returns the bytes the scripts write to `engine_dir/Whisper{Encoder,Decoder}.engine`.  Our engine is the
weight pack of engine_pack.py with the layout transforms the HIP kernels want:
  * conv weights [d, c, 1, 3] -> [d, 3*c] with k-major columns (implicit-GEMM order, csrc/engine.hip)
  * decoder self-attn q|k|v stacked to [3d, d] (k bias = 0), cross-attn k|v stacked to [2d, d]
  * proj_out aliased to embed_tokens when the two arrays are equal (tied embedding, modeling_whisper.py:1335)
  * decoder cross-attention query folded through its LayerNorm and the self-attention out-projection
    (`_fold_cross_query`): one dependent launch fewer per decoder layer
"""
from __future__ import annotations

import numpy as np

from . import _dtypes
from . import engine_pack
from .logger import logger
from .models import WhisperDecoder, WhisperEncoder
from .network import Network


class BuilderConfig:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to_dict(self):
        return dict(self.__dict__)


class Builder:
    _ALLOWED_PRECISIONS = ["float32", "float16"]  # builder.py:55

    def create_network(self) -> Network:
        return Network()

    def create_builder_config(self, name="", precision="float32", timing_cache=None, tensor_parallel=1,
                              parallel_build=False, int8=False, opt_level=None, **kwargs) -> BuilderConfig:
        if precision not in self._ALLOWED_PRECISIONS:
            raise ValueError(f"precision should be one of {self._ALLOWED_PRECISIONS}")
        if tensor_parallel != 1 or int8:
            raise ValueError("the Whisper path builds with tensor_parallel=1 and int8=False (build_encoder.py:59-68)")
        return BuilderConfig(name=name, precision=precision, timing_cache=timing_cache, tensor_parallel=tensor_parallel,
                             parallel_build=parallel_build, int8=int8, opt_level=opt_level, **kwargs)

    def build_engine(self, network: Network, builder_config: BuilderConfig):
        """Returns the serialized engine (bytes) or None on failure, like builder.py:204-238."""
        model = network.model
        if model is None:
            logger.error("build_engine: no model was traced inside net_guard(network)")
            return None
        half = builder_config.precision == "float16"   # both engines: build_encoder.py:25,62 / build_decoder.py:25,62, builder.py:55
        params = dict(network.named_parameters()) or dict(model.named_parameters())
        f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)
        val = lambda name: f32(params[name].value)
        try:
            if isinstance(model, WhisperEncoder):
                blob = self._pack_encoder(model, val, f32, half)
            elif isinstance(model, WhisperDecoder):
                blob = self._pack_decoder(model, val, f32, half)
            else:
                logger.error(f"build_engine: unsupported model {type(model).__name__}")
                return None
        except (KeyError, ValueError) as exc:
            logger.error(f"build_engine failed: {exc}")
            return None
        return blob

    @staticmethod
    def _conv_as_gemm(w4: np.ndarray) -> np.ndarray:
        d, c, one, k = w4.shape
        assert one == 1 and k == 3
        return np.ascontiguousarray(w4[:, :, 0, :].transpose(0, 2, 1).reshape(d, 3 * c))  # [co][k*C + ci]

    def _pack_encoder(self, m: WhisperEncoder, val, f32, half: bool = False) -> bytes:
        t = {
            "conv1.weight": self._conv_as_gemm(val("conv1.weight")), "conv1.bias": val("conv1.bias"),
            "conv2.weight": self._conv_as_gemm(val("conv2.weight")), "conv2.bias": val("conv2.bias"),
            "embed_positions": f32(m.embed_positions_weight).reshape(m.max_source_positions, m.d_model),
        }
        for i in range(len(m.layers)):
            p = f"layers.{i}."
            for n in ("self_attn.qkv.weight", "self_attn.qkv.bias", "self_attn.dense.weight", "self_attn.dense.bias",
                      "self_attn_layer_norm.weight", "self_attn_layer_norm.bias", "fc1.weight", "fc1.bias",
                      "fc2.weight", "fc2.bias", "final_layer_norm.weight", "final_layer_norm.bias"):
                t[p + n] = val(p + n)
        t["layer_norm.weight"], t["layer_norm.bias"] = val("layer_norm.weight"), val("layer_norm.bias")
        cfg = dict(d_model=m.d_model, n_heads=m.encoder_attention_heads, n_layers=len(m.layers),
                   ffn_dim=m.encoder_ffn_dim, n_mels=m.num_mel_bins, max_source_positions=m.max_source_positions)
        if half:  # GEMM operands in fp16; biases, LayerNorm parameters and embed_positions stay fp32
            if m.num_mel_bins % 8 or m.d_model % 8:
                raise ValueError("float16 encoder needs num_mel_bins and d_model to be multiples of 8")
            for name in list(t):
                if name.endswith(".weight") and t[name].ndim == 2:
                    t[name] = t[name].astype(np.float16)
        prec = _dtypes.float16.code if half else _dtypes.float32.code
        return engine_pack.pack(engine_pack.KIND_ENCODER, prec, cfg, t)

    @staticmethod
    def _fold_cross_query(wq, bq, gamma, beta, wo, bo, scale=0.125):
        """The cross-attention query of a decoder layer (model.py:261-272, 283-294; HF modeling_whisper.py:472, 727-735) is
            q = s.(Wq.LN(h1) + bq),   h1 = h + Wo.a + bo   (a = self-attention context, s = head_dim^-0.5)
        With G = s.Wq.diag(gamma) and (mu, rstd) the LayerNorm statistics of h1 this is
            q = rstd.(G.h1 - mu.G.1) + s.(Wq.beta + bq),    G.h1 = [G.Wo | G].[a ; h] + G.bo
        so the engine computes u = W_fold.[a ; h] + c in the SAME launch as h1 (both only need a and h) and the attention
        kernel finishes q = (u - mu.r).rstd + t from the statistics of h1.  Products are formed in float64.
        Returns (W_fold [d, 2d], c [d], r [d], t [d]) as float32."""
        wq64, wo64 = wq.astype(np.float64), wo.astype(np.float64)
        g = scale * wq64 * gamma.astype(np.float64)[None, :]
        w_fold = np.concatenate([g @ wo64, g], axis=1)
        c = g @ bo.astype(np.float64)
        r = g.sum(axis=1)
        t = scale * (wq64 @ beta.astype(np.float64) + bq.astype(np.float64))
        return tuple(np.ascontiguousarray(x, dtype=np.float32) for x in (w_fold, c, r, t))

    def _pack_decoder(self, m: WhisperDecoder, val, f32, half: bool = False) -> bytes:
        """`half` (--engine_precision float16, build_decoder.py:25,62): every GEMV weight matrix -- the self / cross projections, the FFN,
        the folded cross query and the token table tied to the vocabulary projection -- is stored as IEEE half; biases, LayerNorm
        parameters, embed_positions and the folded query's vectors stay fp32, and the kernels accumulate, normalise and soft-max in
        fp32 (the reference forces fp32 scores in fp16 builds too, model.py:292-295).  The fold is formed in float64 from the ROUNDED
        weights -- the fp16 model is the model whose weights are those fp16 values -- and rounded once more as a matrix."""
        d = m.d_model
        if half:
            if d % 8 or m.decoder_ffn_dim % 8:
                raise ValueError("float16 decoder needs d_model and decoder_ffn_dim to be multiples of 8")
            val32 = val
            r16 = lambda a: a.astype(np.float16).astype(np.float32)
            val = lambda name: r16(val32(name)) if name.endswith(".weight") and val32(name).ndim == 2 and not name.startswith("embed_positions") else val32(name)
        zeros = np.zeros((d,), np.float32)
        emb = val("embed_tokens.weight")
        proj = val("proj_out.weight")
        tied = proj is emb or np.array_equal(proj, emb)
        t = {"embed_tokens.weight": emb, "embed_positions.weight": val("embed_positions.weight")}
        if not tied:
            t["proj_out.weight"] = proj
        for i in range(len(m.layers)):
            p = f"layers.{i}."
            sa, ca = p + "self_attn.", p + "encoder_attn."
            t[sa + "qkv.weight"] = np.concatenate([val(sa + "q_proj.weight"), val(sa + "k_proj.weight"), val(sa + "v_proj.weight")], 0)
            t[sa + "qkv.bias"] = np.concatenate([val(sa + "q_proj.bias"), zeros, val(sa + "v_proj.bias")], 0)
            t[sa + "dense.weight"], t[sa + "dense.bias"] = val(sa + "dense.weight"), val(sa + "dense.bias")
            fold = self._fold_cross_query(val(ca + "q_proj.weight"), val(ca + "q_proj.bias"), val(p + "encoder_attn_layer_norm.weight"),
                                          val(p + "encoder_attn_layer_norm.bias"), val(sa + "dense.weight"), val(sa + "dense.bias"))
            for n, a in zip(("weight", "bias", "rowsum", "shift"), fold):
                t[ca + "q_fold." + n] = a
            t[ca + "kv.weight"] = np.concatenate([val(ca + "k_proj.weight"), val(ca + "v_proj.weight")], 0)
            t[ca + "kv.bias"] = np.concatenate([zeros, val(ca + "v_proj.bias")], 0)
            t[ca + "dense.weight"], t[ca + "dense.bias"] = val(ca + "dense.weight"), val(ca + "dense.bias")
            for n in ("self_attn_layer_norm", "final_layer_norm"):
                t[p + n + ".weight"], t[p + n + ".bias"] = val(p + n + ".weight"), val(p + n + ".bias")
            for n in ("fc1", "fc2"):
                t[p + n + ".weight"], t[p + n + ".bias"] = val(p + n + ".weight"), val(p + n + ".bias")
        t["layer_norm.weight"], t["layer_norm.bias"] = val("layer_norm.weight"), val("layer_norm.bias")
        cfg = dict(d_model=d, n_heads=m.decoder_attention_heads, n_layers=len(m.layers), ffn_dim=m.decoder_ffn_dim,
                   n_mels=80, max_source_positions=m.max_source_positions, max_target_positions=m.max_target_positions,
                   vocab_size=m.vocab_size, tied_proj_out=int(tied))
        if half:
            for name in list(t):
                if name.endswith(".weight") and t[name].ndim == 2 and name != "embed_positions.weight":
                    t[name] = t[name].astype(np.float16)
        return engine_pack.pack(engine_pack.KIND_DECODER, (_dtypes.float16 if half else _dtypes.float32).code, cfg, t)
