"""HF `state_dict` -> engine bytes: exactly the weight binding of the reference's build scripts
(build_encoder.py:48-109, build_decoder.py:45-119), factored into functions so tests, bench.py and the
example scripts share it.  `state_dict` maps HF keys to numpy arrays (or torch tensors)."""
from __future__ import annotations

import numpy as np

from .builder import Builder
from .models import WhisperDecoder, WhisperEncoder
from .network import net_guard


def _np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def build_encoder_engine(config: dict, ckpt: dict, precision: str = "float32") -> bytes:
    m = WhisperEncoder(d_model=config["d_model"], num_mel_bins=config["num_mel_bins"],
                       max_source_positions=config["max_source_positions"], encoder_layers=config["encoder_layers"],
                       encoder_attention_heads=config["encoder_attention_heads"],
                       activation_function=config["activation_function"], encoder_ffn_dim=config["encoder_ffn_dim"])
    builder = Builder()
    bcfg = builder.create_builder_config(name="WhisperEncoder", precision=precision, timing_cache="model.cache",
                                         tensor_parallel=1, parallel_build=False, int8=False, opt_level=None)
    g = lambda k: _np(ckpt[k])
    m.conv1.weight.value = g("model.encoder.conv1.weight")[:, :, None, :]
    m.conv1.bias.value = g("model.encoder.conv1.bias")
    m.conv2.weight.value = g("model.encoder.conv2.weight")[:, :, None, :]
    m.conv2.bias.value = g("model.encoder.conv2.bias")
    m.embed_positions_weight = g("model.encoder.embed_positions.weight")[None]
    for i in range(config["encoder_layers"]):
        p, l = f"model.encoder.layers.{i}.", m.layers[i]
        q_b = g(p + "self_attn.q_proj.bias")
        l.self_attn.qkv.weight.value = np.concatenate([g(p + "self_attn.q_proj.weight"), g(p + "self_attn.k_proj.weight"),
                                                       g(p + "self_attn.v_proj.weight")], 0)
        l.self_attn.qkv.bias.value = np.concatenate([q_b, np.zeros_like(q_b), g(p + "self_attn.v_proj.bias")], 0)
        l.self_attn.dense.weight.value = g(p + "self_attn.out_proj.weight")
        l.self_attn.dense.bias.value = g(p + "self_attn.out_proj.bias")
        for n in ("self_attn_layer_norm", "final_layer_norm", "fc1", "fc2"):
            getattr(l, n).weight.value = g(p + n + ".weight")
            getattr(l, n).bias.value = g(p + n + ".bias")
    m.layer_norm.weight.value = g("model.encoder.layer_norm.weight")
    m.layer_norm.bias.value = g("model.encoder.layer_norm.bias")
    network = builder.create_network()
    network.trt_network.name = "WhisperEncoder"
    network.plugin_config.set_identity_plugin(dtype=precision)
    with net_guard(network):
        network.set_named_parameters(m.named_parameters())
        m(m.prepare_inputs())
    engine = builder.build_engine(network, bcfg)
    assert engine is not None, "Failed to build engine"
    return engine


def build_decoder_engine(config: dict, ckpt: dict, precision: str = "float32") -> bytes:
    m = WhisperDecoder(pad_token_id=config["pad_token_id"], max_target_positions=config["max_target_positions"],
                       max_source_positions=config["max_source_positions"], d_model=config["d_model"],
                       scale_embedding=config["scale_embedding"], vocab_size=config["vocab_size"],
                       decoder_layers=config["decoder_layers"], decoder_attention_heads=config["decoder_attention_heads"],
                       activation_function=config["activation_function"], decoder_ffn_dim=config["decoder_ffn_dim"])
    builder = Builder()
    bcfg = builder.create_builder_config(name="WhisperDecoder", precision=precision, timing_cache="model.cache",
                                         tensor_parallel=1, parallel_build=False, int8=False, opt_level=None)
    g = lambda k: _np(ckpt[k])
    m.embed_tokens.weight.value = g("model.decoder.embed_tokens.weight")
    m.embed_positions.weight.value = g("model.decoder.embed_positions.weight")
    for i in range(config["decoder_layers"]):
        p, l = f"model.decoder.layers.{i}.", m.layers[i]
        for attn in ("self_attn", "encoder_attn"):
            a = getattr(l, attn)
            a.q_proj.weight.value = g(p + attn + ".q_proj.weight")
            a.q_proj.bias.value = g(p + attn + ".q_proj.bias")
            a.k_proj.weight.value = g(p + attn + ".k_proj.weight")
            a.v_proj.weight.value = g(p + attn + ".v_proj.weight")
            a.v_proj.bias.value = g(p + attn + ".v_proj.bias")
            a.dense.weight.value = g(p + attn + ".out_proj.weight")
            a.dense.bias.value = g(p + attn + ".out_proj.bias")
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm", "fc1", "fc2"):
            getattr(l, n).weight.value = g(p + n + ".weight")
            getattr(l, n).bias.value = g(p + n + ".bias")
    m.layer_norm.weight.value = g("model.decoder.layer_norm.weight")
    m.layer_norm.bias.value = g("model.decoder.layer_norm.bias")
    m.proj_out.weight.value = g("proj_out.weight")
    network = builder.create_network()
    network.trt_network.name = "WhisperDecoder"
    network.plugin_config.set_identity_plugin(dtype=precision)
    with net_guard(network):
        network.set_named_parameters(m.named_parameters())
        m(*m.prepare_inputs())
    engine = builder.build_engine(network, bcfg)
    assert engine is not None, "Failed to build engine"
    return engine
