"""CPU suite: checkpoint name dialects -> the engine's canonical names (src/name_conversion.cpp).

Expected names are taken from the PUBLIC conversion tables the reference itself cites (diffusers' convert_diffusers_to_original_
stable_diffusion.py / ..._sdxl.py, OpenCLIP <-> HF CLIP layouts), written out by hand below or produced by an independent Python
inverse (LDM -> diffusers) in this file — never by the C++ under test.  The last tests write whole checkpoints in the foreign
dialects and require a bit-identical engine after loading."""
import re

import numpy as np
import pytest

MDM, FSM = "model.diffusion_model.", "first_stage_model."


@pytest.fixture(scope="module")
def e15(sd, oracle):
    return sd.Engine(model=sd.SD15_TINY, backend=oracle)   # same level / res-block / attention layout as SD1.5


@pytest.fixture(scope="module")
def exl(sd, oracle):
    return sd.Engine(model=sd.SDXL_TINY, backend=oracle)


SD1_CASES = [
    ("unet.conv_in.weight", MDM + "input_blocks.0.0.weight"),
    ("unet.time_embedding.linear_2.bias", MDM + "time_embed.2.bias"),
    ("unet.down_blocks.0.resnets.1.norm1.weight", MDM + "input_blocks.2.0.in_layers.0.weight"),
    ("unet.down_blocks.1.resnets.0.conv_shortcut.weight", MDM + "input_blocks.4.0.skip_connection.weight"),
    ("unet.down_blocks.2.attentions.0.transformer_blocks.0.attn1.to_q.weight", MDM + "input_blocks.7.1.transformer_blocks.0.attn1.to_q.weight"),
    ("unet.down_blocks.2.attentions.1.transformer_blocks.0.attn2.to_out.0.bias", MDM + "input_blocks.8.1.transformer_blocks.0.attn2.to_out.0.bias"),
    ("unet.down_blocks.0.attentions.1.transformer_blocks.0.attn1.to_out.weight", MDM + "input_blocks.2.1.transformer_blocks.0.attn1.to_out.0.weight"),
    ("unet.down_blocks.1.downsamplers.0.conv.bias", MDM + "input_blocks.6.0.op.bias"),
    ("unet.down_blocks.3.resnets.1.time_emb_proj.weight", MDM + "input_blocks.11.0.emb_layers.1.weight"),
    ("unet.mid_block.resnets.1.conv2.bias", MDM + "middle_block.2.out_layers.3.bias"),
    ("unet.mid_block.attentions.0.proj_in.weight", MDM + "middle_block.1.proj_in.weight"),
    ("unet.up_blocks.0.resnets.2.norm2.bias", MDM + "output_blocks.2.0.out_layers.0.bias"),
    ("unet.up_blocks.0.upsamplers.0.conv.weight", MDM + "output_blocks.2.1.conv.weight"),       # deepest SD1.x level has no attention
    ("unet.up_blocks.1.upsamplers.0.conv.weight", MDM + "output_blocks.5.2.conv.weight"),
    ("unet.up_blocks.3.attentions.2.norm.weight", MDM + "output_blocks.11.1.norm.weight"),
    ("unet.conv_norm_out.bias", MDM + "out.0.bias"),
    ("unet.conv_out.weight", MDM + "out.2.weight"),
    ("diffusion_model.input_blocks.1.0.in_layers.0.weight", MDM + "input_blocks.1.0.in_layers.0.weight"),
    (MDM + "out.2.bias", MDM + "out.2.bias"),
    ("vae.decoder.conv_in.weight", FSM + "decoder.conv_in.weight"),
    ("vae.decoder.conv_norm_out.weight", FSM + "decoder.norm_out.weight"),
    ("vae.decoder.up_blocks.0.resnets.2.conv1.weight", FSM + "decoder.up.3.block.2.conv1.weight"),
    ("vae.decoder.up_blocks.2.resnets.0.conv_shortcut.weight", FSM + "decoder.up.1.block.0.nin_shortcut.weight"),
    ("vae.decoder.up_blocks.1.upsamplers.0.conv.bias", FSM + "decoder.up.2.upsample.conv.bias"),
    ("vae.decoder.mid_block.resnets.1.norm2.weight", FSM + "decoder.mid.block_2.norm2.weight"),
    ("vae.decoder.mid_block.attentions.0.group_norm.weight", FSM + "decoder.mid.attn_1.norm.weight"),
    ("vae.decoder.mid_block.attentions.0.to_q.weight", FSM + "decoder.mid.attn_1.q.weight"),
    ("vae.decoder.mid_block.attentions.0.to_out.0.bias", FSM + "decoder.mid.attn_1.proj_out.bias"),
    ("vae.decoder.mid_block.attentions.0.proj_attn.weight", FSM + "decoder.mid.attn_1.proj_out.weight"),
    ("vae.decoder.mid_block.attentions.0.value.bias", FSM + "decoder.mid.attn_1.v.bias"),
    ("vae.encoder.down_blocks.1.downsamplers.0.conv.weight", FSM + "encoder.down.1.downsample.conv.weight"),
    ("vae.post_quant_conv.weight", FSM + "post_quant_conv.weight"),
    ("text_encoder.text_model.final_layer_norm.weight", "cond_stage_model.transformer.text_model.final_layer_norm.weight"),
    ("te.text_model.encoder.layers.3.mlp.fc1.bias", "cond_stage_model.transformer.text_model.encoder.layers.3.mlp.fc1.bias"),
    ("cond_stage_model.transformer.text_model.embeddings.token_embedding.weight", "cond_stage_model.transformer.text_model.embeddings.token_embedding.weight"),
    ("cond_stage_model.model.transformer.resblocks.5.mlp.c_fc.weight", "cond_stage_model.transformer.text_model.encoder.layers.5.mlp.fc1.weight"),
    ("cond_stage_model.model.ln_final.bias", "cond_stage_model.transformer.text_model.final_layer_norm.bias"),
    ("cond_stage_model.model.positional_embedding", "cond_stage_model.transformer.text_model.embeddings.position_embedding.weight"),
    ("some.unrelated.tensor", "some.unrelated.tensor"),
]

SDXL_CASES = [
    ("unet.add_embedding.linear_1.weight", MDM + "label_emb.0.0.weight"),
    ("unet.down_blocks.0.resnets.1.conv1.weight", MDM + "input_blocks.2.0.in_layers.2.weight"),
    ("unet.down_blocks.1.attentions.1.transformer_blocks.1.ff.net.0.proj.weight", MDM + "input_blocks.5.1.transformer_blocks.1.ff.net.0.proj.weight"),
    ("unet.down_blocks.1.downsamplers.0.conv.weight", MDM + "input_blocks.6.0.op.weight"),
    ("unet.up_blocks.0.upsamplers.0.conv.weight", MDM + "output_blocks.2.2.conv.weight"),       # deepest SDXL level carries attention
    ("unet.up_blocks.1.upsamplers.0.conv.bias", MDM + "output_blocks.5.2.conv.bias"),
    ("unet.up_blocks.2.resnets.2.conv2.weight", MDM + "output_blocks.8.0.out_layers.3.weight"),
    ("conditioner.embedders.0.transformer.text_model.embeddings.position_embedding.weight", "cond_stage_model.transformer.text_model.embeddings.position_embedding.weight"),
    ("conditioner.embedders.1.model.transformer.resblocks.31.attn.in_proj_weight", "cond_stage_model.1.transformer.text_model.encoder.layers.31.self_attn.in_proj.weight"),
    ("conditioner.embedders.1.model.transformer.resblocks.0.attn.out_proj.bias", "cond_stage_model.1.transformer.text_model.encoder.layers.0.self_attn.out_proj.bias"),
    ("conditioner.embedders.1.model.transformer.resblocks.7.ln_2.weight", "cond_stage_model.1.transformer.text_model.encoder.layers.7.layer_norm2.weight"),
    ("conditioner.embedders.1.model.transformer.resblocks.7.mlp.c_proj.bias", "cond_stage_model.1.transformer.text_model.encoder.layers.7.mlp.fc2.bias"),
    ("conditioner.embedders.1.model.token_embedding.weight", "cond_stage_model.1.transformer.text_model.embeddings.token_embedding.weight"),
    ("conditioner.embedders.1.model.text_projection", "cond_stage_model.1.transformer.text_model.text_projection"),
    ("text_encoder_2.text_model.encoder.layers.2.self_attn.k_proj.weight", "cond_stage_model.1.transformer.text_model.encoder.layers.2.self_attn.k_proj.weight"),
    ("text_encoder_2.text_projection.weight", "cond_stage_model.1.transformer.text_model.text_projection"),
    ("te2.text_model.final_layer_norm.bias", "cond_stage_model.1.transformer.text_model.final_layer_norm.bias"),
]

DIT_CASES = [
    ("model.diffusion_model.joint_blocks.0.x_block.attn.qkv.weight", "model.diffusion_model.joint_blocks.0.x_block.attn.qkv.weight"),
    ("clip_l.text_model.final_layer_norm.weight", "text_encoders.clip_l.transformer.text_model.final_layer_norm.weight"),
    ("text_encoders.clip_g.transformer.text_model.text_projection", "text_encoders.clip_g.transformer.text_model.text_projection"),
    ("text_encoders.t5xxl.transformer.encoder.block.3.layer.1.DenseReluDense.wi_0.weight", "text_encoders.t5xxl.transformer.encoder.block.3.layer.1.DenseReluDense.wi_0.weight"),
    ("text_encoders.t5xxl.transformer.enc.blk.3.ffn_gate.weight", "text_encoders.t5xxl.transformer.encoder.block.3.layer.1.DenseReluDense.wi_0.weight"),
    ("text_encoders.t5xxl.transformer.enc.blk.0.attn_rel_b.weight", "text_encoders.t5xxl.transformer.encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"),
    ("text_encoders.t5xxl.transformer.enc.output_norm.weight", "text_encoders.t5xxl.transformer.encoder.final_layer_norm.weight"),
    ("text_encoders.t5xxl.transformer.token_embd.weight", "text_encoders.t5xxl.transformer.shared.weight"),
    ("te3.encoder.block.0.layer.0.SelfAttention.q.weight", "text_encoders.t5xxl.transformer.encoder.block.0.layer.0.SelfAttention.q.weight"),
    ("vae.decoder.up_blocks.3.resnets.0.norm1.bias", FSM + "decoder.up.0.block.0.norm1.bias"),
]


# diffusers SD3Transformer2DModel names (public diffusers layout) -> original MMDiT names; parts of fused parameters carry ".<index>"
SD3_DIFFUSERS_CASES = [
    ("transformer.time_text_embed.timestep_embedder.linear_1.weight", MDM + "t_embedder.mlp.0.weight"),
    ("transformer.time_text_embed.timestep_embedder.linear_2.bias", MDM + "t_embedder.mlp.2.bias"),
    ("transformer.time_text_embed.text_embedder.linear_1.weight", MDM + "y_embedder.mlp.0.weight"),
    ("transformer.pos_embed.pos_embed", MDM + "pos_embed"),
    ("transformer.pos_embed.proj.weight", MDM + "x_embedder.proj.weight"),
    ("transformer.context_embedder.bias", MDM + "context_embedder.bias"),
    ("transformer.transformer_blocks.3.norm1.linear.weight", MDM + "joint_blocks.3.x_block.adaLN_modulation.1.weight"),
    ("transformer.transformer_blocks.3.norm1_context.linear.bias", MDM + "joint_blocks.3.context_block.adaLN_modulation.1.bias"),
    ("transformer.transformer_blocks.0.attn.to_q.weight", MDM + "joint_blocks.0.x_block.attn.qkv.weight"),
    ("transformer.transformer_blocks.0.attn.to_k.weight", MDM + "joint_blocks.0.x_block.attn.qkv.weight.1"),
    ("transformer.transformer_blocks.0.attn.to_v.bias", MDM + "joint_blocks.0.x_block.attn.qkv.bias.2"),
    ("transformer.transformer_blocks.11.attn.add_k_proj.weight", MDM + "joint_blocks.11.context_block.attn.qkv.weight.1"),
    ("transformer.transformer_blocks.2.attn2.to_v.weight", MDM + "joint_blocks.2.x_block.attn2.qkv.weight.2"),
    ("transformer.transformer_blocks.2.attn.norm_q.weight", MDM + "joint_blocks.2.x_block.attn.ln_q.weight"),
    ("transformer.transformer_blocks.2.attn.norm_added_k.weight", MDM + "joint_blocks.2.context_block.attn.ln_k.weight"),
    ("transformer.transformer_blocks.2.attn2.norm_k.weight", MDM + "joint_blocks.2.x_block.attn2.ln_k.weight"),
    ("transformer.transformer_blocks.5.ff.net.0.proj.weight", MDM + "joint_blocks.5.x_block.mlp.fc1.weight"),
    ("transformer.transformer_blocks.5.ff.net.2.bias", MDM + "joint_blocks.5.x_block.mlp.fc2.bias"),
    ("transformer.transformer_blocks.5.ff_context.net.0.proj.bias", MDM + "joint_blocks.5.context_block.mlp.fc1.bias"),
    ("transformer.transformer_blocks.5.attn.to_out.0.weight", MDM + "joint_blocks.5.x_block.attn.proj.weight"),
    ("transformer.transformer_blocks.5.attn.to_add_out.bias", MDM + "joint_blocks.5.context_block.attn.proj.bias"),
    ("transformer.transformer_blocks.5.attn2.to_out.0.bias", MDM + "joint_blocks.5.x_block.attn2.proj.bias"),
    ("transformer.proj_out.weight", MDM + "final_layer.linear.weight"),
    ("transformer.norm_out.linear.bias", MDM + "final_layer.adaLN_modulation.1.bias"),
    ("model.diffusion_model.transformer_blocks.1.attn.to_k.bias", MDM + "joint_blocks.1.x_block.attn.qkv.bias.1"),
    ("model.diffusion_model.x_embedder.proj.weight", MDM + "x_embedder.proj.weight"),   # already original
]

FLUX_DIFFUSERS_CASES = [
    ("transformer.time_text_embed.timestep_embedder.linear_1.weight", MDM + "time_in.in_layer.weight"),
    ("transformer.time_text_embed.text_embedder.linear_2.bias", MDM + "vector_in.out_layer.bias"),
    ("transformer.time_text_embed.guidance_embedder.linear_1.weight", MDM + "guidance_in.in_layer.weight"),
    ("transformer.context_embedder.weight", MDM + "txt_in.weight"),
    ("transformer.x_embedder.bias", MDM + "img_in.bias"),
    ("transformer.transformer_blocks.7.norm1.linear.weight", MDM + "double_blocks.7.img_mod.lin.weight"),
    ("transformer.transformer_blocks.7.norm1_context.linear.bias", MDM + "double_blocks.7.txt_mod.lin.bias"),
    ("transformer.transformer_blocks.7.attn.to_q.weight", MDM + "double_blocks.7.img_attn.qkv.weight"),
    ("transformer.transformer_blocks.7.attn.to_v.weight", MDM + "double_blocks.7.img_attn.qkv.weight.2"),
    ("transformer.transformer_blocks.7.attn.add_k_proj.bias", MDM + "double_blocks.7.txt_attn.qkv.bias.1"),
    ("transformer.transformer_blocks.7.attn.norm_q.weight", MDM + "double_blocks.7.img_attn.norm.query_norm.scale"),
    ("transformer.transformer_blocks.7.attn.norm_added_k.weight", MDM + "double_blocks.7.txt_attn.norm.key_norm.scale"),
    ("transformer.transformer_blocks.7.ff.net.0.proj.weight", MDM + "double_blocks.7.img_mlp.0.weight"),
    ("transformer.transformer_blocks.7.ff_context.net.2.bias", MDM + "double_blocks.7.txt_mlp.2.bias"),
    ("transformer.transformer_blocks.7.attn.to_out.0.weight", MDM + "double_blocks.7.img_attn.proj.weight"),
    ("transformer.transformer_blocks.7.attn.to_add_out.weight", MDM + "double_blocks.7.txt_attn.proj.weight"),
    ("transformer.single_transformer_blocks.30.norm.linear.weight", MDM + "single_blocks.30.modulation.lin.weight"),
    ("transformer.single_transformer_blocks.30.attn.to_q.weight", MDM + "single_blocks.30.linear1.weight"),
    ("transformer.single_transformer_blocks.30.attn.to_k.bias", MDM + "single_blocks.30.linear1.bias.1"),
    ("transformer.single_transformer_blocks.30.attn.to_v.weight", MDM + "single_blocks.30.linear1.weight.2"),
    ("transformer.single_transformer_blocks.30.proj_mlp.weight", MDM + "single_blocks.30.linear1.weight.3"),
    ("transformer.single_transformer_blocks.30.attn.norm_q.weight", MDM + "single_blocks.30.norm.query_norm.scale"),
    ("transformer.single_transformer_blocks.30.attn.norm_k.weight", MDM + "single_blocks.30.norm.key_norm.scale"),
    ("transformer.single_transformer_blocks.30.proj_out.bias", MDM + "single_blocks.30.linear2.bias"),
    ("transformer.proj_out.weight", MDM + "final_layer.linear.weight"),
    ("transformer.norm_out.linear.weight", MDM + "final_layer.adaLN_modulation.1.weight"),
    ("model.diffusion_model.double_blocks.0.img_attn.norm.query_norm.weight", MDM + "double_blocks.0.img_attn.norm.query_norm.scale"),
    ("model.diffusion_model.single_blocks.4.norm.key_norm.weight", MDM + "single_blocks.4.norm.key_norm.scale"),
    ("model.diffusion_model.double_blocks.0.img_attn.qkv.weight", MDM + "double_blocks.0.img_attn.qkv.weight"),   # already original
]


@pytest.mark.parametrize("raw,want", SD3_DIFFUSERS_CASES)
def test_sd3_diffusers_names(sd, oracle, raw, want):
    e = sd.Engine(model=sd.SD35_TINY, backend=oracle)
    assert e.convert_tensor_name(raw) == want


@pytest.mark.parametrize("raw,want", FLUX_DIFFUSERS_CASES)
def test_flux_diffusers_names(sd, oracle, raw, want):
    e = sd.Engine(model=sd.FLUX_TINY, backend=oracle)
    assert e.convert_tensor_name(raw) == want


# ---- independent inverse: original MMDiT / Flux names -> diffusers (the direction of diffusers' convert_sd3_to_diffusers / convert_flux_to_diffusers),
# splitting every fused parameter into the separate Linears diffusers stores
def mmdit_to_diffusers(name, arr):
    """-> list of (diffusers name, array)"""
    top = {"t_embedder.mlp.0": "time_text_embed.timestep_embedder.linear_1", "t_embedder.mlp.2": "time_text_embed.timestep_embedder.linear_2",
           "y_embedder.mlp.0": "time_text_embed.text_embedder.linear_1", "y_embedder.mlp.2": "time_text_embed.text_embedder.linear_2",
           "x_embedder.proj": "pos_embed.proj", "final_layer.linear": "proj_out", "final_layer.adaLN_modulation.1": "norm_out.linear",
           "context_embedder": "context_embedder"}
    if name == "pos_embed":
        return [("pos_embed.pos_embed", arr)]
    stem, leaf = name.rsplit(".", 1)
    if stem in top:
        return [(f"{top[stem]}.{leaf}", arr)]
    m = re.match(r"joint_blocks\.(\d+)\.(x_block|context_block)\.(.*)\.(weight|bias)$", name)
    assert m, name
    i, blk, member, leaf = m.groups()
    x = blk == "x_block"
    pre = f"transformer_blocks.{i}."
    if member == "adaLN_modulation.1":
        return [(pre + ("norm1" if x else "norm1_context") + f".linear.{leaf}", arr)]
    mm = re.match(r"(attn2?)\.(qkv|proj|ln_q|ln_k)$", member)
    if mm:
        a, what = mm.groups()
        if what == "qkv":
            names = ["to_q", "to_k", "to_v"] if x else ["add_q_proj", "add_k_proj", "add_v_proj"]
            return [(pre + f"{a}.{n}.{leaf}", part) for n, part in zip(names, np.split(arr, 3, axis=0))]
        if what == "proj":
            return [(pre + f"{a}." + ("to_out.0" if x else "to_add_out") + f".{leaf}", arr)]
        return [(pre + f"{a}.norm_" + ("" if x else "added_") + what[-1] + f".{leaf}", arr)]
    ff = "ff" if x else "ff_context"
    if member == "mlp.fc1":
        return [(pre + f"{ff}.net.0.proj.{leaf}", arr)]
    assert member == "mlp.fc2", name
    return [(pre + f"{ff}.net.2.{leaf}", arr)]


def flux_to_diffusers(name, arr, hidden):
    top = {"time_in.in_layer": "time_text_embed.timestep_embedder.linear_1", "time_in.out_layer": "time_text_embed.timestep_embedder.linear_2",
           "vector_in.in_layer": "time_text_embed.text_embedder.linear_1", "vector_in.out_layer": "time_text_embed.text_embedder.linear_2",
           "guidance_in.in_layer": "time_text_embed.guidance_embedder.linear_1", "guidance_in.out_layer": "time_text_embed.guidance_embedder.linear_2",
           "txt_in": "context_embedder", "img_in": "x_embedder", "final_layer.linear": "proj_out", "final_layer.adaLN_modulation.1": "norm_out.linear"}
    stem, leaf = name.rsplit(".", 1)
    if stem in top:
        return [(f"{top[stem]}.{leaf}", arr)]
    m = re.match(r"double_blocks\.(\d+)\.(img|txt)_(mod\.lin|attn\.qkv|attn\.proj|attn\.norm\.query_norm|attn\.norm\.key_norm|mlp\.0|mlp\.2)\.(weight|bias|scale)$", name)
    if m:
        i, side, member, leaf = m.groups()
        img = side == "img"
        pre = f"transformer_blocks.{i}."
        if member == "mod.lin":
            return [(pre + ("norm1" if img else "norm1_context") + f".linear.{leaf}", arr)]
        if member == "attn.qkv":
            names = ["to_q", "to_k", "to_v"] if img else ["add_q_proj", "add_k_proj", "add_v_proj"]
            return [(pre + f"attn.{n}.{leaf}", part) for n, part in zip(names, np.split(arr, 3, axis=0))]
        if member == "attn.proj":
            return [(pre + "attn." + ("to_out.0" if img else "to_add_out") + f".{leaf}", arr)]
        if member.startswith("attn.norm"):
            return [(pre + "attn.norm_" + ("" if img else "added_") + ("q" if "query" in member else "k") + ".weight", arr)]
        ff = "ff" if img else "ff_context"
        return [(pre + (f"{ff}.net.0.proj" if member == "mlp.0" else f"{ff}.net.2") + f".{leaf}", arr)]
    m = re.match(r"single_blocks\.(\d+)\.(modulation\.lin|linear1|linear2|norm\.query_norm|norm\.key_norm)\.(weight|bias|scale)$", name)
    assert m, name
    i, member, leaf = m.groups()
    pre = f"single_transformer_blocks.{i}."
    if member == "modulation.lin":
        return [(pre + f"norm.linear.{leaf}", arr)]
    if member == "linear1":
        parts = np.split(arr, [hidden, 2 * hidden, 3 * hidden], axis=0)   # q, k, v, mlp (unequal sizes)
        return [(pre + f"{n}.{leaf}", part) for n, part in zip(["attn.to_q", "attn.to_k", "attn.to_v", "proj_mlp"], parts)]
    if member == "linear2":
        return [(pre + f"proj_out.{leaf}", arr)]
    return [(pre + "attn.norm_" + ("q" if "query" in member else "k") + ".weight", arr)]


@pytest.mark.parametrize("which", ["SD35_TINY", "FLUX_TINY"])
def test_diffusers_dit_checkpoint_loads_bit_identically(sd, oracle, tmp_path, which):
    """A diffusers-format transformer file (separate to_q / to_k / to_v [/ proj_mlp] Linears, 'transformer.' component prefix) loads into the
    fused original-dialect parameters: every tensor name converts, the parts are stacked in order, and the forward is bit-identical to the
    engine the file was written from.  f16 matrices and f32 vectors, as the engine stores them."""
    from safetensors.numpy import save_file

    model = getattr(sd, which)
    src = sd.Engine(model=model, backend=oracle, weight_seed=11)
    names = [n for n in src.tensor_names() if n.startswith(MDM)]
    hidden = None
    if which == "FLUX_TINY":
        hidden = src.get_tensor(MDM + "single_blocks.0.linear2.weight").shape[0]
    foreign, n_parts = {}, 0
    for name in names:
        arr = src.get_tensor(name)
        arr = arr.astype(np.float16 if src.tensor_info(name)[1] == sd.F16 else np.float32)
        pieces = mmdit_to_diffusers(name[len(MDM):], arr) if which == "SD35_TINY" else flux_to_diffusers(name[len(MDM):], arr, hidden)
        n_parts += len(pieces) - 1
        for dn, part in pieces:
            assert "transformer." + dn not in foreign
            foreign["transformer." + dn] = np.ascontiguousarray(part)
            assert src.convert_tensor_name("transformer." + dn).split(".")[0] == "model"
    assert n_parts > 0
    save_file(foreign, str(tmp_path / "dit.safetensors"))
    e = sd.Engine(model=model, backend=oracle, weight_seed=99)
    r = e.load_weights(tmp_path / "dit.safetensors")
    assert r["loaded"] == len(names) and r["unused"] == 0, r
    for name in names:
        np.testing.assert_array_equal(e.get_tensor(name), src.get_tensor(name), err_msg=name)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((1, 16, 8, 8)).astype(np.float32)
    t = np.array([0.6 if which == "FLUX_TINY" else 600.0], np.float32)
    ctx = rng.standard_normal((1, 12, 96)).astype(np.float32)
    y = rng.standard_normal((1, 64)).astype(np.float32)
    np.testing.assert_array_equal(e.unet_forward(x, t, ctx, y), src.unet_forward(x, t, ctx, y))


@pytest.mark.parametrize("raw,want", SD1_CASES)
def test_sd1_names(e15, raw, want):
    assert e15.convert_tensor_name(raw) == want


@pytest.mark.parametrize("raw,want", SDXL_CASES)
def test_sdxl_names(exl, raw, want):
    assert exl.convert_tensor_name(raw) == want


@pytest.mark.parametrize("raw,want", DIT_CASES)
def test_dit_names(sd, oracle, raw, want):
    e = sd.Engine(model=sd.SD35_TINY, backend=oracle)
    assert e.convert_tensor_name(raw) == want


# ---- independent inverse: LDM -> diffusers (the direction of diffusers' own convert_original_stable_diffusion_to_diffusers) ----------
RES_INV = {"in_layers.0": "norm1", "in_layers.2": "conv1", "out_layers.0": "norm2", "out_layers.3": "conv2", "emb_layers.1": "time_emb_proj",
           "skip_connection": "conv_shortcut"}


def _res_inv(rest):
    for k, v in RES_INV.items():
        if rest.startswith(k + "."):
            return v + rest[len(k):]
    raise AssertionError(rest)


def ldm_unet_to_diffusers(name, R=2):
    top = {"time_embed.0.": "time_embedding.linear_1.", "time_embed.2.": "time_embedding.linear_2.", "label_emb.0.0.": "add_embedding.linear_1.",
           "label_emb.0.2.": "add_embedding.linear_2.", "input_blocks.0.0.": "conv_in.", "out.0.": "conv_norm_out.", "out.2.": "conv_out."}
    for k, v in top.items():
        if name.startswith(k):
            return v + name[len(k):]
    m = re.match(r"(input_blocks|output_blocks)\.(\d+)\.(\d+)\.(.*)", name)
    if m:
        kind, n, sub, rest = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        if kind == "input_blocks":
            i, j = (n - 1) // (R + 1), (n - 1) % (R + 1)
            if j == R:
                assert rest.startswith("op.")
                return f"down_blocks.{i}.downsamplers.0.conv.{rest[3:]}"
            return f"down_blocks.{i}.resnets.{j}.{_res_inv(rest)}" if sub == 0 else f"down_blocks.{i}.attentions.{j}.{rest}"
        i, j = n // (R + 1), n % (R + 1)
        if sub == 0:
            return f"up_blocks.{i}.resnets.{j}.{_res_inv(rest)}"
        if rest.startswith("conv."):
            return f"up_blocks.{i}.upsamplers.0.{rest}"
        return f"up_blocks.{i}.attentions.{j}.{rest}"
    m = re.match(r"middle_block\.(\d)\.(.*)", name)
    assert m, name
    k, rest = int(m.group(1)), m.group(2)
    return f"mid_block.attentions.0.{rest}" if k == 1 else f"mid_block.resnets.{k // 2}.{_res_inv(rest)}"


def ldm_vae_to_diffusers(name, levels=4):
    m = re.match(r"(encoder|decoder)\.(.*)", name)
    if not m:
        return name
    side, rest = m.group(1), m.group(2)
    if rest.startswith("norm_out."):
        return f"{side}.conv_norm_out.{rest[9:]}"
    mm = re.match(r"mid\.block_(\d)\.(.*)", rest)
    if mm:
        return f"{side}.mid_block.resnets.{int(mm.group(1)) - 1}." + mm.group(2).replace("nin_shortcut", "conv_shortcut")
    mm = re.match(r"mid\.attn_1\.(\w+)\.(.*)", rest)
    if mm:
        member = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}[mm.group(1)]
        return f"{side}.mid_block.attentions.0.{member}.{mm.group(2)}"
    mm = re.match(r"(up|down)\.(\d)\.(block\.(\d)|upsample|downsample)\.(.*)", rest)
    if mm:
        up, lvl = mm.group(1) == "up", int(mm.group(2))
        blk = f"{side}.{'up' if up else 'down'}_blocks.{levels - 1 - lvl if up else lvl}"
        if mm.group(3).startswith("block"):
            return f"{blk}.resnets.{mm.group(4)}." + mm.group(5).replace("nin_shortcut", "conv_shortcut")
        return f"{blk}.{'upsamplers' if up else 'downsamplers'}.0.{mm.group(5)}"
    return name


@pytest.mark.parametrize("which", ["SD15_TINY", "SDXL_TINY"])
def test_every_unet_and_vae_name_round_trips(sd, oracle, which):
    e = sd.Engine(model=getattr(sd, which), backend=oracle)
    n_unet = n_vae = 0
    for name in e.tensor_names():
        if name.startswith(MDM):
            foreign = "unet." + ldm_unet_to_diffusers(name[len(MDM):])
            n_unet += 1
        elif name.startswith(FSM):
            foreign = "vae." + ldm_vae_to_diffusers(name[len(FSM):])
            n_vae += 1
        else:
            continue
        assert e.convert_tensor_name(foreign) == name, foreign
    assert n_unet > 300 and n_vae > 60


def test_foreign_dialect_checkpoint_loads_bit_identically(sd, oracle, tmp_path):
    """One checkpoint, three dialects: LDM names; diffusers names for UNet + VAE (VAE attention projections as 2-D Linear weights, the
    way diffusers stores them); OpenCLIP names with the fused in_proj for the text tower.  All must give the same engine."""
    from safetensors.numpy import save_file

    src = sd.Engine(model=sd.SD15_TINY, backend=oracle, weight_seed=5)
    src.text_encoders_init()
    native, foreign = {}, {}
    qkv = {}
    for name in src.tensor_names():
        arr = src.get_tensor(name)
        arr = arr.astype(np.float16 if src.tensor_info(name)[1] == sd.F16 else np.float32)
        native[name] = arr
        if name.startswith(MDM):
            foreign["unet." + ldm_unet_to_diffusers(name[len(MDM):])] = arr
        elif name.startswith(FSM):
            f = "vae." + ldm_vae_to_diffusers(name[len(FSM):])
            if ".attentions.0.to_" in f and f.endswith("weight"):
                arr = arr.reshape(arr.shape[0], arr.shape[1])          # conv 1x1 [O, I, 1, 1] -> Linear [O, I]
            foreign[f] = arr
        else:  # cond_stage_model.transformer.text_model.* -> OpenCLIP layout under cond_stage_model.model.*
            t = name[len("cond_stage_model.transformer.text_model."):]
            m = re.match(r"encoder\.layers\.(\d+)\.(.*)", t)
            if m:
                i, rest = m.group(1), m.group(2)
                mm = re.match(r"self_attn\.([qkv])_proj\.(weight|bias)", rest)
                if mm:
                    qkv.setdefault((i, mm.group(2)), {})[mm.group(1)] = arr
                    continue
                rest = (rest.replace("self_attn.out_proj", "attn.out_proj").replace("layer_norm1", "ln_1").replace("layer_norm2", "ln_2")
                        .replace("mlp.fc1", "mlp.c_fc").replace("mlp.fc2", "mlp.c_proj"))
                foreign[f"cond_stage_model.model.transformer.resblocks.{i}.{rest}"] = arr
            else:
                t = {"embeddings.token_embedding.weight": "token_embedding.weight", "embeddings.position_embedding.weight": "positional_embedding",
                     "final_layer_norm.weight": "ln_final.weight", "final_layer_norm.bias": "ln_final.bias"}[t]
                foreign["cond_stage_model.model." + t] = arr
    for (i, leaf), parts in qkv.items():
        foreign[f"cond_stage_model.model.transformer.resblocks.{i}.attn.in_proj_{leaf}"] = np.concatenate([parts["q"], parts["k"], parts["v"]], axis=0)
    save_file(native, str(tmp_path / "native.safetensors"))
    save_file(foreign, str(tmp_path / "foreign.safetensors"))
    assert not set(native) & set(foreign)   # every name of the foreign file really is in another dialect

    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 4, 16, 16)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    t = np.array([321.0], np.float32)
    ids = np.full(77, 999, np.int32)
    ids[:6] = [998, 4, 8, 15, 16, 23]
    outs = []
    for fname in ("native.safetensors", "foreign.safetensors"):
        e = sd.Engine(model=sd.SD15_TINY, backend=oracle, weight_seed=99)
        r = e.load_weights(tmp_path / fname)
        assert r["loaded"] == len(native) and r["missing"] == 0 and r["unused"] == 0, (fname, r)
        outs.append((e.unet_forward(x, t, ctx), e.vae_decode(x), e.clip_forward(0, ids)))
    ref = (src.unet_forward(x, t, ctx), src.vae_decode(x), src.clip_forward(0, ids))
    for o in outs:
        for a, b in zip(o, ref):
            np.testing.assert_array_equal(a, b)


def test_component_file_with_prefix(sd, oracle, tmp_path):
    """diffusers layout: one un-prefixed file per sub-model (unet/diffusion_pytorch_model.safetensors) loaded with prefix='unet.'"""
    from safetensors.numpy import save_file

    src = sd.Engine(model=sd.SDXL_TINY, backend=oracle, weight_seed=6)
    unet = {}
    for name in src.tensor_names():
        if name.startswith(MDM):
            arr = src.get_tensor(name)
            unet[ldm_unet_to_diffusers(name[len(MDM):])] = arr.astype(np.float16 if src.tensor_info(name)[1] == sd.F16 else np.float32)
    save_file(unet, str(tmp_path / "diffusion_pytorch_model.safetensors"))
    e = sd.Engine(model=sd.SDXL_TINY, backend=oracle, weight_seed=7)
    r = e.load_weights(tmp_path / "diffusion_pytorch_model.safetensors", prefix="unet.")
    assert r["loaded"] == len(unet) and r["unused"] == 0
    # without the prefix nothing matches (bare diffusers names are not guessed)
    r0 = sd.Engine(model=sd.SDXL_TINY, backend=oracle, weight_seed=7).load_weights(tmp_path / "diffusion_pytorch_model.safetensors")
    assert r0["loaded"] == 0
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1, 4, 16, 16)).astype(np.float32)
    ctx = rng.standard_normal((1, 77, 64)).astype(np.float32)
    y = rng.standard_normal((1, 96)).astype(np.float32)
    t = np.array([111.0], np.float32)
    np.testing.assert_array_equal(e.unet_forward(x, t, ctx, y), src.unet_forward(x, t, ctx, y))


# ---- PINNED against the reference's own code (round 5) ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fam,which", [("sd1", "SD15_TINY"), ("sdxl", "SDXL_TINY"), ("sd3", "SD35_TINY"), ("flux", "FLUX_TINY")])
def test_names_against_the_reference_own_convert_tensor_name(sd, oracle, fam, which):
    """the engine's name rules against the REFERENCE's own convert_tensor_name (src/name_conversion.cpp:1346), compiled from where it
    lies into oracle/_ref/libref_names.so (oracle/Makefile) — every raw name of the expectation lists above, and every parameter of the four tiny models in the
    diffusers dialect and in the original one: 3647 names, committed as tests/golden/name_conversion_ref.json (tests/golden/make_names_golden.py), string-equal.
    One documented extension: a single-file checkpoint using the diffusers COMPONENT prefix `text_encoder_2.` — the reference only meets those tensors through
    its directory loader, which prefixes `te.1.` (src/model_loader.cpp:430-450; that spelling is in the table and agrees); it leaves the bare prefix alone."""
    import json
    from pathlib import Path
    here = Path(__file__).resolve().parent
    table = json.loads((here / "golden" / "name_conversion_ref.json").read_text())[fam]
    e = sd.Engine(model=getattr(sd, which), backend=oracle)
    n_conv = 0
    for raw, want in table.items():
        got = e.convert_tensor_name(raw)
        if raw.startswith("text_encoder_2.") and want == raw:
            assert got.startswith("cond_stage_model.1.transformer."), raw   # the extension
            continue
        assert got == want, raw
        n_conv += want != raw
    assert len(table) > 200 and n_conv > 80
    so = here.parent / "oracle" / "_ref" / "libref_names.so"
    if so.exists():   # live, on names outside the table
        import ctypes as C
        R = C.CDLL(str(so))
        R.ref_convert_tensor_name.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int]
        famid = {"sd1": 0, "sdxl": 1, "sd3": 2, "flux": 3}[fam]
        extra = {"sd1": ["unet.up_blocks.3.resnets.2.conv_shortcut.weight", "vae.decoder.up_blocks.2.upsamplers.0.conv.bias", "unet.mid_block.attentions.0.proj_out.weight",
                         "first_stage_model.decoder.mid.attn_1.to_q.weight", "vae.decoder.mid_block.attentions.0.to_out.0.bias"],
                 "sdxl": ["unet.add_embedding.linear_2.weight", "unet.down_blocks.2.attentions.1.transformer_blocks.9.ff.net.0.proj.weight", "unet.time_embedding.linear_1.bias"],
                 "sd3": ["transformer.transformer_blocks.37.attn.add_q_proj.bias", "transformer.norm_out.linear.weight", "transformer.pos_embed.proj.weight",
                         "transformer.transformer_blocks.5.norm1_context.linear.bias", "transformer.context_embedder.weight"],
                 "flux": ["transformer.single_transformer_blocks.37.proj_mlp.weight", "transformer.transformer_blocks.18.ff_context.net.2.bias", "transformer.x_embedder.weight",
                          "transformer.time_text_embed.guidance_embedder.linear_1.weight", "transformer.single_transformer_blocks.0.norm.linear.bias"]}[fam]
        for raw in extra:
            b = C.create_string_buffer(1024)
            assert R.ref_convert_tensor_name(raw.encode(), famid, b, 1024) >= 0
            assert e.convert_tensor_name(raw) == b.value.decode(), raw
