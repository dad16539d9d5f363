"""Text-encoder step in front of the denoise path, CPU suite.

The graphs (csrc/host/text_encoders.hpp, restating src/model/te/clip.hpp and t5.hpp) run on the oracle backend and are compared
with an INDEPENDENT implementation: HuggingFace transformers' CLIPTextModel / CLIPTextModelWithProjection / T5EncoderModel (the
modelling code the reference itself cites, clip.hpp:10) instantiated at the same tiny width and loaded with the engine's weights.
Tolerance: rel-L2 <= 1e-3 with f32 Linear weights (contractions are exact f32 on both sides; what remains is ggml-cpu's f16 lookup
table for GELU / quick-GELU, which the oracle reproduces), <= 3e-3 with f16 weights (activations rounded to f16 at every contraction).  The conditioner composition
(csrc/host/conditioner.hpp, restating src/conditioning/conditioner.hpp:414-544, 842-1015, 1209-1297) is checked against a numpy
restatement built from the per-encoder outputs.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

L = 77
VOCAB = 1000
BOS, EOS = VOCAB - 2, VOCAB - 1
TOL32 = 1e-3


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def prompt_ids(rng, n_words, pad=EOS, length=L):
    ids = np.full(length, pad, dtype=np.int32)
    ids[0] = BOS
    ids[1:1 + n_words] = rng.integers(1, VOCAB - 2, n_words)
    ids[1 + n_words] = EOS
    return ids


def hf_clip(engine, prefix, hidden, heads, layers, act, proj=0):
    cfg = transformers.CLIPTextConfig(vocab_size=VOCAB, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=layers,
                                      num_attention_heads=heads, max_position_embeddings=L, hidden_act=act, eos_token_id=EOS, bos_token_id=BOS,
                                      pad_token_id=EOS, projection_dim=proj or hidden, layer_norm_eps=1e-5, attn_implementation="eager")
    m = (transformers.CLIPTextModelWithProjection(cfg) if proj else transformers.CLIPTextModel(cfg)).eval()
    sdict = {}
    for k in m.state_dict().keys():
        ours = k.replace("text_model.", "")
        if ours == "text_projection.weight":
            ours = "text_projection"
        sdict[k] = torch.from_numpy(engine.get_tensor(prefix + ours).copy()).reshape(m.state_dict()[k].shape)
    m.load_state_dict(sdict)
    return m


@pytest.fixture(scope="module")
def eng15(sd, oracle):
    return sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F32)


@pytest.fixture(scope="module")
def engxl(sd, oracle):
    return sd.Engine(model=sd.SDXL_TINY, backend=oracle, wtype=sd.F32)


@pytest.fixture(scope="module")
def eng35(sd, oracle):
    return sd.Engine(model=sd.SD35_TINY, backend=oracle, wtype=sd.F32)


def test_clip_l_vs_transformers(sd, eng15):
    eng15.text_encoders_init()
    m = hf_clip(eng15, "cond_stage_model.transformer.text_model.", 64, 4, 3, "quick_gelu")
    rng = np.random.default_rng(0)
    ids = prompt_ids(rng, 9)
    with torch.no_grad():
        o = m(torch.from_numpy(ids.astype(np.int64))[None], output_hidden_states=True)
        final_ln = m.text_model.final_layer_norm if hasattr(m, "text_model") else m.final_layer_norm
        penult = final_ln(o.hidden_states[-2])[0].numpy()
    # SD1.x: all layers + final LN (clip_skip 1, with_final_ln) — conditioner.hpp:425-427, clip.hpp:303-306
    assert rel_l2(eng15.clip_forward(0, ids, clip_skip=1), o.last_hidden_state[0].numpy()) < TOL32
    assert rel_l2(eng15.clip_forward(0, ids, clip_skip=-1), o.last_hidden_state[0].numpy()) < TOL32
    # clip_skip 2 stops one layer early; this tower still applies the final LN
    assert rel_l2(eng15.clip_forward(0, ids, clip_skip=2), penult) < TOL32
    # pooled = final-LN'd hidden state at the first EOS (no projection in ViT-L)
    pooled = eng15.clip_forward(0, ids, max_token_idx=10, return_pooled=True, clip_skip=2)
    assert pooled.shape == (64,)
    assert rel_l2(pooled, o.pooler_output[0].numpy()) < TOL32


def test_clip_bigg_vs_transformers(sd, engxl):
    engxl.text_encoders_init()
    # the reference uses ggml_gelu (tanh form) for the OpenCLIP towers (clip.hpp:21-25, 35-36)
    m = hf_clip(engxl, "cond_stage_model.1.transformer.text_model.", 40, 2, 3, "gelu_pytorch_tanh", proj=48)
    rng = np.random.default_rng(1)
    ids = prompt_ids(rng, 5, pad=0)
    with torch.no_grad():
        o = m(torch.from_numpy(ids.astype(np.int64))[None], output_hidden_states=True)
    # SDXL: penultimate layer, NO final LN (with_final_ln = false, conditioner.hpp:186-187)
    assert rel_l2(engxl.clip_forward(1, ids, clip_skip=2), o.hidden_states[-2][0].numpy()) < TOL32
    pooled = engxl.clip_forward(1, ids, max_token_idx=6, return_pooled=True, clip_skip=2)
    assert pooled.shape == (48,)
    assert rel_l2(pooled, o.text_embeds[0].numpy()) < TOL32
    # the ViT-L tower of SDXL: penultimate layer without the final LN
    ml = hf_clip(engxl, "cond_stage_model.transformer.text_model.", 24, 2, 3, "quick_gelu")
    ids_l = prompt_ids(rng, 5)
    with torch.no_grad():
        ol = ml(torch.from_numpy(ids_l.astype(np.int64))[None], output_hidden_states=True)
    assert rel_l2(engxl.clip_forward(0, ids_l, clip_skip=2), ol.hidden_states[-2][0].numpy()) < TOL32


def hf_t5(engine, prefix):
    cfg = transformers.T5Config(vocab_size=VOCAB, d_model=96, d_kv=16, d_ff=128, num_layers=2, num_heads=4, feed_forward_proj="gated-gelu",
                                relative_attention_num_buckets=32, relative_attention_max_distance=128, layer_norm_epsilon=1e-6, dropout_rate=0.0)
    m = transformers.T5EncoderModel(cfg).eval()
    sdict = {}
    for k in m.state_dict().keys():
        ours = "shared.weight" if k == "encoder.embed_tokens.weight" else k
        sdict[k] = torch.from_numpy(engine.get_tensor(prefix + ours).copy()).reshape(m.state_dict()[k].shape)
    m.load_state_dict(sdict)
    return m


def test_t5_encoder_vs_transformers(sd, eng35):
    eng35.text_encoders_init()
    m = hf_t5(eng35, "text_encoders.t5xxl.transformer.")
    rng = np.random.default_rng(2)
    for n in (L, 19):
        ids = rng.integers(0, VOCAB, n).astype(np.int32)
        with torch.no_grad():
            ref = m(torch.from_numpy(ids.astype(np.int64))[None]).last_hidden_state[0].numpy()
        out = eng35.t5_forward(ids)
        assert out.shape == (n, 96)
        assert rel_l2(out, ref) < TOL32


def test_t5_relative_position_buckets_vs_transformers(sd):
    from transformers.models.t5.modeling_t5 import T5Attention

    for q, k in ((77, 77), (256, 256), (5, 300)):
        rel = torch.arange(k)[None, :] - torch.arange(q)[:, None]
        ref = T5Attention._relative_position_bucket(rel, bidirectional=True, num_buckets=32, max_distance=128).numpy()
        np.testing.assert_array_equal(sd.t5_relative_position_buckets(q, k), ref)


@pytest.mark.parametrize("wtype,tol", [("F16", 3e-3), ("Q8_0", 3e-2)])
def test_text_encoders_low_precision_weights(sd, oracle, wtype, tol):
    """f16 / q8_0 Linear weights and embedding tables (GET_ROWS on a quantised table) against transformers on the dequantised weights."""
    e = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=getattr(sd, wtype))
    e.text_encoders_init()
    _, ty, _ = e.tensor_info("cond_stage_model.transformer.text_model.embeddings.token_embedding.weight")
    assert ty == getattr(sd, wtype)   # clip.hpp:132-142: the table keeps the file type when GET_ROWS can read it
    m = hf_clip(e, "cond_stage_model.transformer.text_model.", 64, 4, 3, "quick_gelu")
    ids = prompt_ids(np.random.default_rng(3), 12)
    with torch.no_grad():
        ref = m(torch.from_numpy(ids.astype(np.int64))[None]).last_hidden_state[0].numpy()
    assert rel_l2(e.clip_forward(0, ids), ref) < tol


def np_token_weights(h, w):
    """apply_token_weights, conditioner.hpp:90-125"""
    if np.all(w == 1.0):
        return h
    m0 = np.float32(h.astype(np.float64).mean())
    h = h * w[:, None].astype(np.float32)
    m1 = np.float32(h.astype(np.float64).mean())
    return h * np.float32(m0 / m1)


def np_ts_embedding(ts, dim):
    half = dim // 2
    f = np.exp(-np.log(10000) * np.arange(half) / half).astype(np.float32)
    arg = np.asarray(ts, np.float32)[:, None] * f[None]
    return np.concatenate([np.cos(arg), np.sin(arg)], axis=1).astype(np.float32).ravel()


def test_conditioner_sd1_chunks_and_weights(sd, eng15):
    rng = np.random.default_rng(4)
    a, b = prompt_ids(rng, 20), prompt_ids(rng, 3)
    ids = np.concatenate([a, b])
    w = np.ones(2 * L, np.float32)
    w[3:8] = 1.3
    w[L + 2] = 0.6
    c, y = eng15.get_learned_condition((ids, w))
    assert y is None and c.shape == (1, 2 * L, 64)
    ref = np.concatenate([np_token_weights(eng15.clip_forward(0, a, clip_skip=1), w[:L]), np_token_weights(eng15.clip_forward(0, b, clip_skip=1), w[L:])])
    assert rel_l2(c[0], ref) < 1e-6
    # unit weights leave the hidden states untouched
    c1, _ = eng15.get_learned_condition(a)
    np.testing.assert_array_equal(c1[0], eng15.clip_forward(0, a, clip_skip=1))
    # zero_out_masked (conditioner.hpp:501-503)
    cz, _ = eng15.get_learned_condition(a, zero_out_masked=True)
    assert not cz.any()


def test_conditioner_sdxl_layout(sd, engxl):
    rng = np.random.default_rng(5)
    ids = prompt_ids(rng, 7)
    c, y = engxl.get_learned_condition(ids, width=96, height=160)
    assert c.shape == (1, L, 64) and y.shape == (1, 96)
    ids_g = ids.copy()
    ids_g[9:] = 0          # conditioner.hpp:440-444: zeros after the first EOS for the bigG tower
    hl = engxl.clip_forward(0, ids, clip_skip=2)
    hg = engxl.clip_forward(1, ids_g, clip_skip=2)
    np.testing.assert_array_equal(c[0], np.concatenate([hl, hg], axis=1))
    pooled = engxl.clip_forward(1, ids_g, max_token_idx=8, return_pooled=True, clip_skip=2)
    ref_y = np.concatenate([pooled, np_ts_embedding([160, 96], 8), np_ts_embedding([0, 0], 8), np_ts_embedding([160, 96], 8)])
    np.testing.assert_allclose(y[0], ref_y, rtol=0, atol=1e-6)


def test_conditioner_sd3_layout(sd, eng35):
    rng = np.random.default_rng(6)
    il, ig = prompt_ids(rng, 6), prompt_ids(rng, 6, pad=0)
    it = rng.integers(0, VOCAB, L).astype(np.int32)
    wt = np.ones(L, np.float32)
    wt[:4] = 1.5
    c, y = eng35.get_learned_condition(il, ig, (it, wt))
    assert c.shape == (1, 2 * L, 96) and y.shape == (1, 64)
    hl = eng35.clip_forward(0, il, clip_skip=2)   # 24 wide
    hg = eng35.clip_forward(1, ig, clip_skip=2)   # 40 wide
    ht = np_token_weights(eng35.t5_forward(it), wt)
    lg = np.zeros((L, 96), np.float32)
    lg[:, :24], lg[:, 24:64] = hl, hg             # [clip_l | clip_g | zero pad to the T5 width]  (conditioner.hpp:983-991)
    assert rel_l2(c[0], np.concatenate([lg, ht])) < 1e-6
    pl = eng35.clip_forward(0, il, max_token_idx=7, return_pooled=True)
    pg = eng35.clip_forward(1, ig, max_token_idx=7, return_pooled=True)
    np.testing.assert_array_equal(y[0], np.concatenate([pl, pg]))


def test_conditioner_flux_layout(sd, oracle):
    e = sd.Engine(model=sd.FLUX_TINY, backend=oracle, wtype=sd.F32)
    rng = np.random.default_rng(7)
    il = prompt_ids(rng, 4)
    it = rng.integers(0, VOCAB, 64).astype(np.int32)   # two 32-token T5 chunks at the tiny width (256 per chunk in FLUX.1)
    c, y = e.get_learned_condition(il, None, it)
    assert c.shape == (1, 64, 96) and y.shape == (1, 64)
    np.testing.assert_array_equal(c[0], np.concatenate([e.t5_forward(it[:32]), e.t5_forward(it[32:])]))
    np.testing.assert_array_equal(y[0], e.clip_forward(0, il, max_token_idx=5, return_pooled=True))


def test_text_encoder_errors(sd, eng15):
    with pytest.raises(sd.EngineError, match="multiple"):
        eng15.get_learned_condition(np.zeros(50, np.int32))
    with pytest.raises(sd.EngineError, match="multiple of 77"):
        eng15.clip_forward(0, np.zeros(50, np.int32))
    with pytest.raises(sd.EngineError, match="out of range"):
        eng15.clip_forward(0, np.full(L, VOCAB, np.int32))
    with pytest.raises(sd.EngineError, match="no such CLIP tower"):
        eng15.clip_forward(1, np.zeros(L, np.int32))
    with pytest.raises(sd.EngineError, match="no T5"):
        eng15.t5_forward(np.zeros(8, np.int32))


def test_text_encoder_weights_load_from_safetensors(sd, oracle, tmp_path):
    """A checkpoint naming cond_stage_model.* tensors instantiates the encoders and overwrites their synthetic weights (f2 + f3)."""
    from safetensors.numpy import save_file

    src = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16, weight_seed=77)
    src.text_encoders_init()
    names = [n for n in src.tensor_names() if n.startswith("cond_stage_model.")]
    assert len(names) == 2 + 3 * 16 + 2
    # file dtype = the parameter's own type (f16 Linear weights / token table, f32 biases, norms and position table)
    tensors = {n: src.get_tensor(n).astype(np.float16 if src.tensor_info(n)[1] == sd.F16 else np.float32) for n in names}
    path = tmp_path / "clip_l.safetensors"
    save_file(tensors, str(path))
    dst = sd.Engine(model=sd.SD15_TINY, backend=oracle, wtype=sd.F16, weight_seed=1)
    before = len(dst.tensor_names())
    r = dst.load_weights(path)
    assert r["loaded"] == len(names) and len(dst.tensor_names()) == before + len(names)
    ids = prompt_ids(np.random.default_rng(8), 10)
    np.testing.assert_array_equal(dst.clip_forward(0, ids), src.clip_forward(0, ids))


def test_tokens_to_image_end_to_end(sd, oracle):
    """ids -> conditioner -> denoise -> VAE decode on one engine (tiny SDXL: both towers + the c_vector path)."""
    e = sd.Engine(model=sd.SDXL_TINY, backend=oracle)
    rng = np.random.default_rng(9)
    cond, cy = e.get_learned_condition(prompt_ids(rng, 8), width=64, height=64)
    empty = np.full(L, EOS, np.int32)
    empty[0] = BOS
    uncond, uy = e.get_learned_condition(empty, width=64, height=64)
    img = e.generate_image(cond, uncond, width=64, height=64, steps=2, cfg=5.0, seed=3, cond_y=cy, uncond_y=uy)
    assert img.shape == (1, 64, 64, 3) and img.dtype == np.uint8
