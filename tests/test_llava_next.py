"""LLaVA-NeXT caption pass: rsvld_amd.llava_next against vectors the REFERENCE's vendored model
produced on a tiny seeded LLaMA + CLIP configuration (tests/golden/gen_llava_golden.py): prompt string, token ids with the
image placeholder, anyres pixel views, spliced input embeddings, next-token logits and the generated ids (seeded sampling
at temperature 0.2 and greedy).  CPU fp32; everything offline."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import llava_common as C


@pytest.fixture(scope="module")
def setup(golden_dir, tmp_path_factory):
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    from rsvld_amd import llava_next as LN
    z = np.load(os.path.join(golden_dir, "llava_next.npz"))
    tower_dir = C.save_tiny_clip(str(tmp_path_factory.mktemp("clip")))
    cfg = LN._llama_config_cls()(**C.LLAMA, **C.MM, mm_vision_tower=tower_dir)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    model = LN.build_model(cfg, clip=CLIPVisionModel(CLIPVisionConfig(**C.VISION))).eval()
    C.name_seeded_state(model, C.WEIGHT_SEED)
    return LN, model, C.build_tokenizer(), CLIPImageProcessor.from_pretrained(tower_dir), z


def test_parameter_names_are_the_reference_checkpoint_names(setup):
    LN, model, _, _, z = setup
    mine = sorted(k for k, _ in model.named_parameters())
    assert mine == [str(s) for s in z["param_names"]]
    # a transformers-4 checkpoint spells the CLIP weights with an extra "vision_model." level: renamed on load
    sd = {k.replace(".vision_tower.vision_tower.", ".vision_tower.vision_tower.vision_model."): v for k, v in model.state_dict().items()}
    assert sorted(LN.normalise_checkpoint_keys(sd, model)) == sorted(model.state_dict())


def test_prompt_and_placeholder_tokens(setup):
    LN, _, tok, _, z = setup
    prompt = LN.llama3_prompt(tok, C.QUESTION)
    assert prompt == str(z["prompt"])
    ids = LN.tokenizer_image_token(prompt, tok, LN.IMAGE_TOKEN_INDEX, return_tensors="pt")
    assert ids.tolist() == z["input_ids"][0].tolist() and (ids == LN.IMAGE_TOKEN_INDEX).sum() == 1


@pytest.mark.parametrize("n", [0, 1, 2])
def test_embeddings_logits_and_generation_vs_reference(setup, n):
    LN, model, tok, proc, z = setup
    img = C.test_image(C.IMAGE_SIZES[n], 5 + n)
    px = LN.process_images([img], proc, model.config)
    assert np.array_equal(px[0].numpy(), z[f"i{n}.pixels"])                      # same tiles, bit for bit
    ids = torch.tensor(z["input_ids"])
    images = [x for x in px]
    with torch.no_grad():
        emb = model.multimodal_embeds(ids, images, [img.size])
        assert emb.shape == z[f"i{n}.embeds"].shape
        assert float((emb - torch.tensor(z[f"i{n}.embeds"])).abs().max()) < 1e-5
        logits = model(inputs_embeds=emb).logits[0, -1]
        assert float((logits - torch.tensor(z[f"i{n}.logits"])).abs().max()) < 1e-4
        kw = dict(images=images, image_sizes=[img.size], num_beams=1, max_new_tokens=16, return_dict_in_generate=True, output_scores=True)
        greedy = model.generate(ids, do_sample=False, **kw)[0][0]
        torch.manual_seed(3)
        sampled = model.generate(ids, do_sample=True, temperature=0.2, **kw)[0][0]
    assert greedy.tolist() == z[f"i{n}.greedy"].tolist()
    assert sampled.tolist() == z[f"i{n}.sampled"].tolist()


def test_get_img_describe_is_seeded_and_wired(setup):
    LN, model, tok, proc, z = setup
    img = C.test_image(C.IMAGE_SIZES[0], 5)
    images = [x for x in LN.process_images([img], proc, model.config)]
    a = LN.get_img_describe(images, img, model, tok, C.QUESTION, max_new_tokens=16, device="cpu", seed=3)
    b = LN.get_img_describe(images, img, model, tok, C.QUESTION, max_new_tokens=16, device="cpu", seed=3)
    assert a == b and isinstance(a, list) and len(a) == 1
    assert a[0] == tok.decode(z["i0.sampled"].tolist(), skip_special_tokens=True).lstrip()


def test_merge_lora_equals_explicit_adapter(setup, tmp_path):
    """W x + (alpha / r) B A x == (W + (alpha / r) B A) x on a Linear of the model (peft is not installed: parity unpinned
    against peft itself, pinned against the LoRA definition)."""
    import copy
    import json
    import safetensors.torch
    LN, model, _, _, _ = setup
    m2 = copy.deepcopy(model)
    name = "model.layers.0.self_attn.q_proj"
    lin = dict(m2.named_modules())[name]
    g = torch.Generator().manual_seed(9)
    A, B = torch.randn(4, lin.in_features, generator=g) * 0.1, torch.randn(lin.out_features, 4, generator=g) * 0.1
    json.dump({"peft_type": "LORA", "r": 4, "lora_alpha": 8, "target_modules": ["q_proj"]}, open(tmp_path / "adapter_config.json", "w"))
    safetensors.torch.save_file({f"base_model.model.{name}.lora_A.weight": A, f"base_model.model.{name}.lora_B.weight": B},
                                str(tmp_path / "adapter_model.safetensors"))
    x = torch.randn(3, lin.in_features, generator=g)
    want = lin(x) + (x @ A.T @ B.T) * 2.0
    assert LN.merge_lora(m2, str(tmp_path)) == 1
    assert torch.allclose(lin(x), want, atol=1e-5)


def test_load_llava_fails_loudly_without_a_checkpoint(setup, tmp_path):
    LN = setup[0]
    with pytest.raises(FileNotFoundError, match="--no_llava"):      # not a directory, not in the local hub cache
        LN.load_llava(device="cpu", model_path=str(tmp_path / "nothing-here"))


def _save_checkpoint(model, tok, cfg, path, drop=()):
    import safetensors.torch
    os.makedirs(path, exist_ok=True)
    cfg.save_pretrained(path)
    tok.save_pretrained(path)
    sd = {k: v.detach().clone().contiguous() for k, v in model.state_dict().items() if not any(d in k for d in drop)}
    keys = sorted(sd)
    safetensors.torch.save_file({k: sd[k] for k in keys[: len(keys) // 2]}, os.path.join(path, "model-00001-of-00002.safetensors"))
    safetensors.torch.save_file({k: sd[k] for k in keys[len(keys) // 2:]}, os.path.join(path, "model-00002-of-00002.safetensors"))
    return path


def test_load_llava_from_shards_reproduces_the_model_and_rejects_incomplete_checkpoints(setup, tmp_path):
    """load_llava builds the skeleton on the meta device and assigns every tensor from the shards: the loaded model must give
    the golden logits, and a checkpoint that lacks image_newline / the projector / a tower weight must raise instead of
    leaving that tensor at a random value."""
    LN, model, tok, proc, z = setup
    good = _save_checkpoint(model, tok, model.config, str(tmp_path / "good"))
    tok2, loaded, _ = LN.load_llava(device="cpu", model_path=good, adapter_path=None, dtype=torch.float32)
    img = C.test_image(C.IMAGE_SIZES[0], 5)
    px = LN.process_images([img], proc, loaded.config)
    with torch.no_grad():
        emb = loaded.multimodal_embeds(torch.tensor(z["input_ids"]), [x for x in px], [img.size])
        logits = loaded(inputs_embeds=emb).logits[0, -1]
    assert float((logits - torch.tensor(z["i0.logits"])).abs().max()) < 1e-4
    assert not any(p_.is_meta for p_ in loaded.parameters()) and not any(b.is_meta for b in loaded.buffers())
    for drop in ("image_newline", "mm_projector.2", "vision_tower.vision_tower"):
        bad = _save_checkpoint(model, tok, model.config, str(tmp_path / ("bad_" + drop.replace(".", "_"))), drop=(drop,))
        with pytest.raises(RuntimeError, match="in no shard"):
            LN.load_llava(device="cpu", model_path=bad, adapter_path=None, dtype=torch.float32)


def test_caption_sampling_leaves_the_callers_generator_alone(setup):
    """Stage 2 draws its noise right after the caption pass: the seeded sampling runs in a forked generator, so the
    caller's stream is where it was however many tokens were sampled; seed=None does not reseed anything."""
    LN, model, tok, proc, _ = setup
    img = C.test_image(C.IMAGE_SIZES[0], 5)
    images = [x for x in LN.process_images([img], proc, model.config)]
    torch.manual_seed(123)
    want = torch.randn(4)
    torch.manual_seed(123)
    LN.get_img_describe(images, img, model, tok, C.QUESTION, max_new_tokens=8, device="cpu", seed=3)
    assert torch.equal(torch.randn(4), want)
    torch.manual_seed(5)
    a = LN.get_img_describe(images, img, model, tok, C.QUESTION, max_new_tokens=8, device="cpu", seed=None)
    torch.manual_seed(5)
    b = LN.get_img_describe(images, img, model, tok, C.QUESTION, max_new_tokens=8, device="cpu", seed=None)
    assert a == b


@pytest.mark.parametrize("n", [0, 1, 2])
def test_fast_decoder_reproduces_the_reference_generations(setup, n):
    """FastDecoder (static cache, functional Llama forward over the model's own weights; eager on the CPU, hipGraph replay on a
    GPU) must produce the reference's tokens: greedy and seeded sampling at temperature 0.2, token for token."""
    LN, model, tok, proc, z = setup
    img = C.test_image(C.IMAGE_SIZES[n], 5 + n)
    images = [x for x in LN.process_images([img], proc, model.config)]
    ids = torch.tensor(z["input_ids"])
    with torch.no_grad():
        emb = model.multimodal_embeds(ids, images, [img.size])
        dec = LN.FastDecoder(model, emb.shape[1] + 16)
        logits = dec.forward(emb, torch.arange(emb.shape[1]))
        assert float((logits[0] - torch.tensor(z[f"i{n}.logits"])).abs().max()) < 1e-4
        greedy = dec.generate(emb, 16, do_sample=False, eos_token_id=LN._eos_ids(model, tok))
        torch.manual_seed(3)
        sampled = LN.caption_tokens_fast(model, ids, images, [img.size], 16, True, 0.2, LN._eos_ids(model, tok))
    assert greedy.tolist() == z[f"i{n}.greedy"].tolist()
    assert sampled.tolist() == z[f"i{n}.sampled"].tolist()


def test_fast_decoder_sampling_chain_equals_transformers_warpers():
    """FastDecoder._pick = generate()'s chain for the reference's call (models/util.py:50-60: do_sample, temperature 0.2, the generation
    config's top_k = 50 and top_p): the same masked scores as transformers' own warpers, hence the same draw for one generator state."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    from rsvld_amd.llava_next import FastDecoder
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(4, 1000, generator=g) * 3
    ids = torch.zeros(4, 1, dtype=torch.long)
    for temp, k, p in ((0.2, 50, 1.0), (0.7, 0, 0.9), (1.0, 5, 0.5), (0.2, 0, 1.0)):
        want = TemperatureLogitsWarper(temp)(ids, logits.clone())
        if k:
            want = TopKLogitsWarper(k)(ids, want)
        if p < 1.0:
            want = TopPLogitsWarper(p)(ids, want)
        torch.manual_seed(11)
        tok_w = torch.multinomial(torch.softmax(want, dim=-1), 1)[:, 0]
        torch.manual_seed(11)
        tok = FastDecoder._pick(None, logits, True, temp, k, p)
        assert torch.equal(tok, tok_w), (temp, k, p)
    assert torch.equal(FastDecoder._pick(None, logits, False, 1.0), logits.argmax(-1))
