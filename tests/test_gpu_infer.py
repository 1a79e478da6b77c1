"""End-to-end driver on the GPU: the reference's CLI flow (infer.py:206-215) — Stage 1 SR3 -> uint8 hand-off ->
Stage 2 refinement -> PNG files named like the reference's — on a tiny image with reduced network depth."""
import os
import sys

import numpy as np
import pytest
import torch
import yaml

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import s2_common as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("prec", ["default", "fp32", "split", "vae_split", "tolerance"])
def test_pipeline_cli_flow(cuda, tmp_path, prec):
    """``fp32``: PipelineConfig(ae_dtype / diff_dtype / sr3_dtype = "fp32") = the CLI's --fp32: both stages on the fp32-operand
    kernel family; ``split`` = --split (both stages through the split-operand product path), ``vae_split`` = --vae_split (the VAE
    passes only); ``tolerance`` = --tolerance, the composition bench.py times (Stage 1 fp16 x weight pairs, Stage 2 + VAE split);
    same files, same sizes."""
    from PIL import Image
    from rsvld_amd import infer
    cfg = yaml.safe_load(open(S.YAML.replace("juggernautXL.yaml", "juggernautXL_cached.yaml")))
    for k in ("control_stage_config", "network_config"):
        cfg["model"]["params"][k]["params"].update(S.SMALL)
    c, uc = S.cond_dicts()
    torch.save(c, tmp_path / "c.pth")
    torch.save(uc, tmp_path / "uc.pth")
    cfg["model"]["params"]["conditioner_config"]["params"] = {"cond_pth": str(tmp_path / "c.pth"), "un_cond_pth": str(tmp_path / "uc.pth")}
    cfg["SR_CKPT"] = cfg["SR_CKPT_Q"] = None
    ypath = tmp_path / "model.yaml"
    yaml.safe_dump(cfg, open(ypath, "w"))
    rng = np.random.default_rng(0)
    lr = Image.fromarray(rng.integers(0, 255, (24, 32, 3), dtype=np.uint8))
    lr.save(tmp_path / "tile.png")
    pc = infer.PipelineConfig(input_img=str(tmp_path / "tile.png"), output_dir=str(tmp_path / "out"), model_yaml=str(ypath),
                              allow_random_init=True, no_llava=True,
                              upscale_factor=2, min_size=128, edm_steps=3, sr3_steps=3, seed=1, img_threshold=0.3,
                              **(dict(ae_dtype=prec, diff_dtype=prec, sr3_dtype=prec) if prec in ("fp32", "split") else
                                 dict(ae_dtype="split", diff_dtype="split", sr3_dtype="w2") if prec == "tolerance" else
                                 dict(ae_dtype="split") if prec == "vae_split" else {}))
    pipe = infer.SuperResolutionPipeline(pc)
    want_dt = torch.float32 if prec in ("fp32", "split") else torch.float16
    unet1 = pipe.sr3_model.netG.denoise_fn
    assert unet1.compute_dtype == want_dt
    assert unet1.pack_dtype == (torch.float32 if prec == "tolerance" else want_dt)        # w2: fp16 tensors, weights packed as pairs from fp32
    assert pipe.refinement_model.model.dtype == (torch.float32 if prec == "tolerance" else want_dt)
    assert pipe.refinement_model.first_stage_model.compute_dtype == (torch.bfloat16 if prec == "default" else torch.float32)
    assert (unet1.split is not None) == (prec == "split")
    assert (pipe.refinement_model.model.split is not None) == (prec in ("split", "tolerance"))
    assert (pipe.refinement_model.first_stage_model.split is not None) == (prec in ("split", "vae_split", "tolerance"))
    if prec == "tolerance":     # what main(["--tolerance"]) builds is this composition
        from rsvld_amd import ops
        assert pipe.refinement_model.precision_key() == ("split", "split", ops.VAE_POLICY.key(), ops.UNET_POLICY.key())
    # zero-initialised output convs would make Stage 2 a no-op: give them small seeded weights
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p_ in pipe.refinement_model.parameters():
            if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
                p_.copy_((torch.randn(p_.shape, generator=g) * 0.02).to(p_.device))
    outs = pipe.process()
    sr3 = Image.open(tmp_path / "out" / "sr3_tile.png")
    assert sr3.size == (64, 64)                       # max(w,h)*scale, centre-cropped square (data/dataset.py:16-21)
    assert [os.path.basename(o) for o in outs] == ["tile_final_0.png"]
    final = Image.open(outs[0])
    assert final.size == (64, 64)                     # Tensor2PIL resizes back to the hand-off size (models/util.py:159-166)
    arr = np.asarray(final)
    assert arr.dtype == np.uint8 and arr.std() > 1.0


def test_live_llava_caption_on_device(cuda, tmp_path):
    """The caption pass on the GPU (stock PyTorch-ROCm, SDPA attention, fp16): a tiny seeded LLaVA-NeXT is put where
    load_llava() would put the 8 B one and run_stage2_captioning drives it exactly as the CLI does; the caption is a
    function of the seed.  (Token-level parity with the reference model is pinned on CPU in tests/test_llava_next.py.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import llava_common as C
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    from rsvld_amd import infer, llava_next as LN
    tower_dir = C.save_tiny_clip(str(tmp_path / "clip"))
    cfg = LN._llama_config_cls()(**C.LLAMA, **C.MM, mm_vision_tower=tower_dir)
    cfg._attn_implementation = "sdpa"
    model = LN.build_model(cfg, clip=CLIPVisionModel(CLIPVisionConfig(**C.VISION))).eval()
    C.name_seeded_state(model, C.WEIGHT_SEED)
    model.to(device=cuda, dtype=torch.float16)
    pipe = infer.SuperResolutionPipeline.__new__(infer.SuperResolutionPipeline)      # only the caption stage is under test
    pipe.cfg = infer.PipelineConfig(input_img=str(tmp_path / "x.png"), output_dir=str(tmp_path / "out"), seed=3,
                                    base_model_device="cuda:0")
    pipe.llava_model, pipe.llava_tokenizer = model, C.build_tokenizer()
    pipe.llava_image_processor = CLIPImageProcessor.from_pretrained(tower_dir)
    img = C.test_image((100, 70), 5)
    a, b = pipe.run_stage2_captioning(img), pipe.run_stage2_captioning(img)
    assert isinstance(a, str) and a == b and len(a) > 0
    pipe.cfg.caption = "a river delta"
    assert pipe.run_stage2_captioning(img) == "a river delta"


@pytest.mark.parametrize("n", [0, 1, 2])
def test_llava_next_on_device_vs_reference_golden(cuda, tmp_path, golden_dir, n):
    """The caption model ON THE DEVICE (stock PyTorch-ROCm, SDPA) against the vectors the reference's vendored model produced
    (tests/golden/llava_next.npz): spliced input embeddings and next-token logits in fp32 on the device, and the greedy token
    ids -- the same pins tests/test_llava_next.py holds on the CPU, so the device path is compared with the reference, not
    with itself."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import llava_common as C
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    from rsvld_amd import llava_next as LN
    z = np.load(os.path.join(golden_dir, "llava_next.npz"))
    tower_dir = C.save_tiny_clip(str(tmp_path / "clip"))
    cfg = LN._llama_config_cls()(**C.LLAMA, **C.MM, mm_vision_tower=tower_dir)
    cfg._attn_implementation = "sdpa"
    model = LN.build_model(cfg, clip=CLIPVisionModel(CLIPVisionConfig(**C.VISION))).eval()
    C.name_seeded_state(model, C.WEIGHT_SEED)
    model.to(device=cuda, dtype=torch.float32)
    proc = CLIPImageProcessor.from_pretrained(tower_dir)
    img = C.test_image(C.IMAGE_SIZES[n], 5 + n)
    images = [x.to(cuda) for x in LN.process_images([img], proc, model.config)]
    ids = torch.tensor(z["input_ids"]).to(cuda)
    with torch.no_grad():
        emb = model.multimodal_embeds(ids, images, [img.size])
        e_emb = float((emb.cpu() - torch.tensor(z[f"i{n}.embeds"])).abs().max())
        logits = model(inputs_embeds=emb).logits[0, -1]
        e_log = float((logits.cpu() - torch.tensor(z[f"i{n}.logits"])).abs().max())
        greedy = model.generate(ids, images=images, image_sizes=[img.size], do_sample=False, num_beams=1, max_new_tokens=16,
                                return_dict_in_generate=True, output_scores=True)[0][0]
    print(f"LLaVA-NeXT on the device, image {n}: embeddings max|d| = {e_emb:.2e}, logits max|d| = {e_log:.2e}")
    assert e_emb < 2e-5 and e_log < 5e-4
    assert greedy.cpu().tolist() == z[f"i{n}.greedy"].tolist()
    # the product's token loop on a GPU: FastDecoder, decode step replayed from a hipGraph -- same tokens; a second call
    # re-uses the captured graph on a different prompt length only if the cache is large enough (here: same image, same tokens)
    with torch.no_grad():
        fast = LN.caption_tokens_fast(model, ids, images, [img.size], 16, False, 1.0, LN._eos_ids(model, C.build_tokenizer()))
        again = LN.caption_tokens_fast(model, ids, images, [img.size], 16, False, 1.0, LN._eos_ids(model, C.build_tokenizer()))
    assert model._fast_decoder._graph is not None
    assert fast.cpu().tolist() == z[f"i{n}.greedy"].tolist() == again.cpu().tolist()


def test_fast_decoder_fused_decode_step_equals_the_torch_sequence(cuda):
    """The token loop's decode step through the fused kernels (rsvld_gemv_fused, rsvld_llama_decode_attention: 7 launches per layer) against
    the torch-op sequence it replaces (FastDecoder.fused = False), on a seeded Llama of the 8 B decoder's head geometry (head_dim 128, four
    query heads per kv head) in fp16: the same greedy tokens over 24 steps behind a 300-token prompt -- i.e. across a 256-key chunk boundary --
    next-token logits within the 16-bit kernels' tolerance, and both replayed from a hipGraph."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from rsvld_amd import llava_next as LN
    cfg = LlamaConfig(vocab_size=2000, hidden_size=1024, intermediate_size=2048, num_hidden_layers=3, num_attention_heads=8,
                      num_key_value_heads=2, max_position_embeddings=1024, rms_norm_eps=1e-5, rope_theta=500000.0, attention_bias=False)
    torch.manual_seed(11)
    model = LlamaForCausalLM(cfg).eval()
    with torch.no_grad():
        for p_ in model.parameters():                    # a spread of magnitudes instead of the uniform init
            p_.mul_(2.0)
    model.to(device=cuda, dtype=torch.float16)
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg.vocab_size, (1, 300), generator=g).to(cuda)
    emb = model.model.embed_tokens(ids)
    outs, logits = {}, {}
    for fused in (True, False):
        dec = LN.FastDecoder(model, 400)
        dec.fused = fused
        with torch.no_grad():
            outs[fused] = dec.generate(emb, 24, do_sample=False).cpu().tolist()
            assert dec._graph is not None
            logits[fused] = dec._logits.float().cpu()
        assert (dec._ws is not None) == fused            # the fused operators ran (and only then)
    d = float((logits[True] - logits[False]).abs().max())
    rng = float(logits[False].abs().max())
    print(f"FastDecoder, fused decode step vs the torch sequence: last-step logits max|d| = {d:.3e} (range {rng:.2f})")
    assert outs[True] == outs[False] and len(outs[True]) == 24
    assert d < 2e-2 * rng


def test_pipeline_with_live_caption_end_to_end(cuda, tmp_path):
    """BASELINE configs[3] in miniature: Stage 1 -> uint8 hand-off -> LIVE LLaVA-NeXT caption (tiny seeded model on the device)
    -> Stage 2 with that caption -> PNG, through SuperResolutionPipeline.process() exactly as the CLI runs it."""
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import llava_common as C
    from transformers import CLIPImageProcessor, CLIPVisionConfig, CLIPVisionModel
    from rsvld_amd import infer, llava_next as LN
    cfg = yaml.safe_load(open(S.YAML.replace("juggernautXL.yaml", "juggernautXL_cached.yaml")))
    for k in ("control_stage_config", "network_config"):
        cfg["model"]["params"][k]["params"].update(S.SMALL)
    c, uc = S.cond_dicts()
    torch.save(c, tmp_path / "c.pth")
    torch.save(uc, tmp_path / "uc.pth")
    cfg["model"]["params"]["conditioner_config"]["params"] = {"cond_pth": str(tmp_path / "c.pth"), "un_cond_pth": str(tmp_path / "uc.pth")}
    cfg["SR_CKPT"] = cfg["SR_CKPT_Q"] = None
    yaml.safe_dump(cfg, open(tmp_path / "model.yaml", "w"))
    rng = np.random.default_rng(1)
    Image.fromarray(rng.integers(0, 255, (32, 32, 3), dtype=np.uint8)).save(tmp_path / "tile.png")
    pc = infer.PipelineConfig(input_img=str(tmp_path / "tile.png"), output_dir=str(tmp_path / "out"), model_yaml=str(tmp_path / "model.yaml"),
                              allow_random_init=True, no_llava=True, upscale_factor=2, min_size=128, edm_steps=2, sr3_steps=2,
                              seed=1, img_threshold=0.3, base_model_device="cuda:0")
    pipe = infer.SuperResolutionPipeline(pc)                 # no_llava=True: nothing is loaded from disk ...
    tower_dir = C.save_tiny_clip(str(tmp_path / "clip"))     # ... the tiny captioner is put where load_llava() puts the 8 B one
    lcfg = LN._llama_config_cls()(**C.LLAMA, **C.MM, mm_vision_tower=tower_dir)
    lcfg._attn_implementation = "sdpa"
    llava = LN.build_model(lcfg, clip=CLIPVisionModel(CLIPVisionConfig(**C.VISION))).eval()
    C.name_seeded_state(llava, C.WEIGHT_SEED)
    pipe.llava_model, pipe.llava_tokenizer = llava.to(device=cuda, dtype=torch.float16), C.build_tokenizer()
    pipe.llava_image_processor = CLIPImageProcessor.from_pretrained(tower_dir)
    pipe.cfg.no_llava = False
    seen = {}
    orig = pipe.refinement_model.just_sampling

    def spy(x, p, *a, **k):
        seen["caption"] = p[0]
        return orig(x, p, *a, **k)

    pipe.refinement_model.just_sampling = spy
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p_ in pipe.refinement_model.parameters():
            if p_.dim() >= 2 and float(p_.abs().max()) == 0.0:
                p_.copy_((torch.randn(p_.shape, generator=g) * 0.02).to(p_.device))
    outs = pipe.process()
    assert isinstance(seen["caption"], str) and len(seen["caption"]) > 0          # the live caption reached Stage 2
    assert [os.path.basename(o) for o in outs] == ["tile_final_0.png"]
    assert np.asarray(Image.open(outs[0])).std() > 1.0


def test_build_then_smoke_in_one_process(cuda):
    """__graft_entry__.build() followed by smoke() in ONE fresh process: build() loads librsvld_hip.so before anything has
    touched the GPU, and the library must then share torch's HIP runtime (rsvld_amd._lib.load imports torch first; loaded the
    other way round the process held two HIP runtimes and the first launch failed with RSVLD_ELAUNCH)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "smoke: SR3 p_sample" in r.stdout, (r.stdout + r.stderr)[-2000:]
